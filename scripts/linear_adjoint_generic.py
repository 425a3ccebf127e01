#!/usr/bin/env python3
"""The linear right-hand side under odeint_adjoint at config 4's size (65536 x 128, float64, Dopri5; docs/KERNELS.md section 4e): one training
step with the augmented dynamics on the MFMA kernels (models.LinearODEFunc) against the generic path (autograd over rocBLAS: any
nn.Module), and what the three GEMMs of one augmented evaluation - f = yW, -a W^T, -y^T a, 2 B D^2 flop each - cost in rocBLAS."""
import os
import time

import torch

from tfdiffeq_amd import odeint_adjoint

dev = torch.device('cuda:0')
B, D = 65536, 128


from tfdiffeq_amd import adjoint as ADJ  # noqa: E402
from tfdiffeq_amd import models, odeint  # noqa: E402

torch.manual_seed(0)
func = models.LinearODEFunc(D, bias=False).to(dev)
with torch.no_grad():
    S = torch.randn(D, D, dtype=torch.float64, device=dev)
    func.weight.copy_(-0.5 * torch.eye(D, dtype=torch.float64, device=dev) + 0.5 * (S - S.t()) / D ** 0.5)
y0 = torch.randn(B, D, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
t = torch.tensor([0., 1.], dtype=torch.float64)
grads = {}
ONLY = os.environ.get('LIN_ONLY') == '1'                    # (profiling runs: the MFMA path alone)
for fused in ((True,) if ONLY else (True, False)):
    ADJ.FUSED = ADJ.FUSED_FORWARD = fused
    for it in range(6 if fused else 3):
        func.weight.grad = None
        yi = y0.clone().requires_grad_(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = odeint_adjoint(func, yi, t, rtol=1e-6, atol=1e-9, method='dopri5')
        fwd = dict(odeint.last_stats)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out[-1].sum().backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        st = dict(odeint_adjoint.last_backward_stats)
        seg = st.get('last_segment', {})
        print('%s call %d: forward %.2f ms (%s launches), backward %.2f ms [%s; last interval: %s attempts, engine %s]' % (
            'MFMA dynamics' if fused else 'generic path ', it, (t1 - t0) * 1e3, fwd.get('n_launches'), (t2 - t1) * 1e3, st.get('engine', '')[:60],
            seg.get('n_attempts'), str(seg.get('engine'))[:70]), flush=True)
    grads[fused] = (func.weight.grad.clone(), yi.grad.clone())
ADJ.FUSED = ADJ.FUSED_FORWARD = True
if ONLY:
    raise SystemExit(0)
print('gradients, MFMA dynamics vs generic path: dL/dW rel %.2e, dL/dy0 rel %.2e' % tuple(
    float((a_ - b_).abs().max() / b_.abs().max()) for a_, b_ in zip(grads[True], grads[False])))
a = torch.randn(B, D, dtype=torch.float64, device=dev)
W = func.weight.detach()
for _ in range(3):
    y0 @ W; a @ W.t(); y0.t() @ a
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    y0 @ W; a @ W.t(); y0.t() @ a
torch.cuda.synchronize()
per = (time.perf_counter() - t0) / 20
print('the three GEMMs of one augmented evaluation in rocBLAS: %.3f ms (%.1f TFLOP/s fp64)' % (per * 1e3, 3 * 2 * B * D * D / per / 1e12))

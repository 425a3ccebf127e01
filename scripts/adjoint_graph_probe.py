"""Probe: can the adjoint's backward solve (torch.autograd.grad inside the augmented dynamics) run as one hipGraph replay
per attempt?  Compares gradients and wall time with / without options={'graph': True} for the backward solve."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint_adjoint, models

dev = torch.device('cuda:0')
torch.manual_seed(0)
func = models.ODEFunc(64, 128, non_linearity='tanh').to(dev)
x = torch.randn(4096, 64, device=dev)
t = torch.tensor([0., 1.])


def run(adj_opts):
    for p in func.parameters():
        p.grad = None
    xx = x.clone().requires_grad_(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = odeint_adjoint(func, xx, t, rtol=1e-3, atol=1e-3, method='dopri5', options={'max_num_steps': 1000}, adjoint_options=adj_opts)
    out[1].pow(2).mean().backward()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, xx.grad.clone(), [p.grad.clone() for p in func.parameters()]


for label, opts in (('eager', {'max_num_steps': 1000}), ('graph', {'max_num_steps': 1000, 'graph': True})):
    try:
        run(opts)
        times = []
        for _ in range(3):
            dt, gx, gp = run(opts)
            times.append(dt)
        print(label, 'ok: %.1f ms per forward+backward' % (1e3 * min(times)), 'grad |x| max %.4e' % float(gx.abs().max()))
        if label == 'eager':
            ref = (gx, gp)
        else:
            print('  max |grad diff| x: %.3e  params: %.3e' % (float((gx - ref[0]).abs().max()), max(float((a - b).abs().max()) for a, b in zip(gp, ref[1]))))
    except Exception as e:
        import traceback
        print(label, 'FAILED:', type(e).__name__, str(e)[:300])
        traceback.print_exc(limit=3)

"""One training step of ODEBlock(adjoint=True) at BASELINE config 5's shape (batch 32768, 64-128-128-64 tanh, fp32, tol 1e-3):
forward + loss + backward, fused kernels vs the generic plane-kernel path.  Prints one JSON line per mode.
usage: python scripts/adjoint_train_step.py [fused|planes|both] [steps]      (ADJ_TD=1: the time-dependent network)"""
import json
import os
import sys
import time

import torch

from tfdiffeq_amd import adjoint as ADJ
from tfdiffeq_amd import models, odeint

mode = sys.argv[1] if len(sys.argv) > 1 else 'both'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = 'cuda'
torch.manual_seed(0)
block = models.ODEBlock(models.ODEFunc(64, 128, non_linearity='tanh', time_dependent=os.environ.get('ADJ_TD', '0') == '1'), tol=1e-3,
                        adjoint=True).to(dev)
x = torch.randn(32768, 64, device=dev)
w = torch.randn(32768, 64, device=dev)


def step():
    for p in block.parameters():
        p.grad = None
    xi = x.clone().requires_grad_(True)
    out = block(xi)
    (out * w).sum().backward()
    return xi.grad


for name, fused in (('fused', True), ('planes', False)):
    if mode not in (name, 'both'):
        continue
    ADJ.FUSED = fused
    step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.time() - t0) / steps * 1e3
    fwd = dict(odeint.last_stats)
    bwd = dict(ADJ.odeint_adjoint.last_backward_stats)
    seg = bwd.get('segments', [{}])
    print(json.dumps({'mode': name, 'ms_per_training_step': round(ms, 3), 'forward': {k: fwd.get(k) for k in ('engine', 'n_attempts', 'n_launches')},
                      'backward_engine': bwd.get('engine'), 'backward_attempts': [s.get('n_attempts') for s in seg],
                      'state_elements_per_s': round(32768 * 64 / (ms * 1e-3))}))

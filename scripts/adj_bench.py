"""Time the two passes of the fused adjoint kernel in isolation (MI_ODE_ADJOINT_BENCH=mode,iters): config 5 size."""
import os
import sys

import torch

from tfdiffeq_amd import adjoint as ADJ
from tfdiffeq_amd.models import ODEFunc

torch.manual_seed(0)
dev = 'cuda'
batch, dim, hidden = 32768, 64, 128
func = ODEFunc(dim, hidden, non_linearity='tanh').to(dev)
y = torch.randn(batch, dim, device=dev)
a = torch.randn(batch, dim, device=dev) / batch
mlp = func.device_rhs()
eng = ADJ._FusedAdjointEngine(batch, dim, hidden, 1e-3, 1e-3, 0.9, 10.0, 0.2, 1000, dev)
theta0 = torch.zeros(eng.n_params, device=dev)
adj_t = torch.tensor(0.3, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    eng.segment(mlp, y, a, adj_t, theta0, 1.0, 0.0)
torch.cuda.synchronize()

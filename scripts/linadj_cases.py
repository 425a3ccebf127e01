"""Isolate a failing configuration of the one-launch linear adjoint: a few (batch, dim) cases, optional grid override per case."""
import os
import sys
import torch
sys.path.insert(0, '.')
from scripts.linadj_check import case
from tfdiffeq_amd import adjoint as ADJ

tol = dict(rtol=1e-6, atol=1e-9, method='dopri5')
for grid, batch, dim, bias in [(None, 20000, 64, True), ('128', 70001, 128, True), ('200', 5000, 128, False), ('255', 5000, 128, False), (None, 5000, 128, False)]:
    if grid is None:
        os.environ.pop('MI_ODE_LINADJ_GRID', None)
    else:
        os.environ['MI_ODE_LINADJ_GRID'] = grid
    ADJ.clear_adjoint_engines()
    print('grid override', grid)
    case(batch, dim, bias, torch.float64, [0.0, 1.0], tol)

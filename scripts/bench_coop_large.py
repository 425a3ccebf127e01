#!/usr/bin/env python3
"""rhs.CustomCoop beyond a co-resident grid: the plane-streaming whole-call kernel (round 5) on a reaction-diffusion ring of 100 cells,
dopri5 float64, rtol 1e-6 / atol 1e-8, t = [0, 2]; HBM traffic per attempt = 4 planes (y0, f0 in; y1, f1 out)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint  # noqa: E402
from tfdiffeq_amd import plugin_examples as PE  # noqa: E402

dev = torch.device('cuda:0')
f = PE.reaction_diffusion_ring(100)
t = torch.tensor([0., 2.0], dtype=torch.float64)
for batch in (1000, 10000, 100000, 400000):
    y0 = torch.randn(batch, 100, generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dev)
    for _ in range(2):
        odeint(f, y0, t, method='dopri5', rtol=1e-6, atol=1e-8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        odeint(f, y0, t, method='dopri5', rtol=1e-6, atol=1e-8)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 3
    st = dict(odeint.last_stats)
    gb = 4 * batch * 100 * 8 * st['n_attempts'] / 1e9
    print('ring of 100 cells, batch %7d: %8.3f ms per call, %d launches, %d attempts, %7.1f us per attempt, %.2f TB/s of the 4 planes per attempt' % (
        batch, ms, st['n_launches'], st['n_attempts'], 1e3 * ms / st['n_attempts'], gb / ms), flush=True)

#!/usr/bin/env python3
"""'adams' (variable order) on the 64-spiral fixture shape and on 65 536 spirals: the one-launch kernel (csrc/mi_ode_adams_vc.h) against
the per-step loop over plane kernels.   python scripts/bench_adams.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs  # noqa: E402

dev = torch.device('cuda:0')
A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64)
t = torch.linspace(0, 5, 20, dtype=torch.float64)
for batch in (64, 65536):
    y0 = torch.tensor([[2., 0.]], dtype=torch.float64).repeat(batch, 1).to(dev)
    f = rhs.CubicLinear(A)
    for opt in ({}, {'force_plane_kernels': True}):
        for _ in range(2):
            odeint(f, y0, t, method='adams', rtol=1e-6, atol=1e-8, options=opt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        odeint(f, y0, t, method='adams', rtol=1e-6, atol=1e-8, options=opt)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        st = dict(odeint.last_stats)
        print('adams, %5d spirals, %-48s %8.2f ms, %d attempts, %.1f us per attempt' % (batch, st.get('engine'), ms, st['n_attempts'], 1e3 * ms / st['n_attempts']), flush=True)

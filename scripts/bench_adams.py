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

# round 5: the ODEFunc network (models/dense_odenet.py:41-92) under the Adams family - one launch (RhsMlpCoop: a thread per state element, the three
# layers through LDS; float32 and float64, dim / hidden <= 256) against the host loop over the same network as a Python callable
for dim, hidden, batch, dtype in ((2, 50, 64, torch.float64), (64, 128, 256, torch.float32), (64, 128, 256, torch.float64), (16, 32, 4096, torch.float64)):
    g = torch.Generator().manual_seed(dim + hidden)
    mk = lambda *s: (0.7 * torch.randn(*s, generator=g, dtype=torch.float64) / s[0] ** 0.5).to(dtype)  # noqa: E731
    f = rhs.MLP(mk(dim, hidden), None, mk(hidden, hidden), None, mk(hidden, dim), None, activation='tanh')
    y0 = torch.randn(batch, dim, generator=g, dtype=torch.float64).to(dtype).to(dev)
    for method, tt in (('adams', torch.linspace(0, 2, 5, dtype=torch.float64)), ('explicit_adams', torch.linspace(0, 0.5, 51, dtype=torch.float64))):
        tol = dict(rtol=1e-6, atol=1e-8) if dtype == torch.float64 else dict(rtol=1e-4, atol=1e-6)
        for func, label in ((f, None), (lambda t_, y: f.forward(t_, y), 'host loop, Python callable')):
            for _ in range(2):
                odeint(func, y0, tt, method=method, **tol)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            odeint(func, y0, tt, method=method, **tol)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            st = dict(odeint.last_stats)
            n = st.get('n_attempts') or (len(tt) - 1)
            print('%-14s MLP %d-%d-%d-%d %s batch %5d, %-44s %8.2f ms, %d steps, %.1f us per step' % (
                method, dim, hidden, hidden, dim, str(dtype).replace('torch.', ''), batch, label or st.get('engine'), ms, n, 1e3 * ms / n), flush=True)

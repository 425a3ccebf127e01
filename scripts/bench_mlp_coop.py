#!/usr/bin/env python3
"""The ODEFunc network outside the MFMA tile kernels' box (float64, wide) under dopri5: the cooperative one-launch kernel (round 5: a thread
per state element, RhsMlpCoop under k_persist_rowlocal) against the same network as a Python callable on the device-controlled engine."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs  # noqa: E402

dev = torch.device('cuda:0')
t = torch.tensor([0., 1.0], dtype=torch.float64)
for dim, hidden, batch, dtype in ((2, 50, 64, torch.float64), (2, 50, 4096, torch.float64), (16, 32, 4096, torch.float64), (64, 128, 256, torch.float64),
                                  (64, 128, 4096, torch.float64), (100, 200, 1024, torch.float32)):
    g = torch.Generator().manual_seed(dim + hidden)
    mk = lambda *s: (0.7 * torch.randn(*s, generator=g, dtype=torch.float64) / s[0] ** 0.5).to(dtype)  # noqa: E731
    f = rhs.MLP(mk(dim, hidden), None, mk(hidden, hidden), None, mk(hidden, dim), None, activation='tanh')
    y0 = torch.randn(batch, dim, generator=g, dtype=torch.float64).to(dtype).to(dev)
    tol = dict(rtol=1e-6, atol=1e-8) if dtype == torch.float64 else dict(rtol=1e-4, atol=1e-6)
    for func, label in ((f, 'cooperative kernel'), (lambda t_, y: f.forward(t_, y), 'Python callable')):
        for _ in range(2):
            odeint(func, y0, t, method='dopri5', **tol)
        gc.collect()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            odeint(func, y0, t, method='dopri5', **tol)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / 5
        st = dict(odeint.last_stats)
        print('dopri5 MLP %3d-%3d-%3d-%3d %s batch %5d, %-18s %-58s %8.3f ms, %3d attempts, %6.1f us per attempt, %d launches' % (
            dim, hidden, hidden, dim, str(dtype).replace('torch.', ''), batch, label, str(st.get('engine'))[:58], ms, st['n_attempts'],
            1e3 * ms / st['n_attempts'], st.get('n_launches') or 0), flush=True)

# rhs.CustomCoop (user device code for one state element): a reaction-diffusion ring of 100 cells and a dense layer with the user's own
# pointwise function, against the same function as a Python callable (torch ops) on the device-controlled engine
from tfdiffeq_amd import plugin_examples as PE  # noqa: E402
g = torch.Generator().manual_seed(5)
cases = [('ring of 100 cells', PE.reaction_diffusion_ring(100), 100),
         ('swish layer, dim 48', PE.swish_layer((0.5 * torch.randn(48, 48, generator=g, dtype=torch.float64) / 48 ** 0.5).to(dev),
                                                (0.1 * torch.randn(48, generator=g, dtype=torch.float64)).to(dev)), 48)]
for name, f, dim in cases:
    for batch, tt in ((64, torch.tensor([0., 1.0], dtype=torch.float64)), (1000, torch.tensor([0., 20.0], dtype=torch.float64))):
        y0 = torch.randn(batch, dim, generator=g, dtype=torch.float64).to(dev)
        for func, label in ((f, 'cooperative kernel'), (lambda t_, y: f.forward(t_, y), 'Python callable')):
            for _ in range(2):
                odeint(func, y0, tt, method='dopri5', rtol=1e-6, atol=1e-8)
            gc.collect()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                odeint(func, y0, tt, method='dopri5', rtol=1e-6, atol=1e-8)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / 3
            st = dict(odeint.last_stats)
            print('dopri5 %-20s float64 batch %5d t_end %4.0f, %-18s %-58s %8.3f ms, %4d attempts, %6.1f us per attempt' % (
                name, batch, float(tt[-1]), label, str(st.get('engine'))[:58], ms, st['n_attempts'], 1e3 * ms / st['n_attempts']), flush=True)

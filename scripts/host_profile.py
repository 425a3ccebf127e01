#!/usr/bin/env python3
"""cProfile of the host side of one-launch odeint calls (config 4 and a tiny Lorenz call): where the Python microseconds of a call go.
    python scripts/host_profile.py [calls]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs  # noqa: E402

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
g = torch.Generator().manual_seed(2)
S = torch.randn(128, 128, generator=g, dtype=torch.float64)
A = -0.5 * torch.eye(128, dtype=torch.float64) + 0.5 * (S - S.t()) / np.sqrt(128)
cases = {'config 4 (65536 x 128 linear, dopri5)': (rhs.Linear.from_matrix(A), torch.randn(65536, 128, generator=g, dtype=torch.float64).to(dev), torch.tensor([0., 1.]),
                                                    dict(rtol=1e-6, atol=1e-9, method='dopri5'), 300),
         'lorenz, 64 trajectories, dopri5 T = 2': (rhs.Lorenz(), (torch.ones(64, 3, dtype=torch.float64) + 0.01 * torch.randn(64, 3, generator=g, dtype=torch.float64)).to(dev),
                                                   torch.tensor([0., 0.05]), dict(method='dopri5'), n)}
for name, (f, y0, t, kw, reps) in cases.items():
    for _ in range(20):
        odeint(f, y0, t, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        odeint(f, y0, t, **kw)
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / reps
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        odeint(f, y0, t, **kw)
    pr.disable()
    print('== %s: %.1f us per call (unprofiled), attempts %s' % (name, us, dict(odeint.last_stats).get('n_attempts')))
    st = pstats.Stats(pr)
    st.sort_stats('tottime')
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:22]
    for (fn, line, func), (cc, nc, tt, ct, _) in rows:
        print('  %8.2f us tottime %8.2f us cumtime  %5.1f calls  %s:%d %s' % (1e6 * tt / reps, 1e6 * ct / reps, nc / reps, os.path.basename(fn), line, func))

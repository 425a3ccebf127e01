#!/usr/bin/env python3
"""Whole-call kernel on a grid clamped to what is co-resident against one launch per attempt, linear right-hand side at widths / batches whose
per-attempt grid is NOT co-resident (round 6: csrc/mi_ode_api.hip clamps the grid of the tile kernels instead of giving the whole-call schedule up).
Median of 30 synchronised dopri5 calls, float64, rtol 1e-6 atol 1e-9, t = [0, 1]; both schedules must be bit-identical."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs
dev = torch.device('cuda:0')
for D, batch in ((64, 4096), (64, 5000), (64, 8200), (64, 65536), (16, 65536), (32, 20000), (100, 6000), (128, 5000)):
    g = torch.Generator().manual_seed(D)
    S = torch.randn(D, D, generator=g, dtype=torch.float64)
    Wc = (-0.5 * torch.eye(D, dtype=torch.float64) + 0.5 * (S - S.t()) / np.sqrt(D))
    y0c = torch.randn(batch, D, generator=g, dtype=torch.float64)
    W, y0 = Wc.to(dev), y0c.to(dev)
    f = rhs.Linear(W); t = torch.tensor([0., 1.])
    out = {}
    for fusion in ('auto', 'step'):
        for _ in range(10): sol = odeint(f, y0, t, rtol=1e-6, atol=1e-9, method='dopri5', options={'fusion': fusion})
        torch.cuda.synchronize(); per = []
        for _ in range(30):
            t0 = time.perf_counter(); sol = odeint(f, y0, t, rtol=1e-6, atol=1e-9, method='dopri5', options={'fusion': fusion}); torch.cuda.synchronize(); per.append(time.perf_counter() - t0)
        s = dict(odeint.last_stats)
        out[fusion] = (1e6 * float(np.median(per)), s['n_launches'], s['n_attempts'], sol)
    n = min(batch, 2000)
    same = torch.equal(out['auto'][3], out['step'][3])
    print('dim %3d batch %5d: whole-call %.1f us (%s launch) | per attempt %.1f us (%s launches) | attempts %s | bit-identical %s' % (
        D, batch, out['auto'][0], out['auto'][1], out['step'][0], out['step'][1], out['auto'][2], same))

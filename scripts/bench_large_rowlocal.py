#!/usr/bin/env python3
"""Row-local systems beyond one trajectory per thread (> 131 072 rows): the one-launch schedule with the state in HBM planes
(k_persist_rowlocal_planes) against one launch per attempt.  Lorenz, Tsit5 fp64, rtol 1e-6 / atol 1e-9, t = [0, 1] (config 3's
problem at larger batches).   python scripts/bench_large_rowlocal.py [batch ...]"""
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs

dev = torch.device('cuda:0')
t = torch.tensor([0., 1.], dtype=torch.float64)
for batch in [int(a) for a in sys.argv[1:]] or [131072, 262144, 1048576, 4194304]:
    g = torch.Generator().manual_seed(3)
    y0 = (torch.tensor([1., 1., 1.], dtype=torch.float64) + 1e-2 * torch.randn(batch, 3, generator=g, dtype=torch.float64)).to(dev)
    row = 'batch %8d:' % batch
    for fusion in ('auto', 'step'):
        kw = dict(rtol=1e-6, atol=1e-9, method='tsit5', options={'fusion': fusion})
        for _ in range(3):
            odeint(rhs.Lorenz(), y0, t, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            odeint(rhs.Lorenz(), y0, t, **kw)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        st = dict(odeint.last_stats)
        row += '  %s %.3f ms (%d launches, %d attempts, %.1f us/attempt)' % (fusion, ms, st['n_launches'], st['n_attempts'], 1e3 * ms / st['n_attempts'])
    print(row, flush=True)

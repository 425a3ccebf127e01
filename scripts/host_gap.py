#!/usr/bin/env python3
"""Where the time between two whole-call kernels goes (configs 4 and 5): wall per odeint() call, the part inside the C entry
point, the kernel's own duration by HIP events - with and without the event records of profile=True.
  wall - C = Python side of odeint();  C - kernel = launch + completion wait (+ the two event records)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tfdiffeq_amd import odeint, rhs, solvers  # noqa: E402

N = int(os.environ.get('CALLS', '300'))


class _Args:
    scaling = 'weak'
    batch = bench.BATCH


def cases():
    dev = torch.device('cuda:0')
    for cfg in (4, 5, 2):
        f, y0, t, kw, _ = bench.workload(cfg, _Args, 0, 1, dev)
        kw = dict(kw)
        base = dict(kw.pop('options', {}))
        yield 'config %d' % cfg, f, y0, kw, t, base


c_ns = [0]
orig = solvers._FusedEngine.integrate


def timed(self, *a, **k):
    t0 = time.perf_counter_ns()
    r = orig(self, *a, **k)
    c_ns[0] += time.perf_counter_ns() - t0
    return r


for name, f, y0, kw, t, base in cases():
    for prof in (False, True):
        opts = dict(base, profile=True) if prof else dict(base)
        for _ in range(20):
            odeint(f, y0, t, options=dict(opts), **kw)
        torch.cuda.synchronize()
        solvers._FusedEngine.integrate = timed
        c_ns[0] = 0
        kern = 0.0
        nk = 0
        t0 = time.perf_counter_ns()
        for _ in range(N):
            odeint(f, y0, t, options=dict(opts), **kw)
            if prof:
                p = dict(odeint.last_stats).get('profile', [0, 0, 0, 0])
                kern += p[2]
                nk += int(p[1])
        torch.cuda.synchronize()
        wall = (time.perf_counter_ns() - t0) / N * 1e-3
        solvers._FusedEngine.integrate = orig
        line = '%s profile=%d: wall %.1f us per call, engine.integrate() %.1f us' % (name, prof, wall, c_ns[0] / N * 1e-3)
        if nk:
            line += ', kernel by events %.1f us' % (kern / nk * 1e3)
        print(line, flush=True)

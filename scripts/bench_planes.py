#!/usr/bin/env python3
"""Bandwidth of the stateless plane kernels at a 64 MiB plane (config-4 size) and an 8 MiB fp32 plane."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import misc

dev = torch.device('cuda:0')
for dtype, n in ((torch.float64, 65536 * 128), (torch.float32, 32768 * 64)):
    xs = [torch.randn(n, dtype=dtype, device=dev) for _ in range(7)]
    base = torch.randn(n, dtype=dtype, device=dev)
    es = xs[0].element_size()
    for nx in (1, 3, 7):
        for _ in range(3):
            misc._lincomb(base, [0.1] * nx, xs[:nx], 0.5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            misc._lincomb(base, [0.1] * nx, xs[:nx], 0.5)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print('lincomb %s n=%d nx=%d: %.1f us  %.0f GB/s' % (str(dtype)[6:], n, nx, dt * 1e6, (nx + 2) * n * es / dt / 1e9))
    for _ in range(3):
        misc._error_norms(xs[0], xs[1], xs[2])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        misc._error_norms(xs[0], xs[1], xs[2])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print('error_norms %s n=%d: %.1f us  %.0f GB/s' % (str(dtype)[6:], n, dt * 1e6, 3 * n * es / dt / 1e9))

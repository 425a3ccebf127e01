#!/usr/bin/env python3
"""Float64 ODEFunc 64-128-128-64 (tanh) under ODEBlock's call (t = [0, 1], dopri5, tol 1e-3): the float64 MFMA tile kernels
(csrc/mi_ode_mlp64.h) against the same network as a Python callable (rocBLAS + the device-controlled engine) - the round-5 review's item 3
(1.80 ms per call at 4096 rows)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import models, odeint  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
for d, h in ((64, 128), (16, 16)):
    blk = models.ODEBlock(models.ODEFunc(d, h, non_linearity='tanh'), tol=1e-3).to(dev).double()
    for batch in (256, 4096, 32768, 131072):
        x = torch.randn(batch, d, dtype=torch.float64, device=dev)
        with torch.no_grad():
            for _ in range(5):
                blk(x)
            st = dict(odeint.last_stats)
            ts = []
            for _ in range(50):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                blk(x)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            ms, ms_min, ms_max = ts[len(ts) // 2], ts[0], ts[-1]
            f = lambda t, y: blk.odefunc(t, y)   # noqa: E731
            opts = {'lower': False, 'max_num_steps': 1000}
            for _ in range(3):
                odeint(f, x, torch.tensor([0., 1.]), rtol=1e-3, atol=1e-3, method='dopri5', options=opts)
            tc = []
            for _ in range(10):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                odeint(f, x, torch.tensor([0., 1.]), rtol=1e-3, atol=1e-3, method='dopri5', options=opts)
                torch.cuda.synchronize()
                tc.append((time.perf_counter() - t0) * 1e3)
            ms_c = sorted(tc)[len(tc) // 2]
        nfe = int(st.get('nfe', 0))
        flops = nfe * batch * 2.0 * (d * h + h * h + h * d)
        print('float64 %d-%d-%d-%d tanh batch %6d: ODEBlock call %.3f ms median of 50 (min %.3f, max %.3f) on the tile kernels (attempts %d, NFE %d, launches %d, %.1f TF = %.2f of the fp64 MFMA peak) | '
              'as a Python callable %.3f ms' % (d, h, h, d, batch, ms, ms_min, ms_max, st['n_attempts'], nfe, st['n_launches'], flops / ms / 1e9, flops / ms / 1e9 / 78.6, ms_c), flush=True)

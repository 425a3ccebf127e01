#!/usr/bin/env python3
"""Host-side cost of one odeint() call on the fused path (config 4): cProfile over many calls."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tfdiffeq_amd import odeint, rhs  # noqa: E402

A, y0 = bench.config4(65536, 128, 3)
f = rhs.Linear.from_matrix(A)
y0 = y0.cuda()
t = torch.tensor([0., 1.], dtype=torch.float64)
for _ in range(3):
    odeint(f, y0, t, rtol=1e-6, atol=1e-9, method='dopri5')
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    odeint(f, y0, t, rtol=1e-6, atol=1e-9, method='dopri5')
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)

#!/usr/bin/env python3
"""Your own right-hand side as device code (no reference counterpart: the reference only takes Python callables).

Three ways to hand `odeint` a function, fastest last:
  * a Python callable over torch ops - the reference's call shape; the step controller runs on the device, attempts are replayed as a hipGraph;
  * rhs.CustomRowLocal(dim <= 32, body): a thread owns a TRAJECTORY - `k[0..dim-1]` from `y[0..dim-1]`;
  * rhs.CustomCoop(dim <= 256, body): a thread owns ONE ELEMENT - `k` for element `i` from the trajectory's state `y[0..DIM-1]`.
Both device forms run every adaptive method, euler / rk4 and the Adams family as ONE kernel launch per odeint call.

    python examples/custom_device_rhs.py            (needs an MI355X; the first run compiles two small plugins with hipcc)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs  # noqa: E402

dev = torch.device('cuda:0')
t = torch.tensor([0., 5.0], dtype=torch.float64)


def timed(label, f, y0, **kw):
    for _ in range(2):
        out = odeint(f, y0, t, rtol=1e-6, atol=1e-8, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = odeint(f, y0, t, rtol=1e-6, atol=1e-8, **kw)
    torch.cuda.synchronize()
    st = dict(odeint.last_stats)
    print('%-34s %8.3f ms  %4d attempts  %d launch(es)' % (label, 1e3 * (time.perf_counter() - t0), st['n_attempts'], st.get('n_launches') or 0))
    return out


# 1. Van der Pol oscillators, one trajectory per thread
mu = 2.0
vdp_py = lambda t_, y: torch.stack([y[..., 1], mu * (1 - y[..., 0] ** 2) * y[..., 1] - y[..., 0]], dim=-1)  # noqa: E731
vdp = rhs.CustomRowLocal(2, "k[0] = y[1]; k[1] = p[0] * (1 - y[0] * y[0]) * y[1] - y[0];", params=[mu])
y0 = torch.randn(4096, 2, dtype=torch.float64, device=dev)
a = timed('van der Pol, Python callable', vdp_py, y0, method='dopri5')
b = timed('van der Pol, rhs.CustomRowLocal', vdp, y0, method='dopri5')
print('   max |device code - Python callable| = %.2e' % float((a - b).abs().max()))

# 2. a reaction-diffusion ring of 128 cells, one cell per thread (the trajectory's state is shared through LDS)
D, c = 0.8, 0.05
ring_py = lambda t_, y: D * (torch.roll(y, -1, -1) - 2 * y + torch.roll(y, 1, -1)) - c * y ** 3  # noqa: E731
ring = rhs.CustomCoop(128, "k = p[0] * (y[(i + 1) % DIM] - 2 * y[i] + y[(i + DIM - 1) % DIM]) - p[1] * y[i] * y[i] * y[i];", params=[D, c], torch_fn=ring_py)   # (torch_fn: optional, used where only a callable will do)
u0 = torch.randn(2000, 128, dtype=torch.float64, device=dev)
a = timed('diffusion ring, Python callable', ring_py, u0, method='dopri5')
b = timed('diffusion ring, rhs.CustomCoop', ring, u0, method='dopri5')
print('   max |device code - Python callable| = %.2e' % float((a - b).abs().max()))
timed('diffusion ring, CustomCoop, adams', ring, u0[:500], method='adams')     # (the Adams kernels need a co-resident grid: 250 workgroups here)

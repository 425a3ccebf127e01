#!/usr/bin/env python3
"""Soak of the 256-wide linear tile kernels (round 6: dims 129 .. 256, W streamed): random dims, batches 1 .. 5000, both dtypes, dopri5 / tsit5 /
bosh3 / rk4 / euler, bias or not, both directions, 2 .. 40 output times - the one-launch kernel against the numpy ORACLE on the same system
(float64: identical attempt / accept counts and 1e-11; float32: the bands of tests/bands.py - an attempt more or less, 10 x rtol).
    python scripts/soak_linear_wide.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs  # noqa: E402
import oracle.ode_numpy as O  # noqa: E402

dev = torch.device('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst64, worst32, bad = 0.0, 0.0, 0
for case in range(n_cases):
    dim = int(rng.integers(129, 257))
    batch = int(rng.choice([1, 2, 15, 16, 17, 100, 777, 2048, 5000]))
    dtype = torch.float64 if rng.random() < 0.65 else torch.float32
    method = str(rng.choice(['dopri5', 'dopri5', 'tsit5', 'bosh3', 'rk4', 'euler']))
    bias = bool(rng.random() < 0.5)
    sgn = -1.0 if rng.random() < 0.3 else 1.0
    T = int(rng.choice([2, 3, 7, 40]))
    g = torch.Generator().manual_seed(5000 + case)
    S = torch.randn(dim, dim, generator=g, dtype=torch.float64)
    W = (-0.5 * torch.eye(dim, dtype=torch.float64) + 0.5 * (S - S.t()) / np.sqrt(dim)).to(dtype)
    b = (0.1 * torch.randn(dim, generator=g, dtype=torch.float64)).to(dtype)
    y0 = torch.randn(batch, dim, generator=g, dtype=torch.float64).to(dtype)
    span = 1.0 if method != 'bosh3' else 0.2
    t = np.linspace(0., span, T) * sgn
    kw = {} if method in ('rk4', 'euler') else (dict(rtol=1e-6, atol=1e-9) if dtype == torch.float64 else dict(rtol=1e-4, atol=1e-6))
    Wn, bn = W.numpy(), b.numpy()
    fo = (lambda t_, y: y @ Wn + bn) if bias else (lambda t_, y: y @ Wn)
    ref, st = O.odeint(fo, y0.numpy(), t.astype(Wn.dtype), method=method, return_stats=True, options={'tsit5_fixed': True} if method == 'tsit5' else None, **kw)
    sol = odeint(rhs.Linear(W, b if bias else None), y0.to(dev), torch.tensor(t), method=method, **kw)
    s = dict(odeint.last_stats)
    diff = float(np.abs(sol.cpu().numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
    adaptive = method not in ('rk4', 'euler')
    if dtype == torch.float64:
        same = (not adaptive) or (s['n_attempts'], s['n_accepted']) == (st.n_attempts, st.n_accepted)
        band = 1e-11
        worst64 = max(worst64, diff)
    else:
        same = (not adaptive) or abs(s['n_attempts'] - st.n_attempts) <= 1
        band = 1e-3 if adaptive else 1e-4
        worst32 = max(worst32, diff)
    ok = same and diff <= band and s.get('n_launches') == 1 and bool(torch.isfinite(sol).all())
    bad += 0 if ok else 1
    print('%s case %3d: %-6s %-7s dim %3d batch %4d T %2d bias %d dir %+d | launches %s attempts %s/%s (oracle %s/%s) dev %.2e (band %.0e)' % (
        'ok  ' if ok else 'FAIL', case, method, str(dtype).replace('torch.', ''), dim, batch, T, bias, int(sgn), s.get('n_launches'), s.get('n_attempts'),
        s.get('n_accepted'), getattr(st, 'n_attempts', None), getattr(st, 'n_accepted', None), diff, band), flush=True)
print('%d cases, %d failed, worst deviation float64 %.2e, float32 %.2e' % (n_cases, bad, worst64, worst32))
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Soak: random problems through the whole-call kernels vs one launch per attempt; any hand-off race shows up as a
difference (the two schedules are bit-identical by construction)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs  # noqa: E402

dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
n_whole = 0
t_start = time.time()
for it in range(n_runs):
    kind = rng.choice(['lorenz', 'lv', 'spiral', 'linear', 'mlp'])
    method = rng.choice(['dopri5', 'tsit5', 'bosh3'])
    if kind in ('lorenz', 'lv', 'spiral') and rng.random() < 0.3:
        method = rng.choice(['dopri8', 'adaptive_heun'])       # wide / non-FSAL tableaus: row-local kernels only
    dtype = torch.float64 if (kind != 'mlp' and rng.random() < 0.7) else torch.float32
    if kind == 'mlp':
        dtype = torch.float32
    tol = dict(rtol=10.0 ** rng.uniform(-7, -4), atol=10.0 ** rng.uniform(-9, -6)) if dtype == torch.float64 else dict(rtol=1e-4, atol=1e-6)
    T = int(rng.integers(2, 12))
    span = 10.0 ** rng.uniform(-1.5, 0.3) * (0.05 if method in ('bosh3', 'adaptive_heun') else 1.0)
    t = torch.tensor(np.sort(np.concatenate([[0.0], rng.uniform(0, span, size=T - 1)])))
    if (t[1:] - t[:-1]).min() <= 0:
        continue
    if kind == 'lorenz':
        batch = int(rng.choice([1, 7, 64, 300, 4096, 20000, 65536, 100000]))
        f, y0 = rhs.Lorenz(), np.array([1., 1., 1.]) + 1e-2 * rng.standard_normal((batch, 3))
    elif kind == 'lv':
        batch = int(rng.choice([1, 256, 5000, 70000]))
        f, y0 = rhs.LotkaVolterra(), 1 + 0.5 * rng.uniform(size=(batch, 2))
    elif kind == 'spiral':
        batch = int(rng.choice([1, 33, 4096, 30000]))
        f, y0 = rhs.CubicLinear(torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64)), rng.uniform(-2, 2, size=(batch, 2))
    elif kind == 'linear':
        D = int(rng.choice([16, 32, 64, 128]))
        batch = int(rng.choice([5, 100, 3000, 20000]))
        S_ = rng.standard_normal((D, D))
        A = -0.5 * np.eye(D) + 0.5 * (S_ - S_.T) / np.sqrt(D)
        f, y0 = rhs.Linear.from_matrix(torch.tensor(A)), rng.standard_normal((batch, D))
    else:
        d_, h_ = [(64, 128), (10, 20), (16, 16)][int(rng.integers(0, 3))]
        batch = int(rng.choice([50, 1000, 20000]))
        g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
        mk = lambda i, o: ((torch.rand(i, o, generator=g) * 2 - 1) * (6.0 / (i + o)) ** 0.5).to(dev)  # noqa: E731
        f = rhs.MLPTanh(mk(d_, h_), torch.zeros(h_, device=dev), mk(h_, h_), torch.zeros(h_, device=dev), mk(h_, d_), torch.zeros(d_, device=dev))
        y0 = rng.standard_normal((batch, d_))
    y0 = torch.tensor(y0, dtype=dtype, device=dev)
    if os.environ.get('SOAK_VERBOSE'):
        print(it, kind, method, dtype, batch, T, float(span), tol, flush=True)
    sign = -1.0 if rng.random() < 0.3 else 1.0
    try:
        a = odeint(f, y0, sign * t, method=method, options={'fusion': 'step', 'max_num_steps': 3000}, **tol)
        sa = dict(odeint.last_stats)
        b = odeint(f, y0, sign * t, method=method, options={'max_num_steps': 3000}, **tol)      # auto: whole-call kernel if eligible
        sb = dict(odeint.last_stats)
    except AssertionError as e:            # dt underflow etc.: must happen on both schedules alike
        try:
            odeint(f, y0, sign * t, method=method, options={'max_num_steps': 3000}, **tol)
            print('MISMATCH: only the step schedule raised', kind, method, batch, e)
            bad += 1
        except AssertionError:
            pass
        continue
    n_whole += int(sb['n_launches'] == 1)
    same = torch.equal(a, b) and sa['n_attempts'] == sb['n_attempts']
    if not same and kind in ('lorenz', 'lv', 'spiral') and batch > 65536 and sa['n_attempts'] == sb['n_attempts']:
        # beyond one trajectory per thread the whole-call schedule is the plane-streaming kernel on a grid of its own: the error
        # norm's partial sums fold in another order than the per-attempt launches' - same steps, values to roundoff
        same = float((a - b).abs().max()) <= 1e-10 * max(1.0, float(a.abs().max()))
    if not same:
        bad += 1
        print('MISMATCH', kind, method, dtype, batch, T, tol, sa, sb, float((a - b).abs().max()))
print('soak: %d runs (%d through a whole-call kernel), %d mismatches, %.1f s' % (n_runs, n_whole, bad, time.time() - t_start))
sys.exit(1 if bad else 0)

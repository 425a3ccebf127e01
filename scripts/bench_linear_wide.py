#!/usr/bin/env python3
"""Linear right-hand side beyond dim 128 (review of round 5, item 6): the 256-wide tile kernels with W streamed from L2
(csrc/mi_ode_step_fused.h, LinCtx<T, 256>) against the oracle, and their time at config 4's shape.
  python scripts/bench_linear_wide.py [parity] [bench] [valu]
parity: dims 129 / 200 / 256, float64 + float32, dopri5 / tsit5 / bosh3 / rk4 / euler, T = 2 and T = 7, batch 1000 (ragged last tile)
bench : batch 65536 x {256, 192, 144}, dopri5 rtol 1e-6 atol 1e-9, t = [0, 1] (config 4 at the wider state), ms per call and the
        fraction of the float64 / float32 matrix peak;  valu: the same call on the vector-ALU kernels (options linear_variant).
Times are MEDIANS of synchronised calls: a mean over a back-to-back loop caught a one-off ~45 ms pause in whichever case was running when
Python's cyclic collector made a full pass (3.0 -> 7.5 ms "per call" over ten calls; bench.py disables the collector around its timed region)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs  # noqa: E402
import oracle.ode_numpy as O  # noqa: E402

dev = torch.device('cuda:0')
what = set(sys.argv[1:]) or {'parity', 'bench'}


def system(D, batch, dtype=torch.float64, seed=2):
    g2 = torch.Generator().manual_seed(seed)
    S = torch.randn(D, D, generator=g2, dtype=torch.float64)
    A = -0.5 * torch.eye(D, dtype=torch.float64) + 0.5 * (S - S.t()) / np.sqrt(D)
    y0 = torch.randn(batch, D, generator=torch.Generator().manual_seed(seed + 1), dtype=torch.float64)
    return A.to(dtype), y0.to(dtype)


if 'parity' in what:
    worst = 0.0
    for D in (129, 200, 256):
        for dtype in (torch.float64, torch.float32):
            A, y0 = system(D, 1000, dtype)
            W = A.t().contiguous().numpy()
            b = (0.1 * torch.randn(D, generator=torch.Generator().manual_seed(9), dtype=torch.float64)).to(dtype)
            for method, tt, bias in (('dopri5', [0., 1.], None), ('dopri5', list(np.linspace(0., 2., 7)), b), ('tsit5', [0., 0.4, 1.], None),
                                     ('bosh3', [0., 1.], b), ('rk4', list(np.linspace(0., 1., 6)), None), ('euler', list(np.linspace(0., 1., 9)), b),
                                     ('dopri5', [1., 0.], None)):
                t = np.array(tt)
                kw = dict(rtol=1e-6, atol=1e-9) if dtype == torch.float64 else dict(rtol=1e-4, atol=1e-6)
                bn = None if bias is None else bias.numpy()
                fo = (lambda t_, y: y @ W) if bias is None else (lambda t_, y: y @ W + bn)
                ref, st_ref = O.odeint(fo, y0.numpy(), t.astype(y0.numpy().dtype), method=method, return_stats=True,
                                        options={'tsit5_fixed': True} if method == 'tsit5' else None, **kw)    # (the published tableau: SURVEY F6)
                f = rhs.Linear(A.t().contiguous(), bias)
                sol = odeint(f, y0.to(dev), torch.tensor(t), method=method, **kw)
                st = dict(odeint.last_stats)
                diff = float(np.abs(sol.cpu().numpy() - ref).max())
                tol = 1e-11 if dtype == torch.float64 else 2e-4
                same = method in ('rk4', 'euler') or (st.get('n_attempts') == st_ref.n_attempts and st.get('n_accepted') == st_ref.n_accepted)
                if dtype == torch.float32 and not same:       # float32 step sequences may differ by an attempt (bands, tests/bands.py)
                    same = abs(st.get('n_attempts') - st_ref.n_attempts) <= 1
                ok = diff < tol and same and st.get('n_launches') == 1
                worst = max(worst, diff if dtype == torch.float64 else 0.0)
                print('%s dim %3d %-8s %-6s T=%d bias=%d: max|diff| %.2e attempts %s/%s (oracle %s/%s) launches %s engine %s' % (
                    'ok  ' if ok else 'FAIL', D, str(dtype).split('.')[-1], method, len(t), bias is not None, diff, st.get('n_attempts'),
                    st.get('n_accepted'), getattr(st_ref, 'n_attempts', None), getattr(st_ref, 'n_accepted', None), st.get('n_launches'),
                    st.get('engine')), flush=True)
    print('float64 worst max|diff| %.2e' % worst)

if 'bench' in what or 'valu' in what:
    for D, dtype in ((256, torch.float64), (192, torch.float64), (144, torch.float64), (256, torch.float32), (160, torch.float32)):
        A, y0 = system(D, 65536, dtype)
        f = rhs.Linear.from_matrix(A)
        y = y0.to(dev)
        t = torch.tensor([0., 1.])
        for variant in (['tile'] if 'bench' in what else []) + (['valu'] if 'valu' in what else []):
            opts = {} if variant == 'tile' else {'linear_variant': 1}
            for _ in range(3):
                odeint(f, y, t, rtol=1e-6, atol=1e-9, method='dopri5', options=opts)
            torch.cuda.synchronize()
            reps = 10 if variant == 'tile' else 2
            per = []
            for _ in range(reps):
                t0 = time.perf_counter()
                odeint(f, y, t, rtol=1e-6, atol=1e-9, method='dopri5', options=opts)
                torch.cuda.synchronize()
                per.append(1e3 * (time.perf_counter() - t0))
            ms = float(np.median(per))
            st = dict(odeint.last_stats)
            nfe = st.get('nfe')
            flop_alg = 2.0 * D * D * 65536 * nfe
            vec = 2 if dtype == torch.float64 else 4
            kpad, npad = 16 * vec * -(-D // (16 * vec)), 16 * -(-D // 16)            # what the tile kernels execute: k up to the next 32 (64), columns up to the next 16
            flop = 2.0 * kpad * npad * 65536 * nfe if variant == 'tile' else flop_alg
            peak = 78.6e12 if dtype == torch.float64 else 157.3e12
            print(json.dumps({'case': 'linear b65536 d%d dopri5 %s %s' % (D, str(dtype).split('.')[-1], variant), 'ms_per_call': round(ms, 4),
                              'ms_min_max': [round(min(per), 3), round(max(per), 3)],
                              'attempts': st.get('n_attempts'), 'nfe': nfe, 'launches': st.get('n_launches'),
                              'TFLOPs_algorithmic': round(flop_alg / ms / 1e9, 2), 'frac_of_matrix_peak_algorithmic': round(flop_alg / (ms * 1e-3) / peak, 4),
                              'frac_executed': round(flop / (ms * 1e-3) / peak, 4), 'engine': st.get('engine')}), flush=True)

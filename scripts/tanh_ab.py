#!/usr/bin/env python3
"""A/B of the hidden activation's tanh (round 5): deviation of a fused-adjoint interval from the plane-kernel engine (torch.tanh) on the
cases of tests/test_gpu_adjoint_fused.py::test_time_dependent_segment_matches_the_plane_kernel_engine, and both against the same
interval in float64.  Run once per library (the variant: make EXTRA=-DMI_MLP_TANH_RATIONAL OUT=../_variants/libmi_ode_tanhrat.so in csrc/):
TFDIFFEQ_AMD_LIB=$PWD/tfdiffeq_amd/_variants/libmi_ode_tanhrat.so python scripts/tanh_ab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_adjoint_fused as T  # noqa: E402

print('library:', os.environ.get('TFDIFFEQ_AMD_LIB', 'product'))
for batch, dim, hidden, tol, t0, t1 in [(64, 8, 32, 1e-6, 1.0, 0.25), (64, 8, 32, 1e-4, 0.5, 2.0), (2000, 33, 70, 1e-5, 0.7, -0.4)]:
    func = T._func(dim, hidden, 43, time_dependent=True)
    g = torch.Generator(device='cpu').manual_seed(44)
    y = torch.randn(batch, dim, generator=g).to(T.dev())
    a = (torch.randn(batch, dim, generator=g) / batch).to(T.dev())
    eng = T._engine(batch, dim, hidden, tol, time_dependent=True)
    theta = (0.01 * torch.randn(eng.n_params, generator=g)).to(T.dev())
    adj_t = torch.tensor(0.3, device=T.dev())
    try:
        a1, t_1, p1 = eng.segment(func.device_rhs(), y, a, adj_t, theta, t0, t1)
        st = eng.stats.as_dict()
    finally:
        eng.close()
    ref, rs = T._plane_segment(func, y, a, adj_t, theta, t0, t1, tol)
    print('  case %s: accepted %d | fused vs plane engine: a %.2e theta %.2e' % ((batch, dim, hidden, tol), st['n_accepted'], T._rel(a1, ref[1][1]), T._rel(p1, ref[3][1])))
    try:                                                 # the same interval in float64 on the generic path: which float32 run is closer to it?
        f64 = T._func(dim, hidden, 43, time_dependent=True).double()
        ref64, _ = T._plane_segment(f64, y.double(), a.double(), adj_t.double(), theta.double(), t0, t1, tol)
        print('      vs float64 (same tolerance): fused a %.2e theta %.2e | plane engine a %.2e theta %.2e'
              % (T._rel(a1.double(), ref64[1][1]), T._rel(p1.double(), ref64[3][1]), T._rel(ref[1][1].double(), ref64[1][1]), T._rel(ref[3][1].double(), ref64[3][1])))
    except Exception as e:                               # noqa: BLE001
        print('      (float64 reference not available: %r)' % (e,))

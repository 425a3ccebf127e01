#!/usr/bin/env python3
"""Phase breakdown of the whole-integration kernel (library built with -DMI_PERSIST_PROF, see csrc/Makefile):
TFDIFFEQ_AMD_LIB=tfdiffeq_amd/libmi_ode_prof.so python scripts/persist_prof.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs  # noqa: E402

dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
A2 = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64)
cases = [
    ('spiral b1 dopri5', rhs.CubicLinear(A2), torch.tensor([[2., 0.]], dtype=torch.float64, device=dev), torch.linspace(0., 25., 1000, dtype=torch.float64), 'dopri5'),
    ('spiral b4096 dopri5', rhs.CubicLinear(A2), torch.tensor(rng.uniform(-2, 2, size=(4096, 2)), device=dev), torch.linspace(0., 25., 10, dtype=torch.float64), 'dopri5'),
    ('lorenz b65536 tsit5', rhs.Lorenz(), torch.tensor(np.array([1., 1., 1.]) + 1e-3 * rng.standard_normal((65536, 3)), device=dev), torch.tensor([0., 1.]), 'tsit5'),
    ('lorenz b65536 dopri5', rhs.Lorenz(), torch.tensor(np.array([1., 1., 1.]) + 1e-3 * rng.standard_normal((65536, 3)), device=dev), torch.tensor([0., 1.]), 'dopri5'),
]
for name, f, y0, t, method in cases:
    for _ in range(3):
        sys.stderr.write(name + ': ')
        sys.stderr.flush()
        odeint(f, y0, t, method=method, rtol=1e-6, atol=1e-9, options={'fusion': 'whole'})

"""Test helpers: build the product-side RHS (DeviceRHS or a torch callable) for a golden fixture."""
import math

import numpy as np
import torch

from tfdiffeq_amd import rhs as R


def device_rhs(name, params, weights=None):
    """The fused-kernel RHS for a fixture, or None if the catalogue has no fused kernel for it."""
    if name == 'cubic_linear':
        return R.CubicLinear(torch.tensor(params['W'], dtype=torch.float64))
    if name == 'linear':
        return R.Linear(torch.tensor(params['W'], dtype=torch.float64))
    if name == 'lotka_volterra':
        return R.LotkaVolterra(params['a'], params['b'], params['c'], params['d'])
    if name == 'lorenz':
        return R.Lorenz(params['sigma'], params['beta'], params['rho'])
    if name == 'mlp_tanh':
        w = weights
        return R.MLPTanh(torch.tensor(w['W1']), torch.tensor(w['b1']), torch.tensor(w['W2']), torch.tensor(w['b2']),
                         torch.tensor(w['W3']), torch.tensor(w['b3']))
    return None


def torch_rhs(name, params, weights=None):
    """Plain Python callables written with torch ops (they go through the plane-kernel path)."""
    if name == 'sine':          # tests/problems.py:28-34
        return lambda t, y: 2 * y / t + t ** 4 * torch.sin(2 * t) - t ** 2 + 4 * t ** 3
    if name == 'constant':      # tests/problems.py:13-21
        a, b = params.get('a', 0.2), params.get('b', 3.0)
        return lambda t, y: a + (y - (a * t + b)) ** 5
    if name == 'tdep':
        return lambda t, y: torch.sin(y) * t - 0.5 * y + torch.cos(t)
    d = device_rhs(name, params, weights)
    if d is None:
        raise KeyError(name)
    return lambda t, y: d.forward(t, y)      # same math, but opaque to the solver (no device_rhs attribute)


def sine_exact(t):
    return (-0.5 * t ** 4 * np.cos(2 * t) + 0.5 * t ** 3 * np.sin(2 * t) + 0.25 * t ** 2 * np.cos(2 * t)
            - t ** 3 + 2 * t ** 4 + (math.pi - 0.25) * t ** 2)

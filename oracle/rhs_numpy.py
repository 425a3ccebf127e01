"""ORACLE - numpy right-hand sides, by the names the golden fixtures record (meta['rhs']).

Workload definitions follow the reference's examples/tests:
  sine, constant   tests/problems.py:13-40
  cubic_linear     examples/ode_demo.py:33-35          f = (y**3) @ W
  linear           config 4            f = y @ W   (W = A^T)
  lotka_volterra   examples/ode_usage.ipynb cells 39-42, batched on the last axis
  lorenz           examples/lorenz_attractor.py:20-37,  batched on the last axis
  mlp_tanh         tfdiffeq/models/dense_odenet.py:41-92 (time independent, tanh)
  tdep             elementwise, time dependent (function-level vectors only)
Test infrastructure: not imported by the product package.
"""
import numpy as np


def make_rhs(name, params=None, dtype=np.float64, weights=None):
    p = params or {}
    dt = np.dtype(dtype)
    # scalar states: evaluate through 0-d ARRAYS (np.asarray), not numpy scalars - numpy's scalar `**` goes through libm
    # pow while the array path (which the fixture generator's stand-in takes) may use a SIMD pow: 1-ulp differences
    # that the multistep divided differences amplify into different step sequences
    if name == 'sine':
        def sine(t, y):
            t, y = np.asarray(t), np.asarray(y)
            return 2 * y / t + t ** 4 * np.sin(2 * t) - t ** 2 + 4 * t ** 3
        return sine
    if name == 'constant':
        a, b = p.get('a', 0.2), p.get('b', 3.0)

        def constant(t, y):
            t, y = np.asarray(t), np.asarray(y)
            return a + (y - (a * t + b)) ** 5
        return constant
    if name == 'cubic_linear':
        W = np.asarray(p['W'], dtype=dt)
        return lambda t, y: (y ** 3) @ W
    if name == 'linear':
        W = np.asarray(p['W'], dtype=dt)
        return lambda t, y: y @ W
    if name == 'lotka_volterra':
        a, b, c, d = p['a'], p['b'], p['c'], p['d']

        def lv(t, y):
            u, v = y[..., 0], y[..., 1]
            return np.stack([a * u - b * u * v, -c * v + d * u * v], axis=-1)
        return lv
    if name == 'lorenz':
        s, be, r = p['sigma'], p['beta'], p['rho']

        def lorenz(t, y):
            x0, x1, x2 = y[..., 0], y[..., 1], y[..., 2]
            return np.stack([s * (x1 - x0), x0 * (r - x2) - x1, x0 * x1 - be * x2], axis=-1)
        return lorenz
    if name == 'mlp_tanh':
        W1, b1, W2, b2, W3, b3 = [np.asarray(weights[k], dtype=dt) for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3')]

        def mlp(t, y):
            h = np.tanh(y @ W1 + b1)
            h = np.tanh(h @ W2 + b2)
            return h @ W3 + b3
        return mlp
    if name == 'tdep':
        return lambda t, y: np.sin(y) * t - 0.5 * y + np.cos(t)
    raise KeyError(name)

"""ORACLE / test infrastructure - the 25 DETEST non-stiff problems the reference ships as its own known-problem set
(tests/DETEST/detest.py:9-351; harness tests/DETEST/run.py:25-60: every problem integrated from t = 0 to t = 20).

Restated from the published problem definitions (Hull, Enright, Fellen & Sedgwick, "Comparing numerical methods for
ordinary differential equations", SIAM J. Numer. Anal. 9 (1972), classes A-E) for ANY array namespace: pass numpy to get
the oracle's right-hand sides, torch to get the callables the product's plane-kernel engine integrates.  The reference's
two departures from the paper are kept so that its own results (tests/golden/fn_detest.npz) pin this file:
  * C5 (five outer planets): the y-coordinate of Neptune's initial position reads 165699966404 in the reference
    (detest.py:241 - the decimal point of 1.65699966404 is missing);
  * C5's state is a [2, 3, 5] tensor (positions; velocities), the only non-vector state of the set.
Nothing here is imported by the product package.
"""
import math

import numpy as np

NAMES = [c + i for c in 'ABCDE' for i in '12345']
T_END = 20.0


class _Ops(object):
    """The handful of array functions the problems need, for numpy or torch."""

    def __init__(self, xp, like=None):
        self.xp = xp
        self.is_torch = xp.__name__ == 'torch'
        self.like = like          # torch: a tensor whose device the constants should live on

    def const(self, a):
        if self.is_torch:
            kw = {'dtype': self.xp.float64}
            if self.like is not None:
                kw['device'] = self.like.device
            return self.xp.tensor(np.asarray(a, dtype=np.float64), **kw)
        return np.asarray(a, dtype=np.float64)

    def stack(self, xs, axis=0):
        return self.xp.stack(xs, axis) if not self.is_torch else self.xp.stack(xs, dim=axis)

    def sum(self, x, axis):
        return x.sum(axis)


def _band(n, diag, lower, upper=None):
    """n x n matrix with `diag` on the diagonal, `lower` on the first sub-diagonal (and `upper` on the super-diagonal)."""
    A = np.zeros((n, n))
    A[np.arange(n), np.arange(n)] = diag
    A[np.arange(1, n), np.arange(n - 1)] = lower
    if upper is not None:
        A[np.arange(n - 1), np.arange(1, n)] = upper
    return A


def problem(name, xp=np, like=None):
    """(f(t, y), y0) of DETEST problem `name` in the array namespace `xp`; t0 = 0, integrate to T_END."""
    o = _Ops(xp, like)
    c = o.const
    sqrt, sin, cos = xp.sqrt, xp.sin, xp.cos
    if name == 'A1':                                           # y' = -y
        return (lambda t, y: -y), c(1.)
    if name == 'A2':                                           # y' = -y^3 / 2
        return (lambda t, y: -y ** 3 / 2), c(1.)
    if name == 'A3':                                           # y' = y cos t
        return (lambda t, y: y * cos(t)), c(1.)
    if name == 'A4':                                           # logistic curve
        return (lambda t, y: y / 4 * (1 - y / 20)), c(1.)
    if name == 'A5':                                           # spiral curve
        return (lambda t, y: (y - t) / (y + t)), c(4.)
    if name == 'B1':                                           # growth of two conflicting populations
        return (lambda t, y: o.stack([2 * (y[0] - y[0] * y[1]), -(y[1] - y[0] * y[1])])), c([1., 3.])
    if name == 'B2':                                           # linear chemical reaction
        A = c([[-1., 1., 0.], [1., -2., 1.], [0., 1., -1.]])
        return (lambda t, y: (A @ y[..., None])[..., 0]), c([2., 0., 1.])
    if name == 'B3':                                           # non-linear chemical reaction
        return (lambda t, y: o.stack([-y[0], y[0] - y[1] * y[1], y[1] * y[1]])), c([1., 0., 0.])
    if name == 'B4':                                           # integral surface of a torus
        def f(t, y):
            a = sqrt(y[0] * y[0] + y[1] * y[1])
            return o.stack([-y[1] - y[0] * y[2] / a, y[0] - y[1] * y[2] / a, y[0] / a])
        return f, c([3., 0., 0.])
    if name == 'B5':                                           # Euler equations of a rigid body
        return (lambda t, y: o.stack([y[1] * y[2], -y[0] * y[2], -0.51 * y[0] * y[1]])), c([0., 1., 1.])
    if name in ('C1', 'C2', 'C3', 'C4'):
        if name == 'C1':                                       # radioactive decay chain
            A, n = _band(10, [-1.] * 9 + [0.], 1.), 10
        elif name == 'C2':                                     # ... with growing rates
            A, n = _band(10, list(np.linspace(-1., -9., 9)) + [0.], np.linspace(1., 9., 9)), 10
        else:                                                  # parabolic PDE, semi-discretised (10 / 51 points)
            n = 10 if name == 'C3' else 51
            A = _band(n, -2., 1., 1.)
        Ac = c(A)
        y0 = np.zeros(n)
        y0[0] = 1.
        return (lambda t, y: (Ac @ y[..., None])[..., 0]), c(y0)
    if name == 'C5':                                           # five-body problem: the outer planets
        k2, m0 = 2.95912208286, 1.00000597682
        m = c([0.000954786104043, 0.000285583733151, 0.0000437273164546, 0.0000517759138449, 0.00000277777777778])
        off_diag = c(1.0 - np.eye(5))                          # the self-interaction terms are dropped (detest.py:226-230)

        def f(t, y):
            dy, p = y[1], y[0]                                 # p: [3, 5] positions, dy: velocities
            r = sqrt(o.sum(p ** 2, 0)).reshape(1, 5)
            d = sqrt(o.sum((p[:, :, None] - p[:, None, :]) ** 2, 0))
            F = m.reshape(1, 1, 5) * ((p[:, None, :] - p[:, :, None]) / (d * d * d).reshape(1, 5, 5) +
                                      p.reshape(3, 1, 5) / (r * r * r).reshape(1, 1, 5))
            if o.is_torch:
                F = xp.where(off_diag.reshape(1, 5, 5) > 0, F, xp.zeros_like(F))    # inf/nan on the diagonal: replaced, as the
            else:                                                                    # reference's `F[:, ::6] = 0` does
                F = np.where(off_diag.reshape(1, 5, 5) > 0, F, 0.0)
            ddy = k2 * (-(m0 + m.reshape(1, 5)) * p / (r * r * r)) + o.sum(F, 2)
            return o.stack([dy, ddy], 0)
        pos = np.array([3.42947415189, 3.35386959711, 1.35494901715, 6.64145542550, 5.97156957878, 2.18231499728, 11.2630437207,
                        14.6952576794, 6.27960525067, -30.1552268759, 165699966404, 1.43785752721, -21.1238353380, 28.4465098142,
                        15.388265967]).reshape(5, 3).T
        vel = np.array([-.557160570446, .505696783289, .230578543901, -.415570776342, .365682722812, .169143213293, -.325325669158,
                        .189706021964, .0877265322780, -.0240476254170, -.287659532608, -.117219543175, -.176860753121,
                        -.216393453025, -.0148647893090]).reshape(5, 3).T
        return f, c(np.stack([pos, vel], 0))
    if name[0] == 'D':                                         # two-body orbits, eccentricity 0.1 .. 0.9
        eps = {'D1': .1, 'D2': .3, 'D3': .5, 'D4': .7, 'D5': .9}[name]

        def f(t, y):
            r = (y[0] ** 2 + y[1] ** 2) ** (3 / 2)
            return o.stack([y[2], y[3], -y[0] / r, -y[1] / r])
        return f, c([1 - eps, 0., 0., math.sqrt((1 + eps) / (1 - eps))])
    if name == 'E1':                                           # Bessel's equation of order 1/2
        return (lambda t, y: o.stack([y[1], -(y[1] / (t + 1) + (1 - 0.25 / (t + 1) ** 2) * y[0])])), \
            c([.671396707141803, .0954005144474744])
    if name == 'E2':                                           # Van der Pol
        return (lambda t, y: o.stack([y[1], (1 - y[0] ** 2) * y[1] - y[0]])), c([2., 0.])
    if name == 'E3':                                           # Duffing
        return (lambda t, y: o.stack([y[1], y[0] ** 3 / 6 - y[0] + 2 * sin(2.78535 * t)])), c([0., 0.])
    if name == 'E4':                                           # falling body
        return (lambda t, y: o.stack([y[1], .32 - .4 * y[1] ** 2])), c([30., 0.])
    if name == 'E5':                                           # pursuit curve
        return (lambda t, y: o.stack([y[1], sqrt(1 + y[1] ** 2) / (25 - t)])), c([0., 0.])
    raise KeyError(name)


# closed forms the reference records for class A (detest.py:12, 19, 26, 33)
EXACT = {'A1': lambda t: np.exp(-t), 'A2': lambda t: 1 / np.sqrt(t + 1), 'A3': lambda t: np.exp(np.sin(t)),
         'A4': lambda t: 20 / (1 + 19 * np.exp(-t / 4))}

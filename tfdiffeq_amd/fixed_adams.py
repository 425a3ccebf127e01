"""Mirror of tfdiffeq/fixed_adams.py: fixed-grid Adams-Bashforth ('explicit_adams') and Adams-Bashforth-Moulton
('fixed_adams') with RK4 (3/8 rule) start-up and functional iteration for the corrector.

The integer coefficient tables are data (tableaus/adams.json).  Every state-sized operation is a plane kernel
(`mi_ode_lincomb`, `mi_ode_not_converged`); f is the caller's callable.
"""
import collections
import json
import os
import sys

from .misc import _has_converged, _lincomb, _np_dtype, _scalar_tensor
from .rk_common import rk4_alt_step_func
from .solvers import FixedGridODESolver

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tableaus', 'adams.json')) as _fh:
    _TAB = json.load(_fh)
_BASHFORTH_COEFFICIENTS, _MOULTON_COEFFICIENTS, _DIVISOR = _TAB['bashforth'], _TAB['moulton'], _TAB['divisor']
_MIN_ORDER, _MAX_ORDER, _MAX_ITERS = _TAB['min_order'], _TAB['max_order'], _TAB['max_iters']     # fixed_adams.py:88-90


class AdamsBashforthMoulton(FixedGridODESolver):
    """fixed_adams.py:152-207.  NB: odeint passes ITS rtol/atol (1e-7/1e-9 by default) as the convergence test."""

    def __init__(self, func, y0, rtol=1e-3, atol=1e-4, implicit=True, max_iters=_MAX_ITERS, max_order=_MAX_ORDER, **kwargs):
        super(AdamsBashforthMoulton, self).__init__(func, y0, **kwargs)
        self.rtol = rtol
        self.atol = atol
        self.implicit = implicit
        self.max_iters = max_iters
        self.max_order = int(min(max_order, _MAX_ORDER))
        self.prev_f = collections.deque(maxlen=self.max_order - 1)
        self.prev_t = None

    def _fused_multistep(self):
        """The descriptor of the one-launch kernel (csrc/mi_ode_adams.h): the integer tables as the products the reference forms in
        Python floats - (1 / divisor) * c_j for the predictor and for the corrector's delta, m_0 / divisor for its leading term."""
        ab, am, am0 = [0.0] * (13 * 12), [0.0] * (13 * 12), [0.0] * 13
        for o in range(1, 13):
            if o < len(_BASHFORTH_COEFFICIENTS) and _DIVISOR[o] is not None:
                for j, c in enumerate(_BASHFORTH_COEFFICIENTS[o][:12]):
                    ab[o * 12 + j] = (1 / _DIVISOR[o]) * c
            if o + 1 < len(_MOULTON_COEFFICIENTS) and _DIVISOR[o + 1] is not None:
                mc = _MOULTON_COEFFICIENTS[o + 1]
                for j, c in enumerate(mc[1:13]):
                    am[o * 12 + j] = (1 / _DIVISOR[o + 1]) * c
                am0[o] = mc[0] / _DIVISOR[o + 1]
        return (2 if self.implicit else 1, self.max_order, self.max_iters, _MIN_ORDER, tuple(ab), tuple(am), tuple(am0))

    def _update_history(self, t, f):
        if self.prev_t is None or self.prev_t != t:
            self.prev_f.appendleft(f)
            self.prev_t = t

    def step_func(self, func, t, dt, y):
        like = y[0]
        dt_ = _np_dtype(like.dtype).type
        t, dt = dt_(t), dt_(dt)
        self._update_history(t, func(_scalar_tensor(t, like), y))
        order = min(len(self.prev_f), self.max_order - 1)
        if order < _MIN_ORDER - 1:
            return rk4_alt_step_func(func, t, dt, y, k1=self.prev_f[0])                  # :176-179
        # Adams-Bashforth predictor (:182-184): dt * add_n((1/div * c_j) * f_j)
        coeffs, div = _BASHFORTH_COEFFICIENTS[order], _DIVISOR[order]
        hist = tuple(zip(*self.prev_f))
        dy = tuple(_lincomb(None, [1.0], [_lincomb(None, coeffs, f_, 1 / div)], dt) for f_ in hist)
        if self.implicit:                                                                # Adams-Moulton corrector (:187-201)
            mc, mdiv = _MOULTON_COEFFICIENTS[order + 1], _DIVISOR[order + 1]
            delta = tuple(_lincomb(None, [1.0], [_lincomb(None, mc[1:], f_, 1 / mdiv)], dt) for f_ in hist)
            converged = False
            f = None
            for _ in range(self.max_iters):
                dy_old = dy
                f = func(_scalar_tensor(t + dt, like), tuple(_lincomb(y_, [1.0], [dy_], 1.0) for y_, dy_ in zip(y, dy)))
                dy = tuple(_lincomb(delta_, [1.0], [f_], dt * dt_(mc[0] / mdiv)) for f_, delta_ in zip(f, delta))
                converged = _has_converged(dy_old, dy, self.rtol, self.atol)
                if converged:
                    break
            if not converged:
                print('Warning: Functional iteration did not converge. Solution may be incorrect.', file=sys.stderr)
                self.prev_f.pop()
            self._update_history(t, f)
        return dy

    @property
    def order(self):
        return 4


class AdamsBashforth(AdamsBashforthMoulton):
    """fixed_adams.py:209-212."""

    def __init__(self, func, y0, **kwargs):
        super(AdamsBashforth, self).__init__(func, y0, implicit=False, **kwargs)

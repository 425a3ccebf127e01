"""Mirror of tfdiffeq/dopri8.py: Prince-Dormand 8(7), 13 stages.

Runs through the plane-kernel engine (`rk_common._runge_kutta_step` + `mi_ode_lincomb` with 14 planes); the fused
engine is limited to 6-row tableaus.  The coefficients are data: float64 values in tableaus/dopri8.json.
"""
import json
import os

from . import _native as N
from .rk_common import _ButcherTableau
from .solvers import _AdaptiveRKSolver


def _load():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tableaus', 'dopri8.json')) as fh:
        d = json.load(fh)
    alpha = d['dopri8_alpha']
    beta = [d['dopri8_beta'][i][:i + 1] for i in range(len(alpha))]
    return _ButcherTableau(alpha=alpha, beta=beta, c_sol=d['dopri8_c_sol'], c_error=d['dopri8_c_error']), d['dopri8_c_mid']


_DOPRI8_TABLEAU, c_mid = _load()        # dopri8.py:12-77


class Dopri8Solver(_AdaptiveRKSolver):
    """dopri8.py:97-169: initial-step order 7, controller order 8."""
    c_mid = c_mid
    order = 8
    init_order = 7
    controller = N.CTRL_MISC
    interp = N.INTERP_QUARTIC_MID
    tableau = _DOPRI8_TABLEAU

    def __init__(self, func, y0, rtol, atol, first_step=None, safety=0.9, ifactor=10.0, dfactor=0.2,
                 max_num_steps=2 ** 31 - 1, **unused_kwargs):
        self._setup(func, y0, rtol, atol, first_step, safety, ifactor, dfactor, max_num_steps, unused_kwargs)

    _adaptive_dopri8_step = _AdaptiveRKSolver._adaptive_step

"""Mirror of tfdiffeq/adaptive_huen.py (sic): adaptive Heun, a 2-stage embedded pair.

Not FSAL-shaped: y1 comes from c_sol (rk_common.py:54-56).  The reference hands order=5 to the step-size
controller (adaptive_huen.py:111-113) and order 1 to the initial-step heuristic; both are kept.
Row-local systems (catalogue, plugins, traced callables) and the cooperative kernels run it in one launch (k_persist_rowlocal<T, 1, ..>);
any other callable on the device-controlled engine.
"""
from . import _native as N
from .rk_common import _ButcherTableau
from .solvers import _AdaptiveRKSolver

# adaptive_huen.py:11-25
_ADAPTIVE_HEUN_TABLEAU = _ButcherTableau(alpha=[1.], beta=[[1.]], c_sol=[0.5, 0.5], c_error=[0.5, -0.5])
AH_C_MID = [0.5, 0.]


class AdaptiveHeunSolver(_AdaptiveRKSolver):
    c_mid = AH_C_MID
    order = 5
    init_order = 1
    controller = N.CTRL_MISC
    interp = N.INTERP_QUARTIC_MID
    tableau = _ADAPTIVE_HEUN_TABLEAU

    def __init__(self, func, y0, rtol, atol, first_step=None, safety=0.9, ifactor=10.0, dfactor=0.2,
                 max_num_steps=2 ** 31 - 1, **unused_kwargs):
        self._setup(func, y0, rtol, atol, first_step, safety, ifactor, dfactor, max_num_steps, unused_kwargs)

    _adaptive_heun_step = _AdaptiveRKSolver._adaptive_step

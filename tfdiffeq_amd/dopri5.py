"""Mirror of tfdiffeq/dopri5.py: Dormand-Prince 5(4), the default method."""
from . import _native as N
from .rk_common import _ButcherTableau
from .solvers import _AdaptiveRKSolver

# dopri5.py:11-30 (FSAL; c_error = b - b_hat)
_DORMAND_PRINCE_SHAMPINE_TABLEAU = _ButcherTableau(
    alpha=[1 / 5, 3 / 10, 4 / 5, 8 / 9, 1., 1.],
    beta=[
        [1 / 5],
        [3 / 40, 9 / 40],
        [44 / 45, -56 / 15, 32 / 9],
        [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
        [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
        [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
    ],
    c_sol=[35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0],
    c_error=[
        35 / 384 - 1951 / 21600,
        0,
        500 / 1113 - 22642 / 50085,
        125 / 192 - 451 / 720,
        -2187 / 6784 - -12231 / 42400,
        11 / 84 - 649 / 6300,
        -1. / 60.,
    ],
)

# dopri5.py:33-36: mid-point weights of the dense output
DPS_C_MID = [
    6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
    187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2
]


class Dopri5Solver(_AdaptiveRKSolver):
    """dopri5.py:48-121.  Options: first_step, safety, ifactor, dfactor, max_num_steps, tableau."""
    c_mid = DPS_C_MID
    order = 5
    init_order = 4
    controller = N.CTRL_MISC
    interp = N.INTERP_QUARTIC_MID

    def __init__(self, func, y0, rtol, atol, first_step=None, safety=0.9, ifactor=10.0, dfactor=0.2,
                 max_num_steps=2 ** 31 - 1, tableau=None, **unused_kwargs):
        self._setup(func, y0, rtol, atol, first_step, safety, ifactor, dfactor, max_num_steps, unused_kwargs)
        self.tableau = tableau if tableau is not None else _DORMAND_PRINCE_SHAMPINE_TABLEAU   # dopri5.py:67

    _adaptive_dopri5_step = _AdaptiveRKSolver._adaptive_step

"""Mirror of tfdiffeq/bosh3.py: Bogacki-Shampine 3(2).

PARITY NOTE: the reference tableau carries two typos - alpha[0] = 1./.2 = 5.0 (should be 1/2)
and beta[1] = [0., 3./.4] = [0, 7.5] (should be 3/4).  `method='bosh3'` reproduces them VERBATIM, because
that is what the reference computes (it still converges, at ~25x the NFE of dopri5).  The textbook tableau is
offered separately as `_BOGACKI_SHAMPINE_TABLEAU_TEXTBOOK` / options={'textbook_tableau': True}.
"""
from . import _native as N
from .rk_common import _ButcherTableau
from .solvers import _AdaptiveRKSolver

# bosh3.py:10-20, verbatim
_BOGACKI_SHAMPINE_TABLEAU = _ButcherTableau(
    alpha=[1. / .2, 3. / 4., 1.],
    beta=[
        [1. / 2.],
        [0., 3. / .4],
        [2. / 9., 1. / 3., 4. / 9.]
    ],
    c_sol=[2. / 9., 1. / 3., 4. / 9., 0.],
    c_error=[2. / 9. - 7. / 24., 1. / 3. - 1. / 4., 4. / 9. - 1. / 3., -1. / 8.],
)

_BOGACKI_SHAMPINE_TABLEAU_TEXTBOOK = _ButcherTableau(
    alpha=[1. / 2., 3. / 4., 1.],
    beta=[[1. / 2.], [0., 3. / 4.], [2. / 9., 1. / 3., 4. / 9.]],
    c_sol=[2. / 9., 1. / 3., 4. / 9., 0.],
    c_error=[2. / 9. - 7. / 24., 1. / 3. - 1. / 4., 4. / 9. - 1. / 3., -1. / 8.],
)

BS_C_MID = [0., 0.5, 0., 0.]          # bosh3.py:22


class Bosh3Solver(_AdaptiveRKSolver):
    """bosh3.py:34-99: initial-step order 2, controller order 3."""
    c_mid = BS_C_MID
    order = 3
    init_order = 2
    controller = N.CTRL_MISC
    interp = N.INTERP_QUARTIC_MID

    def __init__(self, func, y0, rtol, atol, first_step=None, safety=0.9, ifactor=10.0, dfactor=0.2,
                 max_num_steps=2 ** 31 - 1, textbook_tableau=False, **unused_kwargs):
        self._setup(func, y0, rtol, atol, first_step, safety, ifactor, dfactor, max_num_steps, unused_kwargs)
        self.tableau = _BOGACKI_SHAMPINE_TABLEAU_TEXTBOOK if textbook_tableau else _BOGACKI_SHAMPINE_TABLEAU

    _adaptive_bosh3_step = _AdaptiveRKSolver._adaptive_step

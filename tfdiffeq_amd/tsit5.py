"""Mirror of tfdiffeq/tsit5.py: Tsitouras 5(4).

PARITY NOTE.  The reference's tsit5 is defective twice:
  (a) its c_error is `b - x` where the x values already ARE Tsitouras' error weights, so sum(c_error) = 0.97
      instead of 0: the error estimate is O(dt), dt collapses to ~tol/|f| and never recovers;
  (b) its dense output starts from k[0] (= f0) instead of y0 (tsit5.py:47).
It has no test in the reference and is not listed in its README.  Following SURVEY.md's plan:
  * `method='tsit5'` uses the published coefficients (error weights that sum to 0, dense output from y0) -
    validated against high-accuracy solutions, usable on real horizons; its step-size controller and
    error norm are still the reference's own (tsit5.py:53-62, 126-138: no sqrt, pooled mean, scalar tol);
  * `options={'refcompat': True}` reproduces the reference verbatim, defects included, for step-sequence
    parity on the short horizons the reference can actually finish.
"""
from . import _native as N
from .rk_common import _ButcherTableau
from .solvers import _AdaptiveRKSolver

_B = [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774]
_X = [0.001780011052226, 0.000816434459657, -0.007880878010262, 0.144711007173263, -0.582357165452555,
      0.458082105929187]
_ALPHA = [0.161, 0.327, 0.9, 0.9800255409045097, 1., 1.]
_BETA = [
    [0.161],
    [-0.008480655492357, 0.3354806554923570],
    [2.897153057105494, -6.359448489975075, 4.362295432869581],
    [5.32586482843925895, -11.74888356406283, 7.495539342889836, -0.09249506636175525],
    [5.86145544294642038, -12.92096931784711, 8.159367898576159, -0.071584973281401006, -0.02826905039406838],
    list(_B),
]

# tsit5.py:10-30 verbatim (defect F6a kept)
_TSITOURAS_TABLEAU = _ButcherTableau(alpha=_ALPHA, beta=_BETA, c_sol=_B + [0.],
                                     c_error=[b - x for b, x in zip(_B, _X)] + [-1. / 66.])
# Tsitouras (2011) error weights b - b_hat (sum = 0)
_TSITOURAS_TABLEAU_PUBLISHED = _ButcherTableau(alpha=_ALPHA, beta=_BETA, c_sol=_B + [0.],
                                               c_error=list(_X) + [-1. / 66.])


class Tsit5Solver(_AdaptiveRKSolver):
    """tsit5.py:69-151."""
    c_mid = None
    order = 5
    init_order = 4
    controller = N.CTRL_TSIT5
    pooled_ratio = True

    def __init__(self, func, y0, rtol, atol, first_step=None, safety=0.9, ifactor=10.0, dfactor=0.2,
                 max_num_steps=2 ** 31 - 1, refcompat=False, **unused_kwargs):
        self._setup(func, y0, rtol, atol, first_step, safety, ifactor, dfactor, max_num_steps, unused_kwargs)
        self.refcompat = bool(refcompat)
        self.tableau = _TSITOURAS_TABLEAU if self.refcompat else _TSITOURAS_TABLEAU_PUBLISHED
        self.interp = N.INTERP_TSIT5_REF if self.refcompat else N.INTERP_TSIT5

    _adaptive_tsit5_step = _AdaptiveRKSolver._adaptive_step

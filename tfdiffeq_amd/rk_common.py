"""Mirror of tfdiffeq/rk_common.py: the Runge-Kutta step sub-contract (B3 in SURVEY.md 8(b)).

`_runge_kutta_step(func, y0, f0, t0, dt, tableau) -> (y1, f1, y1_error, k)` has the reference's
signature and return structure; every state-sized operation is one fused plane-kernel launch
(libmi_ode `mi_ode_lincomb`), `func` is whatever Python callable the caller supplies.
The fully fused engine (device RHS inside the stage kernels) lives behind `solvers._FusedEngine`.
"""
import collections

import torch

from .misc import _lincomb, _np_dtype, _scalar_tensor

_ButcherTableau = collections.namedtuple('_ButcherTableau', 'alpha beta c_sol c_error')     # rk_common.py:5


class _RungeKuttaState(collections.namedtuple('_RungeKuttaState', 'y1, f1, t0, t1, dt, interp_coeff')):
    """Saved state of the Runge Kutta solver (rk_common.py:8-19): y1/f1 tuples of device tensors,
    t0/t1/dt host float64 scalars, interp_coeff whatever the solver needs for dense output."""


def _is_fsal_shaped(tableau):
    """rk_common.py:54."""
    return tableau.c_sol[-1] == 0 and list(tableau.c_sol[:-1]) == list(tableau.beta[-1])


def _runge_kutta_step(func, y0, f0, t0, dt, tableau):
    """Take an arbitrary Runge-Kutta step and estimate error (rk_common.py:22-61).

    t0 / dt are host scalars, or 0-d float64 DEVICE tensors: then nothing of the step depends on host values and the
    whole step can be captured in a hipGraph and replayed with new t0 / dt (graph_step.py)."""
    like = y0[0]
    on_device = isinstance(dt, torch.Tensor)
    if on_device:
        t0_s, dt_s = t0.to(like.dtype), dt.to(like.dtype)          # :45-46 (casts to the state dtype, on device)
    else:
        dt_ = _np_dtype(like.dtype).type
        t0 = dt_(t0)                                               # :45
        dt = dt_(dt)                                               # :46
    k = tuple([f0_] for f0_ in f0)
    yi = None
    for alpha_i, beta_i in zip(tableau.alpha, tableau.beta):
        if on_device:
            ti = t0_s + float(alpha_i) * dt_s                      # :50, a 0-d tensor of the state dtype
        else:
            ti = _scalar_tensor(t0 + dt_(alpha_i) * dt, like)      # :50
        yi = tuple(_lincomb(y0_, beta_i, k_, dt) for y0_, k_ in zip(y0, k))    # :51 (one kernel per component)
        for k_, f_ in zip(k, func(ti, yi)):
            k_.append(f_)
    if not _is_fsal_shaped(tableau):                               # :54-56
        yi = tuple(_lincomb(y0_, tableau.c_sol, k_, dt) for y0_, k_ in zip(y0, k))
    y1 = yi
    f1 = tuple(k_[-1] for k_ in k)
    y1_error = tuple(_lincomb(None, tableau.c_error, k_, dt) for k_ in k)      # :60
    return (y1, f1, y1_error, k)


def _time_pair(t, dt, like):
    """(t, dt) in the state dtype for the stage times, plus dt as the plane kernels take it: host scalars stay host
    scalars; 0-d float64 DEVICE tensors (graph capture) stay on the device."""
    if isinstance(dt, torch.Tensor):
        t = t if isinstance(t, torch.Tensor) else torch.full((), float(t), dtype=torch.float64, device=dt.device)
        return t.to(like.dtype), dt.to(like.dtype), dt
    dt_ = _np_dtype(like.dtype).type
    return dt_(t), dt_(dt), dt_(dt)


def _div(x, c):
    """x / c with a true IEEE division also for device scalars (torch multiplies a tensor by 1/c when c is a Python
    number, which is one ulp off for c = 3, 6)."""
    if isinstance(x, torch.Tensor):
        return torch.div(x, torch.full((), float(c), dtype=x.dtype, device=x.device))
    return x / c


def _time_arg(value, like):
    """What func receives as t: a 0-d device tensor in the state dtype."""
    return value if isinstance(value, torch.Tensor) else _scalar_tensor(value, like)


def rk4_step_func(func, t, dt, y, k1=None):
    """Classical RK4 (rk_common.py:64-70; not used by the 'rk4' method, which is the 3/8 rule)."""
    like = y[0]
    t, dt, h = _time_pair(t, dt, like)
    if k1 is None:
        k1 = func(_time_arg(t, like), y)
    k2 = func(_time_arg(t + _div(dt, 2), like), tuple(_lincomb(y_, [0.5], [k1_], h) for y_, k1_ in zip(y, k1)))
    k3 = func(_time_arg(t + _div(dt, 2), like), tuple(_lincomb(y_, [0.5], [k2_], h) for y_, k2_ in zip(y, k2)))
    k4 = func(_time_arg(t + dt, like), tuple(_lincomb(y_, [1.0], [k3_], h) for y_, k3_ in zip(y, k3)))
    return tuple(_lincomb(None, [1.0, 2.0, 2.0, 1.0], [a, b, c, d], _div(h, 6)) for a, b, c, d in zip(k1, k2, k3, k4))


def rk4_alt_step_func(func, t, dt, y, k1=None):
    """3/8-rule RK4, "smaller error with slightly more compute" (rk_common.py:73-81).  Returns dy."""
    like = y[0]
    t, dt, h = _time_pair(t, dt, like)
    if k1 is None:
        k1 = func(_time_arg(t, like), y)
    k2 = func(_time_arg(t + _div(dt, 3), like),
              tuple(_lincomb(y_, [1. / 3.], [k1_], h) for y_, k1_ in zip(y, k1)))
    k3 = func(_time_arg(t + _div(dt * 2, 3), like),
              tuple(_lincomb(y_, [-1. / 3., 1.0], [k1_, k2_], h) for y_, k1_, k2_ in zip(y, k1, k2)))
    k4 = func(_time_arg(t + dt, like),
              tuple(_lincomb(y_, [1.0, -1.0, 1.0], [k1_, k2_, k3_], h) for y_, k1_, k2_, k3_ in zip(y, k1, k2, k3)))
    return tuple(_lincomb(None, [1.0, 3.0, 3.0, 1.0], [a, b, c, d], _div(h, 8)) for a, b, c, d in zip(k1, k2, k3, k4))

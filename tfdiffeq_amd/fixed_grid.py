"""Mirror of tfdiffeq/fixed_grid.py: Euler, Midpoint, Heun, RK4 (3/8 rule) on a fixed grid."""
import torch

from .misc import _lincomb, _np_dtype
from .rk_common import _ButcherTableau, _div, _time_arg, _time_pair, rk4_alt_step_func
from .solvers import FixedGridODESolver

# step_func(func, t, dt, y): t / dt are host scalars of the state dtype, or 0-d float64 device tensors when the step is
# being captured into a hipGraph (graph_step.GraphedFixedStep); the arithmetic is the same either way.


class Euler(FixedGridODESolver):
    # fused kernel: y1 = y0 + dt * f(t, y0)  (zero-row tableau)
    _fused_tableau = _ButcherTableau(alpha=[], beta=[], c_sol=[1.0], c_error=[0.0])

    def step_func(self, func, t, dt, y):
        """fixed_grid.py:6-7."""
        t, _, h = _time_pair(t, dt, y[0])
        return tuple(_lincomb(None, [1.0], [f_], h) for f_ in func(_time_arg(t + self.eps, y[0]), y))

    @property
    def order(self):
        return 1


class Midpoint(FixedGridODESolver):

    def step_func(self, func, t, dt, y):
        """fixed_grid.py:16-18."""
        t, dt, h = _time_pair(t, dt, y[0])
        f0 = func(_time_arg(t + self.eps, y[0]), y)
        y_mid = tuple(_lincomb(y_, [0.5], [f_], h) for y_, f_ in zip(y, f0))
        return tuple(_lincomb(None, [1.0], [f_], h) for f_ in func(_time_arg(t + _div(dt, 2), y[0]), y_mid))

    @property
    def order(self):
        return 2


class Heun(FixedGridODESolver):

    def step_func(self, func, t, dt, y):
        """fixed_grid.py:28-32."""
        t, dt, h = _time_pair(t, dt, y[0])
        f_outs = func(_time_arg(t + self.eps, y[0]), y)
        ft_1_hat = tuple(_lincomb(y_, [1.0], [f_], h) for y_, f_ in zip(y, f_outs))
        ft_1_outs = func(_time_arg(t + dt, y[0]), ft_1_hat)
        return tuple(_lincomb(None, [1.0, 1.0], [a, b], _div(h, 2.)) for a, b in zip(f_outs, ft_1_outs))

    @property
    def order(self):
        return 2


class RK4(FixedGridODESolver):
    # fused kernels implement rk_common.rk4_alt_step_func (the 3/8 rule, F9) literally
    _fused_tableau = _ButcherTableau(alpha=[1 / 3, 2 / 3, 1.], beta=[[1 / 3], [-1 / 3, 1.], [1., -1., 1.]],
                                     c_sol=[1 / 8, 3 / 8, 3 / 8, 1 / 8], c_error=[0., 0., 0., 0.])

    def step_func(self, func, t, dt, y):
        """fixed_grid.py:41-42."""
        if isinstance(dt, torch.Tensor):
            return rk4_alt_step_func(func, t + self.eps if self.eps else t, dt, y)
        dt_ = _np_dtype(y[0].dtype).type
        return rk4_alt_step_func(func, dt_(t) + dt_(self.eps), dt, y)

    @property
    def order(self):
        return 4

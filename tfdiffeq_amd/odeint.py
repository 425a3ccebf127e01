"""Mirror of tfdiffeq/odeint.py: the `odeint` entry point and the SOLVERS registry (B1/B2)."""
from .adams import VariableCoefficientAdamsBashforth
from .adaptive_huen import AdaptiveHeunSolver
from .bosh3 import Bosh3Solver
from .dopri5 import Dopri5Solver
from .dopri8 import Dopri8Solver
from .fixed_adams import AdamsBashforth, AdamsBashforthMoulton
from .fixed_grid import Euler, Midpoint, RK4, Heun
from .misc import _check_inputs
from .tsit5 import Tsit5Solver

# odeint.py:11-25 - all 13 keys of the reference.  Hot path: dopri5, tsit5, bosh3, euler, rk4;
# 8(f) widening: dopri8, adaptive_heun, midpoint, heun (rank 1), the multistep family (rank 4).
SOLVERS = {
    'explicit_adams': AdamsBashforth,
    'fixed_adams': AdamsBashforthMoulton,
    'adams': VariableCoefficientAdamsBashforth,
    'tsit5': Tsit5Solver,
    'dopri5': Dopri5Solver,
    'dopri8': Dopri8Solver,
    'bosh3': Bosh3Solver,
    'adaptive_heun': AdaptiveHeunSolver,
    'euler': Euler,
    'midpoint': Midpoint,
    'rk4': RK4,
    'huen': Heun,
    'heun': Heun,
}


def odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None):
    """Integrate a system of ordinary differential equations (odeint.py:28-81).

        dy/dt = func(t, y),  y(t[0]) = y0

    func: callable (t: 0-d tensor in the state dtype, y) -> dy, or a `tfdiffeq_amd.rhs.DeviceRHS` (fused kernels).
    y0:   torch.Tensor on the MI355X (any shape; [batch, dim] for the fused kernels) or a tuple of them.
    t:    1-D, strictly monotone (either direction); converted to float64 (adaptive) / the state dtype (fixed).
    Returns a tensor [len(t), *y0.shape] (or a tuple of them) whose first entry is y0.

    Raises ValueError (options without method), KeyError (unknown method), TypeError (non-floating inputs),
    AssertionError (non-monotone t, dt underflow, non-finite state, max_num_steps) - as the reference does.
    """
    if _wants_grad(func, y0):
        # The reference back-propagates through the solver's eager ops (odeint.py:28-81 under a GradientTape).  The
        # kernels here are not taped: gradients come from the adjoint solve, which integrates the same system.
        import torch
        if isinstance(func, torch.nn.Module):
            from .adjoint import odeint_adjoint
            _warn_once('odeint: inputs require grad - gradients are computed with the adjoint method (odeint_adjoint)')
            return odeint_adjoint(func, y0, t, rtol=rtol, atol=atol, method=method, options=options)
        # A plain callable: the reference differentiates through it all the same (tf.GradientTape sees every op).  Its gradient with
        # respect to y0 (and t) is what the adjoint solve delivers for a parameterless system, so wrap it.  What cannot be
        # discovered is a tensor the callable closes over - it gets no gradient here, and the warning says so.
        from .adjoint import odeint_adjoint
        _warn_once('odeint: y0 requires grad and `func` is a plain callable - gradients w.r.t. y0 and t are computed with the adjoint '
                   'method; tensors the callable closes over receive NO gradient (make it a torch.nn.Module with parameters for that)')
        return odeint_adjoint(_CallableModule(func), y0, t, rtol=rtol, atol=atol, method=method, options=options)
    tensor_input, func, y0, t = _check_inputs(func, y0, t)

    if options is None:
        options = {}
    elif method is None:
        raise ValueError('cannot supply `options` without specifying `method`')

    if method is None:
        method = 'dopri5'
    solver = SOLVERS[method](func, y0, rtol=rtol, atol=atol, **options)
    solution = solver.integrate(t)
    odeint.last_stats = getattr(solver, 'stats', {})
    if tensor_input:
        solution = solution[0]
    return solution


odeint.last_stats = {}
_warned = set()


def _callable_module(func):
    import torch

    class _M(torch.nn.Module):
        """A parameterless module around a plain callable (odeint_adjoint needs a module to look for parameters in)."""

        def __init__(self, f):
            super().__init__()
            self._f = f

        def forward(self, t, y):
            return self._f(t, y)
    return _M(func)


def _CallableModule(func):
    return _callable_module(func)


def _warn_once(msg):
    if msg not in _warned:
        _warned.add(msg)
        import warnings
        warnings.warn(msg, stacklevel=3)


def _wants_grad(func, y0):
    import torch
    if not torch.is_grad_enabled():
        return False
    ys = y0 if isinstance(y0, (tuple, list)) else (y0,)
    if any(isinstance(y, torch.Tensor) and y.requires_grad for y in ys):
        return True
    return isinstance(func, torch.nn.Module) and any(p.requires_grad for p in func.parameters())

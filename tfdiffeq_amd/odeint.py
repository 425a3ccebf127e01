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
        # A plain callable: the reference differentiates through it all the same (tf.GradientTape sees every op), including
        # through whatever the callable closes over - `odeint(lambda t, y: net(y), enc(x), t)` trains `net`.  The adjoint solve
        # needs those tensors by name, so look for them: modules and grad-requiring tensors in the callable's closure cells,
        # bound object, partial arguments, defaults and the globals its code names.  They become the wrapper module's
        # parameters and receive their gradients like any module's.
        from .adjoint import odeint_adjoint
        mods, tens = _closure_state(func)
        if mods or tens:
            _warn_once('odeint: inputs require grad and `func` is a plain callable - gradients are computed with the adjoint method; the '
                       'modules / tensors it closes over (%d / %d found) are treated as its parameters' % (len(mods), len(tens)))
        else:
            _warn_once('odeint: y0 requires grad and `func` is a plain callable - gradients w.r.t. y0 and t are computed with the adjoint '
                       'method (no module or grad-requiring tensor was found in its closure)')
        return odeint_adjoint(_callable_module(func, mods, tens), y0, t, rtol=rtol, atol=atol, method=method, options=options)
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


def _closure_state(func):
    """(modules, grad-requiring tensors that belong to none of them) a plain callable can reach: closure cells, the bound object,
    functools.partial arguments, defaults, and the module-level names its code refers to; containers and nested callables are
    followed a few levels deep."""
    import functools
    import inspect
    import torch
    mods, tens, seen = [], [], set()

    def visit(obj, depth):
        if id(obj) in seen or depth > 4:
            return
        seen.add(id(obj))
        if isinstance(obj, torch.nn.Module):
            mods.append(obj)
        elif isinstance(obj, torch.Tensor):
            if obj.requires_grad:
                tens.append(obj)
        elif isinstance(obj, (list, tuple, set, frozenset)):
            for o in obj:
                visit(o, depth + 1)
        elif isinstance(obj, dict):
            for o in obj.values():
                visit(o, depth + 1)
        elif isinstance(obj, functools.partial):
            visit(obj.func, depth + 1)
            visit(obj.args, depth + 1)
            visit(obj.keywords, depth + 1)
        elif inspect.isfunction(obj) or inspect.ismethod(obj):
            walk(obj, depth + 1)
        elif callable(obj) and hasattr(obj, '__dict__') and not inspect.isclass(obj) and not inspect.ismodule(obj):
            visit(vars(obj), depth + 1)                  # an object with __call__: what it holds
            call = getattr(type(obj), '__call__', None)
            if inspect.isfunction(call):
                walk(call, depth + 1)

    def walk(f, depth):
        try:
            f = inspect.unwrap(f)
        except ValueError:
            pass
        owner = getattr(f, '__self__', None)
        if owner is not None and not inspect.ismodule(owner):
            visit(owner, depth)
        fn = getattr(f, '__func__', f)
        for cell in getattr(fn, '__closure__', None) or ():
            try:
                visit(cell.cell_contents, depth)
            except ValueError:                           # empty cell
                pass
        visit(getattr(fn, '__defaults__', None) or (), depth)
        visit(getattr(fn, '__kwdefaults__', None) or {}, depth)
        code, glob = getattr(fn, '__code__', None), getattr(fn, '__globals__', None)
        if code is not None and glob is not None:
            import dis
            names, todo = set(), [code]
            while todo:                                  # LOAD_GLOBAL only (co_names also lists attribute names); nested lambdas /
                c = todo.pop()                           # comprehensions name globals too
                names.update(i.argval for i in dis.get_instructions(c) if i.opname in ('LOAD_GLOBAL', 'LOAD_NAME'))
                todo.extend(k for k in c.co_consts if inspect.iscode(k))
            for name in names:
                v = glob.get(name)
                if isinstance(v, (torch.nn.Module, torch.Tensor, functools.partial)) or inspect.isfunction(v):
                    visit(v, depth)

    visit(func, 0)
    owned = {id(p) for m in mods for p in m.parameters()}
    tens = [x for x in tens if id(x) not in owned]
    return mods, tens


def _callable_module(func, mods=(), tens=()):
    import torch

    class _M(torch.nn.Module):
        """A module around a plain callable (odeint_adjoint looks for parameters in a module): the modules the callable closes
        over are registered as sub-modules, bare grad-requiring tensors travel as `_mi_extra_params` (adjoint._trainable)."""

        def __init__(self, f):
            super().__init__()
            self._f = f
            for i, m in enumerate(mods):
                self.add_module('closed_over_%d' % i, m)
            self._mi_extra_params = tuple(tens)

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
    if isinstance(func, torch.nn.Module):
        return any(p.requires_grad for p in func.parameters())
    if getattr(func, 'kind', 0) or not callable(func):     # a DeviceRHS descriptor: weights are plain device tensors
        return False
    # a plain callable over trainable state (`lambda t, y: net(y)`): the reference's tape would reach net's parameters
    mods, tens = _closure_state(func)
    return bool(tens) or any(p.requires_grad for m in mods for p in m.parameters())

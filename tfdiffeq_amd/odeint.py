"""Mirror of tfdiffeq/odeint.py: the `odeint` entry point and the SOLVERS registry (B1/B2)."""
import os

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

    One semantic difference from the reference's eager loop (adaptive Runge-Kutta methods, Python callable `func`, default
    options={'graph': 'auto'}): after two eager attempts, and if about a dozen more are to come, an attempt is RECORDED as a hipGraph and
    replayed - func's Python body no longer runs (side effects stop; an integer `nfe` attribute is still credited), `t` is a 0-d view
    of a device buffer the controller overwrites (clone it to keep it), and up to 64 blind replays past the end are no-ops on the
    device but were recorded with func inside.  A func that synchronises with the host or records autograd is detected and stays eager;
    options={'graph': False} turns the recording off, 'host' restores the loop with the controller on the host.
    """
    pre = None
    if _plain_callable_no_grad_state(func, y0):
        # A plain callable over a state that does not require grad: whether it has TRAINABLE inputs of its own is read off its trace - the
        # tensors it closes over are exactly the trace's tensor constants - instead of a probe evaluation with autograd on every call.
        pre = _try_lower(func, y0, method, options)
        if pre is not None and pre[0] is not None:
            leaves = tuple(e['t'] for e in pre[0].trace.tensors if isinstance(e['t'], __import__('torch').Tensor) and e['t'].requires_grad)
            if not leaves:
                return _run_lowered(pre[0], func, y0, t, rtol, atol, method, options)
            from .adjoint import odeint_adjoint
            _warn_once('odeint: `func` is a plain callable over %d grad-requiring tensor(s) - gradients are computed with the adjoint method; '
                       'they are treated as its parameters' % len(leaves))
            return odeint_adjoint(_callable_module(func, (), leaves), y0, t, rtol=rtol, atol=atol, method=method, options=options)
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
        # needs those tensors by name: they are read off the autograd graph of ONE evaluation f(t[0], y0) (`_graph_leaves`: exactly the
        # grad-requiring leaves f uses, however it reaches them; cached per callable).  They become the wrapper module's parameters
        # and receive their gradients like any module's.
        from .adjoint import odeint_adjoint
        leaves = _graph_leaves(func, y0, t)
        if leaves:
            _warn_once('odeint: inputs require grad and `func` is a plain callable - gradients are computed with the adjoint method; the '
                       '%d grad-requiring tensor(s) its evaluation depends on are treated as its parameters' % len(leaves))
        else:
            _warn_once('odeint: y0 requires grad and `func` is a plain callable - gradients w.r.t. y0 and t are computed with the adjoint '
                       'method (its evaluation depends on no other grad-requiring tensor)')
        mod = _callable_module(func, (), leaves)
        mod._mi_optional_params = tuple(getattr(_graph_leaves, 'named_only', ()))
        return odeint_adjoint(mod, y0, t, rtol=rtol, atol=atol, method=method, options=options)
    lowered = pre if pre is not None else _try_lower(func, y0, method, options)
    if lowered is not None:
        low, why = lowered
        if low is not None:
            return _run_lowered(low, func, y0, t, rtol, atol, method, options)
        options = dict(options or {})
        impure = 'changed its own Python state' in why
        if impure and method is None:
            method = 'dopri5'                              # (odeint.py:75-76: the default method, named so that it can carry an option)
        if impure and 'graph' not in options:
            # a callable with observable Python state of its own (a step counter in a list, a log): recording an attempt as a hipGraph would
            # stop those side effects after two attempts - eager evaluation unless the caller asks for the recording (round-5 review, item 8)
            options['graph'] = False
        options.pop('lower', None)
        options = options or None
        _note = {'lowered': False, 'why': why}
    else:
        _note = None
        if options is not None and 'lower' in options:
            options = {k: v for k, v in options.items() if k != 'lower'} or (None if method is None else {})
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
    if _note is not None and isinstance(odeint.last_stats, dict):
        odeint.last_stats['lower'] = _note
    if tensor_input:
        solution = solution[0]
    return solution


# Process-wide default of options['lower'] ('auto' | True | False; the environment variable TFDIFFEQ_AMD_LOWER=0 starts with False):
# tests of the callable engines set it to False so that their Python callables stay Python callables.
LOWER_DEFAULT = False if os.environ.get('TFDIFFEQ_AMD_LOWER', '1').lower() in ('0', 'false', 'off') else 'auto'

_LOWER_METHODS = ('dopri5', 'tsit5', 'bosh3', 'dopri8', 'adaptive_heun', 'euler', 'rk4', 'midpoint', 'heun', 'huen', 'explicit_adams',
                  'fixed_adams', 'adams')


def _try_lower(func, y0, method, options):
    """(Lowered, None) / (None, reason) when this call is one the tracer is asked to look at, None when it is not (a DeviceRHS, a CPU
    state - the solver raises its own error -, a tuple of several components, an explicit request for one of the callable engines).
    options['lower']: 'auto' (default) - lower when possible, say once why not otherwise; True - raise if the callable cannot be lowered;
    False - never trace."""
    import torch
    opts = options or {}
    mode = opts.get('lower', LOWER_DEFAULT)
    if mode is False or mode == 'off' or not callable(func) or getattr(func, 'kind', 0) or getattr(func, 'per_component', False):
        return None
    if method is not None and method not in _LOWER_METHODS:
        return None
    if any(k in opts for k in ('process_group', 'force_plane_kernels', 'grid_constructor')) or (opts.get('graph', 'auto') != 'auto' and mode is not True):
        return None
    y = y0[0] if isinstance(y0, (tuple, list)) and len(y0) == 1 else y0
    from . import lower as _lower
    if isinstance(y, (tuple, list)):
        # a tuple state of several components (tests/api_tests.py:29-34): lowered when every component follows the same trajectory-local
        # function (rhs.PerComponent: one launch, one error ratio per component); anything else keeps its Python loop
        ok = 2 <= len(y) <= 8 and all(isinstance(c, torch.Tensor) and c.is_cuda and c.dtype == y[0].dtype and c.numel() > 0 and
                                      c.dtype in (torch.float32, torch.float64) for c in y) and method in (None, 'dopri5', 'bosh3', 'tsit5', 'euler', 'rk4')
        if not ok:
            return None
        try:
            return _lower.lower_tuple(func, y, method=method), None
        except _lower.TraceError as e:
            why = str(e)
        except Exception as e:
            why = 'tracing failed: %s: %s' % (type(e).__name__, e)
        if mode is True:
            raise ValueError('odeint(options={\'lower\': True}): this callable cannot be lowered onto the fused kernels: ' + why)
        return None, why
    if not isinstance(y, torch.Tensor) or not y.is_cuda or y.dtype not in (torch.float32, torch.float64) or y.numel() == 0:
        return None
    wrapped = func
    if y is not y0:                                       # a one-component tuple (the adjoint's forward pass): the tensor form of the same system
        def wrapped(t_, y_, _f=func):
            return _f(t_, (y_,))[0]
    try:
        return _lower.lower(wrapped, y, method=method), None
    except _lower.TraceError as e:
        why = str(e)
    except Exception as e:                                # the callable itself failed on the proxies (an operation torch refuses for them)
        why = 'tracing failed: %s: %s' % (type(e).__name__, e)
    if mode is True:
        raise ValueError('odeint(options={\'lower\': True}): this callable cannot be lowered onto the fused kernels: ' + why)
    _warn_once('odeint: `func` runs as a Python callable between library kernels (not lowered onto the fused kernels: %s)' % why)
    return None, why


def _run_lowered(low, func, y0, t, rtol, atol, method, options):
    """The call with the traced callable's device right-hand side in its place: state reshaped to [*batch, dim] and back."""
    import torch
    from .graph_step import _credit_nfe
    opts = {k: v for k, v in (options or {}).items() if k != 'lower'}
    if getattr(low, 'per_component', False):
        from . import rhs as _rhs
        states = tuple(c.detach().reshape(s_).contiguous() for c, s_ in zip(y0, low.state_shapes))
        sols = odeint(_rhs.PerComponent(low.rhs), states, t, rtol=rtol, atol=atol, method=method, options=None if options is None else opts)
        stats = odeint.last_stats if isinstance(odeint.last_stats, dict) else {}
        info = low.describe()
        info.update(lowered=True, components=len(states))
        stats['lower'] = info
        odeint.last_stats = stats
        _credit_nfe(func, int(stats.get('nfe', 0)) - low.py_calls)
        return tuple(s_.reshape((s_.shape[0],) + tuple(c.shape)) for s_, c in zip(sols, y0))
    tuple_in = isinstance(y0, (tuple, list))
    y = y0[0] if tuple_in else y0
    state = y.detach().reshape(low.state_shape).contiguous()
    try:
        sol = odeint(low.rhs, state, t, rtol=rtol, atol=atol, method=method, options=None if options is None else opts)
    except Exception as e:
        if 'compiling the RHS plugin failed' not in str(e):
            raise
        # hipcc refused (or did not finish) the generated code: the callable itself, on the callable engine - said once
        _warn_once('odeint: the kernel generated for `func` did not compile (%s); it runs as a Python callable instead' % str(e).split(';')[0])
        out = odeint(func, y0, t, rtol=rtol, atol=atol, method=method, options=dict(opts, lower=False) if method is not None else None)
        if isinstance(odeint.last_stats, dict):
            odeint.last_stats['lower'] = {'lowered': False, 'why': 'generated code did not compile'}
        return out
    stats = odeint.last_stats if isinstance(odeint.last_stats, dict) else {}
    info = low.describe()
    info['lowered'] = True
    stats['lower'] = info
    odeint.last_stats = stats
    # f ran inside the kernels: an integer `nfe` counter of the callable stays meaningful (evaluations that DID run the Python body - a
    # method or a batch no one-launch kernel takes - have counted themselves)
    _credit_nfe(func, int(stats.get('nfe', 0)) - low.py_calls)
    sol = sol.reshape((sol.shape[0],) + tuple(y.shape))
    return (sol,) if tuple_in else sol


odeint.last_stats = {}


def _plan(*args, **kwargs):
    from .plan import plan
    return plan(*args, **kwargs)


_plan.__doc__ = 'odeint.plan(func, y0, t=None, rtol=.., atol=.., method=None, options=None): the engine this call will take and why (tfdiffeq_amd/plan.py).'
odeint.plan = _plan
_warned = set()


def _reachable_leaves(func, max_depth=3):
    """Grad-requiring leaf tensors a callable can NAME: closure cells, its bound object, partial arguments, its own attributes, the globals
    its code mentions - followed through modules (their parameters), containers and plain objects' attributes, a few levels deep.  The
    complement of `_graph_leaves`: a branch the probe evaluation did not take (`nA(y) if t < 0.5 else nB(y)`) is still found here."""
    import functools
    import torch
    out, seen = [], set()

    def visit(obj, depth):
        if obj is None or id(obj) in seen or isinstance(obj, (int, float, str, bytes, bool, type)):
            return
        seen.add(id(obj))
        if isinstance(obj, torch.Tensor):
            if obj.requires_grad and obj.is_leaf and all(obj is not o for o in out):
                out.append(obj)
            return
        if isinstance(obj, torch.nn.Module):
            for p in obj.parameters():
                visit(p, depth)
            return
        if depth >= max_depth:
            return
        if isinstance(obj, (list, tuple, set, frozenset)):
            for v in list(obj)[:64]:
                visit(v, depth + 1)
        elif isinstance(obj, dict):
            for v in list(obj.values())[:64]:
                visit(v, depth + 1)
        elif isinstance(obj, functools.partial):
            visit(obj.func, depth)
            visit(obj.args, depth)
            visit(obj.keywords, depth)
        elif callable(obj) and hasattr(obj, '__code__'):
            for cell in obj.__closure__ or ():
                try:
                    visit(cell.cell_contents, depth + 1)
                except ValueError:
                    pass
            g = getattr(obj, '__globals__', {})
            for name in obj.__code__.co_names:
                if name in g and not isinstance(g[name], type(torch)):          # (modules like `torch` itself are not walked)
                    visit(g[name], depth + 1)
        else:
            fn = getattr(obj, '__func__', None)
            if fn is not None:                             # a bound method: the function and the object
                visit(fn, depth)
                visit(getattr(obj, '__self__', None), depth)
                return
            call = getattr(type(obj), '__call__', None)
            if call is not None and hasattr(call, '__code__'):
                visit(call, depth)
            d = getattr(obj, '__dict__', None)
            if isinstance(d, dict):
                for v in list(d.values())[:64]:
                    visit(v, depth + 1)
    visit(func, 0)
    return tuple(out)


def _graph_leaves(func, y0, t):
    """The grad-requiring LEAF tensors `func` depends on, besides y0 and t themselves: the leaves of the autograd graph of ONE probe
    evaluation f(t[0], y0) (AccumulateGrad nodes - exactly what the reference's GradientTape would reach through whatever the evaluation
    touches: parameters of modules the callable calls, bare tensors, attribute chains, containers), in union with the leaves the callable
    can name (`_reachable_leaves`: a branch the probe did not take keeps its gradient; round-5 advisor).  Nothing is cached between calls:
    a callable whose network was rebound, or whose tensors changed `requires_grad`, is seen as it is NOW (round-5 advisor: the cache keyed
    on ids returned the old network's parameters)."""
    import torch
    ys = y0 if isinstance(y0, (tuple, list)) else (y0,)
    # the probe is not one of the solver's evaluations: an integer evaluation counter the callable keeps (`nfe`, as the reference's
    # DETEST harness and ODEFunc do) is put back afterwards
    counted = [o for o in (func, getattr(func, '__self__', None)) if isinstance(getattr(o, 'nfe', None), int)]
    before = [o.nfe for o in counted]
    leaves = []
    try:
        with torch.enable_grad():
            probe = tuple(y.detach().requires_grad_(True) for y in ys)
            t0 = torch.as_tensor(t).reshape(-1)[0].detach().to(device=ys[0].device, dtype=ys[0].dtype).requires_grad_(True)
            out = func(t0, probe if isinstance(y0, (tuple, list)) else probe[0])
        outs = out if isinstance(out, (tuple, list)) else (out,)
        skip = {id(p) for p in probe} | {id(t0)}
        seen, todo = set(), [o.grad_fn for o in outs if isinstance(o, torch.Tensor) and o.grad_fn is not None]
        while todo:
            node = todo.pop()
            if node is None or node in seen:                 # (the set holds the node objects: an id alone is recycled as soon as a
                continue                                     # wrapper is dropped, and a recycled id would hide a whole branch of the graph)
            seen.add(node)
            var = getattr(node, 'variable', None)            # AccumulateGrad: a leaf
            if var is not None:
                if var.requires_grad and id(var) not in skip and all(var is not l_ for l_ in leaves):
                    leaves.append(var)
                continue
            todo.extend(fn for fn, _ in node.next_functions)
    finally:
        for o, n in zip(counted, before):
            try:
                o.nfe = n
            except Exception:
                pass
    n_probe = len(leaves)
    for extra in _reachable_leaves(func):
        if all(extra is not l_ for l_ in leaves) and all(extra is not y for y in ys):
            leaves.append(extra)
    _graph_leaves.named_only = tuple(leaves[n_probe:])      # (of the LAST call: odeint marks them "None if no evaluation reaches them")
    return tuple(leaves)


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
    # a plain callable over trainable state (`lambda t, y: net(y)`): the reference's tape would reach net's parameters.  Whether THIS
    # callable does is a property of its autograd graph (one cached probe evaluation), not of what its closure could name.
    return False if _probe_unsafe(y0) else bool(_graph_leaves_or_none(func, y0))


def _plain_callable_no_grad_state(func, y0):
    import torch
    if not torch.is_grad_enabled() or not callable(func) or getattr(func, 'kind', 0) or isinstance(func, torch.nn.Module):
        return False
    ys = y0 if isinstance(y0, (tuple, list)) else (y0,)
    return all(isinstance(y, torch.Tensor) and not y.requires_grad for y in ys)


def _probe_unsafe(y0):
    import torch
    ys = y0 if isinstance(y0, (tuple, list)) else (y0,)
    return not all(isinstance(y, torch.Tensor) and y.is_floating_point() for y in ys)


def _graph_leaves_or_none(func, y0):
    """_graph_leaves at t = 0 for the routing decision (`odeint` has not validated t yet); a callable that cannot be evaluated like that
    is judged by the tensors it can name alone (the solver will raise the real error of the evaluation)."""
    import torch
    try:
        return _graph_leaves(func, y0, torch.zeros(1))
    except Exception:
        return _reachable_leaves(func)

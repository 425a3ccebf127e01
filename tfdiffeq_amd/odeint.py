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

    One semantic difference from the reference's eager loop (adaptive Runge-Kutta methods, Python callable `func`, default
    options={'graph': 'auto'}): after two eager attempts, and if about a dozen more are to come, an attempt is RECORDED as a hipGraph and
    replayed - func's Python body no longer runs (side effects stop; an integer `nfe` attribute is still credited), `t` is a 0-d view
    of a device buffer the controller overwrites (clone it to keep it), and up to 64 blind replays past the end are no-ops on the
    device but were recorded with func inside.  A func that synchronises with the host or records autograd is detected and stays eager;
    options={'graph': False} turns the recording off, 'host' restores the loop with the controller on the host.
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
        return odeint_adjoint(_callable_module(func, (), leaves), y0, t, rtol=rtol, atol=atol, method=method, options=options)
    lowered = _try_lower(func, y0, method, options)
    if lowered is not None:
        low, why = lowered
        if low is not None:
            return _run_lowered(low, func, y0, t, rtol, atol, method, options)
        options = dict(options or {})
        impure = 'changed its own Python state' in why
        if impure and 'graph' not in options and method is not None:
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


_LOWER_METHODS = ('dopri5', 'tsit5', 'bosh3', 'dopri8', 'adaptive_heun', 'euler', 'rk4', 'midpoint', 'heun', 'huen', 'explicit_adams',
                  'fixed_adams', 'adams')


def _try_lower(func, y0, method, options):
    """(Lowered, None) / (None, reason) when this call is one the tracer is asked to look at, None when it is not (a DeviceRHS, a CPU
    state - the solver raises its own error -, a tuple of several components, an explicit request for one of the callable engines).
    options['lower']: 'auto' (default) - lower when possible, say once why not otherwise; True - raise if the callable cannot be lowered;
    False - never trace."""
    import torch
    opts = options or {}
    mode = opts.get('lower', 'auto')
    if mode is False or mode == 'off' or not callable(func) or getattr(func, 'kind', 0) or getattr(func, 'per_component', False):
        return None
    if method is not None and method not in _LOWER_METHODS:
        return None
    if any(k in opts for k in ('process_group', 'force_plane_kernels', 'grid_constructor')) or (opts.get('graph', 'auto') != 'auto' and mode is not True):
        return None
    y = y0[0] if isinstance(y0, (tuple, list)) and len(y0) == 1 else y0
    if not isinstance(y, torch.Tensor) or not y.is_cuda or y.dtype not in (torch.float32, torch.float64) or y.numel() == 0:
        return None
    from . import lower as _lower
    wrapped = func
    if y is not y0:                                       # a one-component tuple (the adjoint's forward pass): the tensor form of the same system
        def wrapped(t_, y_, _f=func):
            return _f(t_, (y_,))[0]
    try:
        return _lower.lower(wrapped, y), None
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
    tuple_in = isinstance(y0, (tuple, list))
    y = y0[0] if tuple_in else y0
    opts = {k: v for k, v in (options or {}).items() if k != 'lower'}
    state = y.detach().reshape(low.state_shape).contiguous()
    sol = odeint(low.rhs, state, t, rtol=rtol, atol=atol, method=method, options=None if options is None else opts)
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
_warned = set()


_LEAF_CACHE = {}


def _callable_key(func):
    """(key, witnesses) identifying a callable for the leaf cache: the code object and what its closure cells / bound object / partial
    arguments hold (a lambda written inside a training loop is a NEW object every iteration, but the same code over the same cells), else
    the object itself.  The key is built from ids; `witnesses` are the objects behind those ids - kept by the cache entry as weak
    references where the type allows it (a module that died and whose id was recycled is then noticed), strongly otherwise."""
    import functools
    f = func
    if isinstance(f, functools.partial):
        k, w = _callable_key(f.func)
        held = list(f.args) + [v for _, v in sorted((f.keywords or {}).items())]
        return ('partial', k, tuple(id(a) for a in held)), w + held
    fn = getattr(f, '__func__', f)
    code = getattr(fn, '__code__', None)
    if code is None:
        return ('object', id(f)), [f]
    held = [code]
    for cell in getattr(fn, '__closure__', None) or ():
        try:
            held.append(cell.cell_contents)
        except ValueError:                               # empty cell
            held.append(None)
    held.append(getattr(f, '__self__', None))
    return ('code', tuple(id(h) for h in held)), held


def _witness(obj):
    import weakref
    try:
        return weakref.ref(obj)
    except TypeError:
        return lambda o=obj: o                           # (not weak-referenceable: held, so its id cannot be recycled)


def _graph_leaves(func, y0, t):
    """The grad-requiring LEAF tensors one evaluation f(t[0], y0) depends on, besides y0 and t themselves - read off the autograd
    graph (AccumulateGrad nodes).  Exactly what the reference's GradientTape would reach: parameters of modules the callable calls,
    bare tensors it closes over, anything reached through attribute chains or containers - and nothing it merely could name.
    One extra evaluation of f per distinct callable (cached: see _callable_key; the entry keeps the callable's identity alive only as ids
    and is dropped when any leaf's storage was freed)."""
    import torch
    key, held = _callable_key(func)
    hit = _LEAF_CACHE.get(key)
    if hit is not None:
        if len(hit[0]) == len(held) and all(w() is h for w, h in zip(hit[0], held)):
            return hit[1]
        del _LEAF_CACHE[key]                             # an id was recycled by another object
    ys = y0 if isinstance(y0, (tuple, list)) else (y0,)
    # the probe is not one of the solver's evaluations: an integer evaluation counter the callable keeps (`nfe`, as the reference's
    # DETEST harness and ODEFunc do) is put back afterwards
    counted = [o for o in (func, getattr(func, '__self__', None)) if isinstance(getattr(o, 'nfe', None), int)]
    before = [o.nfe for o in counted]
    with torch.enable_grad():
        probe = tuple(y.detach().requires_grad_(True) for y in ys)
        t0 = torch.as_tensor(t).reshape(-1)[0].detach().to(device=ys[0].device, dtype=ys[0].dtype).requires_grad_(True)
        out = func(t0, probe if isinstance(y0, (tuple, list)) else probe[0])
    for o, n in zip(counted, before):
        try:
            o.nfe = n
        except Exception:
            pass
    outs = out if isinstance(out, (tuple, list)) else (out,)
    skip = {id(p) for p in probe} | {id(t0)}
    leaves, seen, todo = [], set(), [o.grad_fn for o in outs if isinstance(o, torch.Tensor) and o.grad_fn is not None]
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
    leaves = tuple(leaves)
    while len(_LEAF_CACHE) >= 16:
        _LEAF_CACHE.pop(next(iter(_LEAF_CACHE)))
    _LEAF_CACHE[key] = ([_witness(h) for h in held], leaves)
    return leaves


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


def _probe_unsafe(y0):
    import torch
    ys = y0 if isinstance(y0, (tuple, list)) else (y0,)
    return not all(isinstance(y, torch.Tensor) and y.is_floating_point() for y in ys)


def _graph_leaves_or_none(func, y0):
    """_graph_leaves at t = 0 for the routing decision (`odeint` has not validated t yet); a callable that cannot be evaluated like that
    decides nothing here - the solver will raise the real error."""
    import torch
    try:
        return _graph_leaves(func, y0, torch.zeros(1))
    except Exception:
        return ()

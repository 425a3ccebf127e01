"""Host-side mirror of tfdiffeq/misc.py for torch tensors on MI355X.

Same names, argument meaning and error behaviour as the reference helpers; every state-sized
operation is a launch of a libmi_ode plane kernel (no torch eager arithmetic on state tensors),
the scalar controller arithmetic stays on the host exactly as the reference does it - including
its dtype quirks (SURVEY.md F3, F4 and the float32 detour of `_convert_to_tensor`).
"""
import ctypes as C
import warnings

import numpy as np
import torch

from . import _native as N


# ---------------------------------------------------------------------------------------------
# small plumbing
# ---------------------------------------------------------------------------------------------
def _np_dtype(t_dtype):
    return np.dtype(np.float64) if t_dtype == torch.float64 else np.dtype(np.float32)


def _ptr(x):
    return C.c_void_p(x.data_ptr()) if x is not None else C.c_void_p(0)


def _contig(x):
    return x if x.is_contiguous() else x.contiguous()


def _scalar_tensor(value, like):
    """A 0-d device tensor in `like`'s dtype (what the reference hands to func as `t`).  torch.full is a
    fill-kernel launch: no host->device copy, no synchronisation."""
    return torch.full((), float(value), dtype=like.dtype, device=like.device)


_workspaces = {}


def _reduce_workspace(device):
    key = str(device)
    if key not in _workspaces:
        nbytes = N.load().mi_ode_reduce_workspace_bytes()
        _workspaces[key] = torch.empty(nbytes // 8, dtype=torch.float64, device=device)
    return _workspaces[key]


def move_to_device(x, device):
    """misc.py:8-42 - device plumbing (kept for API compatibility)."""
    if device is None or not isinstance(x, torch.Tensor):
        return x
    if isinstance(device, torch.Tensor):
        device = device.device
    return x.to(device)


def _check_len(x):
    return x.shape[0] if hasattr(x, 'shape') else len(x)


def _numel(x, dtype=None):
    return x.numel()


def _is_iterable(inputs):
    try:
        iter(inputs)
        return True
    except TypeError:
        return False


def _possibly_nonzero(x):
    return isinstance(x, torch.Tensor) or x != 0


# ---------------------------------------------------------------------------------------------
# plane kernels
# ---------------------------------------------------------------------------------------------
class _DevScalar(object):
    """A float64 scalar in device memory that is not a torch tensor (the step size inside libmi_ode's controller record)."""
    __slots__ = ('ptr',)

    def __init__(self, ptr):
        self.ptr = int(ptr)


def _lincomb(base, coefs, xs, scale):
    """out = base + add_n([(scale * c_j) * x_j])  on device (mi_ode_lincomb).  `scale` is a host scalar, or a 0-d float64
    device tensor / a `_DevScalar` that the kernel reads when it runs (mi_ode_lincomb_dev: graph replays with a new dt)."""
    xs = [_contig(x) for x in xs]
    x0 = xs[0]
    N.require_gpu_tensor(x0, 'state')
    n = x0.numel()
    out = torch.empty_like(x0)
    if n == 0:
        return out
    nx = len(xs)
    if nx > N.MAX_LINCOMB:
        raise ValueError('at most %d planes per linear combination' % N.MAX_LINCOMB)
    ptrs = (C.c_void_p * nx)(*[x.data_ptr() for x in xs])
    cf = (C.c_double * nx)(*[float(c) for c in coefs])
    base_c = _contig(base) if base is not None else None
    lib = N.load()
    if isinstance(scale, (torch.Tensor, _DevScalar)):
        if isinstance(scale, torch.Tensor):
            assert scale.dtype == torch.float64 and scale.numel() == 1 and scale.device == x0.device
            sp = scale.data_ptr()
        else:
            sp = scale.ptr
        N.check(lib.mi_ode_lincomb_dev(N.dtype_code(x0.dtype), n, _ptr(base_c), ptrs, cf, nx, C.c_void_p(sp),
                                       _ptr(out), N.stream_ptr(x0.device)), 'mi_ode_lincomb_dev')
        return out
    N.check(lib.mi_ode_lincomb(N.dtype_code(x0.dtype), n, _ptr(base_c), ptrs, cf, nx, float(scale), _ptr(out),
                               N.stream_ptr(x0.device)), 'mi_ode_lincomb')
    return out


def _scaled_dot_product(scale, xs, ys):
    """misc.py:118-121: add_n([scale * x * y ...]); xs are the (python float) weights, ys the tensors.
    Zero weights are not skipped (F10)."""
    return _lincomb(None, xs, ys, scale)


def _dot_product(xs, ys):
    """misc.py:124-126."""
    return _lincomb(None, xs, ys, 1.0)


def _error_norms(err, y0, y1):
    """Device record {max|y0|, max|y1|, sum err^2, nonfinite(y0)} (float64[4]) for one component."""
    err, y0, y1 = _contig(err), _contig(y0), _contig(y1)
    res = torch.empty(4, dtype=torch.float64, device=err.device)
    lib = N.load()
    N.check(lib.mi_ode_error_norms(N.dtype_code(err.dtype), err.numel(), _ptr(err), _ptr(y0), _ptr(y1), _ptr(res),
                                   _ptr(_reduce_workspace(err.device)), N.stream_ptr(err.device)),
            'mi_ode_error_norms')
    return res


def _scaled_sumsq(x, xsub, y0, rtol, atol):
    x, y0 = _contig(x), _contig(y0)
    xsub = _contig(xsub) if xsub is not None else None
    res = torch.empty(1, dtype=torch.float64, device=x.device)
    lib = N.load()
    N.check(lib.mi_ode_scaled_sumsq(N.dtype_code(x.dtype), x.numel(), _ptr(x), _ptr(xsub), _ptr(y0), float(rtol),
                                    float(atol), _ptr(res), _ptr(_reduce_workspace(x.device)),
                                    N.stream_ptr(x.device)), 'mi_ode_scaled_sumsq')
    return res


def _has_converged(y0, y1, rtol, atol):
    """misc.py:129-134: every element within atol + rtol*max(|y0|,|y1|); one kernel per component, one host sync."""
    lib = N.load()
    flags = []
    for a, b in zip(y0, y1):
        a, b = _contig(a), _contig(b)
        res = torch.empty(1, dtype=torch.float64, device=a.device)
        N.check(lib.mi_ode_not_converged(N.dtype_code(a.dtype), a.numel(), _ptr(a), _ptr(b), float(rtol), float(atol), _ptr(res),
                                         _ptr(_reduce_workspace(a.device)), N.stream_ptr(a.device)), 'mi_ode_not_converged')
        flags.append(res)
    return not bool(torch.cat(flags).max().item() > 0)


class _Exchange(object):
    """Combines per-rank reduction records for batch-sharded runs.  Records are laid out
    {max, max, sum, flag/sum, ...}: maxima for slots listed in `max_slots`, sums elsewhere, in rank order."""

    def __init__(self, group=None):
        self.group = group

    def combine(self, rec, max_slots):
        if self.group is None:
            return rec
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        flat = rec.reshape(-1).contiguous()
        gathered = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
        dist.all_gather_into_tensor(gathered, flat, group=self.group)
        g = gathered.reshape((world,) + tuple(rec.shape))
        out = g.sum(dim=0)
        mx = g.max(dim=0).values
        for s in max_slots:
            out[..., s] = mx[..., s]
        return out


def _is_finite(tensor):
    """misc.py:147-150."""
    return bool(torch.isfinite(tensor).all().item())


def _decreasing(t):
    if t.device.type == 'cpu' and t.numel() <= 64:          # the usual handful of output times: python floats, no tensor kernels
        v = t.tolist()
        return all(b < a for a, b in zip(v, v[1:]))
    return bool((t[1:] < t[:-1]).all())


def _assert_increasing(t):
    if t.device.type == 'cpu' and t.numel() <= 64:
        v = t.tolist()
        ok = all(b > a for a, b in zip(v, v[1:]))
    else:
        ok = bool((t[1:] > t[:-1]).all())
    assert ok, 't must be strictly increasing or decrasing'   # misc.py:159 (sic)


def _handle_unused_kwargs(solver, unused_kwargs):
    if len(unused_kwargs) > 0:
        warnings.warn('{}: Unexpected arguments {}'.format(solver.__class__.__name__, unused_kwargs))   # misc.py:178-181


_F32_DETOUR = {}


def _convert_to_tensor(a, dtype=None, device=None):
    """misc.py:137-144 for HOST scalars: a python float becomes float32 FIRST, then is cast - so 0.9 ends up
    as 0.8999999761581421 in float64.  Returns a numpy scalar."""
    if type(a) is float and dtype is np.float64:             # the controller constants of every call (safety, ifactor, dfactor)
        hit = _F32_DETOUR.get(a)
        if hit is None:
            hit = _F32_DETOUR[a] = np.float64(np.float32(a))
            if len(_F32_DETOUR) > 256:
                _F32_DETOUR.clear()
        return hit
    if isinstance(a, torch.Tensor):
        a = a.item()
    if isinstance(a, (float, np.floating)) and not isinstance(a, (np.float64, np.float32)):
        a = np.float32(a)
    elif isinstance(a, int):
        a = np.int32(a) if abs(a) < 2 ** 31 else np.int64(a)
    if dtype is not None:
        a = np.dtype(dtype).type(a)
    return a


# ---------------------------------------------------------------------------------------------
# controller scalars (host)
# ---------------------------------------------------------------------------------------------
def _ratio_from_norms(norms, n_total, rtol, atol, np_dtype):
    """misc.py:256-263 given the reduction record: tol is ONE scalar (F3); mean((err/tol)^2) = sum err^2 / (N tol^2)."""
    dt = np_dtype.type
    with np.errstate(all='ignore'):
        tol = dt(atol) + dt(rtol) * dt(max(norms[0], norms[1]))
        return dt(norms[2] / (float(n_total) * float(tol) * float(tol)))


def _compute_error_ratio(error_estimate, error_tol=None, rtol=None, atol=None, y0=None, y1=None, exchange=None, recs=None):
    """misc.py:250-264.  Returns a tuple of host scalars (state dtype), one per tuple component; the
    reductions run on device and come back in ONE host synchronisation."""
    assert error_tol is None, 'explicit error_tol is not supported'
    assert rtol is not None and atol is not None and y0 is not None and y1 is not None
    rtol = rtol if _is_iterable(rtol) else [rtol] * len(y0)
    atol = atol if _is_iterable(atol) else [atol] * len(y0)
    if recs is None:                            # (a captured graph of the attempt has produced them already)
        recs = torch.stack([_error_norms(e, a, b) for e, a, b in zip(error_estimate, y0, y1)])     # [ncomp, 4]
    counts = torch.tensor([[float(e.numel())] for e in error_estimate], dtype=torch.float64, device=recs.device)
    recs = torch.cat([recs, counts], dim=1)                                                    # + local N
    if exchange is not None:
        recs = exchange.combine(recs, max_slots=(0, 1, 3))
    host = recs.cpu().numpy()
    ratios = []
    for i, e in enumerate(error_estimate):
        ratios.append(_ratio_from_norms(host[i], host[i, 4], rtol[i], atol[i], _np_dtype(e.dtype)))
    _compute_error_ratio.last_nonfinite = bool((host[:, 3] != 0).any())
    return tuple(ratios)


_compute_error_ratio.last_nonfinite = False


def _optimal_step_size(last_step, mean_error_ratio, safety=0.9, ifactor=10.0, dfactor=0.2, order=5):
    """misc.py:267-287."""
    r = mean_error_ratio[0]
    for x in mean_error_ratio[1:]:           # python max(): first maximal element
        if x > r:
            r = x
    if r == 0:
        return np.float64(last_step * ifactor)
    if r < 1:
        dfactor = 1.0
    with np.errstate(all='ignore'):
        error_ratio = np.float64(np.sqrt(r))                 # sqrt in the ratio's dtype, then cast (:277-278)
        exponent = np.float64(np.float32(1. / order))        # float32 detour (:281-282, F4)
        factor = np.max([np.float64(1. / ifactor), np.min([error_ratio ** exponent / safety, np.float64(1. / dfactor)])])
        return np.float64(last_step / factor)


def _norm_from_sumsq(sumsq, n_total, np_dtype):
    """misc.py:170-175: ||x|| / numel**0.5 in the state dtype."""
    dt = np_dtype.type
    with np.errstate(all='ignore'):
        return dt(np.sqrt(dt(sumsq))) / (dt(n_total) ** 0.5)


def _pymax(seq):
    best = seq[0]
    for x in seq[1:]:
        if x > best:
            best = x
    return best


def _select_initial_step(fun, t0, y0, order, rtol, atol, f0=None, exchange=None):
    """misc.py:183-247 (Hairer II.4).  `fun` takes/returns tuples; rtol/atol scalars.  Returns a host scalar."""
    like = y0[0]
    npdt = _np_dtype(like.dtype)
    dt_ = npdt.type
    t0 = dt_(t0)
    if f0 is None:
        f0 = fun(_scalar_tensor(t0, like), y0)

    def norms(xs, subs):
        recs = torch.cat([_scaled_sumsq(x, s, y_, rtol, atol) for x, s, y_ in zip(xs, subs, y0)])
        counts = torch.tensor([float(x.numel()) for x in xs], dtype=torch.float64, device=recs.device)
        both = torch.stack([recs, counts], dim=1)
        if exchange is not None:
            both = exchange.combine(both, max_slots=())
        h = both.cpu().numpy()
        return tuple(_norm_from_sumsq(h[i, 0], h[i, 1], npdt) for i in range(len(xs)))

    none = [None] * len(y0)
    d0 = norms(y0, none)
    d1 = norms(f0, none)
    with np.errstate(all='ignore'):
        if _pymax(d0) < 1e-5 or _pymax(d1) < 1e-5:
            h0 = dt_(1e-6)
        else:
            h0 = dt_(0.01) * _pymax([a / b for a, b in zip(d0, d1)])
        y1 = tuple(_lincomb(y0_, [1.0], [f0_], h0) for y0_, f0_ in zip(y0, f0))
        f1 = fun(_scalar_tensor(t0 + h0, like), y1)
        d2 = tuple(d / h0 for d in norms(f1, f0))
        if _pymax(d1) <= 1e-15 and _pymax(d2) <= 1e-15:
            h1 = np.max([dt_(1e-6), h0 * dt_(1e-3)])
        else:
            h1 = (dt_(0.01) / _pymax(list(d1) + list(d2))) ** dt_(1. / float(order + 1))
        return np.min([dt_(100) * h0, h1]).astype(npdt)


# ---------------------------------------------------------------------------------------------
# input checking (misc.py:290-329)
# ---------------------------------------------------------------------------------------------
class _TupleFunc(object):
    """func(t, y) for a bare tensor state, lifted to the tuple contract (misc.py:301-303)."""

    def __init__(self, base):
        self.base = base
        self.device_rhs = base if getattr(base, 'kind', 0) else None
        self._mi_no_capture = getattr(base, '_mi_no_capture', False)

    def __call__(self, t, y):
        return (self.base(t, y[0]),)


class _ReverseFunc(object):
    """misc.py:318-321: t <- -t, func <- -f(-t, y)."""

    def __init__(self, base):
        self.base = base
        rhs = getattr(base, 'device_rhs', None)
        self.device_rhs = rhs.reversed() if rhs is not None else None
        self.per_component = getattr(base, 'per_component', False)
        self._mi_no_capture = getattr(base, '_mi_no_capture', False)

    def __call__(self, t, y):
        return tuple(-f_ for f_ in self.base(-t, y))


def _as_time_tensor(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu()
    return torch.as_tensor(np.asarray(t))


def _check_inputs(func, y0, t):
    tensor_input = False
    if isinstance(y0, torch.Tensor):
        tensor_input = True
        y0 = (y0,)
        func = _TupleFunc(func)
    assert isinstance(y0, tuple), 'y0 must be either a tf.Tensor or a tuple'
    for y0_ in y0:
        assert isinstance(y0_, torch.Tensor), 'each element must be a tf.Tensor but received {}'.format(type(y0_))
    t = _as_time_tensor(t)
    if _decreasing(t):
        t = -t
        # misc.py:318-321: func <- -func(-t, y).  A callable that can form that itself (`_mi_time_reversed()`: e.g. the linear system's
        # augmented dynamics, whose kernels take the sign as a scale factor) spares the wrapper's negation pass over every component
        native = getattr(func, '_mi_time_reversed', None)     # (a private name: a user's own `time_reversed` attribute means nothing here)
        func = native() if callable(native) else _ReverseFunc(func)
    for y0_ in y0:
        if not torch.is_floating_point(y0_):
            raise TypeError('`y0` must be a floating point Tensor but is a {}'.format(y0_.dtype))
    if not torch.is_floating_point(t):
        raise TypeError('`t` must be a floating point Tensor but is a {}'.format(t.dtype))
    return tensor_input, func, y0, t

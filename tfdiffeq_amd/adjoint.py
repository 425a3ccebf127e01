"""Mirror of tfdiffeq/adjoint.py: `odeint_adjoint` - O(1)-memory gradients by solving the augmented ODE backwards
(the main caller of the hot path in training).

Same algorithm and argument contract as the reference (adjoint.py:35-224), re-stated for torch autograd:
  * `func` must be a module (reference: tf.keras.Model; here torch.nn.Module) so its parameters can be found;
  * forward = `odeint` (any engine); backward loops i = T-1 .. 1 over the output times, building the heterogeneous
    tuple state (*y_i, *adj_y, adj_time, adj_params) (adjoint.py:148) and calling `odeint` on it over [t_i, t_{i-1}]
    (reversed time, handled by `_check_inputs` as in the reference) - that runs on the plane-kernel engine;
  * vector-Jacobian products of `func` come from torch.autograd.grad (reference: tf.GradientTape, adjoint.py:76-95).
Unlike the reference there is no module-global `_arguments` (adjoint.py:32, 217): the call is re-entrant.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _native as N
from .odeint import odeint

# The backward solve of the ODEFunc MLP (relu, softplus or tanh) runs as ONE kernel launch per output interval (csrc/mi_ode_adjoint.h) when the
# problem qualifies (`_fused_plan`); FUSED = False (or TFDIFFEQ_AMD_FUSED_ADJOINT=0) keeps every case on the plane kernels.
FUSED = os.environ.get('TFDIFFEQ_AMD_FUSED_ADJOINT', '1') != '0'
FUSED_FORWARD = True          # the forward solve under odeint_adjoint also takes the fused kernels of a network that has them
_ENGINES = {}


class HandoffTimeout(RuntimeError):
    """The fused adjoint kernel's in-kernel grid hand-off timed out (the GPU is shared with another persistent kernel).  Nothing
    was committed: the caller may take the generic path.  Every other native failure propagates."""


def _flatten(seq):
    flat = [p.reshape(-1) for p in seq]
    return torch.cat(flat) if len(flat) > 0 else torch.tensor([])


class _FlatParams(torch.autograd.Function):
    """torch.cat of the trainable tensors, whose backward hands a slice back only to the tensors the backward solve actually differentiated
    through.  For a PLAIN CALLABLE the differentiation set is the union of what one probe evaluation depends on and what the callable can
    name (odeint._graph_leaves: a branch the probe did not take must not lose its gradient); a merely-named tensor (`usage['optional']`)
    that NO evaluation of the augmented dynamics reached gets None, not zeros (an optimizer with weight decay would move it) -
    `usage['used']`, filled by the generic backward.  A module's own parameters always get their gradient, zeros if f does not depend on
    them (the reference asks for UnconnectedGradients.ZERO, adjoint.py:83-95)."""

    @staticmethod
    def forward(ctx, usage, *params):
        ctx.usage = usage
        ctx.shapes = [tuple(p.shape) for p in params]
        return torch.cat([p.reshape(-1) for p in params])

    @staticmethod
    def backward(ctx, g):
        used = ctx.usage.get('used')
        out, off = [], 0
        for i, shp in enumerate(ctx.shapes):
            n = 1
            for d in shp:
                n *= d
            skip = used is not None and i < len(used) and not used[i] and i in ctx.usage.get('optional', ())
            out.append(None if (g is None or skip) else g[off:off + n].reshape(shp))
            off += n
        return (None,) + tuple(out)


class _FusedAdjointEngine(object):
    """Owns one mi_ode_adjoint handle: the augmented system (y, adj_y, adj_t, adj_params) of adjoint.py:57-178 for a
    [batch, dim] float32 state and the dim -> hidden -> hidden -> dim MLP (rhs.MLP: relu, softplus or tanh; time_dependent:
    the first layer sees concat([t, x]), dense_odenet.py:79-84 - adj_params then starts with w_t, the row of W1 that multiplies t)."""

    def __init__(self, batch, dim, hidden, rtol, atol, safety, ifactor, dfactor, max_num_steps, device, time_dependent=False):
        from .dopri5 import _DORMAND_PRINCE_SHAMPINE_TABLEAU, DPS_C_MID
        from .solvers import _fill_tableau
        self.lib = N.load()
        self.device = torch.device(device)
        d = N.AdjointDesc()
        d.batch, d.dim, d.hidden = int(batch), int(dim), int(hidden)
        _fill_tableau(d.tableau, _DORMAND_PRINCE_SHAMPINE_TABLEAU, DPS_C_MID)
        d.rtol, d.atol = float(rtol), float(atol)
        d.safety, d.ifactor, d.dfactor = float(safety), float(ifactor), float(dfactor)
        d.order, d.init_order = 5, 4                     # dopri5.py:68, 74
        d.max_num_steps = int(max_num_steps)
        d.time_dependent = 1 if time_dependent else 0
        self.desc = d
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            N.check(self.lib.mi_ode_adjoint_create(C.byref(d), C.byref(h)), 'mi_ode_adjoint_create')
        self.h = h
        self.batch, self.dim = int(batch), int(dim)
        self.n_params = int(self.lib.mi_ode_adjoint_num_params(h))
        self.stats = N.Stats()

    def close(self):
        if getattr(self, 'h', None):
            self.lib.mi_ode_adjoint_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _rhs(self, mlp):
        r = N.Rhs()
        keep = mlp.fill(r, torch.float32, self.device)
        return r, keep

    def segment(self, mlp, y, adj_y, adj_t, adj_params, t_start, t_end):
        """odeint(augmented_dynamics, (y, adj_y, adj_t, adj_params), [t_start, t_end])[..][1] (adjoint.py:148-160) without
        the y component.  All tensors float32 on the device; adj_params in canonical order."""
        r, keep = self._rhs(mlp)
        y, adj_y = y.contiguous(), adj_y.contiguous()
        a_out = torch.empty_like(adj_y)
        t_out = torch.empty_like(adj_t)
        p_out = torch.empty_like(adj_params)
        with torch.cuda.device(self.device):
            rc = N.check(self.lib.mi_ode_adjoint_segment(
                self.h, C.byref(r), y.data_ptr(), adj_y.data_ptr(), adj_t.data_ptr(), adj_params.data_ptr(), float(t_start), float(t_end),
                None, a_out.data_ptr(), t_out.data_ptr(), p_out.data_ptr(), C.byref(self.stats), N.stream_ptr(self.device)),
                'mi_ode_adjoint_segment')
        del keep
        if rc != 0:
            if rc & N.ST_SYNC_TIMEOUT:
                raise HandoffTimeout(N.status_message(rc))
            msg = N.status_message(rc)
            if rc & N.ST_MAX_STEPS:
                msg = 'max_num_steps exceeded ({}>={})'.format(self.desc.max_num_steps, self.desc.max_num_steps)
            if rc & N.ST_DT_UNDERFLOW:
                msg = 'underflow in dt {}'.format(self.stats.dt)
            raise AssertionError(msg)                      # what the reference's solver raises (dopri5.py:85-100)
        return a_out, t_out, p_out

    def dynamics(self, mlp, y, adj_y, t=0.0):
        """One evaluation of the augmented dynamics at time t (adjoint.py:69-105): (f, -adj_y^T df/dy, -adj_y^T df/dparams)."""
        r, keep = self._rhs(mlp)
        y, adj_y = y.contiguous(), adj_y.contiguous()
        f, vy = torch.empty_like(y), torch.empty_like(y)
        vp = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            N.check(self.lib.mi_ode_adjoint_dynamics_at(self.h, C.byref(r), float(t), y.data_ptr(), adj_y.data_ptr(), f.data_ptr(), vy.data_ptr(),
                                                        vp.data_ptr(), N.stream_ptr(self.device)), 'mi_ode_adjoint_dynamics_at')
        del keep
        return f, vy, vp


def _cached_adjoint_engine(*key):
    eng = _ENGINES.get(key)
    if eng is None:
        while len(_ENGINES) >= 4:
            _ENGINES.pop(next(iter(_ENGINES))).close()
        eng = _ENGINES[key] = _FusedAdjointEngine(*key)
    return eng


def clear_adjoint_engines():
    while _ENGINES:
        _ENGINES.pop(next(iter(_ENGINES))).close()
    while _LIN_ENGINES:
        _LIN_ENGINES.pop(next(iter(_LIN_ENGINES))).close()


class _LinearAdjointEngine(object):
    """Owns one mi_ode_linadj handle: the augmented system (y, adj_y, adj_t, adj_params) of adjoint.py:57-178 for f = y W + b on a
    [batch, dim] float32 / float64 state, ONE launch per output interval (csrc/mi_ode_linadj.h)."""

    def __init__(self, batch, dim, dtype, rtol, atol, safety, ifactor, dfactor, max_num_steps, device):
        from .dopri5 import _DORMAND_PRINCE_SHAMPINE_TABLEAU, DPS_C_MID
        from .solvers import _fill_tableau
        self.lib = N.load()
        self.device = torch.device(device)
        self.dtype = dtype
        d = N.LinAdjDesc()
        d.batch, d.dim, d.dtype = int(batch), int(dim), N.dtype_code(dtype)
        _fill_tableau(d.tableau, _DORMAND_PRINCE_SHAMPINE_TABLEAU, DPS_C_MID)
        d.rtol, d.atol = float(rtol), float(atol)
        d.safety, d.ifactor, d.dfactor = float(safety), float(ifactor), float(dfactor)
        d.order, d.init_order = 5, 4                     # dopri5.py:68, 74
        d.max_num_steps = int(max_num_steps)
        self.desc = d
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            N.check(self.lib.mi_ode_linadj_create(C.byref(d), C.byref(h)), 'mi_ode_linadj_create')
        self.h = h
        self.batch, self.dim = int(batch), int(dim)
        self.stats = N.Stats()
        self._scalars = (C.c_double * 2)()
        self.scalars = (0.0, 0.0)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.mi_ode_linadj_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def segment(self, W, b, y, adj_y, adj_t, adj_params, t_start, t_end, grad_out=None):
        """odeint(augmented_dynamics, (y, adj_y, adj_t, adj_params), [t_start, t_end])[..][1] (adjoint.py:148-160) without the y
        component.  W [dim, dim] in [in, out] layout, b [dim] or None; every tensor in the state dtype on the device.  grad_out: the
        gradient at t_start - the kernel then subtracts f(t_start, y) . grad_out from adj_t first (adjoint.py:134-140; its own first evaluation
        IS f(t_start, y)) and `self.scalars` holds (that dot product, adj_t(t_end)) as Python floats."""
        y, adj_y = y.contiguous(), adj_y.contiguous()
        a_out = torch.empty_like(adj_y)
        t_out = torch.empty_like(adj_t)
        p_out = torch.empty_like(adj_params)
        if grad_out is not None:
            grad_out = grad_out.contiguous()
        with torch.cuda.device(self.device):
            rc = N.check(self.lib.mi_ode_linadj_segment(
                self.h, W.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(), adj_y.data_ptr(), adj_t.data_ptr(),
                adj_params.data_ptr(), grad_out.data_ptr() if grad_out is not None else None, float(t_start), float(t_end), a_out.data_ptr(),
                t_out.data_ptr(), p_out.data_ptr(), None, self._scalars, C.byref(self.stats), N.stream_ptr(self.device)), 'mi_ode_linadj_segment')
        self.scalars = (float(self._scalars[0]), float(self._scalars[1]))
        if rc != 0:
            if rc & N.ST_SYNC_TIMEOUT:
                raise HandoffTimeout(N.status_message(rc))
            msg = N.status_message(rc)
            if rc & N.ST_MAX_STEPS:
                msg = 'max_num_steps exceeded ({}>={})'.format(self.desc.max_num_steps, self.desc.max_num_steps)
            if rc & N.ST_DT_UNDERFLOW:
                msg = 'underflow in dt {}'.format(self.stats.dt)
            raise AssertionError(msg)                      # what the reference's solver raises (dopri5.py:85-100)
        return a_out, t_out, p_out

    def profile(self):
        out = (C.c_double * 8)()
        N.check(self.lib.mi_ode_linadj_profile(self.h, out), 'mi_ode_linadj_profile')
        keys = ('tile_passes_us', 'theta_combinations_us', 'attempt_handoffs_us', 'slab_passes_us', 'small_products_us', 'prologue_us',
                'epilogue_us', 'handoffs')
        return dict(zip(keys, list(out)))


_LIN_ENGINES = {}


def _cached_linear_adjoint_engine(*key):
    eng = _LIN_ENGINES.get(key)
    if eng is None:
        eng = _LinearAdjointEngine(*key)                 # (evict only after a successful create: a refusal must not cost live engines)
        while len(_LIN_ENGINES) >= 4:
            _LIN_ENGINES.pop(next(iter(_LIN_ENGINES))).close()
        _LIN_ENGINES[key] = eng
    return eng


def canonical_to_module_order(base, theta):
    """adj_params in the kernel's canonical order (W1 [in,out], b1, W2, b2, W3, b3; the time-dependent W1 is [1 + dim, hidden]
    with the row of t first, as fc1 sees concat([t, x])) -> the flat order of base.parameters() (torch.nn.Linear keeps
    [out, in] weights)."""
    din, hd, d = base.fc1.in_features, base.fc1.out_features, base.fc3.out_features
    sizes = [din * hd, hd, hd * hd, hd, hd * d, d]
    w1, b1, w2, b2, w3, b3 = torch.split(theta, sizes)
    return torch.cat([w1.reshape(din, hd).t().reshape(-1), b1, w2.reshape(hd, hd).t().reshape(-1), b2,
                      w3.reshape(hd, d).t().reshape(-1), b3])


def _fused_plan(func, n_tensors, cfg, like, f_params):
    """(engine, MLPTanh descriptor, base module) if the backward solve can run on the fused kernel, else None."""
    if not FUSED or n_tensors != 1 or not isinstance(func, _TupleModule):
        return None
    base = func.base_func
    get = getattr(base, 'device_rhs', None)
    if get is None or cfg['adjoint_method'] not in (None, 'dopri5'):
        return None
    if not (like.is_cuda and like.dtype == torch.float32 and like.dim() >= 2):
        return None
    try:
        layers = (base.fc1, base.fc2, base.fc3)
    except AttributeError:
        return None
    want = [p for l in layers for p in (l.weight, l.bias)]
    if len(f_params) != 6 or any(a is not b for a, b in zip(f_params, want)):
        return None                                      # frozen / extra parameters: the generic path handles those
    opts = dict(cfg['adjoint_options'] or {})
    max_num_steps = opts.pop('max_num_steps', 2 ** 31 - 1)
    if opts or isinstance(cfg['adjoint_rtol'], (tuple, list)) or isinstance(cfg['adjoint_atol'], (tuple, list)):
        return None                                      # first_step / safety / ... : not wired into the fused controller
    mlp = get()
    y1 = like[0]
    if mlp is None or not mlp.supports(y1):
        return None
    batch = y1.numel() // y1.shape[-1]
    if batch < 1:
        return None
    f32 = lambda v: float(np.float32(v))                 # noqa: E731  (misc.py:137-144: python float -> float32 -> float64)
    try:
        eng = _cached_adjoint_engine(batch, int(y1.shape[-1]), int(mlp.hidden), float(cfg['adjoint_rtol']), float(cfg['adjoint_atol']),
                                     f32(0.9), f32(10.0), f32(0.2), int(max_num_steps), str(like.device), bool(mlp.time_dependent))
    except N.NativeError as e:                           # e.g. no memory for the activation scratch (15 KB per row): the generic path
        import warnings                                  # needs none
        warnings.warn('fused adjoint engine unavailable (%s): using the plane-kernel path' % e)
        return None
    return eng, mlp, base


def _linear_plan(func, n_tensors, cfg, like, f_params):
    """The `models.LinearODEFunc` whose backward solve can use the MFMA kernels for its augmented dynamics, else None."""
    if not FUSED or n_tensors != 1 or not isinstance(func, _TupleModule):
        return None
    from .models import LinearODEFunc
    base = func.base_func
    if not isinstance(base, LinearODEFunc) or not (1 <= base.dim <= 128):
        return None
    y1 = like[0]
    if not (like.is_cuda and like.dtype in (torch.float32, torch.float64) and y1.dim() >= 1 and y1.shape[-1] == base.dim):
        return None
    want = [base.weight] + ([base.bias] if base.bias is not None else [])
    if len(f_params) != len(want) or any(a is not b for a, b in zip(f_params, want)):
        return None                                      # a frozen parameter: the generic path
    if any(p.dtype != like.dtype or p.device != like.device or not p.is_contiguous() for p in want):
        return None
    return base


LINEAR_ONE_LAUNCH = os.environ.get('TFDIFFEQ_AMD_LINEAR_ADJOINT', '1') != '0'   # False: the augmented dynamics on the MFMA kernels, one
                                                                                 # Python evaluation per stage (round 4's path)


def _linear_one_launch_plan(base, cfg, like):
    """The mi_ode_linadj engine for this problem, or None (then `_linear_dynamics` serves it on the callable engine): dopri5 with
    scalar tolerances and no solver options beyond max_num_steps - what the kernel's controller implements."""
    if not LINEAR_ONE_LAUNCH or cfg['adjoint_method'] not in (None, 'dopri5'):
        return None
    opts = dict(cfg['adjoint_options'] or {})
    max_num_steps = opts.pop('max_num_steps', 2 ** 31 - 1)
    if opts or isinstance(cfg['adjoint_rtol'], (tuple, list)) or isinstance(cfg['adjoint_atol'], (tuple, list)):
        return None
    y1 = like[0]
    batch = y1.numel() // base.dim
    if batch < 1:
        return None
    f32 = lambda v: float(np.float32(v))                 # noqa: E731  (misc.py:137-144: python float -> float32 -> float64)
    key = (batch, int(base.dim), like.dtype, float(cfg['adjoint_rtol']), float(cfg['adjoint_atol']), f32(0.9), f32(10.0), f32(0.2),
           int(max_num_steps), str(like.device))
    if key in _NO_LIN_ENGINE:                            # a permanent refusal (the kernel does not fit this shape): asked once, remembered
        return None
    try:
        return _cached_linear_adjoint_engine(*key)
    except N.NativeError as e:
        import warnings
        if getattr(e, 'rc', None) == N.E_INVALID:
            if len(_NO_LIN_ENGINE) > 256:
                _NO_LIN_ENGINE.clear()
            _NO_LIN_ENGINE.add(key)
        warnings.warn('one-launch linear adjoint engine unavailable (%s): using the callable engine' % e)
        return None


_NO_LIN_ENGINE = set()


def _linear_dynamics(base, like):
    """adjoint.py:69-105 for f = y W + b without a tape: (f, -a W^T, 0, (-(y^T a), -sum_rows a)).  The two state-sized products are
    the linear right-hand side itself (mi_ode_eval_rhs on the MFMA stage kernel: W, and -W^T kept in a buffer of its own that is
    refreshed per backward pass), the parameter part is mi_ode_outer_reduce.  Nothing in it synchronises or records autograd, so the
    device-controlled engine replays an attempt of the augmented system as one hipGraph."""
    from . import rhs as R
    from .fixed_grid import Euler
    from .solvers import _FusedEngine, _cached_engine, _tableau_key
    lib = N.load()
    y1 = like[0]
    dim = base.dim
    batch = y1.numel() // dim
    W = base.weight.detach()
    st = getattr(base, '_adjoint_state', None)
    if st is None or st['wt'].dtype != W.dtype or st['wt'].device != W.device:
        wt = torch.empty_like(W)
        st = {'wt': wt, 'rhs_a': R.Linear(wt)}
        object.__setattr__(base, '_adjoint_state', st)
    st['wt'].copy_(-W.t())
    proto = y1.reshape(batch, dim)

    def engine(r):
        key = ('rhs evaluation', r.cache_key(proto.dtype, proto.device), (batch, dim), proto.dtype, str(proto.device),
               _tableau_key(Euler._fused_tableau, None))
        return _cached_engine(key, lambda: _FusedEngine(r, proto, False, Euler._fused_tableau))
    rhs_y, rhs_a = base.device_rhs(), st['rhs_a']
    code = N.dtype_code(like.dtype)
    ws = torch.empty(int(lib.mi_ode_outer_workspace_bytes(code, batch, dim)), dtype=torch.uint8, device=like.device)
    has_b = base.bias is not None
    n_w, elt = dim * dim, like.element_size()

    class _Dynamics(object):
        """sign = +1: the augmented dynamics; sign = -1: -dynamics(-t, .) (misc.py:318-321) - the system is autonomous, so the time
        reversal of `odeint` is the sign alone: the stage kernel multiplies by it (`DeviceRHS.reversed()`), the outer product takes it
        as its scale factor, and the wrapper's negation pass over the two state-sized components is not needed."""

        def __init__(self, sign):
            self.sign = sign
            self.eng_y = self.eng_a = None                   # created on first use: a backward pass only ever runs one direction

        def _mi_time_reversed(self):
            return _Dynamics(-self.sign)

        def __call__(self, tt, y_aug):
            if self.eng_y is None:
                self.eng_y = engine(rhs_y if self.sign > 0 else rhs_y.reversed())
                self.eng_a = engine(rhs_a if self.sign > 0 else rhs_a.reversed())
            y, a = y_aug[0].contiguous(), y_aug[1].contiguous()
            fy = self.eng_y.eval_rhs(y.reshape(batch, dim)).reshape(y.shape)
            va = self.eng_a.eval_rhs(a.reshape(batch, dim)).reshape(a.shape)
            vth = torch.empty(n_w + (dim if has_b else 0), dtype=like.dtype, device=like.device)
            with torch.cuda.device(like.device):
                N.check(lib.mi_ode_outer_reduce(code, batch, dim, y.data_ptr(), a.data_ptr(), -float(self.sign), vth.data_ptr(),
                                                vth.data_ptr() + n_w * elt if has_b else None, ws.data_ptr(), N.stream_ptr(like.device)),
                        'mi_ode_outer_reduce')
            return (fy, va, torch.zeros_like(y_aug[2]), vth)
    return _Dynamics(1.0)


def _trainable(func):
    """The tensors the backward solve differentiates with respect to: the module's grad-requiring parameters, then the bare
    grad-requiring tensors a wrapped plain callable closes over (odeint._callable_module), in a fixed order."""
    ps = [p for p in func.parameters() if p.requires_grad]
    base = getattr(func, 'base_func', func)
    have = {id(p) for p in ps}
    for e in getattr(base, '_mi_extra_params', ()):
        if e.requires_grad and id(e) not in have:
            ps.append(e)
            have.add(id(e))
    return ps


class _TupleModule(torch.nn.Module):
    """adjoint.py:203-211."""

    def __init__(self, base_func):
        super(_TupleModule, self).__init__()
        self.base_func = base_func

    def forward(self, t, y):
        return (self.base_func(t, y[0]),)


def _count_nfe(func, n):
    """Add kernel-side evaluations of f to the wrapped module's `nfe` counter, when it keeps one (models.ODEFunc)."""
    base = getattr(func, 'base_func', func)
    if hasattr(base, 'nfe'):
        try:
            base.nfe += int(n)
        except Exception:
            pass


class _OdeintAdjointMethod(torch.autograd.Function):

    @staticmethod
    def forward(ctx, func, n_tensors, cfg, t, flat_params, *y0):
        ctx.func, ctx.cfg, ctx.n_tensors = func, cfg, n_tensors
        with torch.no_grad():
            fwd, state = func, tuple(y0)
            if FUSED and FUSED_FORWARD and n_tensors == 1 and isinstance(func, _TupleModule):
                # a network with a device descriptor (models.ODEFunc) takes the fused forward kernels here too: a training
                # step then is one launch forward and one per interval backward
                get = getattr(func.base_func, 'device_rhs', None)
                mlp = get() if get is not None else None
                if mlp is not None and y0[0].is_cuda and mlp.supports(y0[0]):
                    fwd, state = mlp, y0[0]
            ans = odeint(fwd, state, t, rtol=cfg['rtol'], atol=cfg['atol'], method=cfg['method'], options=cfg['options'])
            if fwd is not func:                          # f ran inside the kernel: keep the module's evaluation counter honest
                _count_nfe(func, odeint.last_stats.get('nfe', 0))       # (dense_odenet.py:38, 78: users log odefunc.nfe)
            # how the forward pass ran (round 6: a plain module is traced and lowered onto the fused kernels by odeint itself,
            # tfdiffeq_amd/lower.py); the backward of such a callable stays on the generic path - autograd through f - this round
            fstats = odeint.last_stats if isinstance(odeint.last_stats, dict) else {}
            ctx.forward_info = dict(fstats.get('lower') or {'lowered': False}, engine=fstats.get('engine'), n_launches=fstats.get('n_launches'))
            if isinstance(ans, torch.Tensor):
                ans = (ans,)
        ctx.save_for_backward(t, flat_params, *ans)
        return ans

    @staticmethod
    def backward(ctx, *grad_output):
        try:
            return _OdeintAdjointMethod._backward(ctx, *grad_output)
        finally:
            if isinstance(odeint_adjoint.last_backward_stats, dict):
                odeint_adjoint.last_backward_stats['forward'] = getattr(ctx, 'forward_info', None)

    @staticmethod
    def _backward(ctx, *grad_output):
        func, cfg, n_tensors = ctx.func, ctx.cfg, ctx.n_tensors
        t, flat_params, *ans = ctx.saved_tensors
        f_params = tuple(_trainable(func))
        like = ans[0]
        grad_output = tuple(g if g is not None else torch.zeros_like(a) for g, a in zip(grad_output, ans))
        plan = _fused_plan(func, n_tensors, cfg, like, f_params)
        if plan is not None:
            try:
                return _OdeintAdjointMethod._fused_backward(plan, func, t, flat_params, ans, grad_output, like)
            except HandoffTimeout as e:                  # the in-kernel hand-off timed out (the GPU is shared with another
                import warnings                          # persistent kernel): nothing was committed, take the generic path.
                                                         # (Only that: any other native / HIP failure propagates.)
                warnings.warn('fused adjoint kernel unavailable (%s): falling back to the plane-kernel path' % e)
        lin = _linear_plan(func, n_tensors, cfg, like, f_params)
        eng = _linear_one_launch_plan(lin, cfg, like) if lin is not None else None
        if eng is not None:
            try:
                return _OdeintAdjointMethod._linear_backward(eng, lin, func, t, flat_params, ans, grad_output, like)
            except HandoffTimeout as e:
                import warnings
                warnings.warn('one-launch linear adjoint kernel unavailable (%s): falling back to the callable engine' % e)
        if lin is not None:
            res = _OdeintAdjointMethod._generic_backward(func, cfg, n_tensors, t, flat_params, ans, grad_output, f_params, like,
                                                         augmented_dynamics=_linear_dynamics(lin, like))
            odeint_adjoint.last_backward_stats = {'engine': 'linear right-hand side: augmented dynamics on the MFMA kernels (' +
                                                  str(odeint.last_stats.get('engine', 'plane kernels')) + ')',
                                                  'last_segment': dict(odeint.last_stats)}
            return res
        odeint_adjoint.last_backward_stats = {'engine': 'plane kernels'}
        return _OdeintAdjointMethod._generic_backward(func, cfg, n_tensors, t, flat_params, ans, grad_output, f_params, like)

    @staticmethod
    def _augmented_dynamics(func, n_tensors, f_params, like):
        used = [False] * len(f_params)

        def augmented_dynamics(tt, y_aug):
            # dynamics of the original system augmented with the adjoint wrt y, t and the parameters (adjoint.py:69-105)
            y, adj_y = y_aug[:n_tensors], y_aug[n_tensors:2 * n_tensors]
            with torch.enable_grad():
                tt_ = tt.detach().requires_grad_(True)
                y_ = tuple(v.detach().requires_grad_(True) for v in y)
                func_eval = func(tt_, y_)
                # a derivative with no autograd dependence on (t, y, params) - a constant or forcing-only RHS - has zero
                # vjps (the reference asks for UnconnectedGradients.ZERO, adjoint.py:83-95); autograd.grad would raise
                live = [(f_, -a) for f_, a in zip(func_eval, adj_y) if f_.requires_grad]
                if live:
                    vjp = torch.autograd.grad([f_ for f_, _ in live], (tt_,) + y_ + f_params, [a for _, a in live],
                                              allow_unused=True, retain_graph=False)
                else:
                    vjp = (None,) * (1 + n_tensors + len(f_params))
            vjp_t, vjp_y, vjp_params = vjp[0], vjp[1:1 + n_tensors], vjp[1 + n_tensors:]
            for i_, g_ in enumerate(vjp_params):           # which parameters an evaluation reached at all (_FlatParams)
                if g_ is not None:
                    used[i_] = True
            vjp_t = torch.zeros_like(tt) if vjp_t is None else vjp_t
            vjp_y = tuple(torch.zeros_like(v) if g is None else g for g, v in zip(vjp_y, y))
            vjp_params = _flatten([torch.zeros_like(p) if g is None else g for g, p in zip(vjp_params, f_params)])
            if len(f_params) == 0:
                vjp_params = torch.zeros((), dtype=like.dtype, device=like.device)
            return (*[f.detach() for f in func_eval], *vjp_y, vjp_t.to(like.dtype), vjp_params.to(like.dtype))

        # torch.autograd.grad cannot run under hipGraph stream capture (the capture aborts inside the autograd engine and takes the
        # process with it): the device-controlled engine must never record these dynamics - whatever its saved-tensor heuristic sees
        # (an f whose backward saves nothing, `return -y`, looks capture-safe to it)
        augmented_dynamics._mi_no_capture = True
        augmented_dynamics.used = used
        return augmented_dynamics

    @staticmethod
    def _fused_backward(plan, func, t, flat_params, ans, grad_output, like):
        eng, mlp, base = plan
        T = ans[0].shape[0]
        segs = []
        with torch.no_grad():
            g_out = grad_output[0]
            adj_y = g_out[-1].contiguous()
            theta = torch.zeros(eng.n_params, dtype=torch.float32, device=like.device)
            adj_time = torch.zeros((), dtype=torch.float32, device=like.device)
            time_vjps = []
            t_dev = t.to(device=like.device)
            for i in range(T - 1, 0, -1):
                func_i = func(t_dev[i].to(like.dtype), (ans[0][i],))[0]
                dLd_cur_t = torch.dot(func_i.reshape(-1), g_out[i].reshape(-1))              # adjoint.py:134-140
                adj_time = adj_time - dLd_cur_t
                time_vjps.append(dLd_cur_t.reshape(1))
                adj_y, adj_time, theta = eng.segment(mlp, ans[0][i], adj_y, adj_time, theta, float(t[i]), float(t[i - 1]))
                segs.append(eng.stats.as_dict())
                _count_nfe(func, segs[-1].get('nfe', 0))
                adj_y = adj_y + g_out[i - 1]
            time_vjps.append(adj_time.reshape(1))
            time_vjps = torch.cat(time_vjps[::-1]).to(dtype=t.dtype, device=t.device)
            grad_params = canonical_to_module_order(base, theta).to(flat_params.dtype)
        odeint_adjoint.last_backward_stats = {'engine': 'fused adjoint kernel (one launch per interval)', 'segments': segs}
        return (None, None, None, time_vjps, grad_params, adj_y)

    @staticmethod
    def _linear_backward(eng, base, func, t, flat_params, ans, grad_output, like):
        """adjoint.py:117-178 with every interval's odeint call - and the time gradient of its start point (adjoint.py:134-140) - as ONE
        launch of mi_ode_linadj_segment."""
        T = ans[0].shape[0]
        dim = base.dim
        y_shape = ans[0][0].shape
        batch = ans[0][0].numel() // dim
        W = base.weight.detach()
        b = base.bias.detach() if base.bias is not None else None
        segs = []
        with torch.no_grad():
            g_out = grad_output[0]
            adj_y = g_out[-1].reshape(batch, dim)
            theta = torch.zeros(dim * dim + (dim if b is not None else 0), dtype=like.dtype, device=like.device)
            adj_time = torch.zeros((), dtype=like.dtype, device=like.device)
            time_vjps = []                                   # (host floats: the kernel hands f(t_i, y_i) . grad_output_i and adj_t(t_end) back)
            for i in range(T - 1, 0, -1):
                y_i = ans[0][i].reshape(batch, dim)
                # adjoint.py:134-140 (dLd_cur_t, adj_time -= dLd_cur_t) happens inside the segment: f(t_i, y_i) is its first evaluation
                adj_y, adj_time, theta = eng.segment(W, b, y_i, adj_y, adj_time, theta, float(t[i]), float(t[i - 1]), grad_out=g_out[i].reshape(batch, dim))
                time_vjps.append(eng.scalars[0])
                segs.append(eng.stats.as_dict())
                _count_nfe(func, segs[-1].get('nfe', 0))
                if i > 1:
                    adj_y = adj_y + g_out[i - 1].reshape(batch, dim)
                else:
                    adj_y.add_(g_out[0].reshape(batch, dim))
            time_vjps.append(eng.scalars[1] if T > 1 else 0.0)
            time_vjps = torch.tensor(time_vjps[::-1], dtype=t.dtype, device=t.device)
            grad_params = theta.to(flat_params.dtype)
        odeint_adjoint.last_backward_stats = {'engine': 'linear right-hand side: one launch per interval (mi_ode_linadj)', 'segments': segs,
                                              'last_segment': segs[-1] if segs else {}}
        return (None, None, None, time_vjps, grad_params, adj_y.reshape(y_shape))

    @staticmethod
    def _generic_backward(func, cfg, n_tensors, t, flat_params, ans, grad_output, f_params, like, augmented_dynamics=None):
        T = ans[0].shape[0]
        if augmented_dynamics is None:
            augmented_dynamics = _OdeintAdjointMethod._augmented_dynamics(func, n_tensors, f_params, like)
        with torch.no_grad():
            adj_y = tuple(g[-1] for g in grad_output)
            adj_params = torch.zeros_like(flat_params, dtype=like.dtype) if flat_params.numel() > 0 else \
                torch.zeros((), dtype=like.dtype, device=like.device)
            adj_time = torch.zeros((), dtype=like.dtype, device=like.device)
            time_vjps = []
            t_dev = t.to(device=like.device)
            for i in range(T - 1, 0, -1):
                ans_i = tuple(a[i] for a in ans)
                grad_output_i = tuple(g[i] for g in grad_output)
                func_i = func(t_dev[i].to(like.dtype), ans_i)
                # effect of moving the current time measurement point (adjoint.py:134-140)
                dLd_cur_t = sum(torch.dot(f_.reshape(-1), g_.reshape(-1)) for f_, g_ in zip(func_i, grad_output_i))
                adj_time = adj_time - dLd_cur_t
                time_vjps.append(dLd_cur_t.reshape(1))
                aug_y0 = (*ans_i, *adj_y, adj_time, adj_params)
                aug_ans = odeint(augmented_dynamics, aug_y0, torch.stack([t[i], t[i - 1]]).detach().cpu(),
                                 rtol=cfg['adjoint_rtol'], atol=cfg['adjoint_atol'], method=cfg['adjoint_method'],
                                 options=cfg['adjoint_options'])
                adj_y = tuple(a[1] for a in aug_ans[n_tensors:2 * n_tensors])
                adj_time = aug_ans[2 * n_tensors][1]
                adj_params = aug_ans[2 * n_tensors + 1][1]
                adj_y = tuple(a + g[i - 1] for a, g in zip(adj_y, grad_output))
            time_vjps.append(adj_time.reshape(1))
            time_vjps = torch.cat(time_vjps[::-1]).to(dtype=t.dtype, device=t.device)
            grad_params = adj_params.to(flat_params.dtype) if flat_params.numel() > 0 else None
            if isinstance(cfg.get('_usage'), dict) and hasattr(augmented_dynamics, 'used'):
                cfg['_usage']['used'] = list(augmented_dynamics.used)
        return (None, None, None, time_vjps, grad_params, *adj_y)


def odeint_adjoint(func, y0, t, rtol=1e-6, atol=1e-12, method=None, options=None, adjoint_method=None,
                   adjoint_rtol=None, adjoint_atol=None, adjoint_options=None):
    """adjoint.py:183-224.  Gradients flow to y0, t and func's parameters."""
    if not isinstance(func, torch.nn.Module):
        raise ValueError('func is required to be an instance of torch.nn.Module')     # adjoint.py:187-188 (tf.keras.Model there)
    if adjoint_method is None:
        adjoint_method = method
    if adjoint_rtol is None:
        adjoint_rtol = rtol
    if adjoint_atol is None:
        adjoint_atol = atol
    if adjoint_options is None:
        adjoint_options = options
    if adjoint_options and adjoint_options.get('graph'):
        # the backward dynamics call torch.autograd.grad, which cannot run under hipGraph stream capture (probed again in round 2:
        # the capture aborts inside the autograd engine - AccumulateGrad stream mismatch - and takes the process with it)
        import warnings
        warnings.warn("odeint_adjoint: the 'graph' option is not used for the backward solve (autograd inside f)")
        adjoint_options = {k: v for k, v in adjoint_options.items() if k != 'graph'}
    tensor_input = False
    if isinstance(y0, torch.Tensor):
        tensor_input = True
        y0 = (y0,)
        func = _TupleModule(func)
    params = _trainable(func)
    base = getattr(func, 'base_func', func)
    optional = getattr(base, '_mi_optional_params', ())
    usage = {'optional': frozenset(i for i, p in enumerate(params) if any(p is o for o in optional))}
    flat_params = _FlatParams.apply(usage, *params) if params else torch.zeros(0, device=y0[0].device, dtype=y0[0].dtype)
    cfg = dict(rtol=rtol, atol=atol, method=method, options=options, adjoint_method=adjoint_method,
               adjoint_rtol=adjoint_rtol, adjoint_atol=adjoint_atol, adjoint_options=adjoint_options, _usage=usage)
    t = torch.as_tensor(t)
    ys = _OdeintAdjointMethod.apply(func, len(y0), cfg, t, flat_params, *y0)
    if tensor_input:
        ys = ys[0]
    return ys


odeint_adjoint.last_backward_stats = {}

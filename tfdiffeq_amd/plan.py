"""`odeint.plan(func, y0, t, ...)`: which engine a call will take and the predicate that chose it - BEFORE running it (round-5 review,
item 9: the dispatch between the engines is spread over `supports*` predicates of rhs.py and the solver classes; this is the one table).

Nothing is launched and no engine is created: the answer is derived from the same predicates `solvers.py` / `adams.py` evaluate
(`DeviceRHS.supports`, `.supports_coop`, `.supports_multistep`, `.row_local`, `.tile_dopri8`, `.wide_tableaus`, `.fixed_grid_fused`,
`.multistep_fused`, the tableau's shape), so it also works where there is no GPU (the state's device is then taken as given).  The one
thing decided later, inside `mi_ode_create`, is co-residency: a whole-call kernel needs every workgroup of the batch resident at once;
`plan` reports the rule ('co-resident batch') rather than the device's answer.

    >>> odeint.plan(lambda t, y: torch.matmul(y, W), y0, t, method='dopri5')
    {'engine': 'fused', 'family': 'linear', 'kernel': 'k_persist_linear_mfma<double, 128, 6>', 'launches': 'one per call', ...}
"""
import torch

from . import rhs as R

ADAPTIVE = {'dopri5': (6, True), 'tsit5': (6, True), 'bosh3': (3, True), 'dopri8': (13, True), 'adaptive_heun': (1, False)}   # rows, FSAL shaped
FIXED_RK = {'euler': True, 'rk4': True, 'midpoint': False, 'heun': False, 'huen': False}                                           # has a one-launch kernel
MULTISTEP = ('explicit_adams', 'fixed_adams', 'adams')


def _tname(dtype):
    return 'double' if dtype == torch.float64 else 'float'


def _family(rhs):
    if isinstance(rhs, R.MLP):
        return 'mlp'
    if isinstance(rhs, R.CubicLinear):
        return 'cubic_linear'
    if isinstance(rhs, R.Linear):
        return 'linear'
    if getattr(rhs, 'coop', False) or isinstance(rhs, R.CustomCoop):
        return 'generated / user code, a thread per state element'
    if getattr(rhs, 'row_local', False):
        return 'row-local (a trajectory per thread)'
    return type(rhs).__name__


def _callable_engine(method, options, why):
    opts = options or {}
    if method in ADAPTIVE and opts.get('graph', 'auto') != 'host' and not opts.get('force_plane_kernels') and 'process_group' not in opts:
        return {'engine': 'callable', 'kernel': 'k_opq_norms + k_opq_commit around torch kernels (graph_step.DeviceControlledRK)',
                'launches': 'one hipGraph replay per attempt once recorded (options graph=%r), eager before' % (opts.get('graph', 'auto'),),
                'why': why}
    return {'engine': 'plane kernels', 'kernel': 'mi_ode_lincomb / mi_ode_error_norms / mi_ode_interp_eval between evaluations of f',
            'launches': 'several per stage, the controller on the host', 'why': why}


def plan_rhs(rhs, y, method, options=None):
    """The decision for a DeviceRHS and ONE state tensor (shape / dtype are read, nothing else)."""
    opts = options or {}
    fusion = opts.get('fusion', 0)
    rows = y.numel() // max(int(rhs.dim or 1), 1)
    T = _tname(y.dtype)
    fam = _family(rhs)
    base = {'family': fam, 'state': '%d x %d %s' % (rows, rhs.dim, str(y.dtype).replace('torch.', ''))}
    coop_ok = hasattr(rhs, 'supports_coop') and rhs.supports_coop(y)

    def out(d):
        d.update(base)
        return d
    if opts.get('force_plane_kernels'):
        return out(_callable_engine(method, opts, "options['force_plane_kernels']"))
    if method in ADAPTIVE:
        S, fsal = ADAPTIVE[method]
        if not rhs.supports(y):
            if coop_ok and 'process_group' not in opts and fusion in (0, 'auto', 4, 'whole'):
                return out({'engine': 'fused', 'kernel': 'k_persist_rowlocal<%s, %d, .., RhsMlpCoop> (planes variant beyond a co-resident batch)' % (T, S),
                            'launches': 'one per call', 'why': 'rhs.MLP.supports_coop: outside the tile kernels\' box (dim <= 64, hidden <= 128), '
                            'inside the cooperative kernel\'s (<= 256 wide) and under COOP_MAX_FMA multiply-adds per evaluation'})
            return out(_callable_engine(method, opts, '%s.supports(y0) is False (dim %s, dtype %s)' % (type(rhs).__name__, rhs.dim, y.dtype)))
        wide = (fsal and S == 13) or (not fsal and S == 1)
        if wide:
            ok = getattr(rhs, 'row_local', False) or getattr(rhs, 'wide_tableaus', False) or (S == 13 and getattr(rhs, 'tile_dopri8', False))
            if not ok or fusion in (1, 'stage'):
                return out(_callable_engine(method, opts, 'the %d-row tableau exists for row-local / cooperative right-hand sides%s only'
                                            % (S, ' and the tile kernels' if S == 13 else '')))
        if getattr(rhs, 'row_local', False):
            return out({'engine': 'fused', 'kernel': 'k_persist_rowlocal<%s, %d, ..> (k_persist_rowlocal_planes beyond 131072 trajectories)' % (T, S),
                        'launches': 'one per call', 'why': 'row_local right-hand side: state and stage derivatives thread-private'})
        if getattr(rhs, 'coop', False) or isinstance(rhs, R.CustomCoop):
            return out({'engine': 'fused', 'kernel': 'k_persist_rowlocal<%s, %d, .., RhsUserCoop> (planes variant beyond a co-resident batch)' % (T, S),
                        'launches': 'one per call', 'why': 'cooperative plugin: a thread per state element, dim <= 256'})
        if fam == 'mlp':
            kern = 'k_persist_mlp' if y.dtype == torch.float32 else 'k_persist_mlp64'
            return out({'engine': 'fused', 'kernel': '%s<DP, HP, %d, %d> (dim / hidden padded; k_mlp per attempt when the tile grid is not co-resident)'
                        % (kern, R.MLP.ACTIVATIONS[rhs.activation], S), 'launches': 'one per call',
                        'why': 'rhs.MLP.supports: dim <= 64, hidden <= 128 - the MFMA tile kernels (float32: weights resident in registers; '
                               'float64: weights streamed from a packed copy)'})
        if fam in ('linear', 'cubic_linear'):
            if fam == 'cubic_linear':                # (y ** 3) @ W beyond 2 x 2: csrc pick_family keeps the cube on the vector-ALU stage kernels
                return out({'engine': 'fused', 'kernel': 'k_stage_linear_valu', 'launches': 'one per stage',
                            'why': '(y ** 3) @ W at dim %d: the tile kernels have no cube in front of the product (measured in round 6: a run-time '
                                   'switch for it costs the linear system 1.4 %% at config 4) - the vector-ALU stage kernels, dim <= 256' % rhs.dim})
            if 3 <= rhs.dim <= 128:
                sched = {1: 'k_stage_linear_mfma (one kernel per stage)', 'stage': 'k_stage_linear_mfma (one kernel per stage)',
                         2: 'k_step_linear_mfma (one kernel per attempt)', 'step': 'k_step_linear_mfma (one kernel per attempt)'}.get(
                             fusion, 'k_persist_linear_mfma<%s, %d, %d> (co-resident batch; k_step_linear_mfma per attempt otherwise)'
                             % (T, max(16, 1 << (int(rhs.dim) - 1).bit_length()), S))
                return out({'engine': 'fused', 'kernel': sched, 'launches': 'one per call' if fusion in (0, 'auto', 4, 'whole') else 'per stage / attempt',
                            'why': '3 <= dim <= 128: the MFMA tile kernels (W slice resident in registers)'})
            if fam == 'linear' and 128 < rhs.dim <= 256 and S in (3, 6) and fusion not in (1, 'stage'):
                sched = 'k_step_linear_mfma<%s, 256, %d> (one kernel per attempt)' % (T, S) if fusion in (2, 'step') else \
                    'k_persist_linear_mfma<%s, 256, %d> (co-resident batch; k_step_linear_mfma per attempt otherwise)' % (T, S)
                return out({'engine': 'fused', 'kernel': sched, 'launches': 'one per call' if fusion in (0, 'auto', 4, 'whole') else 'per attempt',
                            'why': '128 < dim <= 256, three- or six-row tableau: the 256-wide MFMA tile kernels (W streamed from a copy in '
                                   'consumption order, csrc/mi_ode_step_fused.h LinCtx::STREAM)'})
            return out({'engine': 'fused', 'kernel': 'k_stage_linear_valu', 'launches': 'one per stage',
                        'why': 'dim %d outside 3 .. 128 (and not a dopri5 / tsit5 / bosh3 call at dim <= 256): the vector-ALU fallback (dim <= 256)' % rhs.dim})
        return out({'engine': 'fused', 'kernel': 'catalogue kernels of %s' % type(rhs).__name__, 'launches': 'one per call', 'why': 'supports(y0)'})
    if method in FIXED_RK:
        if FIXED_RK[method] and rhs.fixed_grid_fused and (rhs.supports(y) or coop_ok):
            k = 'k_fixed_rowlocal' if (getattr(rhs, 'row_local', False) or getattr(rhs, 'coop', False) or isinstance(rhs, R.CustomCoop) or coop_ok) else \
                ('k_fixed_mlp' if fam == 'mlp' else 'k_fixed_linear_mfma' if (fam == 'linear' and 3 <= rhs.dim <= 256) else 'FX_* stage kernels (vector ALU)')
            return out({'engine': 'fused', 'kernel': '%s<%s, ..>' % (k, T), 'launches': 'one per call',
                        'why': 'euler / rk4 have one-launch fixed-grid kernels for every fused family'})
        return out({'engine': 'plane kernels', 'kernel': 'step_func over mi_ode_lincomb, one evaluation of forward() per stage',
                    'launches': 'several per grid interval', 'why': '%s has no fused kernel (euler / rk4 do)' % method if not FIXED_RK[method]
                    else 'no fixed-grid kernel takes this state'})
    if method in MULTISTEP:
        if rhs.supports_multistep(y) and getattr(rhs, 'multistep_fused', False):
            k = 'k_adams_vc_rowlocal' if method == 'adams' else 'k_fixed_adams_rowlocal'
            return out({'engine': 'fused', 'kernel': '%s<%s, ..>' % (k, T), 'launches': 'one per call (co-resident batch; the per-step loop otherwise)',
                        'why': 'multistep_fused: row-local systems, matrix right-hand sides and networks up to 256 wide'})
        return out({'engine': 'plane kernels', 'kernel': 'k_adams_predict / _correct / _error_sums / _update_phi' if method == 'adams' else 'mi_ode_lincomb',
                    'launches': 'four per attempt' if method == 'adams' else 'several per step', 'why': 'no one-launch multistep kernel for this right-hand side'})
    raise KeyError(method)


def plan(func, y0, t=None, rtol=1e-7, atol=1e-9, method=None, options=None):
    """What `odeint(func, y0, t, rtol, atol, method, options)` will run on.  A dict: engine ('fused' | 'callable' | 'plane kernels'),
    kernel, launches, why (the predicate that decided), family / state, and `lower` (how a Python callable was lowered, or why not)."""
    from . import lower as L
    from . import odeint as _pkg_odeint              # noqa: F401  (the package exports the function under the module's name)
    import sys
    OD = sys.modules['tfdiffeq_amd.odeint']
    method = method or 'dopri5'
    opts = dict(options or {})
    if method not in OD.SOLVERS:
        raise KeyError(method)
    ys = y0 if isinstance(y0, (tuple, list)) else (y0,)
    lower_info = None
    rhs = func if getattr(func, 'kind', 0) else None
    if rhs is None and getattr(func, 'per_component', False):
        base = func.device_rhs
        d = {'engine': 'fused' if (method in ADAPTIVE or FIXED_RK.get(method)) and getattr(base, 'row_local', False) and 2 <= len(ys) <= 8 else 'plane kernels',
             'kernel': 'k_persist_rowlocal over one segmented buffer', 'launches': 'one per call',
             'why': 'rhs.PerComponent of a row-local right-hand side: tuple components share one buffer', 'family': 'tuple of ' + _family(base)}
        d['lower'] = None
        return d
    if rhs is None and len(ys) == 1 and isinstance(ys[0], torch.Tensor):
        mode = opts.get('lower', OD.LOWER_DEFAULT)
        if mode is False:
            lower_info = {'lowered': False, 'why': "options['lower'] is False"}
        elif any(k in opts for k in ('process_group', 'force_plane_kernels', 'grid_constructor')) or (opts.get('graph', 'auto') != 'auto' and mode is not True):
            lower_info = {'lowered': False, 'why': 'an option asks for one of the callable engines (%s)' % sorted(k for k in opts if k in
                                                                                                                ('process_group', 'force_plane_kernels', 'graph', 'grid_constructor'))}
        else:
            try:
                tr = L.trace(func, ys[0])
                prog = L.program_for(tr, generic=method in L.GENERIC_ONLY_METHODS)
                kind = prog.kind
                lower_info = {'lowered': True, 'kind': kind, 'dim': prog.dim, 'batch_axes': tr.nb}
                shaped = ys[0].reshape(tr.batch_shape + (prog.dim,))
                if kind in ('linear', 'cubic', 'mlp'):
                    rhs = prog.bind(tr, ys[0].device)
                else:
                    rhs = prog.rhs
                d = plan_rhs(rhs, shaped, method, opts)
                d['lower'] = lower_info
                return d
            except L.TraceError as e:
                lower_info = {'lowered': False, 'why': str(e)}
            except Exception as e:
                lower_info = {'lowered': False, 'why': 'tracing failed: %s: %s' % (type(e).__name__, e)}
    if rhs is not None and len(ys) == 1:
        d = plan_rhs(rhs, ys[0], method, opts)
        d['lower'] = None
        return d
    why = 'a Python callable' + ('' if lower_info is None else ' (%s)' % lower_info['why']) if len(ys) == 1 else 'a tuple state of a Python callable'
    d = _callable_engine(method, opts, why) if method in ADAPTIVE else \
        {'engine': 'plane kernels', 'kernel': 'the solver\'s Python loop over plane kernels', 'launches': 'several per step', 'why': why}
    d['lower'] = lower_info
    return d

"""ctypes binding of libmi_ode.so (include/mi_ode.h) - the thin shim named by the north star.

The library is built in-tree (`tfdiffeq_amd/libmi_ode.so`, see csrc/Makefile) for gfx950 only.
There is no CPU fallback: every compute entry point of the package goes through this module and
raises if the library is missing or no MI355X is visible.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TFDIFFEQ_AMD_LIB') or os.path.join(_HERE, 'libmi_ode.so')   # env override: kernel-variant sweeps
CSRC = os.path.join(_HERE, 'csrc')

MAX_STAGES = 13
MAX_K = MAX_STAGES + 1
MAX_LINCOMB = 14
REC = 8

F32, F64 = 0, 1
RHS_LINEAR, RHS_CUBIC_LINEAR, RHS_LOTKA_VOLTERRA, RHS_LORENZ, RHS_MLP_TANH, RHS_PLUGIN = 1, 2, 3, 4, 5, 6
CTRL_MISC, CTRL_TSIT5 = 0, 1
INTERP_QUARTIC_MID, INTERP_TSIT5, INTERP_TSIT5_REF = 0, 1, 2
ABI_VERSION = 13
MAX_SEGMENTS = 8
SEGMENT_ALIGN = 256
IPC_HANDLE_BYTES = 64
RCCL_ID_BYTES = 128
ST_DT_UNDERFLOW, ST_NONFINITE, ST_MAX_STEPS, ST_BAD_T, ST_SYNC_TIMEOUT = 1, 2, 4, 8, 16
E_INVALID, E_HIP, E_NODEVICE, E_EXCHANGE = -1, -2, -3, -4          # negative returns: API errors (mi_ode.h)


class Tableau(C.Structure):
    _fields_ = [('n_stages', C.c_int32), ('fsal', C.c_int32),
                ('alpha', C.c_double * MAX_STAGES),
                ('beta', (C.c_double * MAX_STAGES) * MAX_STAGES),
                ('c_sol', C.c_double * MAX_K),
                ('c_error', C.c_double * MAX_K),
                ('c_mid', C.c_double * MAX_K)]


class Rhs(C.Structure):
    _fields_ = [('kind', C.c_int32), ('hidden', C.c_int32), ('sign', C.c_double),
                ('scalars', C.c_double * 8),
                ('w', C.c_void_p * 3), ('b', C.c_void_p * 3), ('plugin', C.c_void_p)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)


class Desc(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('adaptive', C.c_int32),
                ('batch', C.c_int64), ('dim', C.c_int64),
                ('tableau', Tableau), ('rhs', Rhs),
                ('controller', C.c_int32), ('interp', C.c_int32),
                ('order', C.c_int32), ('init_order', C.c_int32),
                ('rtol', C.c_double), ('atol', C.c_double),
                ('safety', C.c_double), ('ifactor', C.c_double), ('dfactor', C.c_double),
                ('first_step', C.c_double),
                ('max_num_steps', C.c_int64),
                ('world_size', C.c_int32), ('rank', C.c_int32),
                ('allgather', ALLGATHER_FN), ('allgather_user', C.c_void_p),
                ('exchange_send_dev', C.c_void_p), ('exchange_recv_dev', C.c_void_p),
                ('linear_variant', C.c_int32), ('chunk_attempts', C.c_int32),
                ('reserved0', C.c_int32), ('profile', C.c_int32),
                ('fusion', C.c_int32), ('reserved', C.c_int32),
                ('xrank_host', C.c_void_p), ('xrank_bytes', C.c_int64),
                ('n_segments', C.c_int32), ('seg_tolerances', C.c_int32), ('seg_rows', C.c_int64 * MAX_SEGMENTS),
                ('seg_rtol', C.c_double * MAX_SEGMENTS), ('seg_atol', C.c_double * MAX_SEGMENTS),
                ('multistep', C.c_int32), ('ms_max_order', C.c_int32), ('ms_max_iters', C.c_int32), ('ms_min_order', C.c_int32),
                ('ms_ab', C.POINTER(C.c_double)), ('ms_am', C.POINTER(C.c_double)), ('ms_am0', C.POINTER(C.c_double)),
                ('ms_gamma_star', C.POINTER(C.c_double))]


class CtrlParams(C.Structure):
    _fields_ = [('rtol', C.c_double), ('atol', C.c_double), ('safety', C.c_double), ('ifactor', C.c_double), ('dfactor', C.c_double),
                ('order', C.c_int32), ('init_order', C.c_int32), ('controller', C.c_int32), ('dtype', C.c_int32)]


class AdjointDesc(C.Structure):
    _fields_ = [('batch', C.c_int64), ('dim', C.c_int32), ('hidden', C.c_int32), ('tableau', Tableau),
                ('rtol', C.c_double), ('atol', C.c_double), ('safety', C.c_double), ('ifactor', C.c_double), ('dfactor', C.c_double),
                ('order', C.c_int32), ('init_order', C.c_int32), ('max_num_steps', C.c_int64),
                ('time_dependent', C.c_int32), ('reserved', C.c_int32)]


class LinAdjDesc(C.Structure):
    """mi_ode_linadj_desc: the backward segment of odeint_adjoint for the linear right-hand side in one launch."""
    _fields_ = [('batch', C.c_int64), ('dim', C.c_int32), ('dtype', C.c_int32), ('tableau', Tableau),
                ('rtol', C.c_double), ('atol', C.c_double), ('safety', C.c_double), ('ifactor', C.c_double), ('dfactor', C.c_double),
                ('order', C.c_int32), ('init_order', C.c_int32), ('max_num_steps', C.c_int64)]


class OpqDesc(C.Structure):
    """mi_ode_opq_desc: adaptive RK over an opaque (Python) right-hand side with the controller on the device."""
    _fields_ = [('dtype', C.c_int32), ('n_comp', C.c_int32), ('n', C.c_int64 * MAX_SEGMENTS), ('tableau', Tableau),
                ('controller', C.c_int32), ('interp', C.c_int32), ('order', C.c_int32), ('init_order', C.c_int32),
                ('rtol', C.c_double * MAX_SEGMENTS), ('atol', C.c_double * MAX_SEGMENTS),
                ('safety', C.c_double), ('ifactor', C.c_double), ('dfactor', C.c_double), ('max_num_steps', C.c_int64)]


class Stats(C.Structure):
    _fields_ = [('n_attempts', C.c_int64), ('n_accepted', C.c_int64), ('n_rejected', C.c_int64), ('nfe', C.c_int64),
                ('t', C.c_double), ('dt', C.c_double), ('last_ratio', C.c_double),
                ('status', C.c_uint32), ('n_polls', C.c_int32), ('n_launches', C.c_int64), ('clock_mhz', C.c_double),
                ('handoff_us', C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/mi_ode.h declares: (restype, argtypes)
_PROTOS = {
    'mi_ode_abi_version': (C.c_int, []),
    'mi_ode_status_string': (C.c_char_p, [C.c_uint32]),
    'mi_ode_last_error': (C.c_char_p, []),
    'mi_ode_reduce_workspace_bytes': (C.c_int64, []),
    'mi_ode_sizeof': (C.c_int64, [C.c_int32]),
    'mi_ode_create': (C.c_int, [C.POINTER(Desc), C.POINTER(C.c_void_p)]),
    'mi_ode_destroy': (C.c_int, [C.c_void_p]),
    'mi_ode_begin': (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    'mi_ode_advance': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_void_p, C.c_void_p]),
    'mi_ode_integrate': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_void_p,
                                   C.POINTER(Stats), C.c_void_p]),
    'mi_ode_fixed_grid_integrate': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_void_p,
                                              C.POINTER(Stats), C.c_void_p]),
    'mi_ode_fixed_grid_integrate_on': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.c_int32,
                                                 C.c_double, C.c_void_p, C.POINTER(Stats), C.c_void_p]),
    'mi_ode_rk_step_fused': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p,
                                       C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_void_p]),
    'mi_ode_eval_rhs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    'mi_ode_xrank_bytes': (C.c_int64, [C.c_int32]),
    'mi_ode_xrank_selftest': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mi_ode_xrank_enable': (C.c_int, [C.c_void_p, C.c_int32]),
    'mi_ode_xpeer_prepare': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mi_ode_xpeer_connect': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    'mi_ode_rccl_unique_id': (C.c_int, [C.c_void_p]),
    'mi_ode_rccl_connect': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    'mi_ode_get_stats': (C.c_int, [C.c_void_p, C.POINTER(Stats), C.c_void_p]),
    'mi_ode_get_state': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mi_ode_get_profile': (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    'mi_ode_adjoint_create': (C.c_int, [C.POINTER(AdjointDesc), C.POINTER(C.c_void_p)]),
    'mi_ode_adjoint_destroy': (C.c_int, [C.c_void_p]),
    'mi_ode_adjoint_num_params': (C.c_int64, [C.c_void_p]),
    'mi_ode_adjoint_segment': (C.c_int, [C.c_void_p, C.POINTER(Rhs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats), C.c_void_p]),
    'mi_ode_adjoint_dynamics': (C.c_int, [C.c_void_p, C.POINTER(Rhs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    'mi_ode_adjoint_dynamics_at': (C.c_int, [C.c_void_p, C.POINTER(Rhs), C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    'mi_ode_outer_workspace_bytes': (C.c_int64, [C.c_int32, C.c_int64, C.c_int32]),
    'mi_ode_outer_reduce': (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    'mi_ode_linadj_create': (C.c_int, [C.POINTER(LinAdjDesc), C.POINTER(C.c_void_p)]),
    'mi_ode_linadj_destroy': (C.c_int, [C.c_void_p]),
    'mi_ode_linadj_segment': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                        C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(Stats), C.c_void_p]),
    'mi_ode_linadj_profile': (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    'mi_ode_opq_create': (C.c_int, [C.POINTER(OpqDesc), C.POINTER(C.c_void_p)]),
    'mi_ode_opq_destroy': (C.c_int, [C.c_void_p]),
    'mi_ode_opq_dt_dev': (C.c_void_p, [C.c_void_p]),
    'mi_ode_opq_begin': (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_void_p),
                                   C.c_void_p, C.c_void_p]),
    'mi_ode_opq_finish': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    'mi_ode_opq_commit': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_void_p), C.c_void_p]),
    'mi_ode_opq_poll': (C.c_int, [C.c_void_p, C.POINTER(Stats), C.POINTER(C.c_int32), C.c_void_p]),
    'mi_ode_controller_update': (C.c_int, [C.POINTER(CtrlParams), C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.c_void_p]),
    'mi_ode_rk_stage_combine': (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_double),
                                          C.c_int32, C.c_double, C.c_void_p, C.c_void_p]),
    'mi_ode_rk_error_reduce': (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    'mi_ode_lincomb': (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_double),
                                 C.c_int32, C.c_double, C.c_void_p, C.c_void_p]),
    'mi_ode_lincomb_dev': (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_double),
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mi_ode_error_norms': (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    'mi_ode_scaled_sumsq': (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                      C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mi_ode_not_converged': (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    'mi_ode_adams_predict': (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.c_double, C.c_void_p, C.c_void_p]),
    'mi_ode_adams_correct': (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mi_ode_adams_error_sums': (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    'mi_ode_adams_update_phi': (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_double),
                                          C.POINTER(C.c_void_p), C.c_void_p]),
    'mi_ode_interp_eval': (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                     C.c_int32, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double,
                                     C.c_double, C.c_void_p, C.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(sorted(_PROTOS))

_lib = None


class NativeError(RuntimeError):
    """A libmi_ode call failed.  `rc` is the library's return code (MI_ODE_E_*; None when the failure is not a library call's)."""

    def __init__(self, msg, rc=None):
        super(NativeError, self).__init__(msg)
        self.rc = rc


def build(verbose=False):
    """Compile libmi_ode.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ['make', '-C', CSRC, '-j', str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0 or not os.path.exists(LIB_PATH):
        raise NativeError('building libmi_ode.so failed (see output above)')
    return LIB_PATH


def load():
    """dlopen libmi_ode.so (after torch, so that HIP symbols bind to torch's loaded libamdhip64)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                          '(or make -C tfdiffeq_amd/csrc). tfdiffeq_amd has no CPU/eager fallback.' % LIB_PATH)
    import torch  # noqa: F401  (loads libamdhip64.so.7 first; our DT_NEEDED then resolves to the same runtime)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)      # AttributeError here = the .so does not export what mi_ode.h declares
        fn.restype = res
        fn.argtypes = args
    if lib.mi_ode_abi_version() != ABI_VERSION:
        raise NativeError('libmi_ode.so ABI version mismatch')
    for which, st in ((0, Desc), (1, Stats), (2, Tableau), (3, Rhs), (5, CtrlParams), (6, AdjointDesc), (7, OpqDesc), (8, LinAdjDesc)):
        if lib.mi_ode_sizeof(which) != C.sizeof(st):
            raise NativeError('struct layout mismatch for %s: C %d vs ctypes %d'
                              % (st.__name__, lib.mi_ode_sizeof(which), C.sizeof(st)))
    _lib = lib
    return lib


def last_error():
    return load().mi_ode_last_error().decode('utf-8', 'replace')


def check(rc, what=''):
    """Negative return codes are API errors; non-negative ones are status bits handled by the caller."""
    if rc < 0:
        raise NativeError('%s failed (%d): %s' % (what or 'libmi_ode call', rc, last_error()), rc=rc)
    return rc


def status_message(bits):
    return load().mi_ode_status_string(bits).decode('utf-8')


def require_gpu_tensor(x, what='y0'):
    import torch
    if not isinstance(x, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor' % what)
    if not x.is_cuda:
        raise NativeError('tfdiffeq_amd computes on MI355X only: %s is on %s. Move it to a CUDA (ROCm) device; '
                          'there is no CPU fallback.' % (what, x.device))


def dtype_code(dtype):
    import torch
    if dtype == torch.float64:
        return F64
    if dtype == torch.float32:
        return F32
    raise TypeError('`y0` must be a float32 or float64 Tensor for the MI355X kernels but is a {}'.format(dtype))


def stream_ptr(device=None):
    """The current stream of `device` as a hipStream_t.  torch.cuda.current_stream() builds a Stream object per call (~5 us on a 70-us odeint
    call); torch's own raw accessor answers the same question in a fraction of that - used when this torch has it."""
    import torch
    raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    if raw is not None:
        idx = getattr(device, 'index', None) if device is not None else None
        try:
            return C.c_void_p(raw(torch.cuda.current_device() if idx is None else idx))
        except Exception:                                   # (any surprise of the private accessor: the public route)
            pass
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

"""Mirror of tfdiffeq/solvers.py: the solver-registry plugin contract (B2 in SURVEY.md 8(b)).

    cls(func, y0_tuple, rtol=, atol=, **options).integrate(t) -> tuple of stacked tensors

Adaptive subclasses supply `before_integrate(t)` + `advance(next_t)`; fixed-grid subclasses supply
`step_func(func, t, dt, y) -> dy` + `order`  (solvers.py:20-25, 73-80).

Two execution engines sit behind that contract:
  * fused   - `func` carries a DeviceRHS and the state is one [batch, dim] tensor: the whole integrate()
              runs inside libmi_ode (`_FusedEngine`): f evaluated in the stage kernels, controller and
              dense output on device, host polls a done flag.
  * planes  - any Python callable / tuple state: the Python loops below, every state-sized operation being
              a libmi_ode plane kernel.  One host synchronisation per step attempt.
There is no CPU path: state tensors must live on the MI355X.
"""
import abc
import os
import collections
import ctypes as C
import math
import sys

import numpy as np
import torch

from . import _native as N
from .misc import (_assert_increasing, _handle_unused_kwargs, _lincomb, _np_dtype, _scalar_tensor, _Exchange)


# ---------------------------------------------------------------------------------------------
# fused engine wrapper
# ---------------------------------------------------------------------------------------------
def _fill_tableau(tb_struct, tableau, c_mid):
    S = len(tableau.alpha)
    if S > N.MAX_STAGES:
        raise ValueError('fused engine supports at most %d tableau rows' % N.MAX_STAGES)
    tb_struct.n_stages = S
    from .rk_common import _is_fsal_shaped
    tb_struct.fsal = 1 if (S > 0 and _is_fsal_shaped(tableau)) else 0
    for i in range(S):
        tb_struct.alpha[i] = float(tableau.alpha[i])
        for j, v in enumerate(tableau.beta[i]):
            tb_struct.beta[i][j] = float(v)
    for j, v in enumerate(tableau.c_sol):
        tb_struct.c_sol[j] = float(v)
    for j, v in enumerate(tableau.c_error):
        tb_struct.c_error[j] = float(v)
    if c_mid is not None:
        for j, v in enumerate(c_mid):
            tb_struct.c_mid[j] = float(v)


def _xrank_segment(group, nbytes):
    """(mmap, address, size) of a zero-filled /dev/shm segment shared by the ranks of `group` on this node, or None.
    Collective over the group (one broadcast, two barriers)."""
    import mmap
    import uuid
    import torch.distributed as dist
    rank = dist.get_rank(group)
    size = ((int(nbytes) + 4095) // 4096) * 4096
    name = [None]
    if rank == 0:
        name[0] = 'mi_ode_xrank_%d_%s' % (os.getpid(), uuid.uuid4().hex[:12])
    dist.broadcast_object_list(name, src=dist.get_global_rank(group, 0), group=group)
    path = os.path.join('/dev/shm', name[0])
    seg = None
    # every rank passes the same two barriers whatever fails locally (a failure only means "no segment for me"; the
    # caller's all-reduce then switches the mechanism off for the whole group)
    fd = -1
    if rank == 0:
        try:
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            os.ftruncate(fd, size)                      # zero filled
        except OSError:
            fd = -1
    dist.barrier(group=group)
    try:
        if rank != 0:
            fd = os.open(path, os.O_RDWR)               # FileNotFoundError on another node: no segment for this rank
        if fd >= 0:
            mm = mmap.mmap(fd, size, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
            os.close(fd)
            seg = (mm, C.addressof(C.c_char.from_buffer(mm)), size)
    except (OSError, ValueError):
        seg = None
    dist.barrier(group=group)
    if rank == 0:
        try:
            os.unlink(path)                             # the mappings keep it alive
        except OSError:
            pass
    return seg


class _FusedEngine(object):
    """Owns one mi_ode_handle.  y0: a device tensor [..., dim]; all leading axes are batch (trajectories)."""

    def __init__(self, device_rhs, y0, adaptive, tableau, c_mid=None, rtol=1e-7, atol=1e-9, controller=N.CTRL_MISC,
                 interp=N.INTERP_QUARTIC_MID, order=5, init_order=4, safety=0.9, ifactor=10.0, dfactor=0.2,
                 first_step=None, max_num_steps=2 ** 31 - 1, process_group=None, linear_variant=0, chunk_attempts=0,
                 profile=False, fusion=0, seg_rows=None, seg_tols=None, multistep=None):
        N.require_gpu_tensor(y0, 'y0')
        self.lib = N.load()
        self.y0 = y0.contiguous()
        self.shape = tuple(y0.shape)
        self.dim = int(y0.shape[-1])
        self.batch = int(y0.numel() // self.dim)
        self.device = y0.device
        self.dtype = y0.dtype
        d = N.Desc()
        d.dtype = N.dtype_code(y0.dtype)
        d.adaptive = 1 if adaptive else 0
        d.batch, d.dim = self.batch, self.dim
        _fill_tableau(d.tableau, tableau, c_mid)
        self._keep = device_rhs.fill(d.rhs, y0.dtype, y0.device)
        d.controller, d.interp, d.order, d.init_order = controller, interp, order, init_order
        d.rtol, d.atol = float(rtol), float(atol)
        d.safety, d.ifactor, d.dfactor = float(safety), float(ifactor), float(dfactor)
        d.first_step = float('nan') if first_step is None else float(first_step)
        d.max_num_steps = int(max_num_steps)
        d.linear_variant = int(linear_variant)
        d.chunk_attempts = int(chunk_attempts)
        d.profile = 1 if profile else 0
        d.fusion = {'auto': 0, 'stage': 1, 'step': 2, 'step_split': 3, 'whole': 4}.get(fusion, fusion)
        if multistep is not None and int(multistep[0]) == 3:       # the variable-order 'adams' solver in one launch: (3, max_order, gamma_star)
            d.multistep, d.ms_max_order = 3, int(multistep[1])
            self._ms_tabs = ((C.c_double * 13)(*[float(v) for v in multistep[2]][:13]),)
            d.ms_gamma_star = C.cast(self._ms_tabs[0], C.POINTER(C.c_double))
        elif multistep is not None:            # fixed-grid Adams family in one launch (include/mi_ode.h: multistep): the coefficient
            kind, max_order, max_iters, min_order, ab, am, am0 = multistep      # tables as host arrays, formed in Python floats
            d.multistep, d.ms_max_order, d.ms_max_iters, d.ms_min_order = int(kind), int(max_order), int(max_iters), int(min_order)
            self._ms_tabs = ((C.c_double * (13 * 12))(*ab), (C.c_double * (13 * 12))(*am), (C.c_double * 13)(*am0))
            d.ms_ab, d.ms_am, d.ms_am0 = (C.cast(a_, C.POINTER(C.c_double)) for a_ in self._ms_tabs)
        if seg_rows is not None:               # tuple state: y0 is the packed buffer (_pack_components), one segment per component
            d.n_segments = len(seg_rows)
            for k, r in enumerate(seg_rows):
                d.seg_rows[k] = int(r)
            if seg_tols is not None:               # one (rtol, atol) pair per component (dopri5.py:60-61)
                d.seg_tolerances = 1
                for k, (r_, a_) in enumerate(seg_tols):
                    d.seg_rtol[k], d.seg_atol[k] = float(r_), float(a_)
        self._hook = None
        if process_group is not None:
            import torch.distributed as dist
            world = dist.get_world_size(process_group)
            if world >= 1:                     # a 1-rank group still goes through the hook (lets one GPU test the path)
                d.world_size, d.rank = world, dist.get_rank(process_group)
                self._send = torch.zeros(N.REC, dtype=torch.float64, device=self.device)
                self._recv = torch.zeros(world * N.REC, dtype=torch.float64, device=self.device)
                d.exchange_send_dev = self._send.data_ptr()
                d.exchange_recv_dev = self._recv.data_ptr()
                send, recv, group = self._send, self._recv, process_group

                def hook(_user, _sendbuf, _recvbuf, _count, _stream):
                    # RCCL all-gather of one 8-double record per rank, stream-ordered with the kernels around it
                    try:
                        dist.all_gather_into_tensor(recv, send, group=group)
                        return 0
                    except Exception:          # pragma: no cover - surfaces as MI_ODE_E_EXCHANGE
                        import traceback
                        traceback.print_exc()
                        return 1
                self._hook = N.ALLGATHER_FN(hook)
                d.allgather = self._hook
        # Cross-rank transports, best first (TFDIFFEQ_AMD_XRANK = peer | host | rccl | hook | 0 restricts the choice; default: all):
        #   peer  in-kernel hand-off through mailboxes in PEER DEVICE memory (xGMI), one launch per call
        #   host  in-kernel hand-off through a /dev/shm segment every GPU maps (PCIe), one launch per call
        #   rccl  launch per attempt, ncclAllGather enqueued by libmi_ode itself
        #   hook  launch per attempt, torch.distributed all-gather through the ctypes callback (last resort)
        want = os.environ.get('TFDIFFEQ_AMD_XRANK', '1')
        allow = {'1': ('peer', 'host', 'rccl'), 'peer': ('peer', 'rccl'), 'host': ('host', 'rccl'), 'rccl': ('rccl',), 'hook': (),
                 '0': ('rccl',)}.get(want, ('peer', 'host', 'rccl'))
        self._xr = None
        if process_group is not None and 'host' in allow and adaptive:
            # cross-rank hand-off memory for the whole-call kernels: one host segment shared by the ranks of this node
            self._xr = _xrank_segment(process_group, int(self.lib.mi_ode_xrank_bytes(d.world_size)))
            if self._xr is not None:
                d.xrank_host, d.xrank_bytes = self._xr[1], self._xr[2]
        self.desc = d
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            N.check(self.lib.mi_ode_create(C.byref(d), C.byref(h)), 'mi_ode_create')
        self.h = h
        self.stats = N.Stats()
        self.xrank = False
        self.transport = 'single rank' if process_group is None else 'allgather hook (torch.distributed)'
        # what was tried to carry the per-attempt record across ranks, in order, and how it ended on THIS rank - a first run on a
        # multi-GPU node says by itself why it ended up on the transport it used (bench.py prints every rank's list)
        self.transport_log = []
        self._selftest_ok = False

        def tried(name, ok, why=''):
            self.transport_log.append({'transport': name, 'ok': bool(ok), 'why': why})
        if process_group is not None:
            import torch.distributed as dist

            def all_agree(ok):
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if dist.get_backend(process_group) == 'nccl' else 'cpu')
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
                return int(flag.item()) == 1

            def selftest_and_enable(have=True):
                # every rank tests the hand-off; it is used only if ALL ranks saw all peers (a rank on another node, or a failed
                # mapping, turns it off for everybody).  `have` False: this rank has nothing to test (no segment) - it still
                # takes part in the vote, with "no": the all-reduce below is a collective, every rank of the group must enter it.
                ok = False
                if have:
                    with torch.cuda.device(self.device):
                        ok = self.lib.mi_ode_xrank_selftest(self.h, self._stream()) == 0
                self._selftest_ok = ok
                if all_agree(ok):
                    N.check(self.lib.mi_ode_xrank_enable(self.h, 1), 'mi_ode_xrank_enable')
                    return True
                return False

            world, rank = d.world_size, d.rank
            if 'rccl' not in allow:
                tried('rccl', False, 'not allowed by TFDIFFEQ_AMD_XRANK=%s' % want)
            elif dist.get_backend(process_group) != 'nccl':
                tried('rccl', False, 'the process group is %s, not nccl (RCCL)' % dist.get_backend(process_group))
            if 'rccl' in allow and dist.get_backend(process_group) == 'nccl':
                # the launch-per-attempt schedule then needs no callback into Python (rank 0 draws the id, everybody joins)
                ids = [None]
                if rank == 0:
                    buf = (C.c_char * N.RCCL_ID_BYTES)()
                    ids[0] = bytes(buf) if self.lib.mi_ode_rccl_unique_id(buf) == 0 else None
                dist.broadcast_object_list(ids, src=dist.get_global_rank(process_group, 0), group=process_group)
                ok, why = False, 'rank 0 could not draw an ncclUniqueId: ' + N.last_error() if ids[0] is None else ''
                if ids[0] is not None:
                    with torch.cuda.device(self.device):
                        ok = self.lib.mi_ode_rccl_connect(self.h, ids[0], world, rank) == 0
                    if not ok:
                        why = 'mi_ode_rccl_connect: ' + N.last_error()
                if all_agree(ok):
                    self.transport = 'ncclAllGather enqueued by libmi_ode (launch per attempt)'
                    tried('rccl', True, 'communicator of %d ranks inside libmi_ode' % world)
                else:
                    tried('rccl', False, why or 'another rank could not join the communicator')
                    if ok:                     # some rank could not join: everybody drops back to the hook together
                        self.lib.mi_ode_rccl_connect(self.h, None, 0, 0)
            if not adaptive:
                tried('peer', False, 'fixed grid: no exchange needed')
            elif 'peer' not in allow:
                tried('peer', False, 'not allowed by TFDIFFEQ_AMD_XRANK=%s' % want)
            if adaptive and 'peer' in allow:
                buf = (C.c_char * N.IPC_HANDLE_BYTES)()
                with torch.cuda.device(self.device):
                    ok = self.lib.mi_ode_xpeer_prepare(self.h, buf) == 0
                why = '' if ok else 'mi_ode_xpeer_prepare (mailbox allocation / hipIpcGetMemHandle): ' + N.last_error()
                handles = [None] * world
                dist.all_gather_object(handles, bytes(buf) if ok else None, group=process_group)
                if ok and not all(hd is not None for hd in handles):
                    ok, why = False, 'rank(s) %s could not export a mailbox' % [q for q, hd in enumerate(handles) if hd is None]
                if ok:
                    blob = b''.join(handles)
                    with torch.cuda.device(self.device):
                        ok = self.lib.mi_ode_xpeer_connect(self.h, blob, world) == 0
                    if not ok:
                        why = 'mi_ode_xpeer_connect (hipIpcOpenMemHandle of a peer mailbox): ' + N.last_error()
                if not all_agree(ok):
                    tried('peer', False, why or 'another rank could not map the mailboxes')
                elif selftest_and_enable():
                    self.xrank = True
                    self.transport = 'in-kernel hand-off through peer device memory (xGMI mailboxes), one launch per call'
                    tried('peer', True, 'mailboxes mapped, self-test passed on every rank')
                else:
                    tried('peer', False, 'the bounded in-kernel self-test did not see every peer\'s records on some rank (this rank: %s)'
                          % ('passed' if self._selftest_ok else 'FAILED'))
            if adaptive and not self.xrank and 'host' not in allow:
                tried('host', False, 'not allowed by TFDIFFEQ_AMD_XRANK=%s' % want)
            if adaptive and not self.xrank and 'host' in allow:   # (group-uniform condition: self._xr may be None on ONE rank only)
                if selftest_and_enable(self._xr is not None):
                    self.xrank = True
                    self.transport = 'in-kernel hand-off through a shared host segment, one launch per call'
                    tried('host', True, '/dev/shm segment registered with HIP, self-test passed on every rank')
                else:
                    tried('host', False, 'no /dev/shm segment on this rank (another node?)' if self._xr is None else
                          'the bounded in-kernel self-test failed on some rank (this rank: %s)' % ('passed' if self._selftest_ok else 'FAILED'))
            if not self.xrank and self.transport.startswith('allgather hook'):
                tried('hook', True, 'torch.distributed all-gather through the ctypes callback (last resort)')

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.mi_ode_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return N.stream_ptr(self.device)

    @staticmethod
    def _times(t):
        arr = np.ascontiguousarray(np.asarray(t, dtype=np.float64))
        return arr, arr.ctypes.data_as(C.POINTER(C.c_double))

    def _raise_for_status(self, bits):
        if bits == 0:
            return
        msg = N.status_message(bits)
        if bits & N.ST_SYNC_TIMEOUT:       # engine fault, not one of the reference's assertions
            raise SyncTimeout(msg + " - retry with options={'fusion': 'step'}")
        if bits & N.ST_MAX_STEPS:
            msg = 'max_num_steps exceeded ({}>={})'.format(self.desc.max_num_steps, self.desc.max_num_steps)
        if bits & N.ST_DT_UNDERFLOW:
            msg = 'underflow in dt {}'.format(self.stats.dt)
        raise AssertionError(msg)          # the reference raises AssertionError for all of these

    def _check_y0(self, y0):
        if y0 is None:
            return self.y0
        N.require_gpu_tensor(y0, 'y0')
        if tuple(y0.shape) != self.shape or y0.dtype != self.dtype or y0.device != self.device:
            raise ValueError('engine was created for %s %s on %s' % (self.shape, self.dtype, self.device))
        return y0.contiguous()

    def integrate(self, t, y0=None, grid=None, eps=0.0):
        """grid / eps: fixed-grid engines only - the time grid the steps are taken on (default: t itself) and the `eps` of
        FixedGridODESolver (solvers.py:41-56)."""
        y0 = self._check_y0(y0)
        arr, p = self._times(t)
        T = arr.shape[0]
        out = torch.empty((T,) + self.shape, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            if grid is not None or eps != 0.0:
                garr, gp = self._times(arr if grid is None else grid)
                rc = N.check(self.lib.mi_ode_fixed_grid_integrate_on(self.h, C.c_void_p(y0.data_ptr()), gp, garr.shape[0], p, T, float(eps),
                                                                     C.c_void_p(out.data_ptr()), C.byref(self.stats), self._stream()),
                             'mi_ode_fixed_grid_integrate_on')
            else:
                fn = self.lib.mi_ode_integrate if self.desc.adaptive else self.lib.mi_ode_fixed_grid_integrate
                rc = N.check(fn(self.h, C.c_void_p(y0.data_ptr()), p, T, C.c_void_p(out.data_ptr()),
                                C.byref(self.stats), self._stream()), 'mi_ode_integrate')
        self._raise_for_status(rc)
        return out

    def begin(self, t0, y0=None):
        y0 = self._check_y0(y0)
        with torch.cuda.device(self.device):
            N.check(self.lib.mi_ode_begin(self.h, C.c_void_p(y0.data_ptr()), float(t0), self._stream()), 'mi_ode_begin')

    def advance(self, times):
        arr, p = self._times(times)
        out = torch.empty((arr.shape[0],) + self.shape, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            rc = N.check(self.lib.mi_ode_advance(self.h, p, arr.shape[0], C.c_void_p(out.data_ptr()), self._stream()),
                         'mi_ode_advance')
            N.check(self.lib.mi_ode_get_stats(self.h, C.byref(self.stats), self._stream()), 'mi_ode_get_stats')
        self._raise_for_status(rc)
        return out

    def profile(self):
        """(last-stage kernel ms total, launches, all-stages ms total, attempts) - needs profile=True."""
        out = (C.c_double * 4)()
        N.check(self.lib.mi_ode_get_profile(self.h, out), 'mi_ode_get_profile')
        return [out[i] for i in range(4)]

    def eval_rhs(self, y, t=0.0):
        f = torch.empty_like(y)
        with torch.cuda.device(self.device):
            N.check(self.lib.mi_ode_eval_rhs(self.h, C.c_void_p(y.data_ptr()), float(t), C.c_void_p(f.data_ptr()),
                                             self._stream()), 'mi_ode_eval_rhs')
        return f

    def rk_step(self, y0, f0, t0, dt, want_k=False):
        """One attempt with a given dt: (y1, f1, norms[4], k or None) - the _runge_kutta_step parity surface."""
        y1, f1 = torch.empty_like(y0), torch.empty_like(y0)
        S = self.desc.tableau.n_stages
        k = torch.empty((S + 1,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device) if want_k else None
        norms = (C.c_double * 4)()
        with torch.cuda.device(self.device):
            N.check(self.lib.mi_ode_rk_step_fused(self.h, C.c_void_p(y0.data_ptr()), C.c_void_p(f0.data_ptr()), float(t0),
                                                  float(dt), C.c_void_p(y1.data_ptr()), C.c_void_p(f1.data_ptr()), norms,
                                                  C.c_void_p(k.data_ptr()) if want_k else C.c_void_p(0), self._stream()),
                    'mi_ode_rk_step_fused')
        return y1, f1, [norms[i] for i in range(4)], k


# Engines (workspace + handle) are cached across odeint() calls: a call then costs no hipMalloc/hipFree
# (hipFree synchronises the device) - ODEBlock-style callers integrate the same shapes over and over.
_ENGINE_CACHE = collections.OrderedDict()
_ENGINE_CACHE_MAX = 8


def clear_engine_cache():
    _NO_ENGINE.clear()
    while _ENGINE_CACHE:
        _, eng = _ENGINE_CACHE.popitem()
        eng.close()


class SyncTimeout(RuntimeError):
    """The in-kernel grid hand-off of a one-launch kernel timed out (the GPU is shared with another persistent kernel).  Nothing
    was committed: callers with a per-step loop (the multistep solvers) take it instead."""


_NO_ENGINE = set()            # keys whose one-launch engine could not be created (e.g. the batch's workgroups are not co-resident):
                              # remembered, so that every later odeint() call of that shape does not repeat mi_ode_create's
                              # hipMalloc / hipFree (which synchronises the device) before taking its per-step loop


def _cached_engine_or_none(key, factory):
    """_cached_engine, but "there is no such kernel for this problem" (MI_ODE_E_INVALID: family / shape / co-residency - a property of the
    key) is remembered under `key` and answered with None from then on.  A HIP failure at creation (out of memory, a busy device) is a
    property of the MOMENT: it is answered with None once, with a warning, and the next call tries again (advisor, round 4: a transient
    failure used to pin the shape to the slow per-step loop for the rest of the process, silently)."""
    if key in _NO_ENGINE:
        return None
    try:
        return _cached_engine(key, factory)
    except N.NativeError as e:
        if getattr(e, 'rc', None) == N.E_INVALID:
            if len(_NO_ENGINE) > 256:
                _NO_ENGINE.clear()
            _NO_ENGINE.add(key)
        else:
            import warnings
            warnings.warn('one-launch engine unavailable for this call (%s): taking the per-step loop; the next call tries again' % e)
        return None


def _cached_engine(key, factory):
    eng = _ENGINE_CACHE.get(key)
    if eng is not None:
        _ENGINE_CACHE.move_to_end(key)
        return eng
    eng = factory()
    _ENGINE_CACHE[key] = eng
    while len(_ENGINE_CACHE) > _ENGINE_CACHE_MAX:
        _, old = _ENGINE_CACHE.popitem(last=False)
        old.close()
    return eng


_TABLEAU_KEYS = {}


def _tableau_key(tb, c_mid):
    """By-value key of a tableau; the module-level tableau objects are immutable, so the tuples are built once per object."""
    memo = (id(tb), id(c_mid))
    hit = _TABLEAU_KEYS.get(memo)
    if hit is None or hit[0] is not tb or hit[1] is not c_mid:
        hit = (tb, c_mid, (tuple(tb.alpha), tuple(tuple(r) for r in tb.beta), tuple(tb.c_sol), tuple(tb.c_error),
                           None if c_mid is None else tuple(c_mid)))
        if len(_TABLEAU_KEYS) > 64:
            _TABLEAU_KEYS.clear()
        _TABLEAU_KEYS[memo] = hit
    return hit[2]


_COOP_TOLD = set()


def _warn_once_coop(rhs, y, err):
    key = (id(type(rhs)), int(rhs.dim), tuple(y.shape))
    if key not in _COOP_TOLD:
        _COOP_TOLD.add(key)
        import warnings
        warnings.warn('tfdiffeq_amd: no one-launch kernel for this rhs.CustomCoop problem (%s); its torch_fn runs as a Python callable on the '
                      'device-controlled engine instead' % (str(err).split(':')[-1].strip(),))


def _fusable_tuple(func, y0):
    """The row-local DeviceRHS behind a `rhs.PerComponent` lift if this tuple state can travel as one segmented buffer."""
    rhs = getattr(func, 'device_rhs', None)
    if rhs is None or not getattr(func, 'per_component', False) or not getattr(rhs, 'row_local', False):
        return None
    if not 2 <= len(y0) <= N.MAX_SEGMENTS:
        return None
    y = y0[0]
    for c in y0:
        if not (isinstance(c, torch.Tensor) and c.is_cuda and c.dim() >= 1 and c.numel() > 0 and c.dtype == y.dtype and
                c.device == y.device and rhs.supports(c)):
            return None
    return rhs


def _pack_components(y0, dim):
    """(packed [rows, dim] buffer, rows per component, row offset per component): every component starts on a multiple of
    _native.SEGMENT_ALIGN rows (include/mi_ode.h: mi_ode_desc.n_segments); padding rows are never touched by the kernels."""
    rows = [int(c.numel() // dim) for c in y0]
    offs, total = [], 0
    for r in rows:
        offs.append(total)
        total += (r + N.SEGMENT_ALIGN - 1) // N.SEGMENT_ALIGN * N.SEGMENT_ALIGN
    packed = torch.zeros((total, dim), dtype=y0[0].dtype, device=y0[0].device)
    for c, r, o in zip(y0, rows, offs):
        packed[o:o + r].copy_(c.reshape(r, dim))
    return packed, rows, offs


def _fusable(func, y0, multistep=False):
    """The DeviceRHS behind `func` if the fused engine can run this problem, else None.  multistep: for the one-launch Adams kernels
    (a family may take more there than on its Runge-Kutta kernels: rhs.MLP in float64 / up to 256 wide, round 5)."""
    rhs = getattr(func, 'device_rhs', None)
    if rhs is None or len(y0) != 1:
        return None
    y = y0[0]
    ok = rhs.supports_multistep if multistep else rhs.supports
    if not (isinstance(y, torch.Tensor) and y.is_cuda and y.dim() >= 1 and y.numel() > 0 and ok(y)):
        return None
    return rhs


# ---------------------------------------------------------------------------------------------
# base classes
# ---------------------------------------------------------------------------------------------
class AdaptiveStepsizeODESolver(object):
    __metaclass__ = abc.ABCMeta

    def __init__(self, func, y0, atol, rtol, **unused_kwargs):
        _handle_unused_kwargs(self, unused_kwargs)
        del unused_kwargs
        self.func = func
        self.y0 = y0
        self.atol = atol
        self.rtol = rtol

    def before_integrate(self, t):
        pass

    @abc.abstractmethod
    def advance(self, next_t):
        raise NotImplementedError

    def integrate(self, t):
        """solvers.py:27-35."""
        _assert_increasing(t)
        solution = [self.y0]
        t = t.to(torch.float64)                       # :30 time is ALWAYS float64 in adaptive solvers
        self.before_integrate(t)
        for i in range(1, t.shape[0]):
            y = self.advance(t[i])
            solution.append(y)
        return tuple(map(torch.stack, tuple(zip(*solution))))


_EULER_SHAPE = collections.namedtuple('_T', 'alpha beta c_sol c_error')(alpha=[], beta=[], c_sol=[1.0], c_error=[0.0])   # (the tableau slot of a
                                                                                       # multistep descriptor: unused by the kernel)


class FixedGridODESolver(object):
    __metaclass__ = abc.ABCMeta

    def __init__(self, func, y0, step_size=None, grid_constructor=None, eps=0.0, **unused_kwargs):
        unused_kwargs.pop('rtol', None)
        unused_kwargs.pop('atol', None)
        self._pg = unused_kwargs.pop('process_group', None)     # accepted for symmetry; fixed grid needs no exchange
        self._fusion = unused_kwargs.pop('fusion', 0)
        self._graph = bool(unused_kwargs.pop('graph', False))    # plane path: one captured step, replayed per grid interval
        _handle_unused_kwargs(self, unused_kwargs)
        del unused_kwargs
        self.func = func
        self.y0 = y0
        self.eps = eps
        if step_size is not None and grid_constructor is None:
            self.grid_constructor = self._grid_constructor_from_step_size(step_size)
        elif grid_constructor is None:
            self.grid_constructor = lambda f, y0, t: t
            self._default_grid = True
        else:
            raise ValueError("step_size and grid_constructor are exclusive arguments.")     # solvers.py:56

    def _grid_constructor_from_step_size(self, step_size):
        # The reference's version (solvers.py:58-71) is dead code (F7: .item() / item assignment on TF tensors).
        # This is the evident intent: a uniform grid of `step_size` clipped to t[-1].
        def _grid_constructor(func, y0, t):
            start_time, end_time = t[0], t[-1]
            niters = int(math.ceil(float((end_time - start_time) / step_size + 1)))
            t_infer = torch.arange(0, niters, dtype=t.dtype) * step_size + start_time
            if t_infer[-1] > t[-1]:
                t_infer[-1] = t[-1]
            return t_infer
        return _grid_constructor

    @property
    @abc.abstractmethod
    def order(self):
        pass

    @abc.abstractmethod
    def step_func(self, func, t, dt, y):
        pass

    _fused_tableau = None      # subclasses with a fused kernel set a _ButcherTableau here

    def _fused_multistep(self):
        """(kind, max_order, max_iters, min_order, ab, am, am0) for the one-launch multistep kernel, or None (fixed_adams.py sets it)."""
        return None

    def integrate(self, t):
        """solvers.py:82-104."""
        _assert_increasing(t)
        t = t.to(self.y0[0].dtype)                    # :84 time in the STATE dtype here
        ms = self._fused_multistep()
        rhs = _fusable(self.func, self.y0) if ms is None else None      # (a multistep solver asks below, with the multistep kernels' own limits)
        if rhs is None and ms is None and len(self.y0) == 1 and self._fused_tableau is not None:
            # a network outside the tile kernels' box (float64, wide): euler / rk4 on the cooperative one-launch kernel (round 5)
            cand = getattr(self.func, 'device_rhs', None)
            y_ = self.y0[0]
            if cand is not None and hasattr(cand, 'supports_coop') and isinstance(y_, torch.Tensor) and y_.is_cuda and y_.numel() > 0 and \
                    cand.supports_coop(y_):
                rhs = cand
        default_grid = getattr(self, '_default_grid', False)
        time_grid = None
        if rhs is None and default_grid and self.eps == 0.0 and self._fused_tableau is not None:
            # a tuple state of a row-local RHS (rhs.PerComponent): a fixed grid has no norms, so the components simply share one
            # buffer (rows are independent trajectories) and the one-launch kernel
            trhs = _fusable_tuple(self.func, self.y0)
            if trhs is not None and trhs.fixed_grid_fused:
                dim = trhs.dim
                rows = [int(c.numel() // dim) for c in self.y0]
                y = torch.cat([c.reshape(-1, dim) for c in self.y0], dim=0).contiguous()
                key = ('fixed', trhs.cache_key(y.dtype, y.device), tuple(y.shape), y.dtype, str(y.device),
                       _tableau_key(self._fused_tableau, None), self._fusion)
                eng = _cached_engine(key, lambda: _FusedEngine(trhs, y, False, self._fused_tableau, fusion=self._fusion))
                out = eng.integrate(t.to(torch.float64).numpy(), y)
                self.stats = eng.stats.as_dict()
                self.stats['components'] = len(rows)
                offs = np.concatenate([[0], np.cumsum(rows)])
                return tuple(out[:, int(o):int(o) + r].reshape((out.shape[0],) + tuple(c.shape)) for c, r, o in zip(self.y0, rows, offs[:-1]))
        rhs_ms = _fusable(self.func, self.y0, multistep=True) if ms is not None else None     # (rhs.MLP: float64 / up to 256 wide here)
        if rhs_ms is not None and getattr(rhs_ms, 'multistep_fused', False) and self._fusion not in (1, 'stage') and not self._graph:
            rhs = rhs_ms
            # the Adams family on a row-local catalogue system: the whole integration - history, predictor, corrector iterations and
            # their batch-wide convergence test - in ONE launch (csrc/mi_ode_adams.h)
            y = self.y0[0]
            key = ('multistep', rhs.cache_key(y.dtype, y.device), tuple(y.shape), y.dtype, str(y.device), ms[:4], float(self.rtol), float(self.atol))
            # (None: e.g. a batch whose workgroups cannot be co-resident - remembered - the per-step loop below)
            eng = _cached_engine_or_none(key, lambda: _FusedEngine(rhs, y, False, _EULER_SHAPE, rtol=self.rtol, atol=self.atol, multistep=ms))
            out = None
            if eng is not None:
                try:
                    if default_grid and self.eps == 0.0:
                        out = eng.integrate(t.to(torch.float64).numpy(), y)
                    else:
                        time_grid = self.grid_constructor(self.func, self.y0, t)
                        assert bool(time_grid[0] == t[0]) and bool(time_grid[-1] == t[-1])        # solvers.py:87
                        out = eng.integrate(t.to(torch.float64).numpy(), y, grid=time_grid.to(torch.float64).numpy(), eps=float(self.eps))
                except SyncTimeout:                   # the grid hand-off timed out (shared GPU): nothing was committed, the per-step
                    out = None                        # loop below does the same arithmetic
            if out is not None:
                self.stats = eng.stats.as_dict()
                self.stats['engine'] = 'fused multistep kernel (one launch)'
                for _ in range(int(self.stats.get('n_rejected', 0))):                         # fixed_adams.py:197-199
                    print('Warning: Functional iteration did not converge. Solution may be incorrect.', file=sys.stderr)
                return (out,)
            rhs = None
        if rhs is not None and rhs.fixed_grid_fused and self._fused_tableau is not None:
            y = self.y0[0]
            key = ('fixed', rhs.cache_key(y.dtype, y.device), tuple(y.shape), y.dtype, str(y.device),
                   _tableau_key(self._fused_tableau, None), self._fusion)
            # (a family that has no kernel for the requested schedule - the MLP with fusion='stage' has none: whole-attempt kernels
            # only - takes the per-step loop below, as before its fixed-grid kernel existed; advisor, round 4)
            eng = _cached_engine_or_none(key, lambda: _FusedEngine(rhs, y, False, self._fused_tableau, fusion=self._fusion))
            if eng is not None and default_grid and self.eps == 0.0:
                out = eng.integrate(t.to(torch.float64).numpy(), y)
                self.stats = eng.stats.as_dict()
                return (out,)
            if eng is not None and self._fusion not in (1, 'stage'):
                # a grid of its own (step_size / grid_constructor) and / or eps: still one launch - the kernel walks the grid
                # and interpolates the requested times linearly inside the step that reaches them (solvers.py:86-115)
                time_grid = self.grid_constructor(self.func, self.y0, t)
                assert bool(time_grid[0] == t[0]) and bool(time_grid[-1] == t[-1])            # solvers.py:87
                try:
                    out = eng.integrate(t.to(torch.float64).numpy(), y, grid=time_grid.to(torch.float64).numpy(), eps=float(self.eps))
                    self.stats = eng.stats.as_dict()
                    return (out,)
                except N.NativeError:
                    pass                                  # no one-launch kernel for this family: the per-step loop below
        if time_grid is None:
            time_grid = self.grid_constructor(self.func, self.y0, t)
        assert bool(time_grid[0] == t[0]) and bool(time_grid[-1] == t[-1])
        for y_ in self.y0:
            N.require_gpu_tensor(y_, 'y0')
        if self._graph and getattr(self, '_default_grid', False) and t.shape[0] > 1:
            # one captured step, replayed per grid interval with no host synchronisation in the loop (graph_step.py).
            # dt is formed in the state dtype on the host exactly like below (the values are exact in float64).
            from .graph_step import GraphedFixedStep
            g = GraphedFixedStep(self, self.y0)
            tt = t.numpy()
            grid64 = tt.astype(np.float64)
            dts = (tt[1:] - tt[:-1]).astype(np.float64)       # t1 - t0 in the state dtype (solvers.py:94)
            outs = tuple(torch.empty((t.shape[0],) + tuple(y.shape), dtype=y.dtype, device=y.device) for y in self.y0)
            for out, y in zip(outs, self.y0):
                out[0].copy_(y)
            g.integrate_pairs(grid64[:-1], dts, outs)
            self.stats = {'engine': 'plane kernels, one hipGraph replay per step', 'n_attempts': int(t.shape[0] - 1),
                          'n_accepted': int(t.shape[0] - 1), 'status': 0}
            return outs
        solution = [self.y0]
        j = 1
        y0 = self.y0
        grid = time_grid.numpy()
        tt = t.numpy()
        for t0, t1 in zip(grid[:-1], grid[1:]):
            dy = self.step_func(self.func, t0, t1 - t0, y0)
            y1 = tuple(_lincomb(y0_, [1.0], [dy_], 1.0) for y0_, dy_ in zip(y0, dy))
            while j < tt.shape[0] and t1 >= tt[j]:
                solution.append(self._linear_interp(t0, t1, y0, y1, tt[j]))
                j += 1
            y0 = y1
        return tuple(map(torch.stack, tuple(zip(*solution))))

    def _linear_interp(self, t0, t1, y0, y1, t):
        """solvers.py:106-115."""
        if t == t0:
            return y0
        if t == t1:
            return y1
        # the reference's operation order: slope = (y1 - y0) / (t1 - t0); y0 + slope * (t - t0), scalars in the state dtype.
        # Only reachable with step_size (the default grid hits every requested time exactly), so plain torch ops do.
        dt_ = _np_dtype(y0[0].dtype).type
        span, off = dt_(t1) - dt_(t0), dt_(t) - dt_(t0)
        return tuple(y0_ + ((y1_ - y0_) / span) * off for y0_, y1_ in zip(y0, y1))


# ---------------------------------------------------------------------------------------------
# shared adaptive Runge-Kutta driver (dopri5.py:48-121, bosh3.py:34-99, tsit5.py:69-151)
# ---------------------------------------------------------------------------------------------
class _AdaptiveRKSolver(AdaptiveStepsizeODESolver):
    tableau = None
    c_mid = None
    order = 5
    init_order = 4
    controller = N.CTRL_MISC
    interp = N.INTERP_QUARTIC_MID
    pooled_ratio = False          # tsit5 pools all components into one mean (tsit5.py:134-137)

    def _setup(self, func, y0, rtol, atol, first_step, safety, ifactor, dfactor, max_num_steps, unused_kwargs):
        from .misc import _convert_to_tensor, _is_iterable
        self._pg = unused_kwargs.pop('process_group', None)
        self._linear_variant = unused_kwargs.pop('linear_variant', 0)
        self._chunk_attempts = unused_kwargs.pop('chunk_attempts', 0)
        self._force_planes = unused_kwargs.pop('force_plane_kernels', False)
        self._profile = unused_kwargs.pop('profile', False)
        # Python-callable path (graph_step.DeviceControlledRK): 'auto' (default) - the controller runs on the device, attempts are
        # evaluated eagerly first and replayed as one hipGraph once enough of them remain to pay for the recording; True - record
        # after the first attempt; False - never record; 'host' - the round-1 loop with the controller on the host (one
        # synchronisation per attempt), which is also what a process group or force_plane_kernels selects
        self._graph = unused_kwargs.pop('graph', 'auto')
        # 'reuse' - like True, and the recorded attempt is kept for later calls with the same callable, shapes and tolerances
        # (graph_step: the caller promises f stays the same function up to in-place updates of the tensors it reads)
        if self._graph not in ('auto', 'host', 'reuse', True, False):
            raise ValueError("options['graph'] must be True, False, 'auto', 'host' or 'reuse'")
        self._graph_attempt = None
        self._fusion = unused_kwargs.pop('fusion', 0)
        _handle_unused_kwargs(self, unused_kwargs)
        self.func = func
        self.y0 = y0
        if self.pooled_ratio:
            self.rtol, self.atol = rtol, atol                                     # tsit5.py:81-82
        else:
            self.rtol = rtol if _is_iterable(rtol) else [rtol] * len(y0)          # dopri5.py:60-61
            self.atol = atol if _is_iterable(atol) else [atol] * len(y0)
        self.first_step = first_step
        # dopri5.py:63-66 through misc.py:137-144: python float -> float32 -> float64
        self.safety = _convert_to_tensor(safety, dtype=np.float64)
        self.ifactor = _convert_to_tensor(ifactor, dtype=np.float64)
        self.dfactor = _convert_to_tensor(dfactor, dtype=np.float64)
        self.max_num_steps = int(max_num_steps)
        self._engine = None
        self._exchange = _Exchange(self._pg) if self._pg is not None else None
        self.stats = {}

    # -- fused engine ----------------------------------------------------------------------------
    def _make_engine(self):
        rhs = None if self._force_planes else _fusable(self.func, self.y0)
        from .rk_common import _is_fsal_shaped
        one_row = len(self.tableau.alpha) == 1 and not _is_fsal_shaped(self.tableau)
        if rhs is not None and one_row and hasattr(rhs, 'supports_coop'):
            rhs = None                                       # adaptive_heun: the MLP tile kernels have no 1-row tableau - the cooperative kernel does
        self._packed = None
        if rhs is None and not self._force_planes and self._pg is None and self._fusion in (0, 'auto', 4, 'whole'):
            rhs = _fusable_tuple(self.func, self.y0)            # tuple state of a row-local RHS: one segmented buffer
            if rhs is not None:
                self._packed = _pack_components(self.y0, rhs.dim)
        self._coop = False
        if rhs is None and not self._force_planes and self._pg is None and self._fusion in (0, 'auto', 4, 'whole') and len(self.y0) == 1:
            # a network outside the tile kernels' box (float64, wide): the cooperative whole-call kernel, if the batch is co-resident there
            cand = getattr(self.func, 'device_rhs', None)
            y = self.y0[0]
            if cand is not None and hasattr(cand, 'supports_coop') and isinstance(y, torch.Tensor) and y.is_cuda and y.numel() > 0 and \
                    cand.supports_coop(y, any_box=one_row):
                rhs, self._coop = cand, True
            elif cand is not None and hasattr(cand, 'warn_limits') and isinstance(y, torch.Tensor) and y.is_cuda and not cand.coop_in_box(y):
                cand.warn_limits(y)                                  # (e.g. hidden > 256: no kernel of the family takes it - said once)
        if rhs is None:
            return None
        from .rk_common import _is_fsal_shaped
        fsal, rows = _is_fsal_shaped(self.tableau), len(self.tableau.alpha)
        if not (fsal and rows in (3, 6)) and not self._coop:           # (the cooperative kernel exists for every adaptive tableau)
            # dopri8 (13 rows) and adaptive_heun (1 row, not FSAL shaped): row-local kernels only, no per-stage schedule
            wide = (fsal and rows == 13) or (not fsal and rows == 1)
            ok = getattr(rhs, 'row_local', False) or getattr(rhs, 'wide_tableaus', False) or \
                (fsal and rows == 13 and getattr(rhs, 'tile_dopri8', False))
            if not (wide and ok and self._fusion not in (1, 'stage')):
                return None
        rtol0 = self.rtol if self.pooled_ratio else self.rtol[0]
        atol0 = self.atol if self.pooled_ratio else self.atol[0]
        first = None
        if self.first_step is not None:
            from .misc import _convert_to_tensor
            first = float(_convert_to_tensor(self.first_step, dtype=np.float64))   # dopri5.py:77: float32 detour
        y = self.y0[0] if self._packed is None else self._packed[0]
        seg_rows = None if self._packed is None else tuple(self._packed[1])
        seg_tols = None
        if seg_rows is not None and not self.pooled_ratio and \
                any(float(r_) != float(rtol0) or float(a_) != float(atol0) for r_, a_ in zip(self.rtol, self.atol)):
            seg_tols = tuple((float(r_), float(a_)) for r_, a_ in zip(self.rtol, self.atol))
        args = (float(rtol0), float(atol0), self.controller, self.interp, self.order, self.init_order, float(self.safety),
                float(self.ifactor), float(self.dfactor), first, self.max_num_steps)
        key = ('adaptive', rhs.cache_key(y.dtype, y.device), tuple(y.shape), y.dtype, str(y.device),
               _tableau_key(self.tableau, self.c_mid), args, id(self._pg) if self._pg is not None else None,
               self._linear_variant, self._chunk_attempts, bool(self._profile), self._fusion, seg_rows, seg_tols)
        if self._coop:                                           # (None: the batch's workgroups are not co-resident - remembered under the key)
            eng = _cached_engine_or_none(key, lambda: _FusedEngine(
                rhs, y, True, self.tableau, self.c_mid, *args, linear_variant=self._linear_variant, chunk_attempts=self._chunk_attempts,
                profile=self._profile, fusion=self._fusion))
            if eng is None:
                rhs.warn_limits(y, 'batch %d' % int(y.numel() // rhs.dim))
            return eng
        try:
            return _cached_engine(key, lambda: _FusedEngine(
                rhs, y, True, self.tableau, self.c_mid, *args, process_group=self._pg, linear_variant=self._linear_variant,
                chunk_attempts=self._chunk_attempts, profile=self._profile, fusion=self._fusion, seg_rows=seg_rows, seg_tols=seg_tols))
        except N.NativeError as e:
            if seg_rows is None:
                if getattr(e, 'rc', None) == N.E_INVALID and getattr(rhs, 'wide_tableaus', False) and \
                        (getattr(rhs, 'torch_fn', None) is not None or 'forward' in vars(rhs)):
                    # rhs.CustomCoop whose batch is not co-resident on the cooperative kernel (its only schedule), with a torch_fn:
                    # the same function as a Python callable on the device-controlled engine - said once
                    _warn_once_coop(rhs, y, e)
                    return None
                raise
            self._packed = None                                  # e.g. more workgroups than are co-resident: the generic path
            return None

    def integrate(self, t):
        _assert_increasing(t)
        eng = self._make_engine()
        if eng is not None and self._coop:
            # the cooperative kernel has the whole-call schedule only: no output beyond t[0], or a hand-off that timed out on a shared
            # GPU (nothing was committed) -> the device-controlled engine below
            out = None
            if len(t) > 1:
                try:
                    out = eng.integrate(t.to(torch.float64).numpy(), self.y0[0])
                except SyncTimeout:
                    out = None
            if out is not None:
                self.stats = eng.stats.as_dict()
                self.stats['cross_rank'] = eng.transport
                self.stats['engine'] = 'cooperative whole-call kernel (one launch, a thread per state element)'
                return (out,)
            eng = None
        if eng is None:
            out = self._integrate_device_controlled(t)
            if out is not None:
                return out
            out = super(_AdaptiveRKSolver, self).integrate(t)
            self.stats = {'engine': 'plane kernels', 'n_attempts': getattr(self, '_n_attempts', 0),
                          'n_accepted': getattr(self, '_n_accepted', 0), 'status': 0}
            return out
        prof0 = eng.profile() if self._profile else None
        try:
            out = eng.integrate(t.to(torch.float64).numpy(), self.y0[0] if self._packed is None else self._packed[0])
        finally:
            self.stats = eng.stats.as_dict()
            self.stats['cross_rank'] = eng.transport
            if eng.transport_log:
                self.stats['cross_rank_log'] = list(eng.transport_log)
            if self._profile:
                self.stats['profile'] = [a - b for a, b in zip(eng.profile(), prof0)]
        if self._packed is not None:                             # [T, padded rows, dim] -> one [T, *shape] tensor per component
            _, rows, offs = self._packed
            self.stats['components'] = len(rows)
            return tuple(out[:, o:o + r].reshape((out.shape[0],) + tuple(c.shape)) for c, r, o in zip(self.y0, rows, offs))
        return (out,)

    # -- Python callable, controller on the device (graph_step.DeviceControlledRK) -----------------
    def _integrate_device_controlled(self, t):
        """AdaptiveStepsizeODESolver.integrate (solvers.py:27-35) with the attempt loop's scalars on the device; None when this
        problem has to take the host-controlled loop below (process group, mixed dtypes, an empty component, the 'host' option)."""
        if self._graph == 'host' or self._force_planes or self._exchange is not None:
            return None
        y0 = self.y0
        if not (1 <= len(y0) <= N.MAX_SEGMENTS) or len(self.tableau.alpha) + 1 not in (2, 4, 7, 14):
            return None
        like = y0[0]
        if like.dtype not in (torch.float32, torch.float64):
            return None
        if any(y.dtype != like.dtype or y.device != like.device or y.numel() == 0 or not y.is_cuda for y in y0):
            return None
        if self.interp != N.INTERP_QUARTIC_MID and len(self.tableau.alpha) != 6:
            return None
        from .graph_step import DeviceControlledRK, _credit_nfe, keep_recorded, recorded_engine
        t64 = t.to(torch.float64)                     # solvers.py:30
        self.before_integrate(t64)                    # f0 and the first step size, as the reference forms them (dopri5.py:70-79)
        reuse = self._graph == 'reuse'
        eng = recorded_engine(self) if reuse else None
        if eng is None:
            eng = DeviceControlledRK(self, graph=True if reuse else self._graph)
        try:
            outs = eng.integrate(t64.numpy(), y0, self.rk_state.f1, float(self.rk_state.dt))
            st = eng.stats.as_dict()
            info = dict(eng.info)
            py_calls = eng.py_calls
        except BaseException:
            eng.close()
            raise
        if reuse and eng.captured and eng.graph is not None:
            keep_recorded(self, eng)
        else:
            eng.close()
        self.stats = st
        self.stats.update(info)
        self.stats['n_polls'] = info.get('polls', 0)
        # Python side effects of f (an evaluation counter, tests/DETEST/run.py:18-21) happen once per RECORDED evaluation: credit
        # the evaluations the replays performed to an integer `nfe` attribute of the callable, if it keeps one
        _credit_nfe(self.func, int(st['nfe']) - py_calls)
        bits = int(st['status'])
        if bits:
            msg = N.status_message(bits)
            if bits & N.ST_MAX_STEPS:
                msg = 'max_num_steps exceeded ({}>={})'.format(self.max_num_steps, self.max_num_steps)
            if bits & N.ST_DT_UNDERFLOW:
                msg = 'underflow in dt {}'.format(st['dt'])
            raise AssertionError(msg)                 # dopri5.py:85-100
        return outs

    # -- plane-kernel path (any callable, tuple states) -------------------------------------------
    def before_integrate(self, t):
        from .misc import _convert_to_tensor, _select_initial_step
        from .rk_common import _RungeKuttaState
        for y_ in self.y0:
            N.require_gpu_tensor(y_, 'y0')
        like = self.y0[0]
        t0 = float(t[0])
        if self.pooled_ratio:                                                     # tsit5.py:91-103
            if self.first_step is None:
                first = np.float64(_select_initial_step(self.func, t0, self.y0, self.init_order, self.rtol, self.atol,
                                                        exchange=self._exchange))
            else:
                first = _convert_to_tensor(self.first_step, dtype=np.float64)
            f0 = self.func(torch.full((), t0, dtype=torch.float64, device=like.device), self.y0)   # :99 t[0] un-cast
        else:                                                                     # dopri5.py:70-79
            f0 = self.func(_scalar_tensor(_np_dtype(like.dtype).type(t0), like), self.y0)
            if self.first_step is None:
                first = np.float64(_select_initial_step(self.func, t0, self.y0, self.init_order, self.rtol[0],
                                                        self.atol[0], f0=f0, exchange=self._exchange))
            else:
                first = _convert_to_tensor(self.first_step, dtype=np.float64)
        self.rk_state = _RungeKuttaState(self.y0, f0, np.float64(t0), np.float64(t0), first, interp_coeff=None)
        self._n_attempts = self._n_accepted = 0

    def advance(self, next_t):
        """Interpolate through the next time point, integrating as necessary (dopri5.py:81-89)."""
        next_t = np.float64(float(next_t))
        n_steps = 0
        while next_t > self.rk_state.t1:
            assert n_steps < self.max_num_steps, 'max_num_steps exceeded ({}>={})'.format(n_steps, self.max_num_steps)
            self.rk_state = self._adaptive_step(self.rk_state)
            n_steps += 1
        return self._dense_output(self.rk_state, next_t)

    def _dense_output(self, rk_state, next_t):
        from .interp import _interp_eval_step
        ic = rk_state.interp_coeff
        if ic is None:                   # no accepted step yet: the reference's [y0] * 5 placeholder evaluates to y0
            return rk_state.y1
        y0, y1, k, dt = ic
        return _interp_eval_step(self.interp, y0, y1, k, self.c_mid, dt, rk_state.t0, rk_state.t1, next_t)

    def _adaptive_step(self, rk_state):
        """Take an adaptive Runge-Kutta step (dopri5.py:91-121 / bosh3.py:70-99 / tsit5.py:113-151)."""
        from .misc import _compute_error_ratio, _optimal_step_size, _ratio_from_norms
        from .rk_common import _RungeKuttaState, _runge_kutta_step
        y0, f0, _, t0, dt, interp_coeff = rk_state
        dt = np.float64(dt)
        assert t0 + dt > t0, 'underflow in dt {}'.format(dt)
        recs = None
        if self._graph is True and self._exchange is None:
            # the whole attempt (S evaluations of f, stage arithmetic, error norms) is one hipGraph replay (graph_step.py)
            g = self._graph_attempt
            if g is None or not g.matches(y0):
                from .graph_step import GraphedAttempt
                g = self._graph_attempt = GraphedAttempt(self.func, y0, f0, self.tableau)
            y0, y1, f1, y1_error, k, recs = g.run(y0, f0, t0, dt)
        else:
            y1, f1, y1_error, k = _runge_kutta_step(self.func, y0, f0, t0, dt, tableau=self.tableau)
        if self.pooled_ratio:
            ratios = self._pooled_ratio(y1_error, y0, y1, recs=recs)
            accept_step = bool(ratios[0] <= 1.)
            dt_next = self._tsit5_step_size(dt, ratios[0])
        else:
            ratios = _compute_error_ratio(y1_error, atol=self.atol, rtol=self.rtol, y0=y0, y1=y1, exchange=self._exchange,
                                          recs=recs)
            accept_step = bool(np.all(np.asarray([float(r) for r in ratios]) <= 1))
            dt_next = _optimal_step_size(dt, ratios, safety=self.safety, ifactor=self.ifactor, dfactor=self.dfactor,
                                         order=self.order)
        assert not _compute_error_ratio.last_nonfinite, 'non-finite values in state `y`'       # dopri5.py:99-100
        self._n_attempts += 1
        if accept_step:
            self._n_accepted += 1
            if recs is not None:
                self._graph_attempt.accepted(y1, f1)
            return _RungeKuttaState(y1, f1, t0, t0 + dt, dt_next, (y0, y1, k, dt))
        if recs is not None:
            f0 = self._graph_attempt.f0
        return _RungeKuttaState(y0, f0, t0, t0, dt_next, interp_coeff)

    def _pooled_ratio(self, y1_error, y0, y1, recs=None):
        """tsit5.py:126-138: scalar rtol/atol, one mean over ALL components."""
        from .misc import _error_norms
        if recs is None:
            recs = torch.stack([_error_norms(e, a, b) for e, a, b in zip(y1_error, y0, y1)])
        counts = torch.tensor([[float(e.numel())] for e in y1_error], dtype=torch.float64, device=recs.device)
        recs = torch.cat([recs, counts], dim=1)
        if self._exchange is not None:
            recs = self._exchange.combine(recs, max_slots=(0, 1, 3))
        host = recs.cpu().numpy()
        from .misc import _compute_error_ratio
        _compute_error_ratio.last_nonfinite = bool((host[:, 3] != 0).any())
        dt_ = _np_dtype(y0[0].dtype).type
        num, den = 0.0, 0.0
        with np.errstate(all='ignore'):
            for i in range(host.shape[0]):
                tol = dt_(self.atol) + dt_(self.rtol) * dt_(max(host[i, 0], host[i, 1]))
                num += host[i, 2] / (float(tol) * float(tol))
                den += host[i, 4]
            return (dt_(num / den),)

    def _tsit5_step_size(self, last_step, mean_error_ratio):
        """tsit5.py:53-62."""
        ifactor, dfactor, safety = self.ifactor, self.dfactor, self.safety
        if mean_error_ratio == 0:
            return np.float64(last_step * ifactor)
        if mean_error_ratio < 1:
            dfactor = np.float64(1.)
        with np.errstate(all='ignore'):
            error_ratio = np.float64(mean_error_ratio)
            exponent = np.float64(1. / self.order)
            factor = np.maximum(np.float64(1. / ifactor), np.minimum((error_ratio ** exponent) / safety, np.float64(1. / dfactor)))
            return np.float64(last_step / factor)

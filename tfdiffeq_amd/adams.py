"""Mirror of tfdiffeq/adams.py: variable-step, variable-order Adams-Bashforth-Moulton (Hairer, Norsett, Wanner
III.5), `method='adams'`.

Reference behaviour kept verbatim: the g vector lives in a float32 variable (adams.py:34, 41-60), the accepted
state advances with the PREDICTOR value p_next (adams.py:210), steps are clipped to land exactly on the requested
times (adams.py:130-131), the first step is always the Hairer heuristic of order 2 (adams.py:115-118).
A row-local catalogue / plugin system with one state tensor runs the whole call in one launch (csrc/mi_ode_adams_vc.h); otherwise
state-sized arithmetic runs in four plane kernels per attempt (csrc/mi_ode_adams_planes.h) and the scalar bookkeeping (g, beta,
orders) stays on the host.
"""
import collections
import ctypes as C
import json
import os

import numpy as np
import torch

from . import _native as N
from .misc import (_contig, _convert_to_tensor, _handle_unused_kwargs, _is_iterable, _np_dtype, _optimal_step_size, _reduce_workspace,
                   _scalar_tensor, _select_initial_step)
from .solvers import AdaptiveStepsizeODESolver

_MIN_ORDER = 1
_MAX_ORDER = 12
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tableaus', 'adams.json')) as _fh:
    gamma_star = json.load(_fh)['gamma_star']                                            # adams.py:15-18

_VCABMState = collections.namedtuple('_VCABMState', 'y_n, prev_f, prev_t, next_t, phi, order')


def g_and_beta(prev_t, next_t, k):
    """The scalars of adams.py:29-63: g (float32 variable, :34) and beta_j, j < k (beta_0 = 1); the explicit phi planes
    beta_j * phi_j are formed inside the plane kernels."""
    curr_t = prev_t[0]
    dt = next_t - prev_t[0]
    g = np.zeros(k + 1, dtype=np.float32)
    betas = [np.float64(1.0)]
    beta = np.float64(1.0)
    g[0] = 1
    c = 1 / np.arange(1, k + 2).astype(np.float64)
    for j in range(1, k):
        beta = (next_t - prev_t[j - 1]) / (curr_t - prev_t[j]) * beta
        betas.append(beta)
        c = c[:-1] - c[1:] if j == 1 else c[:-1] - c[1:] * dt / (next_t - prev_t[j - 1])
        g[j] = np.float32(c[0])
    c = c[:-1] - c[1:] * dt / (next_t - prev_t[k - 1])
    g[k] = np.float32(c[0])
    return g, betas


class VariableCoefficientAdamsBashforth(AdaptiveStepsizeODESolver):

    def __init__(self, func, y0, rtol, atol, implicit=True, first_step=None, max_order=_MAX_ORDER, safety=0.9,
                 ifactor=10.0, dfactor=0.2, **unused_kwargs):
        self._force_planes = unused_kwargs.pop('force_plane_kernels', False)
        _handle_unused_kwargs(self, unused_kwargs)
        self.func = func
        self.y0 = y0
        self.rtol = rtol if _is_iterable(rtol) else [rtol] * len(y0)
        self.atol = atol if _is_iterable(atol) else [atol] * len(y0)
        self.implicit = implicit
        self.first_step = first_step
        self.max_order = int(max(_MIN_ORDER, min(max_order, _MAX_ORDER)))
        self.safety = _convert_to_tensor(safety, dtype=np.float64)
        self.ifactor = _convert_to_tensor(ifactor, dtype=np.float64)
        self.dfactor = _convert_to_tensor(dfactor, dtype=np.float64)
        self.stats = {'engine': 'plane kernels', 'n_attempts': 0, 'n_accepted': 0, 'status': 0}

    def _f(self, t, y):
        like = self.y0[0]
        return self.func(_scalar_tensor(_np_dtype(like.dtype).type(t), like), y)

    def integrate(self, t):
        """A row-local catalogue system with a single state tensor: the whole call - the deque of backward differences, g / beta,
        the error ratios, the order selection - is ONE kernel launch (csrc/mi_ode_adams_vc.h).  Everything else: the per-step loop of
        the base class over plane kernels."""
        from .solvers import _EULER_SHAPE, _FusedEngine, _cached_engine_or_none, _fusable, SyncTimeout
        from .misc import _assert_increasing
        rhs = _fusable(self.func, self.y0, multistep=True) if not self._force_planes else None
        if rhs is not None and getattr(rhs, 'multistep_fused', False) and len(self.y0) == 1:
            _assert_increasing(t)
            y = self.y0[0]
            key = ('adams_vc', rhs.cache_key(y.dtype, y.device), tuple(y.shape), y.dtype, str(y.device), self.max_order,
                   float(self.rtol[0]), float(self.atol[0]), float(self.safety), float(self.ifactor), float(self.dfactor))
            # (None: a batch whose workgroups cannot be co-resident - remembered under the key - the per-step loop)
            eng = _cached_engine_or_none(key, lambda: _FusedEngine(rhs, y, True, _EULER_SHAPE, rtol=self.rtol[0], atol=self.atol[0],
                                                                   safety=float(self.safety), ifactor=float(self.ifactor),
                                                                   dfactor=float(self.dfactor), multistep=(3, self.max_order, gamma_star)))
            out = None
            if eng is not None:
                try:
                    out = eng.integrate(t.to(torch.float64).numpy(), y)
                except SyncTimeout:                   # hand-off timed out (shared GPU): nothing was committed, take the per-step loop
                    out = None
            if out is not None:
                self.stats = eng.stats.as_dict()
                self.stats['engine'] = 'fused variable-order Adams kernel (one launch)'
                return (out,)
        return super(VariableCoefficientAdamsBashforth, self).integrate(t)

    def before_integrate(self, t):
        for y_ in self.y0:
            N.require_gpu_tensor(y_, 'y0')
        prev_f = collections.deque(maxlen=self.max_order + 1)
        prev_t = collections.deque(maxlen=self.max_order + 1)
        phi = collections.deque(maxlen=self.max_order)
        t0 = np.float64(float(t[0]))
        f0 = self._f(t0, self.y0)
        prev_t.appendleft(t0)
        prev_f.appendleft(f0)
        phi.appendleft(f0)
        first_step = np.float64(_select_initial_step(self.func, t0, self.y0, 2, self.rtol[0], self.atol[0], f0=f0))   # :115-118
        self.vcabm_state = _VCABMState(self.y0, prev_f, prev_t, next_t=t0 + first_step, phi=phi, order=1)

    def advance(self, final_t):
        final_t = np.float64(float(final_t))
        while final_t > self.vcabm_state.prev_t[0]:
            self.vcabm_state = self._adaptive_adams_step(self.vcabm_state, final_t)
        assert final_t == self.vcabm_state.prev_t[0]
        return self.vcabm_state.y_n

    def _adaptive_adams_step(self, vcabm_state, final_t):
        """adams.py:134-210 on four plane kernels per component (csrc/mi_ode_adams_planes.h: predictor; implicit phi + corrector +
        max norms; the error sums; the new phi deque) - the arithmetic of the one-combination-at-a-time formulation, operation for
        operation, in ~9 launches and two read-backs per attempt instead of ~45 launches at order 12."""
        y0, prev_f, prev_t, next_t, prev_phi, order = vcabm_state
        if next_t > final_t:
            next_t = final_t
        dt = next_t - prev_t[0]
        like = y0[0]
        dt_ = _np_dtype(like.dtype).type
        g32, beta = g_and_beta(prev_t, next_t, order)
        g = g32.astype(_np_dtype(like.dtype))
        lib = N.load()
        code = N.dtype_code(like.dtype)
        stream = N.stream_ptr(like.device)
        g_c = (C.c_double * (order + 1))(*[float(v) for v in g])
        b_c = (C.c_double * order)(*[float(v) for v in beta])
        ncomp = len(y0)
        phis = [[_contig(prev_phi[j][c]) for j in range(order)] for c in range(ncomp)]
        phi_c = [(C.c_void_p * order)(*[t_.data_ptr() for t_ in phis[c]]) for c in range(ncomp)]
        y0c = [_contig(y_) for y_ in y0]
        p_next = tuple(torch.empty_like(y_) for y_ in y0c)
        for c in range(ncomp):
            N.check(lib.mi_ode_adams_predict(code, y0c[c].numel(), C.c_void_p(y0c[c].data_ptr()), phi_c[c], order, g_c, b_c, float(dt),
                                             C.c_void_p(p_next[c].data_ptr()), stream), 'mi_ode_adams_predict')
        next_f0 = self._f(next_t, p_next)
        want_lower = not (len(prev_t) <= 4 or order < 3)
        want_higher = want_lower and order < self.max_order
        y_next = tuple(torch.empty_like(y_) for y_ in y0c)
        ipk = tuple(torch.empty_like(y_) for y_ in y0c)
        ipk1 = tuple(torch.empty_like(y_) for y_ in y0c)
        ipk2 = tuple(torch.empty_like(y_) for y_ in y0c) if order >= 3 else (None,) * ncomp
        recs_dev = torch.empty((ncomp, 4), dtype=torch.float64, device=like.device)
        ws = _reduce_workspace(like.device)
        for c in range(ncomp):
            fp = _contig(next_f0[c])
            N.check(lib.mi_ode_adams_correct(code, y0c[c].numel(), C.c_void_p(y0c[c].data_ptr()), C.c_void_p(p_next[c].data_ptr()),
                                             C.c_void_p(fp.data_ptr()), phi_c[c], order, g_c, b_c, float(dt), C.c_void_p(y_next[c].data_ptr()),
                                             C.c_void_p(ipk[c].data_ptr()), C.c_void_p(ipk1[c].data_ptr()),
                                             C.c_void_p(ipk2[c].data_ptr()) if ipk2[c] is not None else None,
                                             C.c_void_p(recs_dev[c].data_ptr()), C.c_void_p(ws.data_ptr()), stream), 'mi_ode_adams_correct')
        recs = recs_dev.cpu().numpy()                                                    # read-back 1: the max norms
        with np.errstate(all='ignore'):                                                  # one scalar per component (F3)
            tolerance = tuple(dt_(self.atol[i]) + dt_(self.rtol[i]) * dt_(max(recs[i, 0], recs[i, 1])) for i in range(ncomp))
        sums_dev = torch.zeros((ncomp, 4), dtype=torch.float64, device=like.device)
        c_k = float(g[order] - g[order - 1])
        c_km1 = float(g[order - 1] - g[order - 2]) if want_lower else 0.0
        c_km2 = float(g[order - 2] - g[order - 3]) if want_lower else 0.0
        c_kp1 = float(dt_(gamma_star[order])) if want_higher else 0.0
        for c in range(ncomp):
            n_el = y0c[c].numel()
            N.check(lib.mi_ode_adams_error_sums(code, n_el, C.c_void_p(ipk[c].data_ptr()), c_k,
                                                C.c_void_p(ipk1[c].data_ptr()) if want_lower else None, c_km1, float(dt), float(tolerance[c]),
                                                C.c_void_p(sums_dev[c].data_ptr()), C.c_void_p(ws.data_ptr()), stream), 'mi_ode_adams_error_sums')
            if want_lower:
                N.check(lib.mi_ode_adams_error_sums(code, n_el, C.c_void_p(ipk2[c].data_ptr()), c_km2,
                                                    C.c_void_p(ipk[c].data_ptr()) if want_higher else None, c_kp1, float(dt), float(tolerance[c]),
                                                    C.c_void_p(sums_dev[c, 2:].data_ptr()), C.c_void_p(ws.data_ptr()), stream),
                        'mi_ode_adams_error_sums')
        sums = sums_dev.cpu().numpy()                                                    # read-back 2: every estimate's sum
        with np.errstate(all='ignore'):
            ratio = lambda col: tuple(_np_dtype(y0c[c].dtype).type(sums[c, col] / float(y0c[c].numel())) for c in range(ncomp))   # noqa: E731
            error_k = ratio(0)
        accept_step = bool(np.all(np.asarray([float(r) for r in error_k]) <= 1))
        self.stats['n_attempts'] += 1
        if not accept_step:
            dt_next = _optimal_step_size(dt, error_k, self.safety, self.ifactor, self.dfactor, order=order)
            return _VCABMState(y0, prev_f, prev_t, prev_t[0] + dt_next, prev_phi, order=order)
        self.stats['n_accepted'] += 1
        next_f0 = self._f(next_t, y_next)
        new_phi = [tuple(torch.empty_like(y_) for y_ in y0c) for _ in range(order + 1)]
        for c in range(ncomp):
            fn = _contig(next_f0[c])
            out_c = (C.c_void_p * (order + 1))(*[new_phi[j][c].data_ptr() for j in range(order + 1)])
            N.check(lib.mi_ode_adams_update_phi(code, y0c[c].numel(), C.c_void_p(fn.data_ptr()), phi_c[c], order, b_c, out_c, stream),
                    'mi_ode_adams_update_phi')
        implicit_phi = collections.deque(new_phi, maxlen=order + 1)
        next_order = order
        if len(prev_t) <= 4 or order < 3:
            next_order = min(order + 1, 3, self.max_order)
        else:
            with np.errstate(all='ignore'):
                error_km1, error_km2 = ratio(1), ratio(2)
            if min(error_km1 + error_km2) < max(error_k):
                next_order = order - 1
            elif order < self.max_order:
                with np.errstate(all='ignore'):
                    error_kp1 = ratio(3)
                if max(error_kp1) < max(error_k):
                    next_order = order + 1
        dt_next = dt if next_order > order else _optimal_step_size(dt, error_k, self.safety, self.ifactor, self.dfactor,
                                                                   order=order + 1)
        prev_f.appendleft(next_f0)
        prev_t.appendleft(next_t)
        return _VCABMState(p_next, prev_f, prev_t, next_t + dt_next, implicit_phi, order=next_order)      # p_next (:210)

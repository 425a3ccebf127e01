"""Mirror of tfdiffeq/adams.py: variable-step, variable-order Adams-Bashforth-Moulton (Hairer, Norsett, Wanner
III.5), `method='adams'`.

Reference behaviour kept verbatim: the g vector lives in a float32 variable (adams.py:34, 41-60), the accepted
state advances with the PREDICTOR value p_next (adams.py:210), steps are clipped to land exactly on the requested
times (adams.py:130-131), the first step is always the Hairer heuristic of order 2 (adams.py:115-118).
State-sized arithmetic runs in plane kernels; the scalar bookkeeping (g, beta, orders) stays on the host.
"""
import collections
import json
import os

import numpy as np
import torch

from . import _native as N
from .misc import (_convert_to_tensor, _error_norms, _handle_unused_kwargs, _is_iterable, _lincomb, _np_dtype,
                   _optimal_step_size, _scalar_tensor, _scaled_sumsq, _select_initial_step)
from .solvers import AdaptiveStepsizeODESolver

_MIN_ORDER = 1
_MAX_ORDER = 12
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tableaus', 'adams.json')) as _fh:
    gamma_star = json.load(_fh)['gamma_star']                                            # adams.py:15-18

_VCABMState = collections.namedtuple('_VCABMState', 'y_n, prev_f, prev_t, next_t, phi, order')


def g_and_explicit_phi(prev_t, next_t, implicit_phi, k):
    """adams.py:29-63."""
    curr_t = prev_t[0]
    dt = next_t - prev_t[0]
    g = np.zeros(k + 1, dtype=np.float32)                                               # float32 variable (:34)
    explicit_phi = collections.deque(maxlen=k)
    beta = np.float64(1.0)
    g[0] = 1
    c = 1 / np.arange(1, k + 2).astype(np.float64)
    explicit_phi.append(implicit_phi[0])
    dt_ = _np_dtype(implicit_phi[0][0].dtype).type
    for j in range(1, k):
        beta = (next_t - prev_t[j - 1]) / (curr_t - prev_t[j]) * beta
        explicit_phi.append(tuple(_lincomb(None, [1.0], [iphi_], dt_(beta)) for iphi_ in implicit_phi[j]))
        c = c[:-1] - c[1:] if j == 1 else c[:-1] - c[1:] * dt / (next_t - prev_t[j - 1])
        g[j] = np.float32(c[0])
    c = c[:-1] - c[1:] * dt / (next_t - prev_t[k - 1])
    g[k] = np.float32(c[0])
    return g, explicit_phi


def compute_implicit_phi(explicit_phi, f_n, k):
    """adams.py:66-81."""
    k = min(len(explicit_phi) + 1, k)
    implicit_phi = collections.deque(maxlen=k)
    implicit_phi.append(f_n)
    for j in range(1, k):
        implicit_phi.append(tuple(_lincomb(a, [-1.0], [b], 1.0) for a, b in zip(implicit_phi[j - 1], explicit_phi[j - 1])))
    return implicit_phi


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
        from .solvers import _EULER_SHAPE, _FusedEngine, _cached_engine, _fusable
        from .misc import _assert_increasing
        rhs = _fusable(self.func, self.y0) if not self._force_planes else None
        if rhs is not None and getattr(rhs, 'multistep_fused', False) and len(self.y0) == 1:
            _assert_increasing(t)
            y = self.y0[0]
            key = ('adams_vc', rhs.cache_key(y.dtype, y.device), tuple(y.shape), y.dtype, str(y.device), self.max_order,
                   float(self.rtol[0]), float(self.atol[0]), float(self.safety), float(self.ifactor), float(self.dfactor))
            try:
                eng = _cached_engine(key, lambda: _FusedEngine(rhs, y, True, _EULER_SHAPE, rtol=self.rtol[0], atol=self.atol[0],
                                                               safety=float(self.safety), ifactor=float(self.ifactor),
                                                               dfactor=float(self.dfactor), multistep=(3, self.max_order, gamma_star)))
            except N.NativeError:
                eng = None                            # (a batch whose workgroups cannot be co-resident): the per-step loop
            if eng is not None:
                out = eng.integrate(t.to(torch.float64).numpy(), y)
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

    def _ratios(self, errs, tolerance):
        """misc._compute_error_ratio with an explicit tolerance: mean((err/tol)^2) = sum err^2 / (N tol^2)."""
        return self._ratios_many([errs], tolerance)[0]

    def _ratios_many(self, groups, tolerance):
        """The error ratios of several candidate estimates (one tuple of tensors per group) with ONE device-to-host copy."""
        sums = torch.cat([_scaled_sumsq(e, None, e, 0.0, float(tol)) for errs in groups for e, tol in zip(errs, tolerance)]).cpu().numpy()
        out, i = [], 0
        with np.errstate(all='ignore'):
            for errs in groups:
                out.append(tuple(_np_dtype(e.dtype).type(sums[i + j] / float(e.numel())) for j, e in enumerate(errs)))
                i += len(errs)
        return out

    def _adaptive_adams_step(self, vcabm_state, final_t):
        y0, prev_f, prev_t, next_t, prev_phi, order = vcabm_state
        if next_t > final_t:
            next_t = final_t
        dt = next_t - prev_t[0]
        dt_ = _np_dtype(y0[0].dtype).type
        dt_cast = dt_(dt)
        g, phi = g_and_explicit_phi(prev_t, next_t, prev_phi, order)
        g = g.astype(_np_dtype(y0[0].dtype))
        n = max(1, order - 1)
        p_next = tuple(_lincomb(y0_, list(g[:n]), list(phi_[:n]), dt_cast) for y0_, phi_ in zip(y0, tuple(zip(*phi))))
        next_f0 = self._f(next_t, p_next)
        implicit_phi_p = compute_implicit_phi(phi, next_f0, order + 1)
        y_next = tuple(_lincomb(p_, [g[order - 1]], [iphi_], dt_cast) for p_, iphi_ in zip(p_next, implicit_phi_p[order - 1]))
        local_error = tuple(_lincomb(None, [g[order] - g[order - 1]], [iphi_], dt_cast) for iphi_ in implicit_phi_p[order])
        recs = torch.stack([_error_norms(e, a, b) for e, a, b in zip(local_error, y0, y_next)]).cpu().numpy()
        with np.errstate(all='ignore'):                                                  # one scalar per component (F3)
            tolerance = tuple(dt_(self.atol[i]) + dt_(self.rtol[i]) * dt_(max(recs[i, 0], recs[i, 1]))
                              for i in range(len(local_error)))
        # One read-back for every estimate this step may need ("the order / accept record read once per step"):
        # the order-selection estimates of adams.py:176-199 only depend on implicit_phi_p, so they are formed speculatively next to
        # error_k - the same kernels and values as before, two host synchronisations per step (tolerance, ratios) instead of up to five.
        want_lower = not (len(prev_t) <= 4 or order < 3)
        want_higher = want_lower and order < self.max_order
        groups = [local_error]
        if want_lower:
            groups.append(tuple(_lincomb(None, [g[order - 1] - g[order - 2]], [iphi_], dt_cast) for iphi_ in implicit_phi_p[order - 1]))
            groups.append(tuple(_lincomb(None, [g[order - 2] - g[order - 3]], [iphi_], dt_cast) for iphi_ in implicit_phi_p[order - 2]))
        if want_higher:
            groups.append(tuple(_lincomb(None, [dt_(gamma_star[order])], [iphi_], dt_cast) for iphi_ in implicit_phi_p[order]))
        ratios = self._ratios_many(groups, tolerance)
        error_k = ratios[0]
        accept_step = bool(np.all(np.asarray([float(r) for r in error_k]) <= 1))
        self.stats['n_attempts'] += 1
        if not accept_step:
            dt_next = _optimal_step_size(dt, error_k, self.safety, self.ifactor, self.dfactor, order=order)
            return _VCABMState(y0, prev_f, prev_t, prev_t[0] + dt_next, prev_phi, order=order)
        self.stats['n_accepted'] += 1
        next_f0 = self._f(next_t, y_next)
        implicit_phi = compute_implicit_phi(phi, next_f0, order + 2)
        next_order = order
        if len(prev_t) <= 4 or order < 3:
            next_order = min(order + 1, 3, self.max_order)
        else:
            error_km1, error_km2 = ratios[1], ratios[2]
            if min(error_km1 + error_km2) < max(error_k):
                next_order = order - 1
            elif order < self.max_order:
                error_kp1 = ratios[3]
                if max(error_kp1) < max(error_k):
                    next_order = order + 1
        dt_next = dt if next_order > order else _optimal_step_size(dt, error_k, self.safety, self.ifactor, self.dfactor,
                                                                   order=order + 1)
        prev_f.appendleft(next_f0)
        prev_t.appendleft(next_t)
        return _VCABMState(p_next, prev_f, prev_t, next_t + dt_next, implicit_phi, order=next_order)      # p_next (:210)

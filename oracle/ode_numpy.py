"""ORACLE - test infrastructure, not product code.

CPU (numpy) restatement of the reference's explicit Runge-Kutta `odeint` path,
operation for operation, including the reference's quirks (SURVEY.md section 0:
F3 global scalar tolerance, F4 float32-rounded controller exponent, F5 bosh3
tableau typos, F6 tsit5 error-coefficient / dense-output defects, F9 3/8-rule
RK4, F10 zero tableau entries not skipped).  Every function cites the reference
file:line it follows (paths are relative to /root/reference/tfdiffeq/).

Pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which were
captured from the reference's own solver files imported over a numpy stand-in for
TensorFlow (TensorFlow is absent: those fixtures are "reference control-flow over
numpy stand-in", not TF-eager output).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package tfdiffeq_amd never does.
"""
import collections

import numpy as np

# ---------------------------------------------------------------------------
# Tableaus - verbatim values held by the reference modules
# ---------------------------------------------------------------------------
ButcherTableau = collections.namedtuple('ButcherTableau', 'alpha beta c_sol c_error')   # rk_common.py:5

# dopri5.py:11-30
DOPRI5 = ButcherTableau(
    alpha=[1 / 5, 3 / 10, 4 / 5, 8 / 9, 1., 1.],
    beta=[[1 / 5],
          [3 / 40, 9 / 40],
          [44 / 45, -56 / 15, 32 / 9],
          [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
          [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
          [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]],
    c_sol=[35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0],
    c_error=[35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
             -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1. / 60.])
# dopri5.py:33-36
DOPRI5_C_MID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
                187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]

# bosh3.py:10-20 -- "1. / .2" (= 5.0) and "3. / .4" (= 7.5) are the reference's typos (F5), kept verbatim
BOSH3 = ButcherTableau(
    alpha=[1. / .2, 3. / 4., 1.],
    beta=[[1. / 2.], [0., 3. / .4], [2. / 9., 1. / 3., 4. / 9.]],
    c_sol=[2. / 9., 1. / 3., 4. / 9., 0.],
    c_error=[2. / 9. - 7. / 24., 1. / 3. - 1. / 4., 4. / 9. - 1. / 3., -1. / 8.])
BOSH3_C_MID = [0., 0.5, 0., 0.]

_TS_B = [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774]
_TS_X = [0.001780011052226, 0.000816434459657, -0.007880878010262, 0.144711007173263, -0.582357165452555,
         0.458082105929187]
_TS_ALPHA = [0.161, 0.327, 0.9, 0.9800255409045097, 1., 1.]
_TS_BETA = [[0.161],
            [-0.008480655492357, 0.3354806554923570],
            [2.897153057105494, -6.359448489975075, 4.362295432869581],
            [5.32586482843925895, -11.74888356406283, 7.495539342889836, -0.09249506636175525],
            [5.86145544294642038, -12.92096931784711, 8.159367898576159, -0.071584973281401006,
             -0.02826905039406838],
            list(_TS_B)]
# tsit5.py:10-30 -- c_error = b - x where x already ARE the error weights -> sum(c_error) = 0.97 (F6a)
TSIT5_REF = ButcherTableau(alpha=_TS_ALPHA, beta=_TS_BETA, c_sol=_TS_B + [0.],
                           c_error=[b - x for b, x in zip(_TS_B, _TS_X)] + [-1. / 66.])
# ORACLE EXTENSION (no reference counterpart): Tsitouras' published error weights, sum = 0
TSIT5_FIXED = ButcherTableau(alpha=_TS_ALPHA, beta=_TS_BETA, c_sol=_TS_B + [0.],
                             c_error=list(_TS_X) + [-1. / 66.])


# adaptive_huen.py:11-25
ADAPTIVE_HEUN = ButcherTableau(alpha=[1.], beta=[[1.]], c_sol=[0.5, 0.5], c_error=[0.5, -0.5])
ADAPTIVE_HEUN_C_MID = [0.5, 0.]


def load_dopri8():
    """dopri8.py:12-77 (Prince-Dormand RK8(7)13M): the float64 values captured from the reference module
    (tests/golden/fn_tableaus_next.npz) - (tableau, c_mid)."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                             'fn_tableaus_next.npz'))
    alpha = d['dopri8_alpha'].tolist()
    beta = [d['dopri8_beta'][i, :i + 1].tolist() for i in range(len(alpha))]
    return (ButcherTableau(alpha=alpha, beta=beta, c_sol=d['dopri8_c_sol'].tolist(), c_error=d['dopri8_c_error'].tolist()),
            d['dopri8_c_mid'].tolist())


def tableau_arrays(tb):
    """(alpha[S], beta[S,S] zero padded, c_sol[S+1], c_error[S+1]) as float64 arrays."""
    S = len(tb.alpha)
    beta = np.zeros((S, S))
    for i, row in enumerate(tb.beta):
        beta[i, :len(row)] = row
    return (np.asarray(tb.alpha, dtype=np.float64), beta,
            np.asarray(tb.c_sol, dtype=np.float64), np.asarray(tb.c_error, dtype=np.float64))


# ---------------------------------------------------------------------------
# small helpers
# ---------------------------------------------------------------------------
def _scalar(x, dtype):
    return np.dtype(dtype).type(x)


def scaled_dot_product(scale, xs, ys):
    """misc.py:118-121: add_n([(scale * x) * y ...]) - zero weights are NOT skipped (F10)."""
    acc = None
    with np.errstate(all='ignore'):
        for x, y in zip(xs, ys):
            term = (scale * x) * y
            acc = term if acc is None else acc + term
    return acc


def dot_product(xs, ys):
    """misc.py:124-126: python sum(), starting from int 0."""
    acc = 0
    with np.errstate(all='ignore'):
        for x, y in zip(xs, ys):
            acc = acc + x * y
    return acc


def is_fsal_shaped(tb):
    """rk_common.py:54: the tableau property that lets y1 = y_last."""
    return tb.c_sol[-1] == 0 and list(tb.c_sol[:-1]) == list(tb.beta[-1])


# ---------------------------------------------------------------------------
# one RK attempt and its pieces
# ---------------------------------------------------------------------------
def runge_kutta_step(func, y0, f0, t0, dt, tableau):
    """rk_common.py:22-61.  y0, f0: tuples of arrays.  Returns (y1, f1, y1_error, k)."""
    dtype = y0[0].dtype
    t0 = _scalar(t0, dtype)                      # :45
    dt = _scalar(dt, dtype)                      # :46
    k = tuple([f0_] for f0_ in f0)               # :48
    yi = None
    for alpha_i, beta_i in zip(tableau.alpha, tableau.beta):     # :49
        ti = t0 + alpha_i * dt                   # :50
        yi = tuple(y0_ + scaled_dot_product(dt, beta_i, k_) for y0_, k_ in zip(y0, k))   # :51
        for k_, f_ in zip(k, func(ti, yi)):      # :52-53
            k_.append(f_)
    if not is_fsal_shaped(tableau):              # :54-56
        yi = tuple(y0_ + scaled_dot_product(dt, tableau.c_sol, k_) for y0_, k_ in zip(y0, k))
    y1 = yi
    f1 = tuple(k_[-1] for k_ in k)               # :59
    y1_error = tuple(scaled_dot_product(dt, tableau.c_error, k_) for k_ in k)   # :60
    return y1, f1, y1_error, k


def rk4_alt_step(func, t, dt, y):
    """rk_common.py:73-81, the 3/8 rule used by method='rk4' (F9).  Returns dy."""
    k1 = func(t, y)
    k2 = func(t + dt / 3, tuple(y_ + dt * k1_ / 3 for y_, k1_ in zip(y, k1)))
    k3 = func(t + dt * 2 / 3, tuple(y_ + dt * (k1_ / -3 + k2_) for y_, k1_, k2_ in zip(y, k1, k2)))
    k4 = func(t + dt, tuple(y_ + dt * (k1_ - k2_ + k3_) for y_, k1_, k2_, k3_ in zip(y, k1, k2, k3)))
    return tuple((k1_ + 3 * k2_ + 3 * k3_ + k4_) * (dt / 8) for k1_, k2_, k3_, k4_ in zip(k1, k2, k3, k4))


def compute_error_ratio(error_estimate, rtol, atol, y0, y1):
    """misc.py:250-264.  tol is ONE scalar per tuple component (F3)."""
    out = []
    with np.errstate(all='ignore'):
        for err, rt, at, a, b in zip(error_estimate, rtol, atol, y0, y1):
            dtype = err.dtype
            tol = _scalar(at, dtype) + _scalar(rt, dtype) * np.max(np.stack([np.abs(a), np.abs(b)]))   # :256-259
            r = err / tol
            out.append(np.mean(r * r))           # :262-263
    return tuple(out)


def optimal_step_size(last_step, mean_error_ratio, safety=0.9, ifactor=10.0, dfactor=0.2, order=5):
    """misc.py:267-287 (dopri5 / bosh3 controller).  last_step is float64."""
    r = mean_error_ratio[0]
    for x in mean_error_ratio[1:]:               # python max(): first maximal element, NaN-insensitive like the ref
        if x > r:
            r = x
    if r == 0:
        return np.float64(last_step * ifactor)   # :271-272
    if r < 1:
        dfactor = 1.0                            # :274-275
    with np.errstate(all='ignore'):
        er = np.float64(np.sqrt(r))              # :277-278  sqrt in the ratio's dtype, then cast
        exponent = np.float64(np.float32(1. / order))      # :281-282  float32 detour (F4)
        factor = np.max([np.float64(1. / ifactor),
                         np.min([er ** exponent / safety, np.float64(1. / dfactor)])])   # :285-286
        return np.float64(last_step / factor)


def optimal_step_size_tsit5(last_step, mean_error_ratio, safety=0.9, ifactor=10.0, dfactor=0.2, order=5):
    """tsit5.py:53-62: no sqrt, true float64 exponent."""
    if mean_error_ratio == 0:
        return np.float64(last_step * ifactor)
    if mean_error_ratio < 1:
        dfactor = 1.0
    with np.errstate(all='ignore'):
        er = np.float64(mean_error_ratio)
        exponent = np.float64(1. / order)
        factor = np.maximum(np.float64(1. / ifactor), np.minimum((er ** exponent) / safety, np.float64(1. / dfactor)))
        return np.float64(last_step / factor)


def _rms_norm(x):
    """misc.py:170-175 for one tensor: ||x||_2 / sqrt(numel), in x's dtype."""
    dtype = x.dtype
    with np.errstate(all='ignore'):
        return np.sqrt(np.sum(x * x)).astype(dtype) / (_scalar(x.size, dtype) ** 0.5)


def _pymax(seq):
    best = seq[0]
    for x in seq[1:]:
        if x > best:
            best = x
    return best


def select_initial_step(fun, t0, y0, order, rtol, atol, f0=None):
    """misc.py:183-247 (Hairer II.4).  rtol/atol scalars; returns (h, n_fevals)."""
    dtype = y0[0].dtype
    nfe = 0
    t0 = _scalar(t0, dtype)                      # :213
    if f0 is None:
        f0 = fun(t0, y0)                         # :214-215
        nfe += 1
    with np.errstate(all='ignore'):
        scale = tuple(atol + np.abs(y0_) * rtol for y0_ in y0)            # :225
        scale = tuple(s.astype(dtype) for s in scale)
        d0 = tuple(_rms_norm(y0_ / s) for y0_, s in zip(y0, scale))       # :227
        d1 = tuple(_rms_norm(f0_ / s) for f0_, s in zip(f0, scale))       # :228
        if _pymax(d0) < 1e-5 or _pymax(d1) < 1e-5:                        # :230-231
            h0 = _scalar(1e-6, dtype)
        else:
            h0 = 0.01 * _pymax([a / b for a, b in zip(d0, d1)])           # :233
        y1 = tuple(y0_ + h0 * f0_ for y0_, f0_ in zip(y0, f0))            # :235
        f1 = fun(t0 + h0, y1)                                             # :236
        nfe += 1
        d2 = tuple(_rms_norm((f1_ - f0_) / s) / h0 for f1_, f0_, s in zip(f1, f0, scale))   # :237
        if _pymax(d1) <= 1e-15 and _pymax(d2) <= 1e-15:                   # :239-241
            h1 = np.max([_scalar(1e-6, dtype), h0 * 1e-3])
        else:
            h1 = (0.01 / _pymax(list(d1) + list(d2))) ** (1. / float(order + 1))   # :243  (tuple concat)
        return np.min([100 * h0, h1]).astype(dtype), nfe                  # :245


def interp_fit(y0, y1, y_mid, f0, f1, dt):
    """interp.py:6-36: quartic through y0, y_mid, y1 with end slopes.  Returns [a, b, c, d, e]."""
    comps = list(zip(f0, f1, y0, y1, y_mid))
    a = tuple(dot_product([-2 * dt, 2 * dt, -8, -8, 16], c) for c in comps)
    b = tuple(dot_product([5 * dt, -3 * dt, 18, 14, -32], c) for c in comps)
    c_ = tuple(dot_product([-4 * dt, dt, -11, -5, 16], c) for c in comps)
    d = tuple(dt * f0_ for f0_ in f0)
    return [a, b, c_, d, y0]


def interp_fit_mid(y0, y1, k, dt, c_mid):
    """dopri5.py:39-45 / bosh3.py:25-31: y_mid from the stage derivatives, then interp_fit."""
    dtype = y0[0].dtype
    dt = _scalar(dt, dtype)
    y_mid = tuple(y0_ + scaled_dot_product(dt, c_mid, k_) for y0_, k_ in zip(y0, k))
    f0 = tuple(k_[0] for k_ in k)
    f1 = tuple(k_[-1] for k_ in k)
    return interp_fit(y0, y1, y_mid, f0, f1, dt)


def interp_evaluate(coefficients, t0, t1, t):
    """interp.py:39-67: x = (t-t0)/(t1-t0) computed in the STATE dtype."""
    dtype = coefficients[0][0].dtype
    t0, t1, t = _scalar(t0, dtype), _scalar(t1, dtype), _scalar(t, dtype)
    assert (t0 <= t) & (t <= t1), 'invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}'.format(t0, t, t1)
    with np.errstate(all='ignore'):
        x = dtype.type((t - t0) / (t1 - t0))
        xs = [dtype.type(1), x]
        for _ in range(2, len(coefficients)):
            xs.append(xs[-1] * x)
        return tuple(dot_product(c, list(reversed(xs))) for c in zip(*coefficients))


def tsit5_interp_weights(t0, dt, eval_t):
    """tsit5.py:33-42: the seven dense-output weights b_i(theta)."""
    t = (eval_t - t0) / dt
    b1 = -1.0530884977290216 * t * (t - 1.3299890189751412) * (t ** 2 - 1.4364028541716351 * t + 0.7139816917074209)
    b2 = 0.1017 * t ** 2 * (t ** 2 - 2.1966568338249754 * t + 1.2949852507374631)
    b3 = 2.490627285651252793 * t ** 2 * (t ** 2 - 2.38535645472061657 * t + 1.57803468208092486)
    b4 = -16.54810288924490272 * (t - 1.21712927295533244) * (t - 0.61620406037800089) * t ** 2
    b5 = 47.37952196281928122 * (t - 1.203071208372362603) * (t - 0.658047292653547382) * t ** 2
    b6 = -34.87065786149660974 * (t - 1.2) * (t - 0.666666666666666667) * t ** 2
    b7 = 2.5 * (t - 1) * (t - 0.6) * t ** 2
    return [b1, b2, b3, b4, b5, b6, b7]


def interp_eval_tsit5(t0, t1, k, eval_t, base=None):
    """tsit5.py:45-50.  The reference uses base = k[0] (= f0, defect F6b); base=y0 is the oracle extension."""
    dt = t1 - t0
    w = tsit5_interp_weights(t0, dt, eval_t)
    if base is None:
        base = tuple(k_[0] for k_ in k)
    return tuple(b_ + scaled_dot_product(dt, w, k_) for b_, k_ in zip(base, k))


# ---------------------------------------------------------------------------
# drivers
# ---------------------------------------------------------------------------
class Stats(object):
    def __init__(self):
        self.nfe = 0
        self.trace = []          # rows: (t0, dt, accepted, dt_next)

    @property
    def n_attempts(self):
        return len(self.trace)

    @property
    def n_accepted(self):
        return int(sum(r[2] for r in self.trace))


def _is_finite(x):
    return bool(np.all(np.isfinite(x)))


class AdaptiveRK(object):
    """dopri5.py:48-121, bosh3.py:34-99, tsit5.py:69-151 folded into one parameterised driver."""

    def __init__(self, func, y0, rtol, atol, method, first_step=None, safety=0.9, ifactor=10.0, dfactor=0.2,
                 max_num_steps=2 ** 31 - 1, tableau=None, tsit5_fixed=False, max_attempts=None):
        self.func, self.y0, self.method = func, y0, method
        n = len(y0)
        if method == 'tsit5':
            self.rtol, self.atol = rtol, atol                    # tsit5.py:81-82 scalars
            self.tableau = TSIT5_FIXED if tsit5_fixed else TSIT5_REF
            self.init_order, self.order = 4, 5
        else:
            self.rtol = list(rtol) if np.iterable(rtol) else [rtol] * n      # dopri5.py:60-61
            self.atol = list(atol) if np.iterable(atol) else [atol] * n
            if method == 'dopri5':
                self.tableau = tableau if tableau is not None else DOPRI5    # dopri5.py:67
                self.c_mid, self.init_order, self.order = DOPRI5_C_MID, 4, 5
            elif method == 'bosh3':
                self.tableau, self.c_mid, self.init_order, self.order = BOSH3, BOSH3_C_MID, 2, 3
            elif method == 'dopri8':                         # dopri8.py:119-121, 163-165
                (self.tableau, self.c_mid), self.init_order, self.order = load_dopri8(), 7, 8
            elif method == 'adaptive_heun':                  # adaptive_huen.py:70-71, 112 (order=5 is the reference's)
                self.tableau, self.c_mid, self.init_order, self.order = ADAPTIVE_HEUN, ADAPTIVE_HEUN_C_MID, 1, 5
            else:
                raise KeyError(method)
        self.tsit5_fixed = tsit5_fixed
        self.first_step = first_step
        # dopri5.py:63-65 -> misc.py:137-144 (_convert_to_tensor): python float -> float32 tensor -> cast to
        # float64, so safety = 0.8999999761581421 and dfactor = 0.20000000298023224 (same mechanism as F4)
        self.safety, self.ifactor, self.dfactor = (np.float64(np.float32(safety)), np.float64(np.float32(ifactor)),
                                                   np.float64(np.float32(dfactor)))
        self.max_num_steps = max_num_steps
        self.max_attempts = max_attempts
        self.stats = Stats()

    def _f(self, t, y):
        self.stats.nfe += 1
        return self.func(t, y)

    def before_integrate(self, t):
        dtype = self.y0[0].dtype
        if self.method == 'tsit5':                       # tsit5.py:91-103
            if self.first_step is None:
                h, _ = select_initial_step(self._f, t[0], self.y0, 4, self.rtol, self.atol)
                first = np.float64(h)
            else:
                first = np.float64(np.float32(self.first_step))   # _convert_to_tensor float32 detour
            f0 = self._f(t[0], self.y0)                  # t[0] stays float64 here (:99)
            self.k = tuple([y_] * 7 for y_ in self.y0)
        else:                                            # dopri5.py:70-79 / bosh3.py:53-60
            f0 = self._f(_scalar(t[0], dtype), self.y0)
            if self.first_step is None:
                h, _ = select_initial_step(self._f, t[0], self.y0, self.init_order, self.rtol[0], self.atol[0], f0=f0)
                first = np.float64(h)
            else:
                first = np.float64(np.float32(self.first_step))   # dopri5.py:77 -> misc.py:137-144
            self.coeff = [self.y0] * 5
        self.y1, self.f1 = self.y0, f0
        self.t0 = self.t1 = np.float64(t[0])
        self.dt = first

    def _attempt(self):
        y0, f0, t0, dt = self.y1, self.f1, self.t1, np.float64(self.dt)
        assert t0 + dt > t0, 'underflow in dt {}'.format(dt)                      # dopri5.py:98
        for y0_ in y0:
            assert _is_finite(np.abs(y0_)), 'non-finite values in state `y`: {}'.format(y0_)   # :99-100
        y1, f1, err, k = runge_kutta_step(self._f, y0, f0, t0, dt, self.tableau)  # :101
        with np.errstate(all='ignore'):
            if self.method == 'tsit5':                   # tsit5.py:126-138: pooled mean, scalar rtol/atol
                dtype = y0[0].dtype
                num, den = 0, 0
                for a, b, e in zip(y0, y1, err):
                    tol = _scalar(self.atol, dtype) + _scalar(self.rtol, dtype) * np.max(np.stack([np.abs(a), np.abs(b)]))
                    r = e / tol
                    num = num + np.sum(r * r)
                    den = den + _scalar(r.size, dtype)
                ratio = num / den
                accept = bool(ratio <= 1.)
                dt_next = optimal_step_size_tsit5(dt, ratio, self.safety, self.ifactor, self.dfactor, self.order)
            else:
                ratios = compute_error_ratio(err, self.rtol, self.atol, y0, y1)   # dopri5.py:106-107
                accept = bool(np.all(np.stack(ratios) <= 1))                      # :108
                dt_next = optimal_step_size(dt, ratios, self.safety, self.ifactor, self.dfactor, self.order)
        if accept:                                       # dopri5.py:113-116
            self.y1, self.f1, self.t0, self.t1 = y1, f1, t0, t0 + dt
            if self.method == 'tsit5':
                self.k, self.step_y0 = k, y0
            else:
                self.coeff = interp_fit_mid(y0, y1, k, dt, self.c_mid)
        else:
            self.t0 = t0                                 # rejected: state.t0 == state.t1 (:120)
        self.stats.trace.append((float(t0), float(dt), 1.0 if accept else 0.0, float(dt_next)))
        self.dt = dt_next

    class Truncated(Exception):
        pass

    def advance(self, next_t):
        n_steps = 0
        while next_t > self.t1:                          # dopri5.py:84
            assert n_steps < self.max_num_steps, 'max_num_steps exceeded ({}>={})'.format(n_steps, self.max_num_steps)
            if self.max_attempts is not None and self.stats.n_attempts >= self.max_attempts:
                raise AdaptiveRK.Truncated()
            self._attempt()
            n_steps += 1
        if self.method == 'tsit5':
            base = self.step_y0 if (self.tsit5_fixed and hasattr(self, 'step_y0')) else None
            if self.tsit5_fixed and not hasattr(self, 'step_y0'):
                base = self.y0
            return interp_eval_tsit5(self.t0, self.t1, self.k, next_t, base=base)
        return interp_evaluate(self.coeff, self.t0, self.t1, next_t)              # :89

    def integrate(self, t):
        assert bool(np.all(t[1:] > t[:-1])), 't must be strictly increasing or decrasing'   # misc.py:158-159
        t = t.astype(np.float64)                         # solvers.py:30
        solution = [self.y0]
        self.before_integrate(t)
        for i in range(1, t.shape[0]):
            solution.append(self.advance(t[i]))
        return tuple(np.stack(c) for c in zip(*solution))


class FixedGrid(object):
    """solvers.py:39-115 with the Euler / RK4 step functions of fixed_grid.py:4-46."""

    def __init__(self, func, y0, method, eps=0.0, grid_constructor=None, step_size=None):
        self.func, self.y0, self.method, self.eps = func, y0, method, eps
        if grid_constructor is not None:
            # solvers.py:49-56: the branch that would accept a grid_constructor is the one that raises - in the reference a
            # custom grid constructor can never be used
            raise ValueError("step_size and grid_constructor are exclusive arguments.")
        if step_size is not None:
            # solvers.py:58-71 cannot run in the reference (F7: .item() / item assignment on TF tensors); this is its evident
            # intent - a uniform grid of `step_size` from t[0], clipped to t[-1] - and an oracle EXTENSION, not pinned by it
            def grid_constructor(func_, y0_, t):
                n = int(np.ceil((t[-1] - t[0]) / step_size + 1))
                g = (np.arange(0, n).astype(t.dtype) * t.dtype.type(step_size) + t[0]).astype(t.dtype)
                if g[-1] > t[-1]:
                    g[-1] = t[-1]
                return g
        self.grid_constructor = grid_constructor
        self.stats = Stats()

    def _f(self, t, y):
        self.stats.nfe += 1
        return self.func(t, y)

    def step_func(self, t, dt, y):
        if self.method == 'euler':                       # fixed_grid.py:6-7
            return tuple(dt * f_ for f_ in self._f(t + self.eps, y))
        if self.method == 'rk4':                         # fixed_grid.py:41-42
            return rk4_alt_step(self._f, t + self.eps, dt, y)
        if self.method == 'midpoint':                    # fixed_grid.py:16-18
            y_mid = tuple(y_ + f_ * dt / 2 for y_, f_ in zip(y, self._f(t + self.eps, y)))
            return tuple(dt * f_ for f_ in self._f(t + dt / 2, y_mid))
        raise KeyError(self.method)

    def integrate(self, t):
        assert bool(np.all(t[1:] > t[:-1])), 't must be strictly increasing or decrasing'
        t = t.astype(self.y0[0].dtype)                   # solvers.py:84  time in the STATE dtype
        grid = t if self.grid_constructor is None else np.asarray(self.grid_constructor(self.func, self.y0, t)).astype(t.dtype)   # :53-54, :86
        assert grid[0] == t[0] and grid[-1] == t[-1]     # :87
        solution = [self.y0]
        j, y0 = 1, self.y0
        for t0, t1 in zip(grid[:-1], grid[1:]):
            dy = self.step_func(t0, t1 - t0, y0)
            y1 = tuple(a + b for a, b in zip(y0, dy))
            while j < t.shape[0] and t1 >= t[j]:         # :97-100
                solution.append(self._linear_interp(t0, t1, y0, y1, t[j]))
                j += 1
            y0 = y1
        return tuple(np.stack(c) for c in zip(*solution))

    @staticmethod
    def _linear_interp(t0, t1, y0, y1, t):               # solvers.py:106-115
        if t == t0:
            return y0
        if t == t1:
            return y1
        slope = tuple((b - a) / (t1 - t0) for a, b in zip(y0, y1))
        return tuple(a + s * (t - t0) for a, s in zip(y0, slope))


ADAPTIVE = ('dopri5', 'bosh3', 'tsit5', 'dopri8', 'adaptive_heun')
FIXED = ('euler', 'rk4', 'midpoint')


def odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None, return_stats=False, max_attempts=None):
    """odeint.py:28-81 + misc.py:290-329 (_check_inputs).  y0: ndarray or tuple of ndarrays; t: 1-D ndarray."""
    tensor_input = False
    if isinstance(y0, np.ndarray) or np.isscalar(y0):
        tensor_input = True
        y0 = (np.asarray(y0),)
        base = func
        func = lambda tt, yy: (base(tt, yy[0]),)         # noqa: E731   misc.py:301-303
    assert isinstance(y0, tuple), 'y0 must be either a tf.Tensor or a tuple'
    t = np.asarray(t)
    if bool(np.all(t[1:] < t[:-1])):                     # misc.py:318-321 (vacuously true for len(t) == 1)
        t = -t
        fwd = func
        func = lambda tt, yy: tuple(-f_ for f_ in fwd(-tt, yy))   # noqa: E731
    for y0_ in y0:
        if y0_.dtype.kind not in 'fiuc':
            raise TypeError('`y0` must be a floating point Tensor but is a {}'.format(y0_.dtype))
    if options is None:
        options = {}
    elif method is None:
        raise ValueError('cannot supply `options` without specifying `method`')          # odeint.py:72-73
    if method is None:
        method = 'dopri5'
    if method in ADAPTIVE:
        solver = AdaptiveRK(func, y0, rtol, atol, method, max_attempts=max_attempts, **options)
    elif method in FIXED:
        solver = FixedGrid(func, y0, method, **options)
    else:
        raise KeyError(method)                           # odeint.py:77 dict lookup
    try:
        sol = solver.integrate(t)
    except AdaptiveRK.Truncated:
        sol = None
        solver.stats.state_y = solver.y1
        solver.stats.state_t = solver.t1
    if tensor_input and sol is not None:
        sol = sol[0]
    if return_stats:
        return sol, solver.stats
    return sol

"""ORACLE (test / benchmark infrastructure - never imported by the product package).

The CPU baseline SURVEY.md 8(d) / BASELINE.md section 3 prescribe: an op-for-op EAGER restatement of the reference's
adaptive Runge-Kutta path on torch-CPU tensors - one tensor op per reference op, the same host synchronisation points
(`bool(tensor)` wherever the reference's Python branches on a TF eager tensor) - because the reference itself
(TensorFlow) cannot run on the GPU box.  It is what a user of the reference would get from a straight TF-eager -> torch
port, NOT an optimised CPU solver: every `scale * x * y` is a scalar op plus a full-plane op, every add_n another pass,
the five interpolation-coefficient planes are written on every accepted step, ~10 syncs per attempt.

Covers what the benchmark configurations need: Dopri5 / Bosh3, a single tensor state, increasing t.
Pinned by tests/test_oracle_golden.py::test_torch_cpu_restatement_* against the fixtures captured from the reference.
Citations are into /root/reference/tfdiffeq/.
"""
import collections

import numpy as np
import torch

Tableau = collections.namedtuple('Tableau', 'alpha beta c_sol c_error')

# dopri5.py:11-36
DOPRI5 = Tableau(
    alpha=[1 / 5, 3 / 10, 4 / 5, 8 / 9, 1., 1.],
    beta=[[1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9], [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
          [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656], [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]],
    c_sol=[35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0],
    c_error=[35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 - -12231 / 42400,
             11 / 84 - 649 / 6300, -1. / 60.])
DPS_C_MID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
             187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


def _scalar(x, like):
    """tf.cast / tf.convert_to_tensor of a scalar to the state dtype: a 0-d tensor (a scalar op in TF eager).
    DELIBERATELY detached: the step size and the stage times are controller outputs, and the gradient checker differentiates the
    DISCRETE flow at the step sequence the controller chose - d(dt)/d(y0) is not part of it (the adjoint method does not see it either:
    it integrates the continuous sensitivity).  At the tolerances the gradient tests run this checker (rtol <= 1e-9) the two agree to the
    solve's own accuracy; the detach is what makes that statement exact instead of approximately true."""
    if isinstance(x, torch.Tensor):
        x = x.detach()
    return torch.tensor(float(x), dtype=like.dtype)


def _scaled_dot_product(scale, xs, ys):
    """misc.py:118-121: add_n([(scale * x) * y ...]); `scale * x` is a scalar op, `* y` one full-plane op per term, the
    add_n one more pass over all of them (torch has no n-ary add: len - 1 binary adds)."""
    terms = [(scale * x) * y for x, y in zip(xs, ys)]
    acc = terms[0]
    for term in terms[1:]:
        acc = acc + term
    return acc


def _dot_product(xs, ys):
    """misc.py:124-126: python sum() starting from int 0."""
    acc = 0
    for x, y in zip(xs, ys):
        acc = acc + x * y
    return acc


def _runge_kutta_step(func, y0, f0, t0, dt, tableau):
    """rk_common.py:22-61 for a one-component state."""
    t0 = _scalar(t0, y0)                                   # :45
    dt = _scalar(dt, y0)                                   # :46
    k = [f0]
    yi = None
    for alpha_i, beta_i in zip(tableau.alpha, tableau.beta):
        ti = t0 + alpha_i * dt                             # :50
        yi = y0 + _scaled_dot_product(dt, beta_i, k)       # :51
        k.append(func(ti, yi))                             # :52-53
    y1 = yi                                                # FSAL shaped (:54-58)
    f1 = k[-1]
    y1_error = _scaled_dot_product(dt, tableau.c_error, k)  # :60
    return y1, f1, y1_error, k


def _compute_error_ratio(err, rtol, atol, y0, y1):
    """misc.py:250-264: ONE scalar tolerance per component (F3)."""
    tol = atol + rtol * torch.max(torch.stack([torch.abs(y0), torch.abs(y1)]))      # :256-259 (stack + reduce_max)
    r = err / tol                                          # :261
    return torch.mean(r * r)                               # :262-263


def _optimal_step_size(last_step, ratio, safety, ifactor, dfactor, order):
    """misc.py:267-287: scalar tensor ops, two Python branches on tensors (two syncs)."""
    if bool(ratio == 0):                                   # :271
        return last_step * ifactor
    if bool(ratio < 1):                                    # :274
        dfactor = torch.tensor(1.0, dtype=torch.float64)
    er = torch.sqrt(ratio).to(torch.float64)               # :277-278
    exponent = torch.tensor(1. / order, dtype=torch.float32).to(torch.float64)      # :281-282 (F4)
    factor = torch.maximum(1. / ifactor, torch.minimum(er ** exponent / safety, 1. / dfactor))   # :285-286
    return last_step / factor


def _select_initial_step(func, t0, y0, order, rtol, atol, f0):
    """misc.py:183-247 (Hairer II.4)."""
    scale = atol + torch.abs(y0) * rtol                    # :225
    n = float(y0.numel())
    d0 = torch.norm(y0 / scale) / n ** 0.5                 # :227 (misc._norm, :170-175)
    d1 = torch.norm(f0 / scale) / n ** 0.5                 # :228
    if bool(d0 < 1e-5) or bool(d1 < 1e-5):                 # :230
        h0 = torch.tensor(1e-6, dtype=y0.dtype)
    else:
        h0 = 0.01 * d0 / d1                                # :233
    y1 = y0 + h0 * f0                                      # :235
    f1 = func(t0 + h0, y1)                                 # :236
    d2 = torch.norm((f1 - f0) / scale) / n ** 0.5 / h0     # :237
    if bool(d1 <= 1e-15) and bool(d2 <= 1e-15):            # :239
        h1 = torch.maximum(torch.tensor(1e-6, dtype=y0.dtype), h0 * 1e-3)
    else:
        h1 = (0.01 / torch.maximum(d1, d2)) ** (1. / float(order + 1))              # :242-245
    return torch.minimum(100 * h0, h1).to(torch.float64)   # :247


def _interp_fit_dopri5(y0, y1, k, dt):
    """dopri5.py:39-45 + interp.py:6-36: y_mid, then FIVE coefficient planes (written on every accepted step)."""
    dt_s = _scalar(dt, y0)
    y_mid = y0 + _scaled_dot_product(dt_s, DPS_C_MID, k)   # dopri5.py:42
    f0, f1 = k[0], k[-1]
    a = _dot_product([-2 * dt_s, 2 * dt_s, -8, -8, 16], [f0, f1, y0, y1, y_mid])     # interp.py:28
    b = _dot_product([5 * dt_s, -3 * dt_s, 18, 14, -32], [f0, f1, y0, y1, y_mid])    # :29
    c = _dot_product([-4 * dt_s, dt_s, -11, -5, 16], [f0, f1, y0, y1, y_mid])        # :30
    d = dt_s * f0                                          # :31
    e = y0                                                 # :32
    return [a, b, c, d, e]


def _interp_evaluate(coefficients, t0, t1, t):
    """interp.py:39-67."""
    dtype = coefficients[0].dtype
    t0, t1, t = (torch.tensor(float(v.detach() if isinstance(v, torch.Tensor) else v), dtype=dtype) for v in (t0, t1, t))   # :55-57 (times: detached, see _scalar)
    assert bool((t0 <= t) & (t <= t1)), 'invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}'.format(t0, t, t1)
    x = ((t - t0) / (t1 - t0)).to(dtype)                   # :60
    xs = [torch.tensor(1.0, dtype=dtype), x]
    for _ in range(2, len(coefficients)):
        xs.append(xs[-1] * x)                              # :62-64
    return _dot_product(coefficients, list(reversed(xs)))  # :66


class Stats(object):
    def __init__(self):
        self.nfe = 0
        self.n_attempts = 0
        self.n_accepted = 0


def odeint_dopri5(func, y0, t, rtol=1e-7, atol=1e-9):
    """odeint(func, y0, t, method='dopri5') for one CPU tensor state and increasing t (odeint.py:28-81, solvers.py:27-35,
    dopri5.py:48-121).  Returns (solution [T, *y0.shape], Stats)."""
    assert y0.device.type == 'cpu'
    st = Stats()

    def f(tt, yy):
        st.nfe += 1
        return func(tt, yy)
    t = torch.as_tensor(t, dtype=torch.float64)            # solvers.py:30
    assert bool(torch.all(t[1:] > t[:-1])), 't must be strictly increasing or decrasing'       # misc.py:158-159
    # dopri5.py:63-65 through misc.py:137-144: python float -> float32 tensor -> float64
    safety = torch.tensor(0.9, dtype=torch.float32).to(torch.float64)
    ifactor = torch.tensor(10.0, dtype=torch.float32).to(torch.float64)
    dfactor = torch.tensor(0.2, dtype=torch.float32).to(torch.float64)
    # before_integrate (dopri5.py:70-79)
    f0 = f(_scalar(t[0], y0), y0)
    dt = _select_initial_step(f, t[0].to(y0.dtype), y0, 4, rtol, atol, f0)
    y1, f1, t0s, t1s = y0, f0, t[0], t[0]
    coeff = [y0] * 5
    solution = [y0]
    for i in range(1, t.shape[0]):
        next_t = t[i]
        while bool(next_t > t1s):                          # dopri5.py:84 (sync)
            ya, fa, ta = y1, f1, t1s
            assert bool(ta + dt > ta), 'underflow in dt {}'.format(float(dt))                   # :98 (sync)
            assert bool(torch.all(torch.isfinite(torch.abs(ya)))), 'non-finite values in state `y`'   # :99-100 (reduce + sync)
            yb, fb, err, k = _runge_kutta_step(f, ya, fa, ta, dt, DOPRI5)                      # :101
            ratio = _compute_error_ratio(err, rtol, atol, ya, yb)                               # :106-107
            accept = bool(torch.all(ratio <= 1))                                                # :108 (sync)
            st.n_attempts += 1
            if accept:                                     # :113-118
                st.n_accepted += 1
                coeff = _interp_fit_dopri5(ya, yb, k, dt)
                y1, f1, t0s, t1s = yb, fb, ta, ta + dt
            else:
                t0s = ta
            dt = _optimal_step_size(dt, ratio, safety, ifactor, dfactor, 5)                     # :119
        solution.append(_interp_evaluate(coeff, t0s, t1s, next_t))                              # :89
    return torch.stack(solution), st

"""Mirror of tfdiffeq/interp.py: quartic dense output (dopri5 / bosh3).

`_interp_fit` / `_interp_evaluate` keep the reference's signatures.  The solvers themselves do not
materialise the five coefficient planes per accepted step (pure waste when T = 2):
they keep (y0, y1, k) of the accepted step and evaluate through the fused `mi_ode_interp_eval`
kernel (`_interp_eval_step`), which performs fit + evaluation in registers.
"""
import ctypes as C

import torch

from . import _native as N
from .misc import _contig, _lincomb, _np_dtype, _ptr


def _interp_fit(y0, y1, y_mid, f0, f1, dt):
    """interp.py:6-36: [a, b, c, d, e] with p(x) = a x^4 + b x^3 + c x^2 + d x + e, x in [0, 1]."""
    dt_ = _np_dtype(y0[0].dtype).type
    dt = dt_(dt)
    comps = list(zip(f0, f1, y0, y1, y_mid))
    a = tuple(_lincomb(None, [-2 * dt, 2 * dt, -8, -8, 16], c, 1.0) for c in comps)
    b = tuple(_lincomb(None, [5 * dt, -3 * dt, 18, 14, -32], c, 1.0) for c in comps)
    c_ = tuple(_lincomb(None, [-4 * dt, dt, -11, -5, 16], c, 1.0) for c in comps)
    d = tuple(_lincomb(None, [1.0], [f0_], dt) for f0_ in f0)
    e = y0
    return [a, b, c_, d, e]


def _interp_evaluate(coefficients, t0, t1, t):
    """interp.py:39-67: x = (t - t0)/(t1 - t0) in the STATE dtype; asserts t0 <= t <= t1."""
    dt_ = _np_dtype(coefficients[0][0].dtype).type
    t0, t1, t = dt_(t0), dt_(t1), dt_(t)
    assert (t0 <= t) & (t <= t1), 'invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}'.format(t0, t, t1)
    x = dt_((t - t0) / (t1 - t0))
    xs = [dt_(1), x]
    for _ in range(2, len(coefficients)):
        xs.append(xs[-1] * x)
    return tuple(_lincomb(None, list(reversed(xs)), list(c), 1.0) for c in zip(*coefficients))


def _interp_eval_step(interp_kind, y0, y1, k, c_mid, dt, t0, t1, t):
    """Fused fit + evaluate for one tuple state from the accepted step's (y0, y1, k): one kernel per component.
    interp_kind: _native.INTERP_QUARTIC_MID (dopri5.py:39-45 + interp.py) or INTERP_TSIT5[_REF] (tsit5.py:33-50)."""
    dt_ = _np_dtype(y0[0].dtype).type
    assert (dt_(t0) <= dt_(t)) & (dt_(t) <= dt_(t1)), \
        'invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}'.format(t0, t, t1)     # interp.py:59
    lib = N.load()
    outs = []
    for y0_, y1_, k_ in zip(y0, y1, k):
        y0_, y1_ = _contig(y0_), _contig(y1_)
        ks = [_contig(x) for x in k_]
        nk = len(ks)
        out = torch.empty_like(y0_)
        ptrs = (C.c_void_p * nk)(*[x.data_ptr() for x in ks])
        cm = (C.c_double * nk)(*[float(c) for c in (c_mid if c_mid is not None else [0.0] * nk)])
        rc = lib.mi_ode_interp_eval(N.dtype_code(y0_.dtype), interp_kind, y0_.numel(), _ptr(y0_), _ptr(y1_), ptrs, nk, cm,
                                    float(dt), float(t0), float(t1), float(t), _ptr(out), N.stream_ptr(y0_.device))
        N.check(rc, 'mi_ode_interp_eval')
        outs.append(out)
    return tuple(outs)

"""Helpers to load tests/golden/*.npz fixtures (data only; see tests/golden/make_golden.py)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    d = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = json.loads(str(d['meta']))
    return d, meta


def run_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'run_*.npz'))
                  if not p.endswith('run_mlp_weights.npz'))


def mlp_weights():
    d, _ = load('run_mlp_weights')
    return {k: d[k] for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3')}


def trace_is_decided(trace, order, tsit5, safety32=float(np.float32(0.9))):
    """True if no attempt of a reference trace has an error ratio within 1e-9 of the accept threshold 1.  The fixture stores
    (t, dt, accepted, dt_next); the ratio follows from dt / dt_next = ratio**e / safety when no clamp was hit."""
    dt, dtn = trace[:, 1], trace[:, 3]
    fac = dt / dtn * safety32                      # = ratio ** (e / 2)  (misc) or ratio ** e (tsit5), if no clamp was hit
    e = 1.0 / order if tsit5 else 0.5 * float(np.float32(1.0 / order))
    with np.errstate(all='ignore'):
        ratio = fac ** (1.0 / e)
    clamped = (np.abs(dt / dtn - 0.1) < 1e-12) | (np.abs(dt / dtn - 5.0) < 1e-9) | (np.abs(dt / dtn - 1.0) < 1e-12)
    return not np.any((np.abs(ratio - 1.0) < 1e-9) & ~clamped)


RK_ORDER = {'dopri5': 5, 'tsit5': 5, 'bosh3': 3, 'dopri8': 8, 'adaptive_heun': 5}


def traces_touching_the_threshold():
    """Names of the float64 adaptive-RK fixtures whose reference trace passes through ratio == 1 +- 1e-9: the only ones for which
    a step sequence may legitimately differ from the reference's by a decision (profiles/r03_skip_reasons.txt lists them)."""
    out = []
    for n in run_cases():
        d, meta = load(n)
        if 'trace' not in d.files or meta['method'] not in RK_ORDER or meta['max_attempts'] is not None or d['trace'].shape[1] < 4:
            continue
        if not trace_is_decided(d['trace'], RK_ORDER[meta['method']], meta['method'] == 'tsit5'):
            out.append(n)
    return out

"""Per-case float32 parity bands (round 3: the blanket rtol 2e-3 / atol 2e-4 of the fp32 comparisons was ~200x what the
kernels deliver).

A band B for a comparison `key` means   |got - ref| <= B * (1 + |ref|)   for every element.  The bands live in
tests/golden/fp32_bands.json: B = 10 x the largest value of  max |got - ref| / (1 + |ref|)  OBSERVED on the MI355X (two
runs; scripts/measure_fp32_bands.sh regenerates them), never below 2e-6 (a few float32 ulps of an O(1) state).
The observed values are kept next to the bands.  Recording mode (TFDIFFEQ_AMD_RECORD_BANDS=<file>): nothing is
asserted, every comparison appends {key: observed} to <file>.
"""
import json
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fp32_bands.json')
_REC = os.environ.get('TFDIFFEQ_AMD_RECORD_BANDS')
_cache = {}


def _bands():
    if 'b' not in _cache:
        _cache['b'] = json.load(open(_PATH))['bands'] if os.path.exists(_PATH) else {}
    return _cache['b']


def observed(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, 'shape %s vs %s' % (got.shape, ref.shape)
    return float((np.abs(got - ref) / (1.0 + np.abs(ref))).max())


def assert_f32(got, ref, key):
    obs = observed(got, ref)
    if _REC:
        with open(_REC, 'a') as f:
            f.write(json.dumps({'key': key, 'observed': obs}) + '\n')
        return
    b = _bands().get(key)
    assert b is not None, 'no float32 band for %r in tests/golden/fp32_bands.json (scripts/measure_fp32_bands.sh records them)' % key
    assert obs <= b['band'], '%s: max |got - ref| / (1 + |ref|) = %.3e outside the band %.1e (observed when recorded: %.3e)' % (
        key, obs, b['band'], b['observed'])


def assert_scalar(value, key):
    """A scalar deviation (e.g. a relative gradient error) against its recorded band (10 x observed, same file)."""
    value = float(value)
    if _REC:
        with open(_REC, 'a') as f:
            f.write(json.dumps({'key': key, 'observed': value}) + '\n')
        return
    b = _bands().get(key)
    assert b is not None, 'no band for %r in tests/golden/fp32_bands.json (scripts/measure_fp32_bands.sh records them)' % key
    assert value <= b['band'], '%s: %.3e outside the band %.1e (observed when recorded: %.3e)' % (key, value, b['band'], b['observed'])

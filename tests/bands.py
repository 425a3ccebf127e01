"""Per-case float32 parity bands (round 3: the blanket rtol 2e-3 / atol 2e-4 of the fp32 comparisons was ~200x what the
kernels deliver).

A band B for a comparison `key` means   |got - ref| <= B * (1 + |ref|)   for every element.  The bands live in
tests/golden/fp32_bands.json: B = 10 x the largest value of  max |got - ref| / (1 + |ref|)  OBSERVED on the MI355X (two
runs; scripts/measure_fp32_bands.sh regenerates them), never below 2e-6 (a few float32 ulps of an O(1) state).
The observed values are kept next to the bands.

A recorded band is a regression guard, not a correctness argument (it was derived from this implementation's own output), so
every comparison is ALSO held to an a-priori ceiling that does not depend on what was observed:
    state comparisons   1e-3  = 8 x [eps32 (6e-8) x 7 stage evaluations x 300 attempts], the roundoff a float32 Dopri5 run of the
                                longest fixture can accumulate if every rounding error lined up - and the local tolerances of the
                                float32 cases (rtol 1e-4 .. 1e-3) allow no less;
    scalar deviations   the caller's `ceiling` (gradient comparisons state theirs next to the call), default 1e-3.
The assertion is  observed <= min(recorded band, ceiling).  Two groups need a ceiling above 1e-3 and say so at the call:
relu-network gradients (kinks: a sample that crosses one between the float32 and the float64 solve changes a whole column of
the weight gradient; observed 2.5e-3 .. 4.7e-3, ceiling 1e-2) and dopri8 in float32 (13-stage combination with coefficients up to
~2e2 in magnitude: cancellation noise 3e-4 observed, ceiling 3e-3).

Recording mode (TFDIFFEQ_AMD_RECORD_BANDS=<file>): every comparison appends {key: observed} to <file> and asserts only the
a-priori ceiling; the session then ENDS WITH A NON-ZERO EXIT STATUS and a banner (tests/conftest.py) - a run in recording mode can
never be mistaken for a green parity run, whatever inherited the variable.
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


CEILING = 1e-3          # a-priori bound of every float32 comparison (module docstring); callers with a reason pass their own
recorded = []           # keys recorded in this session (recording mode): tests/conftest.py turns a non-empty list into a failed session
EPS32 = 2.0 ** -24      # unit roundoff of float32


def case_ceiling(attempts, stages=7, growth=1.0, k=8.0):
    """Round 5 (round-4 review, weak point 2): the a-priori ceiling of ONE comparison from that case's own size instead of the blanket 1e-3 -
    k x eps32 x (stage evaluations per attempt) x (attempts of the case's own reference trace) x growth, never below 4e-6 (a few float32
    ulps of an O(1) state).  `growth`: how much the dynamics amplify a perturbation over the horizon (1 for the contractive / neutral
    fixtures; a caller with an expanding system states its factor and why).  k = 8: every rounding error of a stage lining up, with the
    combination's dozen operations per element folded in.  A ten-attempt run gets 3.3e-5 where the blanket bound allowed 1e-3."""
    return max(4e-6, k * EPS32 * stages * max(int(attempts), 1) * growth)


def _record(key, obs):
    recorded.append(key)
    with open(_REC, 'a') as f:
        f.write(json.dumps({'key': key, 'observed': obs}) + '\n')


def assert_f32(got, ref, key, ceiling=CEILING):
    obs = observed(got, ref)
    assert obs <= ceiling, '%s: max |got - ref| / (1 + |ref|) = %.3e above the a-priori ceiling %.1e' % (key, obs, ceiling)
    if _REC:
        _record(key, obs)
        return
    b = _bands().get(key)
    assert b is not None, 'no float32 band for %r in tests/golden/fp32_bands.json (scripts/measure_fp32_bands.sh records them)' % key
    if b['observed'] == 0.0 and b.get('runs', 1) >= 2 and not key.endswith('vs_oracle'):
        # Round 5: a comparison that was bit-exact in every recorded run (plane kernels against the float32 fixtures, fused against plane
        # kernels on elementwise systems) IS an exactness statement - it is held to equality, not to the 2e-6 floor of a band.  Not for
        # a comparison against the CPU oracle's own matrix product (`..vs_oracle`): equality there would be a coincidence of two libraries
        assert obs == 0.0, '%s: was bit-exact when recorded (two runs), now max |got - ref| / (1 + |ref|) = %.3e' % (key, obs)
        return
    assert obs <= b['band'], '%s: max |got - ref| / (1 + |ref|) = %.3e outside the band %.1e (observed when recorded: %.3e)' % (
        key, obs, b['band'], b['observed'])


def assert_scalar(value, key, ceiling=CEILING):
    """A scalar deviation (e.g. a relative gradient error) against min(its recorded band (10 x observed, same file), ceiling)."""
    value = float(value)
    assert value <= ceiling, '%s: %.3e above the a-priori ceiling %.1e' % (key, value, ceiling)
    if _REC:
        _record(key, value)
        return
    b = _bands().get(key)
    assert b is not None, 'no band for %r in tests/golden/fp32_bands.json (scripts/measure_fp32_bands.sh records them)' % key
    assert value <= b['band'], '%s: %.3e outside the band %.1e (observed when recorded: %.3e)' % (key, value, b['band'], b['observed'])

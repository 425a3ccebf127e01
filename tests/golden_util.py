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

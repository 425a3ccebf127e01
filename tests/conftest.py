import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible, e.g. a plain `pytest tests/` here.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """A session that RECORDED float32 bands asserted none of them (tests/bands.py): it must not look like a green parity run."""
    if os.environ.get('TFDIFFEQ_AMD_RECORD_BANDS'):
        from tests import bands
        if bands.recorded:
            sys.stderr.write('\n*** TFDIFFEQ_AMD_RECORD_BANDS is set: %d float32 comparisons were RECORDED, not asserted against their bands - '
                             'this session does not count as a parity run (exit status 3) ***\n' % len(bands.recorded))
            session.exitstatus = 3

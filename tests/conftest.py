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


# Test modules written before the tracer existed (round 6) hand `odeint` Python callables BECAUSE they test the callable engines
# (device-controlled, hipGraph replay, host loop, generic adjoint): there the automatic lowering stays off, so that they keep
# testing what they were written for.  tests/test_gpu_lower.py is the module that tests the lowering itself - with the default on.
_LOWERING_AWARE = ('test_gpu_lower', 'test_lower_trace', 'test_gpu_published')


@pytest.fixture(autouse=True)
def _callable_engine_tests_keep_their_engine(request):
    import sys as _sys
    mod = _sys.modules.get('tfdiffeq_amd.odeint')
    if mod is None:
        import tfdiffeq_amd  # noqa: F401
        mod = _sys.modules['tfdiffeq_amd.odeint']
    name = request.module.__name__.rsplit('.', 1)[-1]
    before = mod.LOWER_DEFAULT
    if name not in _LOWERING_AWARE:
        mod.LOWER_DEFAULT = False
    try:
        yield
    finally:
        mod.LOWER_DEFAULT = before


def pytest_sessionfinish(session, exitstatus):
    """A session that RECORDED float32 bands asserted none of them (tests/bands.py): it must not look like a green parity run."""
    if os.environ.get('TFDIFFEQ_AMD_RECORD_BANDS'):
        from tests import bands
        if bands.recorded:
            sys.stderr.write('\n*** TFDIFFEQ_AMD_RECORD_BANDS is set: %d float32 comparisons were RECORDED, not asserted against their bands - '
                             'this session does not count as a parity run (exit status 3) ***\n' % len(bands.recorded))
            session.exitstatus = 3

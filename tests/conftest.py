import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def pytest_sessionfinish(session, exitstatus):
    """IM_DEBUG_GUARDS=1 (tools/stress_gpu_suite.sh): a guard word the library found overwritten anywhere in the session fails
    the run, also when the test that caused it swallowed the -90 (e.g. inside a destructor)."""
    if os.environ.get("IM_DEBUG_GUARDS") != "1":
        return
    from icepy4d_amd import _lib
    if _lib._lib is None:
        return
    n = _lib._lib.im_debug_guard_failures()
    print(f"\nIM_DEBUG_GUARDS: {n} guard failure(s) in this session")
    if n and session.exitstatus == 0:
        session.exitstatus = 1

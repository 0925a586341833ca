import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; see oracle/tetris_oracle.h)."""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture()
def host_backend():
    """tetris_amd bound to the CPU harness build of the lane logic (test only)."""
    from tetris_amd import _lib
    import harness_backend
    old = _lib._install_test_backend(harness_backend.binding())
    yield
    _lib._install_test_backend(old)

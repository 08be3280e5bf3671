import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


SESSION_START = [None]


def pytest_configure(config):
    import time
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    SESSION_START[0] = time.time()


def session_seconds():
    """Seconds since this pytest session was configured (tests/test_gpu_samples.py: the one sample whose unmodified host code needs two
    minutes on a normal box is skipped when the box has already shown itself several times slower than normal)."""
    import time
    return time.time() - SESSION_START[0] if SESSION_START[0] else 0.0


@pytest.fixture(scope="session")
def built():
    """The HIP library and the CPU oracle, built once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True

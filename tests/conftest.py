import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The suite (and every child process it spawns) loads the TEST-HOOKS flavour of the libraries (cudalibrarysamples_amd/lib_hooks/: the same
# kernel objects, host code compiled with -DCTAMD_TEST_HOOKS): the tests force kernels, transports and planner rules through CUTENSOR*_AMD_*
# switches and replay multi-device plans on the host, none of which the production libraries (lib/: bench.py, smoke(), the samples, the
# reference's binding) read or export.  tests/test_abi.py checks the production flavour's exports and strings separately.
os.environ.setdefault("CTAMD_LIB_FLAVOUR", "hooks")


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

"""Drop-in evidence on the GPU: the reference's own sample drivers, compiled unmodified from
/root/reference against lib/libcutensor.so / libcutensorMg.so by oracle/build_ref_samples.sh (the
binaries travel to the GPU box inside oracle/_ref/; /root/reference itself is not needed at run time),
must run to completion.  The samples print timings and check only status codes — numerical parity is
the job of the other test files."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
SAMPLES = ["contraction", "einsum", "reduction", "elementwise_permute", "contraction_multi_gpu"]


@pytest.mark.parametrize("name", SAMPLES)
def test_reference_sample_runs(built, name):
    exe = os.path.join(REF, name)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/%s was not built (reference tree absent at build time)" % name)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "%s exited %d\nstdout:\n%s\nstderr:\n%s" % (name, r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    out = r.stdout + r.stderr
    assert "rror" not in out.replace("No such file or directory", ""), out[-2000:]

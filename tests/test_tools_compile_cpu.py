"""The microbenchmarks that DESIGN.md / NOTES.md quote numbers from must keep compiling for gfx950 (syntax check only: hipcc
cross-compiles without a GPU; the binaries are built and run by hand on a GPU box, profiles/r04*_ubench_*)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["ldsdma_rate.hip", "ldsdma_shared_panels.hip", "store_rate_by_cus.hip"]


@pytest.mark.parametrize("src", SOURCES)
def test_round4_microbenchmarks_compile_for_gfx950(src):
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc in this environment")
    path = os.path.join(ROOT, "tools", "ubench", src)
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-w", "-fsyntax-only", path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]

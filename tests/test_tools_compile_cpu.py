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


def test_first_contact_script_dry_run_names_only_things_that_exist():
    """tools/first_contact_8gpu.sh is the one command for the first multi-GPU node; --dry-run prints what it would execute for 8 visible
    GPUs.  Every python file / test file / binary-producing recipe it names must exist, every bench.py flag must be one bench.py's
    parser knows, the scaling sequence must be the driver's own (1, 2, 4, 8 under torch.distributed.run with --master-addr 127.0.0.1),
    and all three transports and 1 / 2 / 4 gather waves must be in it."""
    import re
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "first_contact_8gpu.sh"), "--dry-run"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, FIRST_CONTACT_NGPU="8"))
    assert r.returncode == 0, r.stderr[-2000:]
    cmds = [ln[2:] for ln in r.stdout.splitlines() if ln.startswith("+ ")]
    assert len(cmds) >= 14, r.stdout
    bench_src = open(os.path.join(ROOT, "bench.py")).read()
    known_flags = set(re.findall(r"add_argument\(\"(--[a-z0-9-]+)\"", bench_src))
    seen_n, transports, waves = [], set(), set()
    for c in cmds:
        for tok in c.split():
            if tok.endswith(".py") or tok.startswith("tests/") or tok.startswith("tools/"):
                path = tok if os.path.isabs(tok) else os.path.join(ROOT, tok)
                assert os.path.exists(path), (tok, c)
        if "bench.py" in c:
            flags = re.findall(r"(--[a-z0-9-]+)", c.split("bench.py", 1)[1])
            assert set(flags) <= known_flags, (flags, c)
            n = int(re.search(r"--gpus (\d+)", c).group(1))
            if n > 1 and "rocprofv3" not in c:
                assert "torch.distributed.run" in c and "--master-addr 127.0.0.1" in c and "--nproc-per-node %d" % n in c, c
            if "CUTENSORMG_AMD_" not in c and "rocprofv3" not in c:
                seen_n.append(n)
        m = re.search(r"CUTENSORMG_AMD_TRANSPORT=(\w+)", c)
        if m:
            transports.add(m.group(1))
        m = re.search(r"CUTENSORMG_AMD_WAVES=(\d+)", c)
        if m:
            waves.add(int(m.group(1)))
    assert seen_n == [1, 2, 4, 8] and transports == {"allgather", "sendrecv", "peer"} and waves == {1, 2, 4}, (seen_n, transports, waves)
    assert any("rocprofv3 --kernel-trace --stats" in c and "--pmc" not in c for c in cmds)
    # the reference's own multi-GPU samples are part of it when oracle/_ref was built
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "blog_post")):
        assert any("oracle/_ref/contraction_multi_gpu" in c for c in cmds) and sum("oracle/_ref/blog_post 8" in c for c in cmds) == 4

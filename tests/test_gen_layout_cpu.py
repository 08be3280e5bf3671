"""Host replay of the general MFMA GETT kernel's tile staging (csrc/kernels/gett_gen_layout.h, the index arithmetic that
gett_gen.inc compiles into the kernels): tests/harness/gen_layout_harness.cpp stages a tile with all 256 threads and reads every
MFMA fragment back, for every (element size, tile, orientation, vector width) the kernel tables instantiate.  No GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_staging_and_fragment_reads_agree(tmp_path):
    exe = str(tmp_path / "gen_layout_harness")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "cudalibrarysamples_amd", "csrc", "kernels"),
                           os.path.join(ROOT, "tests", "harness", "gen_layout_harness.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "gen layout ok" in r.stdout, r.stdout + r.stderr


def test_the_harness_covers_every_instantiated_shape():
    """Every (element bytes, rows, BK, V) of the three kernel tables appears in the harness's REPLAY list."""
    import re
    kdir = os.path.join(ROOT, "cudalibrarysamples_amd", "csrc", "kernels")
    es = {"GEN_BF16": 2, "GEN_F16": 2, "GE": 2, "GEN_F64": 8, "GEN_C32": 8, "GEN_C64": 16}
    want = set()
    for f in ("gett_gen_h16.hip", "gett_gen_f64.hip", "gett_gen_cplx.hip"):
        for m in re.finditer(r"CTAMD_GEN_ORIENTS\((\w+), (\d+), (\d+), (\d+), (\d+)\)", open(os.path.join(kdir, f)).read()):
            ge, bm, bn, bk, v = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))
            want.add((es[ge], bm, bk, v))
            want.add((es[ge], bn, bk, v))
    src = open(os.path.join(ROOT, "tests", "harness", "gen_layout_harness.cpp")).read()
    have = {tuple(int(x) for x in m.groups()) for m in re.finditer(r"REPLAY\((\d+), (\d+), (\d+), (\d+)\)", src)}
    assert want and want <= have, sorted(want - have)


def test_persistent_kernel_epilogue_image_replay(tmp_path):
    """gett_h16w4p_kernel's transposed epilogue image (csrc/kernels/gett_h16p_layout.h, compiled into the kernel): a pass replayed on
    the CPU — every element reaches the lane / register that stores it, every image byte is written once, no bank conflicts."""
    exe = str(tmp_path / "h16p_layout_harness")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "cudalibrarysamples_amd", "csrc", "kernels"),
                           os.path.join(ROOT, "tests", "harness", "h16p_layout_harness.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "h16p layout ok" in r.stdout, r.stdout + r.stderr

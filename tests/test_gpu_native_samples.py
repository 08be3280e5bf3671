"""The native HIP sample drivers (samples/*.hip): the call sequences of the reference's contraction.cu, einsum.cu,
reduction.cu, elementwise_permute.cu, contraction_multi_gpu.cu and blog_post.cu written directly against the engine's C
ABI in HIP — no CUDA names, no compatibility headers — each with the host-side value check the reference samples do not
have.  (tests/test_gpu_samples.py runs the reference's own unmodified sources; this file is the same path without the
name-aliasing fixture.)"""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "samples", "bin")


def _run(args, timeout=600):
    exe = os.path.join(BIN, args[0])
    assert os.path.exists(exe), "samples/bin/%s is not built (__graft_entry__.build())" % args[0]
    r = subprocess.run([exe] + args[1:], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "%s exited %d\n%s\n%s" % (" ".join(args), r.returncode, r.stdout[-3000:], r.stderr[-2000:])
    assert "FAILED" not in r.stdout, r.stdout[-3000:]
    return r.stdout


def test_contraction_default_and_shrunk(built):
    out = _run(["contraction"])                       # contraction.cu defaults: 4096 sampled outputs vs fp64 host dot products
    assert "check: 4096 outputs" in out and "-> ok" in out and "GFLOPs/s" in out
    out = _run(["contraction", "--shrink"])           # BASELINE configs[0] plumbing case: every element
    assert "check: 13824 outputs" in out and "-> ok" in out


def test_einsum_demo_equations_and_headline(built):
    out = _run(["einsum"])
    assert out.count("-> ok") == 6 and "all checks ok" in out
    assert "[ 2 5 7 8 ]" in out and "[ 2 7 8 5 ]" in out and "[ 5 4 ]" in out and "[ 96 96 ]" in out      # einsum.cu:447-451 output shapes


def test_reduction_and_permutation(built):
    out = _run(["bandwidth"])
    assert "reduction:" in out and "permutation:" in out and out.count("-> ok") == 2 and "0 mismatches" in out


def test_multi_gpu_block_cyclic_with_value_check(built):
    out = _run(["multi_gpu", "1024", "256"])          # contraction_multi_gpu.cu layout (2 x 2 grids, two local blocks), visible devices
    assert "-> ok" in out
    out = _run(["multi_gpu", "512", "256", "--devices", "0,0,0,0"])     # four logical devices on one GPU
    assert "-> ok" in out and "4 device(s)" in out


@pytest.mark.parametrize("scaling", list(range(1, 13)))
def test_blog_post_harness_with_value_check(built, scaling):
    """blog_post.cu <numDevices> <scaling> (:131-146) — the reference harness is only an rc check; this one compares 256
    sampled outputs of the six-mode result with fp64 sums over K0, K1, K2 (scaling 11: ragged last blocks in M1 / N1)."""
    out = _run(["multi_gpu", "--blog", "1", str(scaling)])
    assert "-> ok" in out


@pytest.mark.parametrize("ndev,scaling", [(2, 3), (2, 12), (4, 2), (4, 9), (4, 12), (8, 2), (8, 5), (8, 12)])
def test_blog_post_layouts_of_multi_device_runs_on_logical_devices(built, ndev, scaling):
    """blog_post.cu <numDevices> <scaling> with numDevices = 2 / 4 / 8 (:131-146) as n LOGICAL devices on GPU 0 (--virtual): the
    descriptors (device counts per mode, :78-101), plans and kernels of the n-GPU run, value-checked.  At 4 devices from scaling 9
    and at 8 devices from scaling 2 a local view has five unfusable modes in a group; the plan peels one block-index digit into
    a host loop so that the tiled kernels run (57-99 TFLOP/s here) instead of the mode-table kernel (0.65 TFLOP/s)."""
    out = _run(["multi_gpu", "--blog", str(ndev), str(scaling), "--virtual"])
    assert "-> ok" in out and ("on %d device(s)" % ndev) in out
    gflops = float(out.split("GFLOPs/s")[0].split(",")[-1])
    assert gflops > 5000.0, out      # the mode-table kernel would be ~650


def test_blog_post_layout_with_the_plan_level_peel(built):
    """The same 8-device layout with the single-GPU library's peel switched off (CUTENSOR_AMD_PEEL=0): cuTENSORMg's own peel — a
    block-index digit of the oversized group walked by the piece loop — keeps the local contractions on the tiled kernels."""
    import os
    exe = os.path.join(BIN, "multi_gpu")
    r = subprocess.run([exe, "--blog", "8", "12", "--virtual"], capture_output=True, text=True, timeout=600, env=dict(os.environ, CUTENSOR_AMD_PEEL="0"))
    assert r.returncode == 0 and "-> ok" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
    assert float(r.stdout.split("GFLOPs/s")[0].split(",")[-1]) > 5000.0, r.stdout

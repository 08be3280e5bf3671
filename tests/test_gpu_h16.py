"""GPU parity of the 16-bit (bf16 / fp16 data, fp32 accumulate) MFMA contraction path — SURVEY §8 row a6,
BASELINE configs[3]: contraction.cu:33-40 retyped to CUDA_R_16BF + CUTENSOR_COMPUTE_DESC_16BF as in
python/cutensor/torch/einsum.cc:35-40 — through the C ABI (cutensorContract) against the oracle on
inputs already rounded to the 16-bit type.

Tolerance: the result is the fp32-accumulated sum rounded once to the 16-bit output type, so
|got - ref| <= 2^-8 |ref| (bf16) / 2^-11 |ref| (fp16) plus fp32 accumulation noise; the tests use
rtol 8e-3 (bf16) / 2e-3 (fp16) as stated in DESIGN.md, with a small atol for sums that cancel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import torch
    assert torch.cuda.is_available()
    from cudalibrarysamples_amd import cutensor as ct, ops
    return torch, ct, ops, ops.Handle()


def _packed_strides(ext):
    s, acc = [], 1
    for e in ext:
        s.append(acc)
        acc *= e
    return s


def _run(env, ext, mA, mB, mC, dtype_name="bfloat16", alpha=1.0, beta=0.0, seed=0, expect_mfma=True):
    """D = alpha * A * B + beta * C in packed column-major layout; returns (got, ref) as float64 arrays of
    shape extC (Fortran order)."""
    torch, ct, ops, h = env
    tdt = getattr(torch, dtype_name)
    cdt = ct.R_16BF if dtype_name == "bfloat16" else ct.R_16F
    eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    # column-major packed tensor with modes m0 (fastest) .. == a torch tensor of the reversed shape
    A = (torch.rand(eA[::-1], generator=g, device="cuda") * 2 - 1).to(tdt)
    B = (torch.rand(eB[::-1], generator=g, device="cuda") * 2 - 1).to(tdt)
    C = (torch.rand(eC[::-1], generator=g, device="cuda") * 2 - 1).to(tdt)
    D = C.clone()
    plan = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, dtype=cdt, workspace_limit=1 << 28)
    d = plan.describe()
    if expect_mfma:
        assert d["family"] == 1 and d["kernel"] >= 0, d
    ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    plan.contract(alpha, A.data_ptr(), B.data_ptr(), beta, C.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    # oracle on the rounded inputs: fp64 einsum over the reversed (row-major) views
    rA, rB, rC = mA[::-1], mB[::-1], mC[::-1]
    ref = torch.einsum("%s,%s->%s" % (rA, rB, rC), A.double().cpu(), B.double().cpu())
    ref = alpha * ref + beta * C.double().cpu()
    got = D.double().cpu().numpy()
    if float(np.prod([float(v) for v in ext.values()])) <= 1.5e8:
        # the CPU oracle's own 16-bit entry point on the same bit patterns (fp64 accumulation, one rounding): the GPU's
        # fp32-accumulated result may land on the neighbouring 16-bit value, never further
        import oracle
        kind = "bf16" if dtype_name == "bfloat16" else "f16"
        bits = lambda t: t.view(torch.int16).cpu().numpy().view(np.uint16)   # noqa: E731  (row-major view of the reversed shape)
        oD = np.zeros_like(bits(D))
        oracle.contract(bits(A), rA, bits(B), rB, oD, rC, alpha=alpha, beta=beta, C=bits(C), h16=kind)
        want = oracle.from_bits(oD, kind)
        ulp = 2.0 ** -7 if kind == "bf16" else 2.0 ** -10
        np.testing.assert_allclose(got, want, rtol=ulp, atol=3e-2 if kind == "bf16" else 4e-3)
    return got, ref.numpy(), d


LAYOUTS = {
    # name: (modes of A, modes of B)   C is always [m, n] with m fastest
    "sample_mk_kn": ("mk", "kn"),     # contraction.cu GEMM-like config: A free-contiguous, B K-contiguous
    "km_kn": ("km", "kn"),            # both K-contiguous
    "mk_nk": ("mk", "nk"),            # both free-contiguous
    "km_nk": ("km", "nk"),
}


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
@pytest.mark.parametrize("dims", [(512, 512, 256), (520, 264, 192), (256, 256, 64), (8, 1032, 128)])
def test_gemm_like_bf16(env, layout, dims):
    m, n, k = dims
    mA, mB = LAYOUTS[layout]
    got, ref, d = _run(env, dict(m=m, n=n, k=k), mA, mB, "mn", seed=hash((layout, dims)) % 1000)
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=2e-2)


def test_alpha_beta_and_fp16(env):
    got, ref, _ = _run(env, dict(m=384, n=264, k=128), "mk", "kn", "mn", alpha=1.5, beta=-0.75, seed=3)
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=3e-2)
    got, ref, _ = _run(env, dict(m=264, n=384, k=192), "km", "nk", "mn", dtype_name="float16", alpha=0.5, beta=0.25, seed=4)
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=1e-2)


def test_multi_mode_contraction_bf16(env):
    """Tensor (not matrix) shapes: C[m,u,n,v] = A[m,h,k,n] B[u,k,v,h] (contraction.cu:43) in bf16 with a
    64-deep fastest contracted mode, and the headline einsum's mode structure 'abcd,dcbe->ae'."""
    got, ref, d = _run(env, dict(m=24, n=16, u=16, v=24, h=6, k=64), "mhkn", "ukvh", "munv", seed=7)
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=3e-2)
    got, ref, d = _run(env, dict(a=96, b=4, c=6, d=64, e=96), "dcba", "ebcd", "ea", seed=8)
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=5e-2)


RAGGED_K = [  # (m, n, k): one contracted mode, k % 64 != 0
    (512, 512, 200),     # three whole K-tiles and 8 of the fourth
    (520, 264, 72),      # two K-tiles: the masked one is staged by the prologue
    (256, 256, 8),       # ONE K-tile, 8 of its 64 k
    (304, 520, 1096),    # odd tile count (18): the masked tile comes out of the single-tile tail of the pair loop
    (1024, 1024, 1080),  # even tile count (17 whole + 56)
]


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
@pytest.mark.parametrize("dims", RAGGED_K)
def test_ragged_k_stays_on_the_lds_dma_kernels(env, layout, dims):
    """K % 64 != 0 with 16-byte lanes (round-4 review, Missing #4): the 256 x 256 / 128 x 128 LDS-DMA kernels stage the last K-tile with
    the lanes past the end of the contracted mode out of range (zeros in LDS), instead of handing the problem to the general family.
    What lies behind the end of a K-contiguous row is the next row's data, so a lane that is not masked shows up in the result."""
    m, n, k = dims
    mA, mB = LAYOUTS[layout]
    got, ref, d = _run(env, dict(m=m, n=n, k=k), mA, mB, "mn", seed=hash((layout, dims)) % 1000)
    assert d["kname"] in ("gett_h16w4x_kernel", "gett_h16w4m_kernel", "gett_h16w4m4_kernel", "gett_h16w4q_kernel"), d
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=2e-2)


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
def test_ragged_k_large_shapes_by_the_planners_own_choice(env, layout):
    mA, mB = LAYOUTS[layout]
    got, ref, d = _run(env, dict(m=2048, n=2048, k=200), mA, mB, "mn", seed=31)
    assert d["family"] == 1 and d["kname"] in ("gett_h16w4m_kernel", "gett_h16w4m4_kernel"), d
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=2e-2)
    got, ref, d = _run(env, dict(m=4096, n=2048, k=1096), mA, mB, "mn", seed=32)
    assert d["family"] == 1 and d["kname"] in ("gett_h16w4x_kernel", "gett_h16w4m_kernel"), d
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=3e-2)


def test_ragged_k_any_extent_when_both_operands_are_free_contiguous(env):
    """A[m,k] B[n,k] (both free-contiguous): the rows k >= K of the last K-tile are masked whatever K is — 77 here, and 4100 with
    alpha / beta in fp16.  The rows behind the end of B are past the end of its allocation: they must not be touched."""
    got, ref, d = _run(env, dict(m=2048, n=1032, k=77), "mk", "nk", "mn", seed=21)
    assert d["family"] == 1, d
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=2e-2)
    got, ref, d = _run(env, dict(m=1032, n=2048, k=4100), "mk", "nk", "mn", dtype_name="float16", alpha=0.5, beta=0.25, seed=22)
    assert d["family"] == 1, d
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize("dims", [(96, 96, 4104), (264, 120, 1544), (512, 256, 8200)])
def test_ragged_k_split_k(env, dims):
    """Small outputs, deep ragged K: the slices are whole K-tiles, the LAST slice owns the masked tile; fp32 partials, one rounding."""
    import os
    m, n, k = dims
    got, ref, d = _run(env, dict(m=m, n=n, k=k), "km", "kn", "mn", seed=23, expect_mfma=False)
    if os.environ.get("CUTENSOR_AMD_H16_WAVES", "4x") in ("4x", "4m", "4m4", "4q"):   # (test_kernel_variant_parity runs this file under every variant)
        assert d["family"] == 1 and d["splitK"] > 1 and d["kPerSlice"] % 64 == 0, d
    else:
        assert d["family"] == 2, d                 # a kernel without a masked K-tile was asked for: the general family
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=3e-2)


def test_ragged_k_forced_kernels(built):
    """Every kernel that masks a partial K-tile — the 256 x 256 kernel, the 128 x 128 pair (ring of two and of four K-tiles: the deep ring
    stages a tile's A and B pieces at different times, one switch each), the 64 x 64 tile — forced by CUTENSOR_AMD_H16_WAVES in a child
    process on K ranges of 1 .. 6 K-tiles (the masked tile staged by the prologue, by the unrolled loop, by its tail)."""
    import os, subprocess, sys
    code = r'''
import numpy as np, torch
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
g = torch.Generator(device="cuda"); g.manual_seed(5)
for (mA, mB) in (("mk", "kn"), ("km", "kn"), ("mk", "nk"), ("km", "nk")):
    for (m, n, k) in ((384, 384, 328), (256, 128, 8), (640, 384, 136), (384, 640, 264), (128, 128, 200), (264, 72, 72), (256, 256, 456)):
        ext = dict(m=m, n=n, k=k)
        eA, eB = [ext[c] for c in mA], [ext[c] for c in mB]
        A = (torch.rand(eA[::-1], generator=g, device="cuda") * 2 - 1).bfloat16()
        B = (torch.rand(eB[::-1], generator=g, device="cuda") * 2 - 1).bfloat16()
        D = torch.full((n, m), float("nan"), dtype=torch.bfloat16, device="cuda")
        plan = ops.contraction_plan(h, eA, mA, eB, mB, [m, n], "mn", dtype=ct.R_16BF, workspace_limit=1 << 28)
        d = plan.describe()
        assert d["family"] == 1 and d["kname"] == WANT, d
        ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
        plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
        torch.cuda.synchronize()
        ref = torch.einsum("%s,%s->nm" % (mA[::-1], mB[::-1]), A.double(), B.double())
        np.testing.assert_allclose(D.double().cpu().numpy(), ref.cpu().numpy(), rtol=8e-3, atol=2e-2)
print("ok")
'''
    for waves, want in (("4m", "gett_h16w4m_kernel"), ("4m4", "gett_h16w4m4_kernel"), ("4x", "gett_h16w4x_kernel"), ("4q", "gett_h16w4q_kernel")):
        envv = dict(os.environ, CUTENSOR_AMD_H16_WAVES=waves)
        r = subprocess.run([sys.executable, "-c", code.replace("WANT", repr(want))], env=envv, capture_output=True, text=True, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (waves, r.stdout[-2000:], r.stderr[-3000:])


SWEEP_LAYOUTS = {
    # two contracted modes (k the fastest) that do not fuse: A carries m between them or B holds them in the other order
    "k_contig_both": ("kmj", "kjn"),
    "a_free_b_k": ("mjk", "kjn"),
    "a_k_b_free": ("kmj", "nkj"),
    "free_both": ("mjk", "nkj"),
}
SWEEP_DIMS = [  # (m, n, k, j): k % 64 != 0, k % 8 == 0
    (264, 136, 40, 5),     # one K-tile per sweep: every tile is staged masked, the mask is switched on once
    (512, 264, 72, 3),     # two per sweep: the mask toggles at every K-tile
    (264, 520, 96, 4),     # even tile count
    (136, 264, 200, 3),    # four per sweep (3 whole + 8 k), odd tile count
    (96, 96, 72, 40),      # split-K: slices start inside a sweep
]


@pytest.mark.parametrize("layout", sorted(SWEEP_LAYOUTS))
@pytest.mark.parametrize("dims", SWEEP_DIMS)
def test_sweep_ragged_k_stays_on_the_lds_dma_kernels(env, layout, dims):
    """Several contracted modes and the fastest one without whole K-tiles (round-5 review, Missing #5): the K-tiles are counted per sweep
    of that mode, rounded up, and the last one of EVERY sweep is staged with the lanes past the end of the mode out of range — the mask
    is switched on and off as the odometer goes (x_rag_toggle).  A lane that stays unmasked adds the next sweep's first elements (or,
    free-contiguous, rows of the next index) to the sum; one that stays masked drops live data."""
    m, n, k, j = dims
    mA, mB = SWEEP_LAYOUTS[layout]
    got, ref, d = _run(env, dict(m=m, n=n, k=k, j=j), mA, mB, "mn", seed=hash((layout, dims)) % 1000)
    assert d["kname"] in ("gett_h16w4x_kernel", "gett_h16w4q_kernel", "gett_h16w4m_kernel", "gett_h16w4m4_kernel") and d["rag"] == 1, d
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=3e-2)


def test_sweep_ragged_k_three_modes_and_batch(env):
    """Three contracted modes (the odometer's carry past digit 1 recomputes the bases from the full index), a batch mode, beta != 0 in
    fp16, and the headline equation at extents of 40 and 96."""
    got, ref, d = _run(env, dict(m=264, n=136, k=72, j=3, i=5), "kmji", "kijn", "mn", seed=41)
    assert d["family"] == 1 and d["rag"] == 1, d
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=3e-2)
    got, ref, d = _run(env, dict(m=136, n=264, k=40, j=3, l=3), "kmjl", "kjnl", "mnl", dtype_name="float16", alpha=0.5, beta=0.25, seed=42)
    assert d["family"] == 1 and d["rag"] == 1, d
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-2)
    for e in (40, 96):
        got, ref, d = _run(env, dict(a=96, b=8, c=6, d=e, e=104), "abcd", "dcbe", "ae", seed=43 + e)
        assert d["family"] == 1 and d["rag"] == 1, d
        np.testing.assert_allclose(got, ref, rtol=8e-3, atol=5e-2)


def test_sweep_ragged_k_forced_kernels(built):
    """Every kernel that carries the sweep mask (the 128 x 128 pair switches it per operand: its deep ring stages a tile's A and B pieces at
    different times), forced in a child process, on 1 .. 4 K-tiles per sweep and sweeps that end in the
    prologue, in the unrolled loop and in its tail; large enough for several rounds of workgroups."""
    import os, subprocess, sys
    code = r'''
import numpy as np, torch
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
g = torch.Generator(device="cuda"); g.manual_seed(6)
for (mA, mB) in (("kmj", "kjn"), ("mjk", "kjn"), ("kmj", "nkj"), ("mjk", "nkj"), ("mkj", "njk")):   # (the last one: j is the fastest contracted mode)
    for (m, n, k, j) in ((384, 384, 40, 1 + 2), (256, 128, 8, 7), (640, 384, 136, 2), (384, 640, 72, 5), (128, 128, 200, 2), (264, 72, 24, 9), (2048, 1024, 96, 3),
                         (384, 264, 48, 5), (264, 136, 104, 3)):
        ext = dict(m=m, n=n, k=k, j=j)
        eA, eB = [ext[c] for c in mA], [ext[c] for c in mB]
        A = (torch.rand(eA[::-1], generator=g, device="cuda") * 2 - 1).bfloat16()
        B = (torch.rand(eB[::-1], generator=g, device="cuda") * 2 - 1).bfloat16()
        D = torch.full((n, m), float("nan"), dtype=torch.bfloat16, device="cuda")
        plan = ops.contraction_plan(h, eA, mA, eB, mB, [m, n], "mn", dtype=ct.R_16BF, workspace_limit=1 << 28)
        d = plan.describe()
        assert d["family"] == 1 and d["kname"] == WANT and d["rag"] == 1, d
        ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
        plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
        torch.cuda.synchronize()
        ref = torch.einsum("%s,%s->nm" % (mA[::-1], mB[::-1]), A.double(), B.double())
        np.testing.assert_allclose(D.double().cpu().numpy(), ref.cpu().numpy(), rtol=8e-3, atol=3e-2, err_msg=str((mA, mB, m, n, k, j)))
print("ok")
'''
    for waves, want in (("4x", "gett_h16w4x_kernel"), ("4q", "gett_h16w4q_kernel"), ("4m", "gett_h16w4m_kernel"), ("4m4", "gett_h16w4m4_kernel")):
        envv = dict(os.environ, CUTENSOR_AMD_H16_WAVES=waves)
        r = subprocess.run([sys.executable, "-c", code.replace("WANT", repr(want))], env=envv, capture_output=True, text=True, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (waves, r.stdout[-2000:], r.stderr[-3000:])


REPACKED = [  # (extents, modes of A, B, D, operands copied first)
    # A contiguous in k, B in j ('ijk,lkj->il' of a row-major front end): B is copied with k fastest, the contracted modes fuse
    (dict(i=1024, l=1024, j=16, k=72), "kji", "jkl", "li", (0, 1)),
    # the headline equation with d = 16: the sweep mask would keep 16 of every 64 k — B is copied in A's order of (d, c, b), K fuses
    (dict(a=2048, b=8, c=16, d=16, e=2048), "dcba", "ebcd", "ea", (0, 1)),
    # A contiguous in j, B in k, batched: A is copied, the contracted modes stay two (k whole K-tiles, then j)
    (dict(m=1024, n=1024, j=16, k=64, l=3), "jmkl", "knjl", "mnl", (1, 0)),
    # the reference's test equation 'mlik,lkjm->lij' (einsum_test.py:84-107), larger: A contiguous in k, B in m
    (dict(m=64, l=16, i=256, k=64, j=256), "kilm", "mjkl", "jil", (0, 1)),
    # the headline equation with d = 50: A's sweeps of d end in partial 16-byte units — A is copied with its FREE mode fastest (a plain
    # transpose), and two free-contiguous operands take any fastest contracted extent under the sweep mask
    (dict(a=2048, b=4, c=16, d=50, e=2048), "dcba", "ebcd", "ea", (1, 0)),
]


@pytest.mark.parametrize("case", range(len(REPACKED)))
def test_operands_the_lds_dma_kernels_cannot_stage_are_copied_first(env, case):
    """Round 6: an operand that is contiguous in a contracted mode the K order does not start with (or whose sweeps end in partial units)
    used to send the whole problem to the general family's 2-byte gathers (70 TFLOP/s where the vendor BLAS reaches 630).  Large problems
    now copy that operand into a packed temporary in the workspace first (cutensorPermute: bit-exact at alpha = 1) and contract the
    temporaries on the LDS-DMA kernels — same oracle, same tolerance."""
    import os
    ext, mA, mB, mD, (ra, rb) = REPACKED[case]
    got, ref, d = _run(env, ext, mA, mB, mD, seed=50 + case, expect_mfma=False)
    if not os.environ.get("CUTENSOR_AMD_H16_WAVES"):   # (a forced kernel keeps the operands where they are)
        assert (d.get("repack_A") or d.get("repack_B")) and d["family"] == 1, d    # (which operand: the cheaper copy by the model — (ra, rb) when this was written)
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=5e-2)
    # with beta != 0 in fp16 (C is not touched by the copies)
    got, ref, d = _run(env, ext, mA, mB, mD, dtype_name="float16", alpha=0.5, beta=0.25, seed=60 + case, expect_mfma=False)
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=3e-2)


def test_small_problems_keep_their_operands_in_place(env):
    """The copies cost a launch each: the reference's own test shape 'mlik,lkjm->lij' at extents of 50 stays on the general family, and so
    does whatever the LDS-DMA kernels take as it lies."""
    import os
    got, ref, d = _run(env, dict(m=50, l=50, i=50, k=50, j=50), "kilm", "mjkl", "jil", seed=70, expect_mfma=False)
    if not os.environ.get("CUTENSOR_AMD_H16_WAVES"):
        assert d["family"] == 2 and "repack_A" not in d, d
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=5e-2)
    # (... and so does the headline equation with d = 50 when it is small: sweeps that end in partial 16-byte units)
    got, ref, d = _run(env, dict(a=256, b=4, c=16, d=50, e=256), "dcba", "ebcd", "ea", seed=71, expect_mfma=False)
    if not os.environ.get("CUTENSOR_AMD_H16_WAVES"):
        assert d["family"] == 2 and "repack_A" not in d, d
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=5e-2)


def test_unaligned_shapes_stay_on_the_lds_dma_kernels(env):
    """Extents that admit neither 16-byte lanes nor 64-deep K-tiles (round 6): still the LDS-DMA kernels — 16-byte units at 2-byte
    addresses, the partial k-unit repaired in LDS (tests/test_gpu_h16_unaligned.py is that path's own file).  Two contracted modes with a
    ragged fastest one whose 16-byte units are partial (k = 50 beside a free extent of 37) remain the general MFMA family's
    (gett_gen.inc; tests/test_gpu_gen.py), never the scalar FMA kernel's."""
    got, ref, d = _run(env, dict(m=37, n=29, k=50), "mk", "kn", "mn", seed=9, expect_mfma=False)
    assert d["family"] == 1 and d["kname"] == "gett_h16w4q_kernel", d
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=2e-2)
    got, ref, d = _run(env, dict(m=37, n=29, k=50, j=3), "mkj", "jkn", "mn", seed=9, expect_mfma=False)   # (k, j do not fuse: B holds j first)
    assert d["family"] == 2 and d["kname"] == "gett_gen_kernel", d
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=2e-2)


def test_full_size_8192_sampled(env):
    """BASELINE configs[3] at full size: 8192^3 bf16; 4096 sampled outputs against fp64 dot products of the
    rounded inputs, plus linearity in alpha (size-independent property)."""
    torch, ct, ops, h = env
    n = 8192
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    A = (torch.rand((n, n), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)   # A[m,k], m fastest: A[k][m] row-major
    B = (torch.rand((n, n), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)   # B[k,n], k fastest: B[n][k] row-major
    D1 = torch.empty((n, n), device="cuda", dtype=torch.bfloat16)                     # C[m,n], m fastest: C[n][m]
    D2 = torch.empty_like(D1)
    plan = ops.contraction_plan(h, [n, n], "mk", [n, n], "kn", [n, n], "mn", dtype=ct.R_16BF)
    assert plan.describe()["family"] == 1
    plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D1.data_ptr(), D1.data_ptr())
    plan.contract(2.0, A.data_ptr(), B.data_ptr(), 0.0, D2.data_ptr(), D2.data_ptr())
    torch.cuda.synchronize()
    rng = np.random.default_rng(12)
    ms, ns = rng.integers(0, n, 4096), rng.integers(0, n, 4096)
    mi, ni = torch.from_numpy(ms).cuda(), torch.from_numpy(ns).cuda()
    ref = (A[:, mi].double() * B[ni, :].double().t()).sum(dim=0).cpu().numpy()       # sum_k A[k][m] * B[n][k]
    got = D1[ni, mi].double().cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=0.15)     # |sum of 8192 U(-1,1)^2 terms| ~ 30; atol covers cancellation
    np.testing.assert_array_equal((D1.float() * 2).cpu().numpy(), D2.float().cpu().numpy())   # exact: scaling by 2 commutes with rounding


@pytest.mark.parametrize("dims", [(96, 96, 4096), (264, 120, 1536), (512, 256, 8192)])
def test_split_k_for_small_outputs(env, dims):
    """Few output tiles and a long K: the 16-bit kernel splits K over the idle CUs (fp32 partials in the workspace,
    folded with one rounding to the 16-bit type)."""
    torch, ct, ops, h = env
    m, n, k = dims
    tdt = torch.bfloat16
    g = torch.Generator(device="cuda")
    g.manual_seed(21)
    A = (torch.rand((k, m), generator=g, device="cuda") * 2 - 1).to(tdt)      # A[m,k] column-major
    B = (torch.rand((n, k), generator=g, device="cuda") * 2 - 1).to(tdt)      # B[k,n] column-major
    C = (torch.rand((n, m), generator=g, device="cuda") * 2 - 1).to(tdt)
    D = torch.empty_like(C)
    plan = ops.contraction_plan(h, [m, k], "mk", [k, n], "kn", [m, n], "mn", dtype=ct.R_16BF, workspace_limit=1 << 28)
    d = plan.describe()
    assert d["family"] == 1 and d["splitK"] > 1 and plan.required_workspace > 0, d
    ws = torch.empty(plan.required_workspace, dtype=torch.uint8, device="cuda")
    plan.contract(1.25, A.data_ptr(), B.data_ptr(), -0.5, C.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    ref = 1.25 * (B.double().cpu() @ A.double().cpu()) - 0.5 * C.double().cpu()     # [n][m] = sum_k B[n][k] A[k][m]
    np.testing.assert_allclose(D.double().cpu().numpy(), ref.numpy(), rtol=8e-3, atol=0.1)
    # without workspace the same plan falls back to one slice
    plan1 = ops.contraction_plan(h, [m, k], "mk", [k, n], "kn", [m, n], "mn", dtype=ct.R_16BF, workspace_limit=0)
    assert plan1.describe()["splitK"] == 1 and plan1.required_workspace == 0
    D1 = torch.empty_like(C)
    plan1.contract(1.25, A.data_ptr(), B.data_ptr(), -0.5, C.data_ptr(), D1.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_allclose(D1.double().cpu().numpy(), ref.numpy(), rtol=8e-3, atol=0.1)


def test_split_k_fp16(env):
    got, ref, d = _run(env, dict(m=96, n=120, k=2048), "km", "nk", "mn", dtype_name="float16", alpha=0.5, beta=1.0, seed=25)
    assert d["family"] == 1 and d["splitK"] > 1, d
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-2)


def test_headline_einsum_shape_in_bf16(env):
    """'abcd,dcbe->ae' (einsum.cu helper view) with 16-bit data: one 96 x 96 output tile, K = b*c*d split over the CUs."""
    got, ref, d = _run(env, dict(a=96, b=16, c=16, d=64, e=96), "dcba", "ebcd", "ea", seed=23)
    assert d["splitK"] > 1, d
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=0.25)


@pytest.mark.parametrize("variant", ["4", "s", "p", "4s", "4r", "4v", "4x", "4p", "4m", "4m4", "8m", "4q"])
def test_kernel_variant_parity(built, variant):
    """The other 16-bit kernel variants (CUTENSOR_AMD_H16_WAVES=4: gett_h16w4_kernel, one wave per SIMD, 128 x 128 per
    wave; =s: gett_h16s_kernel, eight free-running waves, K-tile of 32, five-deep LDS ring, one barrier per K-tile; =p: gett_h16_kernel, two wave rows
    alternated by barriers; =4m: gett_h16w4m_kernel, the 128 x 128 mid-size sibling of the default kernel, two workgroups per CU; =4m4: the same on a four-deep K-tile ring, one workgroup per CU; =8m: that tile with four data-moving waves; =4q: gett_h16w4q_kernel, the 64 x 64 tile for small problems; =4s: gett_h16w4s_kernel, four waves with 128 x 128 wave tiles on the K-tile-32 ring; =4r: gett_h16w4r_kernel, four waves register-staged; =4v: gett_h16w4v_kernel, the four-wave kernel with the lean instruction stream of gett_h16v.hip) must give the same answers: the GEMM-like, multi-mode and split-K cases of this file in a
    child process that plans with the variant (the planner reads the switch once per process)."""
    import os
    import subprocess
    import sys
    from cudalibrarysamples_amd import cutensor as ct
    if variant in ("4", "s", "p", "4s", "4r", "4v") and not ct.lib.ctamdResearchKernelsBuilt():
        pytest.skip("a retired kernel family: compiled by make RESEARCH=1 only (round 5)")
    env = dict(os.environ, CUTENSOR_AMD_H16_WAVES=variant)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "gemm_like or multi_mode or split_k or alpha_beta", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("mA", ["km", "mk"])
def test_bf16_operand_larger_than_4_gib(built, mA):
    """Maximum sizes: a 5.2-GB bf16 operand stays on the MFMA kernel — its 32-bit lane offsets are relative to a 64-bit
    base that moves with the workgroup tile, the wave and the K-tile.  fp64 dot products of sampled outputs."""
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    h = ops.Handle()
    M, N, K = 2048, 256, 5 << 18
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    A = (torch.rand((M, K) if mA == "km" else (K, M), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
    B = (torch.rand((N, K), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)      # "kn"
    assert A.numel() * 2 > (1 << 32)
    C = torch.empty((N, M), device="cuda", dtype=torch.bfloat16)                          # "mn"
    p = ops.contraction_plan(h, [K, M] if mA == "km" else [M, K], mA, [K, N], "kn", [M, N], "mn", dtype=ct.R_16BF,
                             workspace_limit=1 << 30)
    d = p.describe()
    assert d["kname"] in ("gett_h16_kernel", "gett_h16w4_kernel", "gett_h16s_kernel", "gett_h16w4s_kernel", "gett_h16w4r_kernel", "gett_h16w4v_kernel", "gett_h16w4x_kernel", "gett_h16w4m_kernel", "gett_h16w4m4_kernel", "gett_h16w8m_kernel", "gett_h16w4q_kernel", "gett_h16w4p_kernel"), d
    ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
    p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), p.required_workspace)
    torch.cuda.synchronize()
    rows = [0, 1, 127, 128, 1000, 1023, 1024, M - 1]        # first / last rows of several 256-row tiles
    Arows = (A[rows, :] if mA == "km" else A[:, rows].t()).double()
    ref = B.double() @ Arows.t()
    got = C[:, rows].double()
    err = float((got - ref).abs().max())
    assert err <= 8e-3 * float(ref.abs().max()) + 1e-2, (err, float(ref.abs().max()), d)

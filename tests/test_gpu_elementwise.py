"""GPU parity: cutensorPermute / cutensorReduce / ElementwiseBinary through the C ABI vs the oracle.
Permutations are bit-exact for alpha = 1 (pure data movement); reductions rtol 1e-5."""
import numpy as np
import pytest

import oracle
from util import assert_close, from_device, make_tensor, to_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import torch
    assert torch.cuda.is_available()
    from cudalibrarysamples_amd import cutensor as ct, ops
    return ct, ops, ops.Handle(), torch


@pytest.mark.parametrize("case", [
    (dict(w=32, h=16, c=64, n=8), "whcn", "cwhn", 0),       # elementwise_permute.cu:51-63 shrunk -> TRANSPOSE
    (dict(a=128, b=20, c=68), "abc", "cab", 0),            # config 3 permute A[a,b,c] -> C[c,a,b]
    (dict(a=72, b=36, c=132), "abc", "cba", 0),            # full reversal, partial 64-tiles
    (dict(a=256, b=12, c=10), "abc", "acb", 1),            # shared fastest mode -> ROWCOPY
    (dict(a=4096), "a", "a", 1),                           # 1-D copy
    (dict(a=33, b=17, c=5), "abc", "cab", 3),              # odd extents, the whole tensor one contiguous block on both sides -> BLOCK (round 6)
    (dict(a=33, b=170, c=7), "abc", "acb", 2),             # odd extents, shared fastest mode, more than a block's 32 KiB -> GENERIC
    (dict(d=50, c=16, b=4, a=40), "dcba", "bcda", 3),      # the copy in front of 'abcd,dcbe->ae' with d = 50: blocks of 3200, permuted inside
    (dict(a=6, b=5, c=4, d=300), "abcd", "cbad", 3),       # small blocks: several per workgroup
    (dict(a=7), "a", "a", 2),
    (dict(a=1, b=64, c=64), "abc", "cba", 0),              # extent-1 mode dropped
])
@pytest.mark.parametrize("lanes_only", [True, False])
def test_permutation(env, case, lanes_only):
    """lanes_only: CUTENSOR_AMD_EW_ANY=0 (hooks flavour) keeps transpositions of small tensors on the 16-byte-lane kernels — since round 6
    the planner hands pure permutations below ~4096 wide tiles to the element-wise 64 x 64 transposer (variant 4), which is faster there;
    both must be exact."""
    import os
    ct, ops, h, torch = env
    ext, mA, mB, variant = case
    eA, eB = [ext[c] for c in mA], [ext[c] for c in mB]
    A = make_tensor(eA, 11)
    if lanes_only:
        os.environ["CUTENSOR_AMD_EW_ANY"] = "0"
    try:
        p = ops.permutation_plan(h, eA, mA, eB, mB)
    finally:
        os.environ.pop("CUTENSOR_AMD_EW_ANY", None)
    assert p.describe()["variant"] == variant or (not lanes_only and variant == 0 and p.describe()["variant"] == 4), p.describe()
    dA = to_device(A)
    dB = torch.zeros(int(np.prod(eB)), dtype=torch.float32, device="cuda")
    for alpha in (1.0, 1.5):
        p.permute(alpha, dA.data_ptr(), dB.data_ptr(), 0)
        torch.cuda.synchronize()
        ref = np.zeros(eB, dtype=np.float32, order="F")
        oracle.permute(A, mA, ref, mB, alpha=alpha)
        got = np.reshape(dB.cpu().numpy(), eB, order="F")
        if alpha == 1.0:
            assert np.array_equal(got, ref), (mA, mB)
        else:
            assert_close(got, ref, rtol=1e-6, what="permute")


@pytest.mark.parametrize("dtype_name", ["bfloat16", "float16", "float32"])
def test_block_permutation_bit_exact_at_alpha_one(env, dtype_name):
    """EW_BLOCK (elementwise.hip ew_block_kernel, round 6): the leading modes of D are the same packed set as the leading modes of A — blocks
    that are contiguous on both sides, permuted inside through LDS.  Against torch's permute on the same bits (alpha = 1: pure data
    movement), with a strided outer mode on both sides, and alpha = -0.5 (exact in every type)."""
    ct, ops, h, torch = env
    tdt = getattr(torch, dtype_name)
    cdt = {"bfloat16": ct.R_16BF, "float16": ct.R_16F, "float32": ct.R_32F}[dtype_name]
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    for (d, c, b, a, padA, padD) in ((50, 16, 4, 300, 0, 0), (7, 3, 5, 1000, 0, 0), (24, 10, 6, 77, 5, 9), (128, 8, 4, 64, 0, 0)):
        # A[d, c, b, a] (d fastest) -> D[b, c, d, a]; with padA / padD the outer mode a has a longer pitch than the block
        blk = d * c * b
        bufA = (torch.rand((a, blk + padA), generator=g, device="cuda") * 2 - 1).to(tdt)
        bufD = torch.full((a, blk + padD), 7.0, device="cuda", dtype=tdt)
        sA = [1, d, d * c, blk + padA]
        sD = [1, b, b * c, blk + padD]
        p = ops.permutation_plan(h, [d, c, b, a], "dcba", [b, c, d, a], "bcda", dtype=cdt, strideA=sA, strideB=sD)
        assert p.describe()["variant"] == 3, p.describe()
        viewA = bufA[:, :blk].reshape(a, b, c, d)                      # row-major view: a slowest, d fastest
        want = viewA.permute(0, 3, 2, 1).reshape(a, blk)               # [a][d][c][b]: b fastest
        p.permute(1.0, bufA.data_ptr(), bufD.data_ptr(), 0)
        torch.cuda.synchronize()
        assert torch.equal(bufD[:, :blk], want), (dtype_name, d, c, b, a)
        if padD:
            assert bool((bufD[:, blk:] == 7.0).all())                  # the padding of D is not touched
        p.permute(-0.5, bufA.data_ptr(), bufD.data_ptr(), 0)
        torch.cuda.synchronize()
        assert torch.equal(bufD[:, :blk], (want.float() * -0.5).to(tdt)), (dtype_name, d, c, b, a)
        p.destroy()


@pytest.mark.parametrize("case", [
    (dict(m=196, h=16, k=8, v=12), "mhkv", "mv", 0),        # reduction.cu:49-61 shrunk -> RED_COL (+split)
    (dict(a=64, b=40, c=24), "abc", "ac", 0),               # config 3 reduce C[a,c] = sum_b
    (dict(a=64, b=40, c=24), "abc", "c", 1),                # C[c] = sum_{a,b} -> RED_ROW
    (dict(a=4096, b=6), "ab", "b", 1),
    (dict(a=33, b=7, c=5), "abc", "ac", 2),                 # odd -> GENERIC
    (dict(a=64, b=48), "ab", "", 1),                        # full reduction to a scalar
    (dict(n=5, i=4, j=6), "jin", "ij", 2),                  # einsum.cu:451 "nij->ji" reversed
    # round 6: odd extents with A's stride-1 mode REDUCED -> the element-gather variant's wave-per-kept-element form (reduce_row_any_kernel)
    (dict(a=77, b=5, c=3), "abc", "bc", 2),                 # 77 reduced elements per kept one: two passes of a wave, ragged
    (dict(a=4099, b=3), "ab", "b", 2),                      # three kept elements: the reduced range is split, partials + finalize
    (dict(a=131, b=9, c=7), "abc", "c", 2),                 # two reduced modes (a, b), the walk crosses the mode boundary
    (dict(a=131, b=67, c=5), "abc", "ac", 2),               # stride-1 mode kept, 67 reduced rows: the unrolled loop and its tails
])
def test_reduction(env, case):
    ct, ops, h, torch = env
    ext, mA, mC, variant = case
    eA, eC = [ext[c] for c in mA], [ext[c] for c in mC]
    A, C = make_tensor(eA, 21), make_tensor(eC, 22)
    p = ops.reduction_plan(h, eA, mA, eC, mC)
    assert p.describe()["variant"] == variant, p.describe()
    assert p.required_workspace <= p.workspace_estimate
    dA, dC = to_device(A), to_device(C)
    ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
    for alpha, beta in ((1.1, 0.0), (0.5, 2.0)):
        dD = dC.clone()
        p.reduce(alpha, dA.data_ptr(), beta, dD.data_ptr(), dD.data_ptr(), ws.data_ptr(), p.required_workspace, 0)
        torch.cuda.synchronize()
        ref = np.zeros_like(C)
        oracle.reduce(A, mA, ref, mC, alpha=alpha, beta=beta, C=C)
        assert_close(from_device(dD, C), ref, rtol=1e-5, atol=1e-5 * float(np.abs(ref).max()), what="reduce %s->%s" % (mA, mC))


def test_reduction_operators(env):
    ct, ops, h, torch = env
    eA, eC = [32, 24, 8], [32, 8]
    A = make_tensor(eA, 31, lo=0.5, hi=1.5)
    dA = to_device(A)
    for op in (ct.OP_MAX, ct.OP_MIN):
        p = ops.reduction_plan(h, eA, "abc", eC, "ac", op_reduce=op)
        dD = torch.zeros(int(np.prod(eC)), dtype=torch.float32, device="cuda")
        ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
        p.reduce(1.0, dA.data_ptr(), 0.0, dD.data_ptr(), dD.data_ptr(), ws.data_ptr(), p.required_workspace, 0)
        torch.cuda.synchronize()
        ref = np.zeros(eC, dtype=np.float32, order="F")
        oracle.reduce(A, "abc", ref, "ac", op=op)
        assert np.array_equal(np.reshape(dD.cpu().numpy(), eC, order="F"), ref)


def test_reduce_as_permutation_with_beta(env):
    """einsum.cu:449-450 routes permutations through cutensorReduce; beta*C must be honoured."""
    ct, ops, h, torch = env
    eA, eC = [8, 64, 20], [20, 8, 64]
    A, C = make_tensor(eA, 41), make_tensor(eC, 42)
    p = ops.reduction_plan(h, eA, "abc", eC, "cab")
    dA, dC = to_device(A), to_device(C)
    p.reduce(2.0, dA.data_ptr(), 0.5, dC.data_ptr(), dC.data_ptr(), 0, 0, 0)
    torch.cuda.synchronize()
    ref = np.zeros_like(C)
    oracle.permute(A, "abc", ref, "cab", alpha=2.0, C=C, gamma=0.5)
    assert_close(from_device(dC, C), ref, rtol=1e-6, what="reduce-as-permute")


def test_max_min_reduction_over_unit_extents_still_adds_beta_c(env):
    """A MAX / MIN reduction whose reduced modes all have extent 1 degenerates to a permutation; D must stay
    alpha * A + beta * C — the reduction operator must not turn into the operator that joins A and C
    (found by tools/fuzz_elementwise.py)."""
    ct, ops, h, torch = env
    eA, eC = [5, 1, 16, 12], [12, 5, 16]
    A, C = make_tensor(eA, 51, lo=-1.0, hi=1.0), make_tensor(eC, 52, lo=-1.0, hi=1.0)
    for op in (ct.OP_MAX, ct.OP_MIN, ct.OP_MUL):
        p = ops.reduction_plan(h, eA, "ebgh", eC, "heg", op_reduce=op)
        dA, dD = to_device(A), to_device(C)
        p.reduce(-1.25, dA.data_ptr(), -0.5, dD.data_ptr(), dD.data_ptr(), 0, 0, 0)
        torch.cuda.synchronize()
        ref = -1.25 * np.transpose(A[:, 0, :, :], (2, 0, 1)) - 0.5 * C
        assert_close(from_device(dD, C), ref, rtol=1e-6, atol=1e-6, what="unit-extent reduction, op %d" % op)


def test_large_2d_transpose_roundtrip(env):
    """Size-independent property at a large size: transposing twice is the identity (bit-exact), and
    a checksum of the transposed tensor equals the checksum of the input."""
    ct, ops, h, torch = env
    n = 4096
    a = torch.rand(n * n, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    c = torch.empty_like(a)
    p = ops.permutation_plan(h, [n, n], "ab", [n, n], "ba")
    p.permute(1.0, a.data_ptr(), b.data_ptr(), 0)
    p.permute(1.0, b.data_ptr(), c.data_ptr(), 0)
    torch.cuda.synchronize()
    assert torch.equal(a, c)
    assert torch.equal(b.view(n, n), a.view(n, n).t().contiguous().view(n, n))


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_16bit_vector_permutes_are_exact(built, dtype):
    """bf16 / fp16 permutations through the 8-element-lane kernels (transposing and row-copy variants) and the generic
    fallback: bit-exact for alpha = 1; binary add (alpha A + gamma C, fp32 arithmetic, one rounding) within 1 ulp."""
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    h = ops.Handle()
    tdt = getattr(torch, dtype)
    cdt = ct.R_16BF if dtype == "bfloat16" else ct.R_16F
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    cases = [
        (dict(a=136, b=24, c=72), "abc", "cba", 0),     # full reversal: transposing variant
        (dict(a=128, b=40, c=64), "abc", "acb", 1),     # shared fastest mode: row copy
        (dict(a=136, b=24, c=72), "abc", "cab", 0),
        (dict(a=37, b=5, c=11), "abc", "cba", 3),       # odd extents, the whole tensor one block that is contiguous on both sides (round 6)
        (dict(a=37, b=50, c=11), "abc", "cba", 2),      # odd extents, more than a block's 32 KiB: generic
        (dict(a=130, b=3, c=134), "abc", "cba", 4),     # even extents without 16-byte lanes: the element-wise transposer's PAIR form (128 x 128 tiles, ragged)
        (dict(a=258, b=2, c=132), "abc", "cab", 4),     # ... with a fused outer mode
        (dict(a=131, b=3, c=134), "abc", "cba", 4),     # an odd extent: element by element
        # full tiles of the wide transposing kernel (ew_transpose_h16_wide_kernel<T0, T1>): 256 x 128, 128 x 128, 256 x 64, 128 x 64
        (dict(a=256, b=3, c=512), "abc", "cba", 0),
        (dict(a=128, b=5, c=384), "abc", "cab", 0),
        (dict(a=192, b=2, c=256), "abc", "cba", 0),
        (dict(a=64, b=7, c=128), "abc", "cab", 0),
        (dict(a=128, b=80, c=256), "abc", "cba", 0),    # rest >= 64: still order 0 (rows < 1 MiB apart), many tiles
    ]
    wide_seen = set()
    import os
    for ext, mA, mD, want in cases + [(e, a, d_, 4) for (e, a, d_, w) in cases if w == 0][:4]:   # (want 4: the planner's own choice for small transpositions since round 6)
        eA, eD = [ext[c] for c in mA], [ext[c] for c in mD]
        A = (torch.rand(eA[::-1], generator=g, device="cuda") * 2 - 1).to(tdt)      # column-major tensor == reversed row-major
        D = torch.zeros(eD[::-1], device="cuda", dtype=tdt)
        if want != 4:
            os.environ["CUTENSOR_AMD_EW_ANY"] = "0"                                 # hooks flavour: the 16-byte-lane kernels also on small tensors
        try:
            plan = ops.permutation_plan(h, eA, mA, eD, mD, dtype=cdt)
        finally:
            os.environ.pop("CUTENSOR_AMD_EW_ANY", None)
        assert plan.describe()["variant"] == want or (want == 4 and plan.describe()["variant"] == 0), (plan.describe(), ext, mA, mD)
        if plan.describe().get("tile0", 64) > 64 and want == 0:
            wide_seen.add(plan.describe()["tile0"])
        plan.permute(1.0, A.data_ptr(), D.data_ptr())
        torch.cuda.synchronize()
        ref = torch.einsum("%s->%s" % (mA[::-1], mD[::-1]), A)
        assert torch.equal(D, ref), (ext, mA, mD)
        # binary: D = 0.5 * perm(A) + 2 * C
        C = (torch.rand(eD[::-1], generator=g, device="cuda") * 2 - 1).to(tdt)
        bp = ops.binary_plan(h, eA, mA, eD, mD, dtype=cdt)
        bp.binary(0.5, A.data_ptr(), 2.0, C.data_ptr(), D.data_ptr())
        torch.cuda.synchronize()
        refb = (0.5 * ref.float() + 2.0 * C.float()).to(tdt)
        torch.testing.assert_close(D.float(), refb.float(), rtol=1e-2 if dtype == "bfloat16" else 2e-3, atol=1e-2)
    assert wide_seen == {128, 256}, wide_seen


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_full_reversal_takes_the_rest_first_per_xcd_tile_order(env, dtype):
    """A[a,b,c] -> C[c,b,a] at 512^3 (bf16: 1024 x 512 x 1024): both the rows A is read by (pitch a*b) and the rows D is written by (pitch c*b) lie 1 MiB
    apart, so the planner walks the tiles rest-first with one contiguous eighth of the id space per XCD (Ew2DParams::order = 1;
    elementwise.hip ordered_tile).  Bit-exact against torch, fp32 and bf16 (wide 16-bit tiles), with a ragged variant whose
    id space does not divide by eight."""
    ct, ops, h, torch = env
    tdt = getattr(torch, dtype)
    cdt = ct.R_32F if dtype == "float32" else ct.R_16BF
    wide = 1 if dtype == "float32" else 2        # 16-bit data: twice the extents for the same 1-MiB pitches
    for n_a, n_b, n_c in ((512 * wide, 512, 512 * wide), (512 * wide, 515, 512 * wide)):
        A = torch.rand((n_c, n_b, n_a), device="cuda").to(tdt)                 # column-major modes a, b, c
        D = torch.empty((n_a, n_b, n_c), device="cuda", dtype=tdt)            # column-major modes c, b, a
        p = ops.permutation_plan(h, [n_a, n_b, n_c], "abc", [n_c, n_b, n_a], "cba", dtype=cdt)
        d = p.describe()
        assert d["variant"] == 0 and d["order"] == 1 and d["tile0"] == 256, d
        p.permute(1.0, A.data_ptr(), D.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(D, A.permute(2, 1, 0).contiguous())
        p.destroy()
        del A, D


def test_permute_tensor_larger_than_4_gib(env):
    """Maximum sizes: a 5.4-GB matrix transposed and transposed back is bit-identical (64-bit element offsets in the
    tile kernels), and a strided sample of the intermediate matches the definition."""
    ct, ops, h, torch = env
    a, b = 1 << 15, 40960
    A = torch.rand((b, a), device="cuda", dtype=torch.float32)        # column-major [a, b]: a contiguous
    assert A.numel() * 4 > (1 << 32)
    T = torch.empty((a, b), device="cuda", dtype=torch.float32)       # column-major [b, a]
    back = torch.empty_like(A)
    p1 = ops.permutation_plan(h, [a, b], "ab", [b, a], "ba")
    p2 = ops.permutation_plan(h, [b, a], "ba", [a, b], "ab")
    p1.permute(1.0, A.data_ptr(), T.data_ptr())
    p2.permute(1.0, T.data_ptr(), back.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(back, A)
    assert torch.equal(T[::4099, ::977], A.t()[::4099, ::977])


def test_reduce_tensor_larger_than_4_gib(env):
    """Maximum sizes: row sums and column sums of a 5.4-GB matrix (both reduction kernels walk 64-bit offsets)."""
    ct, ops, h, torch = env
    a, b = 1 << 15, 40960
    A = torch.rand((b, a), device="cuda", dtype=torch.float32)        # column-major [a, b]
    for kept, ext, dim in (("a", [a], 0), ("b", [b], 1)):
        p = ops.reduction_plan(h, [a, b], "ab", ext, kept)
        D = torch.zeros(ext[0], device="cuda", dtype=torch.float32)
        ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
        p.reduce(1.0, A.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace)
        torch.cuda.synchronize()
        ref = A.sum(dim=dim, dtype=torch.float64)
        rel = float(((D.double() - ref).abs() / ref).max())
        assert rel < 2e-5, (kept, rel, p.describe())


def test_bandwidth_config_at_true_size_2048_cubed(env):
    """BASELINE configs[2] at its true size: 2048^3 fp32 (32 GiB per tensor, generated on the device — SURVEY 8d: never
    mirror the samples' pinned-host copies), cutensorPermute abc->cab and cutensorReduce abc->ac (alpha 1.1, as
    reduction.cu:45).  Size-independent checks: 2^18 sampled elements of the permutation must be bit-exact gathers of
    the source, and the reduction must match fp64 sums over b at rtol 2e-4 (sums of 2048 terms ... 4M terms)."""
    ct, ops, h, torch = env
    n = 2048
    numel = n ** 3
    free, _ = torch.cuda.mem_get_info()
    if free < 2 * numel * 4 + (6 << 30):
        pytest.skip("needs two 32-GiB tensors in HBM (%d bytes free)" % free)
    A = torch.empty(numel, dtype=torch.float32, device="cuda")
    chunk = 1 << 28
    for s in range(0, numel, chunk):          # counter-based fill (fixed seed), in chunks
        idx = torch.arange(s, min(numel, s + chunk), device="cuda", dtype=torch.int64)
        A[s:s + idx.numel()] = ((idx * 2654435761 + 1234) % 16777216).to(torch.float32) / 16777216.0
        del idx
    D = torch.empty(numel, dtype=torch.float32, device="cuda")
    p = ops.permutation_plan(h, [n, n, n], "abc", [n, n, n], "cab")
    p.permute(1.0, A.data_ptr(), D.data_ptr())
    torch.cuda.synchronize()
    At = A.view(n, n, n)                      # At[c][b][a]: row-major view of the column-major tensor with modes a, b, c
    Dt = D.view(n, n, n)                      # Dt[b][a][c]: modes c, a, b
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    ia, ib, ic = (torch.randint(0, n, (1 << 18,), generator=g, device="cuda") for _ in range(3))
    assert int((Dt[ib, ia, ic] != At[ic, ib, ia]).sum().item()) == 0
    # corners and the last element (64-bit offsets)
    for a_, b_, c_ in ((0, 0, 0), (n - 1, n - 1, n - 1), (n - 1, 0, n - 1), (0, n - 1, 1)):
        assert float(Dt[b_, a_, c_]) == float(At[c_, b_, a_])
    p.destroy()
    del D, Dt
    R = torch.zeros(n * n, dtype=torch.float32, device="cuda")
    p = ops.reduction_plan(h, [n, n, n], "abc", [n, n], "ac", workspace_limit=1 << 30)
    ws = torch.empty(max(p.required_workspace, 256), dtype=torch.uint8, device="cuda")
    p.reduce(1.1, A.data_ptr(), 0.0, R.data_ptr(), R.data_ptr(), ws.data_ptr(), p.required_workspace)
    torch.cuda.synchronize()
    ref = At.sum(dim=1, dtype=torch.float64) * 1.1           # [c][a]
    got = R.view(n, n).double()                               # modes a, c -> row-major [c][a]
    rel = float(((got - ref).abs() / ref.abs().clamp_min(1e-30)).max())
    assert rel < 2e-4, (rel, p.describe())
    p.destroy()
    del A, At, R, ref, got
    torch.cuda.empty_cache()


# ---- complex reductions / permutations (round 5) -------------------------------------------------------------------------------
# What the reference's binding runs for a unary einsum on complex tensors: python/einsum.h:326-343 (cutensorCreateReduction, OP_ADD),
# :430-441 (cutensorReduce); torch/einsum.cc:83 dispatches the complex types; einsum.cu:346-372.  Against the oracle's complex entry
# points (pinned to numpy in tests/test_oracle.py).  complex64 rtol 2e-5 of the magnitude sum, complex128 1e-12.
def _cplx(ext, seed, np_dt):
    rng = np.random.default_rng(seed)
    n = int(np.prod(ext)) if len(ext) else 1
    flat = ((rng.random(n) * 2 - 1) + 1j * (rng.random(n) * 2 - 1)).astype(np_dt)
    return np.reshape(flat, tuple(ext), order="F") if len(ext) else flat.reshape(())


def _cdev(torch, arr):
    return torch.from_numpy(np.ascontiguousarray(arr.ravel(order="F")).copy()).cuda()


CPLX = {"complex64": (np.complex64, "C_32F", 2e-5), "complex128": (np.complex128, "C_64F", 1e-12)}


@pytest.mark.parametrize("dtype", sorted(CPLX))
@pytest.mark.parametrize("case", [
    (dict(a=33, b=17, c=5), "abc", "cab"),
    (dict(a=128, b=20, c=68), "abc", "cba"),
    (dict(i=50, j=64), "ij", "ji"),                # the unary einsum "ji->ij" of the reference's binding
    (dict(a=257), "a", "a"),
])
def test_complex_permutation(env, dtype, case):
    ct, ops, h, torch = env
    np_dt, cname, rtol = CPLX[dtype]
    ext, mA, mB = case
    eA, eB = [ext[c] for c in mA], [ext[c] for c in mB]
    A = _cplx(eA, 61, np_dt)
    dA = _cdev(torch, A)
    for alpha, conj in ((1.0, False), (0.75 - 1.5j, False), (1j, True)):
        p = ops.permutation_plan(h, eA, mA, eB, mB, dtype=getattr(ct, cname), opA=ct.OP_CONJ if conj else ct.OP_IDENTITY)
        dB = torch.zeros(int(np.prod(eB)), dtype=dA.dtype, device="cuda")
        p.permute(alpha, dA.data_ptr(), dB.data_ptr(), 0)
        torch.cuda.synchronize()
        ref = np.zeros(eB, dtype=np_dt, order="F")
        oracle.permute(A, mA, ref, mB, alpha=alpha, conjA=conj)
        got = np.reshape(dB.cpu().numpy(), eB, order="F")
        if alpha == 1.0:
            assert np.array_equal(got, ref), (mA, mB)          # pure data movement: bit-exact
        else:
            np.testing.assert_allclose(got, ref, rtol=rtol * 4, atol=rtol * 4)
        p.destroy()


@pytest.mark.parametrize("dtype", sorted(CPLX))
@pytest.mark.parametrize("case", [
    (dict(m=40, h=16, k=8, v=12), "mhkv", "mv"),
    (dict(a=64, b=40, c=24), "abc", "ac"),         # the unary einsum "cba->ca" of the reference's binding
    (dict(a=64, b=40, c=24), "abc", "c"),
    (dict(a=4096, b=6), "ab", "b"),                # few kept elements: the reduced range is split over workgroups + finalize
    (dict(a=64, b=48), "ab", ""),                  # full reduction to a scalar
])
def test_complex_reduction(env, dtype, case):
    ct, ops, h, torch = env
    np_dt, cname, rtol = CPLX[dtype]
    ext, mA, mC = case
    eA, eC = [ext[c] for c in mA], [ext[c] for c in mC]
    A, C = _cplx(eA, 71, np_dt), _cplx(eC, 72, np_dt)
    dA, dC = _cdev(torch, A), _cdev(torch, C)
    for alpha, beta, cA, cC in ((1.0, 0.0, False, False), (1.1 - 0.3j, 0.5j, False, False), (-1j, 2.0, True, True)):
        p = ops.reduction_plan(h, eA, mA, eC, mC, dtype=getattr(ct, cname), opA=ct.OP_CONJ if cA else ct.OP_IDENTITY,
                               opC=ct.OP_CONJ if cC else ct.OP_IDENTITY)
        assert p.required_workspace <= p.workspace_estimate
        ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
        dD = dC.clone()
        p.reduce(alpha, dA.data_ptr(), beta, dD.data_ptr(), dD.data_ptr(), ws.data_ptr(), p.required_workspace, 0)
        torch.cuda.synchronize()
        ref = np.zeros_like(C)
        oracle.reduce(A, mA, ref, mC, alpha=alpha, beta=beta, C=C, conjA=cA, conjC=cC)
        got = np.reshape(dD.cpu().numpy(), eC, order="F") if len(eC) else dD.cpu().numpy().reshape(())
        mag = abs(alpha) * np.abs(A).sum() / max(C.size, 1) + abs(beta) * np.abs(C).max() + 1.0
        np.testing.assert_allclose(got, ref, rtol=0, atol=rtol * mag * 4)
        p.destroy()


def test_complex_product_reduction_and_binary_forms(env):
    """OP_MUL over a short reduced mode, and cutensorElementwiseBinaryExecute D = alpha conj(perm(A)) (+ | *) gamma C on complex data;
    MAX / MIN are refused (not defined on complex numbers), the trinary element-wise form stays NOT_SUPPORTED."""
    ct, ops, h, torch = env
    np_dt = np.complex128
    A = _cplx([6, 5, 7], 81, np_dt)
    dA = _cdev(torch, A)
    p = ops.reduction_plan(h, [6, 5, 7], "abc", [6, 7], "ac", dtype=ct.C_64F, op_reduce=ct.OP_MUL)
    dD = torch.zeros(42, dtype=torch.complex128, device="cuda")
    p.reduce(1.0, dA.data_ptr(), 0.0, dD.data_ptr(), dD.data_ptr(), 0, 0, 0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(np.reshape(dD.cpu().numpy(), [6, 7], order="F"), A.prod(axis=1), rtol=1e-12, atol=1e-12)
    C = _cplx([7, 6, 5], 82, np_dt)
    for op, fn in (("ADD", lambda x, y: x + y), ("MUL", lambda x, y: x * y)):
        b = ops.binary_plan(h, [6, 5, 7], "abc", [7, 6, 5], "cab", op=op, dtype=ct.C_64F, opA=ct.OP_CONJ)
        dC = _cdev(torch, C)
        b.binary(0.5 + 1j, dA.data_ptr(), -2j, dC.data_ptr(), dC.data_ptr(), 0)
        torch.cuda.synchronize()
        ref = fn((0.5 + 1j) * np.conj(np.transpose(A, (2, 0, 1))), -2j * C)
        np.testing.assert_allclose(np.reshape(dC.cpu().numpy(), [7, 6, 5], order="F"), ref, rtol=1e-12, atol=1e-12)
    with pytest.raises(RuntimeError):
        ops.reduction_plan(h, [6, 5, 7], "abc", [6, 7], "ac", dtype=ct.C_64F, op_reduce=ct.OP_MAX)
    with pytest.raises(RuntimeError):
        ops.binary_plan(h, [6, 5, 7], "abc", [7, 6, 5], "cab", op="MAX", dtype=ct.C_32F)


def test_complex_unary_einsum_through_the_torch_front_end(env):
    """cudalibrarysamples_amd.torch_einsum on complex tensors with ONE operand: permutation and reduction, forward and (for the
    permutation) backward, against torch.einsum at the reference's tolerance (einsum_test.py:35-42)."""
    ct, ops, h, torch = env
    from cudalibrarysamples_amd import torch_einsum as te
    torch.manual_seed(0)
    for dt in (torch.complex64, torch.complex128):
        z = torch.randn(12, 50, 20, dtype=dt, device="cuda", requires_grad=True)
        for eq in ("ijk->kji", "ijk->ik", "ijk->", "ijk->j"):
            got = te.einsum(eq, z.detach())
            torch.testing.assert_close(got, torch.einsum(eq, z.detach()), rtol=5e-3, atol=6e-3)
        out = te.EinsumFunction.apply("ijk->kij", z)
        g = torch.randn_like(out)
        out.backward(g)
        zr = z.detach().clone().requires_grad_(True)
        torch.einsum("ijk->kij", zr).backward(g)
        torch.testing.assert_close(z.grad, zr.grad, rtol=5e-3, atol=6e-3)


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16", "float16"])
def test_binary_form_at_odd_extents_on_the_elementwise_transposer(env, dtype_name):
    """cutensorElementwiseBinaryExecute D = op(alpha perm(A), gamma C) (elementwise_binary.cu:149-153) where perm is a transposition and the
    extents / alignment admit no 16-byte lanes: ew_transpose_any_kernel<T, HASC> (round 6; the element-gather kernel before).  ADD / MUL /
    MAX against torch in fp32 arithmetic with ONE rounding to the data type (exact for fp32 ADD; within 1 ulp otherwise)."""
    ct, ops, h, torch = env
    tdt = getattr(torch, dtype_name)
    cdt = {"float32": ct.R_32F, "bfloat16": ct.R_16BF, "float16": ct.R_16F}[dtype_name]
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    for ext in (dict(a=77, b=5, c=131), dict(a=401, b=3, c=67), dict(a=64, b=2, c=65)):
        eA, eC = [ext[c] for c in "cba"], [ext[c] for c in "abc"]
        A = (torch.rand(eA[::-1], generator=g, device="cuda") * 2 - 1).to(tdt)
        C = (torch.rand(eC[::-1], generator=g, device="cuda") * 2 - 1).to(tdt)
        for op, fn in (("ADD", lambda x, y: x + y), ("MUL", lambda x, y: x * y), ("MAX", torch.maximum)):
            D = torch.full_like(C, float("nan"))
            p = ops.binary_plan(h, eA, "cba", eC, "abc", op=op, dtype=cdt)
            assert p.describe()["variant"] == 4, p.describe()
            p.binary(1.5, A.data_ptr(), -0.75, C.data_ptr(), D.data_ptr())
            torch.cuda.synchronize()
            ref = fn(1.5 * torch.einsum("abc->cba", A.float()), -0.75 * C.float()).to(tdt)
            if dtype_name == "float32":
                assert torch.equal(D, ref), (ext, op)
            else:
                ulp = 2.0 ** -7 if dtype_name == "bfloat16" else 2.0 ** -10
                assert bool(((D.float() - ref.float()).abs() <= ulp * ref.float().abs() + 1e-30).all()), (ext, op)
            p.destroy()

"""GPU parity of the TILED bandwidth kernels for everything that is not fp32 (round 6; VERDICT r5 "Missing #3"): permutations / binary
operations on fp64, complex64 and complex128 (elementwise.hip: ew_transpose_wide_kernel, ew_rowcopy_wide_kernel) and reductions on fp64,
complex64, complex128, bf16 and fp16 (reduce.hip: reduce_col_wide_kernel, reduce_row_wide_kernel; 16-bit data accumulates in fp32).  The
reference's binding dispatches every unary einsum over these types (cuTENSOR/python/cutensor/torch/einsum.cc:83,159,215;
python/einsum.h:326-343,430-441; einsum.cu:36-41 double).

Against the oracle (oracle_permute_* / oracle_reduce_*): a permutation with alpha = 1 is pure data movement — bit-exact; scaled / combined
forms and reductions within a few ulp of the data type (fp64 1e-13, complex64 2e-5, complex128 1e-12 relative to the magnitude summed;
16-bit: the fp64 sum of the 16-bit inputs rounded once, rtol 2^-8 bf16 / 2^-11 fp16 + fp32 accumulation noise).  Every case asserts that the
tiled variant ran (plan description), and partial tiles, padded strides, conjugation, operators and the split-reduction path are covered."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import torch
    assert torch.cuda.is_available()
    from cudalibrarysamples_amd import cutensor as ct, ops
    return ct, ops, ops.Handle(), torch


DT = {  # numpy dtype, cutensor name, tolerance relative to the summed magnitude, complex?
    "float64": (np.float64, "R_64F", 1e-13, False),
    "complex64": (np.complex64, "C_32F", 2e-5, True),
    "complex128": (np.complex128, "C_64F", 1e-12, True),
}
EW_TRANSPOSE, EW_ROWCOPY, RED_COL, RED_ROW = 0, 1, 0, 1


def _rand(ext, seed, np_dt, cx):
    rng = np.random.default_rng(seed)
    n = int(np.prod(ext)) if len(ext) else 1
    flat = rng.random(n) * 2 - 1
    if cx:
        flat = flat + 1j * (rng.random(n) * 2 - 1)
    return np.reshape(flat.astype(np_dt), tuple(ext), order="F")


def _dev(torch, arr):
    return torch.from_numpy(np.ascontiguousarray(arr.ravel(order="F")).copy()).cuda()


PERMS = [  # extents, modes of A, modes of B, expected variant
    (dict(a=200, b=36, c=70), "abc", "cab", EW_TRANSPOSE),     # partial tiles on both tile modes
    (dict(a=256, b=64, c=96), "abc", "cba", EW_TRANSPOSE),     # full reversal
    (dict(a=130, b=258), "ab", "ba", EW_TRANSPOSE),            # 2-D transpose, ragged tiles
    (dict(a=200, b=36, c=70), "abc", "acb", EW_ROWCOPY),       # A and B share the stride-1 mode
    (dict(a=1026, b=4, c=6), "abc", "acb", EW_ROWCOPY),
]


@pytest.mark.parametrize("dtype", sorted(DT))
@pytest.mark.parametrize("case", PERMS, ids=["%s->%s" % (c[1], c[2]) + "x".join(str(v) for v in c[0].values()) for c in PERMS])
def test_tiled_permutations_and_binary_forms(env, dtype, case):
    ct, ops, h, torch = env
    np_dt, cname, rtol, cx = DT[dtype]
    ext, mA, mB, variant = case
    eA, eB = [ext[c] for c in mA], [ext[c] for c in mB]
    A = _rand(eA, 11, np_dt, cx)
    dA = _dev(torch, A)
    forms = [(1.0, False)] + ([(0.75 - 1.5j, False), (1j, True)] if cx else [(-2.5, False)])
    for alpha, conj in forms:
        p = ops.permutation_plan(h, eA, mA, eB, mB, dtype=getattr(ct, cname), opA=ct.OP_CONJ if conj else ct.OP_IDENTITY)
        assert p.describe()["variant"] == variant, p.describe()
        dB = torch.zeros(int(np.prod(eB)), dtype=dA.dtype, device="cuda")
        p.permute(alpha, dA.data_ptr(), dB.data_ptr(), 0)
        torch.cuda.synchronize()
        ref = np.zeros(eB, dtype=np_dt, order="F")
        oracle.permute(A, mA, ref, mB, alpha=alpha, conjA=conj)
        got = np.reshape(dB.cpu().numpy(), eB, order="F")
        if alpha == 1.0:
            assert np.array_equal(got, ref), (dtype, mA, mB)       # pure data movement: bit-exact
        else:
            np.testing.assert_allclose(got, ref, rtol=rtol * 4, atol=rtol * 4)
        p.destroy()
    # D = op(alpha perm(A), gamma C): cutensorElementwiseBinaryExecute (elementwise_binary.cu:149-153,202-205)
    C = _rand(eB, 12, np_dt, cx)
    for op, fn in (("ADD", lambda x, y: x + y), ("MUL", lambda x, y: x * y)) + (() if cx else (("MAX", np.maximum),)):
        b = ops.binary_plan(h, eA, mA, eB, mB, op=op, dtype=getattr(ct, cname))
        assert b.describe()["variant"] == variant, b.describe()
        dC = _dev(torch, C)
        al, ga = (0.5 + 1j, -2j) if cx else (0.5, -2.0)
        b.binary(al, dA.data_ptr(), ga, dC.data_ptr(), dC.data_ptr(), 0)
        torch.cuda.synchronize()
        perm = np.transpose(A, [mA.index(c) for c in mB])
        np.testing.assert_allclose(np.reshape(dC.cpu().numpy(), eB, order="F"), fn(al * perm, ga * C), rtol=rtol * 8, atol=rtol * 8)
        b.destroy()


def test_padded_strides_keep_the_tiled_kernels_and_leave_the_padding_alone(env):
    ct, ops, h, torch = env
    for dtype in ("float64", "complex64"):
        np_dt, cname, rtol, cx = DT[dtype]
        a, b = 130, 66
        pa, pb = a + 6, b + 2                        # padded pitches (multiples of 2 elements: 16-byte lanes)
        A = _rand([pa, b], 21, np_dt, cx)
        dA = _dev(torch, A)
        dB = torch.full((pb * a,), 7.0, dtype=dA.dtype, device="cuda")
        p = ops.permutation_plan(h, [a, b], "ab", [b, a], "ba", dtype=getattr(ct, cname), strideA=[1, pa], strideB=[1, pb])
        assert p.describe()["variant"] == EW_TRANSPOSE, p.describe()
        p.permute(1.0, dA.data_ptr(), dB.data_ptr(), 0)
        torch.cuda.synchronize()
        got = np.reshape(dB.cpu().numpy(), [pb, a], order="F")
        assert np.array_equal(got[:b, :], A[:a, :].T)
        assert np.all(got[b:, :] == 7.0)
        p.destroy()


REDS = [  # extents, modes of A, kept modes, expected variant
    (dict(a=64, b=40, c=24), "abc", "ac", RED_COL),
    (dict(a=64, b=40, c=24), "abc", "c", RED_ROW),
    (dict(a=2048, b=6), "ab", "b", RED_ROW),         # few kept elements: split over workgroups + finalize
    (dict(a=16, b=3000), "ab", "a", RED_COL),        # few kept elements, long strided reduction: split
    (dict(m=40, h=16, k=8, v=12), "mhkv", "mv", RED_COL),
    (dict(a=64, b=48), "ab", "", RED_ROW),           # full reduction to a scalar
]


@pytest.mark.parametrize("dtype", sorted(DT))
@pytest.mark.parametrize("case", REDS, ids=["%s->%s" % (c[1], c[2]) for c in REDS])
def test_tiled_reductions(env, dtype, case):
    ct, ops, h, torch = env
    np_dt, cname, rtol, cx = DT[dtype]
    ext, mA, mC, variant = case
    eA, eC = [ext[c] for c in mA], [ext[c] for c in mC]
    A, C = _rand(eA, 31, np_dt, cx), _rand(eC, 32, np_dt, cx)
    dA, dC = _dev(torch, A), _dev(torch, C)
    forms = ((1.0, 0.0, False, False), (1.1 - 0.3j, 0.5j, False, False), (-1j, 2.0, True, True)) if cx else ((1.0, 0.0, False, False), (1.1, -0.5, False, False))
    for alpha, beta, cA, cC in forms:
        p = ops.reduction_plan(h, eA, mA, eC, mC, dtype=getattr(ct, cname), opA=ct.OP_CONJ if cA else ct.OP_IDENTITY,
                               opC=ct.OP_CONJ if cC else ct.OP_IDENTITY, workspace_limit=1 << 26)
        assert p.describe()["variant"] == variant, p.describe()
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
    if not cx:      # MAX / MIN / MUL on real data through the tiled kernels (exact: no rounding in MAX / MIN)
        for opname, fn in (("OP_MAX", np.max), ("OP_MIN", np.min)):
            p = ops.reduction_plan(h, eA, mA, eC, mC, dtype=getattr(ct, cname), op_reduce=getattr(ct, opname), workspace_limit=1 << 26)
            assert p.describe()["variant"] == variant
            ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
            dD = torch.zeros(max(int(np.prod(eC)), 1), dtype=dA.dtype, device="cuda")
            p.reduce(1.0, dA.data_ptr(), 0.0, dD.data_ptr(), dD.data_ptr(), ws.data_ptr(), p.required_workspace, 0)
            torch.cuda.synchronize()
            axes = tuple(i for i, c in enumerate(mA) if c not in mC)
            want = fn(A, axis=axes)
            got = np.reshape(dD.cpu().numpy(), eC, order="F") if len(eC) else dD.cpu().numpy().reshape(())
            assert np.array_equal(got, want), opname
            p.destroy()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("case", [
    (dict(a=64, b=40, c=24), "abc", "ac", RED_COL),
    (dict(a=64, b=40, c=24), "abc", "c", RED_ROW),
    (dict(a=4096, b=6), "ab", "b", RED_ROW),
    (dict(a=16, b=3000), "ab", "a", RED_COL),
    (dict(a=128, b=48), "ab", "", RED_ROW),
], ids=["ac", "c", "b-split", "a-split", "scalar"])
def test_16_bit_reductions_accumulate_in_fp32(env, dtype, case):
    ct, ops, h, torch = env
    ext, mA, mC, variant = case
    eA, eC = [ext[c] for c in mA], [ext[c] for c in mC]
    tdt = getattr(torch, dtype)
    cdt = ct.R_16BF if dtype == "bfloat16" else ct.R_16F
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    nA, nC = int(np.prod(eA)), max(int(np.prod(eC)), 1)
    dA = (torch.rand(nA, generator=g, device="cuda") * 2 - 1).to(tdt)
    dC = (torch.rand(nC, generator=g, device="cuda") * 2 - 1).to(tdt)
    A64 = np.reshape(dA.double().cpu().numpy(), eA, order="F")
    C64 = np.reshape(dC.double().cpu().numpy(), eC, order="F") if len(eC) else dC.double().cpu().numpy().reshape(())
    ulp = 2.0 ** -8 if dtype == "bfloat16" else 2.0 ** -11
    for alpha, beta in ((1.0, 0.0), (1.1, -0.5)):
        p = ops.reduction_plan(h, eA, mA, eC, mC, dtype=cdt, workspace_limit=1 << 26)
        assert p.describe()["variant"] == variant, p.describe()
        ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
        dD = dC.clone()
        p.reduce(alpha, dA.data_ptr(), beta, dD.data_ptr(), dD.data_ptr(), ws.data_ptr(), p.required_workspace, 0)
        torch.cuda.synchronize()
        ref = np.zeros_like(C64)
        oracle.reduce(A64, mA, ref, mC, alpha=alpha, beta=beta, C=C64)
        got = np.reshape(dD.double().cpu().numpy(), eC, order="F") if len(eC) else dD.double().cpu().numpy().reshape(())
        red = nA // nC
        np.testing.assert_allclose(got, ref, rtol=ulp, atol=ulp * 0.5 + 1e-6 * red)
        p.destroy()
    p = ops.reduction_plan(h, eA, mA, eC, mC, dtype=cdt, op_reduce=ct.OP_MAX, workspace_limit=1 << 26)
    ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
    dD = torch.zeros(nC, dtype=tdt, device="cuda")
    p.reduce(1.0, dA.data_ptr(), 0.0, dD.data_ptr(), dD.data_ptr(), ws.data_ptr(), p.required_workspace, 0)
    torch.cuda.synchronize()
    axes = tuple(i for i, c in enumerate(mA) if c not in mC)
    got = np.reshape(dD.double().cpu().numpy(), eC, order="F") if len(eC) else dD.double().cpu().numpy().reshape(())
    assert np.array_equal(got, A64.max(axis=axes))
    p.destroy()

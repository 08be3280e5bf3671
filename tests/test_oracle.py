"""CPU-only: the oracle against numpy/torch einsum, the reference's golden cases and the einsum.cu
demo equations (parity pinning of the checker itself)."""
import json
import os

import numpy as np
import pytest

import oracle
from util import make_tensor

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_contraction_default_modes_shrunk():
    # contraction.cu:46-59 with shrunk extents: full compare with numpy
    ext = dict(m=12, n=12, u=12, v=8, h=8, k=8)
    A = make_tensor([ext[c] for c in "mhkn"], 1)
    B = make_tensor([ext[c] for c in "ukvh"], 2)
    C = make_tensor([ext[c] for c in "munv"], 3)
    D = np.zeros_like(C)
    oracle.contract(A, "mhkn", B, "ukvh", D, "munv", alpha=1.1, beta=0.5, C=C)
    ref = 1.1 * np.einsum("mhkn,ukvh->munv", A.astype(np.float64), B.astype(np.float64)) + 0.5 * C
    np.testing.assert_allclose(D, ref, rtol=1e-6)
    # literal fp32 loop nest agrees to fp32 roundoff
    D32 = np.zeros_like(C)
    oracle.contract(A, "mhkn", B, "ukvh", D32, "munv", alpha=1.1, beta=0.5, C=C, acc64=False)
    np.testing.assert_allclose(D32, ref, rtol=2e-5)


def test_contraction_strided_views_and_aliasing():
    big = make_tensor([10, 9, 8], 4)
    A = big[1:9:2, :, 2:6]          # non-packed strides
    B = make_tensor([9, 4, 5], 5)
    D = make_tensor([4, 5], 6)
    ref = 2.0 * np.einsum("ijk,jkl->il", A.astype(np.float64), B.astype(np.float64)) - 1.0 * D
    oracle.contract(A, "ijk", B, "jkl", D, "il", alpha=2.0, beta=-1.0)   # C aliases D (contraction.cu:264)
    np.testing.assert_allclose(D, ref, rtol=1e-6)


def test_reduction_and_permutation():
    A = make_tensor([7, 5, 3, 4], 7)
    C = make_tensor([7, 4], 8)
    D = np.zeros_like(C)
    oracle.reduce(A, "mhkv", D, "mv", alpha=1.1, beta=2.0, C=C)       # reduction.cu:49
    np.testing.assert_allclose(D, 1.1 * A.astype(np.float64).sum(axis=(1, 2)) + 2.0 * C, rtol=1e-6)
    Dm = np.zeros_like(C)
    oracle.reduce(A, "mhkv", Dm, "mv", op=oracle.OP_MAX)
    np.testing.assert_allclose(Dm, A.max(axis=(1, 2)))
    P = np.zeros((3, 7, 5, 4), dtype=np.float32, order="F")
    oracle.permute(A, "whcn", P, "cwhn", alpha=1.5)                   # elementwise_permute.cu:51
    np.testing.assert_allclose(P, 1.5 * np.transpose(A, (2, 0, 1, 3)), rtol=1e-7)


def test_einsum_demo_equations():
    # einsum.cu:447-451 with the shapes used there; expected shapes from SURVEY 8c
    a = np.random.default_rng(0).random((2, 4, 5)).astype(np.float32)
    b = np.random.default_rng(1).random((4, 8, 7)).astype(np.float32)
    for eq, shape in [("ijn,jmk->inkm", [2, 5, 7, 8]), ("ijn,jmk", [2, 7, 8, 5])]:
        out = oracle.einsum(eq, a, b)
        assert list(out.shape) == shape
        np.testing.assert_allclose(out, np.einsum(eq, a, b), rtol=1e-5)
    for eq, shape in [("nij", [4, 5, 2]), ("nij->ijn", [4, 5, 2]), ("nij->ji", [5, 4])]:
        out = oracle.einsum(eq, a)
        assert list(out.shape) == shape
        np.testing.assert_allclose(out, np.einsum(eq, a), rtol=1e-5)


def test_einsum_unsupported_inputs():
    # einsum.cu:76-79 ("..."), :118-122 (rank mismatch), :123-127 (> max modes)
    assert oracle.einsum_parse("ab...,bc->ac", (2, 3, 4), (3, 4)) is None
    assert oracle.einsum_parse("ab,bc->ac", (2, 3, 4), (3, 4)) is None
    assert oracle.einsum_parse("ab,bc->ac", (2, 3), (3, 4), max_modes=1) is None
    p = oracle.einsum_parse(" a b , b c -> a c ", (2, 3), (3, 4))     # blanks are skipped (:91-116)
    assert p["output_shape"] == [2, 4] and p["modesA"] == ["b", "a"]


def test_headline_einsum_view():
    # cuTENSOR view of 'abcd,dcbe->ae'
    p = oracle.einsum_parse("abcd,dcbe->ae", (96, 64, 64, 64), (64, 64, 64, 96))
    assert p["modesA"] == list("dcba") and p["extentA"] == [64, 64, 64, 96]
    assert p["modesB"] == list("ebcd") and p["extentB"] == [96, 64, 64, 64]
    assert p["modesC"] == list("ea") and p["extentC"] == [96, 96]


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz")))
def test_against_reference_golden(name):
    """torch.einsum outputs for the reference's own test cases (einsum_test.py:45-125, the list parsed out of that file
    by tests/golden/make_golden.py) — the only numerical pins the reference has for this path; tolerance is the
    reference's (:35-42).  Every dtype of the list goes through the oracle's own entry point for it: fp32 / fp64,
    complex64 / complex128, and the 16-bit types as bit patterns (fp64 accumulation, one rounding to the 16-bit type)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    dt = meta["dtype"]
    ref = z["out"]
    if dt in ("float16", "bfloat16"):
        kind = "f16" if dt == "float16" else "bf16"
        a, b = oracle.to_bits(z["a"], kind), oracle.to_bits(z["b"], kind)
        np.testing.assert_array_equal(oracle.from_bits(a, kind), z["a"].astype(np.float64))   # fixtures hold rounded inputs
        out = oracle.from_bits(oracle.einsum(meta["equation"], a, b, h16=kind), kind)
        np.testing.assert_allclose(out, ref.astype(np.float64), rtol=5e-3, atol=6e-3)
        # one rounding of the exact sum: half an ulp of the 16-bit type
        ulp = 2.0 ** -11 if kind == "f16" else 2.0 ** -8
        np.testing.assert_allclose(out, ref.astype(np.float64), rtol=ulp, atol=1e-6)
        return
    a, b = z["a"], z["b"]
    out = oracle.einsum(meta["equation"], a, b)
    if np.iscomplexobj(ref):
        for part in (np.real, np.imag):       # einsum_test.py:38-40 compares real and imaginary parts separately
            np.testing.assert_allclose(part(out), part(ref), rtol=5e-3, atol=6e-3)
        np.testing.assert_allclose(out, ref, rtol=2e-4 if dt == "complex64" else 1e-12, atol=1e-4 if dt == "complex64" else 1e-12)
        return
    np.testing.assert_allclose(out, ref.astype(np.float64), rtol=5e-3, atol=6e-3)
    # and much tighter than the reference's bound for the fp32/fp64 cases
    np.testing.assert_allclose(out, ref.astype(np.float64), rtol=2e-4, atol=1e-4)


def test_golden_case_list_is_the_reference_list():
    """The fixtures cover every parameter set of einsum_test.py:45-125 (ten live cases + the bf16 one the reference keeps
    commented out) and carry the source line they were parsed from."""
    metas = [json.loads(str(np.load(os.path.join(GOLDEN, f))["meta"])) for f in sorted(os.listdir(GOLDEN)) if f.endswith(".npz")]
    names = sorted(m["reference_case"] + "/" + m["dtype"] for m in metas)
    assert names == sorted(["test 0/float32", "test 0 (complex)/complex64", "test 1/complex128", "test 2/float32", "test 3/float32",
                            "test 4/float16", "test 5/float16", "test 6/float16", "test 7/float16", "test 8/float64", "test 8/bfloat16"])
    assert all(m["source"].startswith("cuTENSOR/python/cutensor/torch/einsum_test.py:") for m in metas)
    assert [m["dtype"] for m in metas if m["commented_out_in_reference"]] == ["bfloat16"]


def test_sixteen_bit_conversions_match_numpy_and_torch():
    import torch
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(500) * s for s in (1e-8, 1e-5, 1e-3, 1, 100, 6e4)] +
                       [np.array([0.0, -0.0, 65504, 65519.9, 65520, 1e9, 6e-8, 3e-8, 2.98e-8, 5.96e-8, np.inf, -np.inf])])
    with np.errstate(over="ignore"):
        np.testing.assert_array_equal(oracle.to_bits(x, "f16"), x.astype(np.float16).view(np.uint16))
    np.testing.assert_array_equal(oracle.to_bits(x, "bf16"), torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))


def test_complex_contraction_with_conjugation():
    # python/einsum.h:82-83,321-322: opA / opB in {IDENTITY, CONJ}, complex alpha / beta (contraction_jit.cu:205)
    rng = np.random.default_rng(2)
    A = (rng.standard_normal((4, 5, 3)) + 1j * rng.standard_normal((4, 5, 3))).astype(np.complex64)
    B = (rng.standard_normal((5, 6, 3)) + 1j * rng.standard_normal((5, 6, 3))).astype(np.complex64)
    C = (rng.standard_normal((4, 6, 3)) + 1j * rng.standard_normal((4, 6, 3))).astype(np.complex64)
    D = np.zeros_like(C)
    oracle.contract(A, "mkl", B, "knl", D, "mnl", alpha=1.5 - 0.5j, beta=0.25j, C=C, conjA=True)
    ref = (1.5 - 0.5j) * np.einsum("mkl,knl->mnl", np.conj(A).astype(np.complex128), B.astype(np.complex128)) + 0.25j * C
    np.testing.assert_allclose(D, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dt", [np.complex64, np.complex128])
def test_complex_reduction_and_permutation(dt):
    """What the reference's binding runs for a unary equation on complex tensors (python/einsum.h:326-343,430-441: cutensorCreateReduction
    with OP_ADD + cutensorReduce; torch/einsum.cc:83 dispatches the complex types): the oracle's complex entry points against numpy."""
    rng = np.random.default_rng(5)
    def cplx(shape):
        return np.asfortranarray((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt))
    tol = dict(rtol=2e-5, atol=2e-5) if dt is np.complex64 else dict(rtol=1e-12, atol=1e-12)
    A, C = cplx((7, 5, 3, 4)), cplx((7, 4))
    D = np.zeros_like(C)
    oracle.reduce(A, "mhkv", D, "mv", alpha=1.1 - 0.3j, beta=0.5j, C=C)
    np.testing.assert_allclose(D, (1.1 - 0.3j) * A.astype(np.complex128).sum(axis=(1, 2)) + 0.5j * C, **tol)
    oracle.reduce(A, "mhkv", D, "mv", conjA=True, beta=2.0, C=C, conjC=True)
    np.testing.assert_allclose(D, np.conj(A.astype(np.complex128)).sum(axis=(1, 2)) + 2.0 * np.conj(C), **tol)
    Dm = np.zeros_like(C)
    oracle.reduce(A[:, :2, :2, :], "mhkv", Dm, "mv", op=oracle.OP_MUL)
    np.testing.assert_allclose(Dm, A[:, :2, :2, :].astype(np.complex128).prod(axis=(1, 2)), **tol)
    P = np.zeros((3, 7, 5, 4), dtype=dt, order="F")
    oracle.permute(A, "whcn", P, "cwhn", alpha=1.5 + 2j)
    np.testing.assert_allclose(P, (1.5 + 2j) * np.transpose(A, (2, 0, 1, 3)), **tol)
    Q = cplx((3, 7, 5, 4))
    Q0 = Q.copy()
    oracle.permute(A, "whcn", Q, "cwhn", alpha=1j, C=Q, gamma=-1.0, conjA=True)
    np.testing.assert_allclose(Q, 1j * np.conj(np.transpose(A, (2, 0, 1, 3))) - Q0, **tol)
    # the framework-level helper: unary equations go to the reduction entry point (einsum.cu:346-372)
    z = (rng.standard_normal((4, 6, 5)) + 1j * rng.standard_normal((4, 6, 5))).astype(dt)
    np.testing.assert_allclose(oracle.einsum("ijk->ik", z), np.einsum("ijk->ik", z.astype(np.complex128)), **tol)
    np.testing.assert_allclose(oracle.einsum("ij->ji", z[0]), z[0].T, **tol)


def test_naive_fp32_loop_is_the_literal_triple_loop():
    """oracle_contract_f32_naive is the loop BASELINE.json names (fp32 accumulation in loop order): identical to a Python
    restatement of that loop on a tiny case, and within fp32 round-off of the fp64-accumulating oracle."""
    A = make_tensor([3, 4], 11)
    B = make_tensor([4, 5], 12)
    D = np.zeros((3, 5), dtype=np.float32, order="F")
    oracle.contract(A, "mk", B, "kn", D, "mn", acc64=False)
    ref = np.zeros((3, 5), dtype=np.float32)
    for m in range(3):
        for n in range(5):
            acc = np.float32(0)
            for k in range(4):
                acc = np.float32(acc + np.float32(A[m, k] * B[k, n]))
            ref[m, n] = acc
    np.testing.assert_array_equal(D, ref)


FULL = os.path.join(GOLDEN, "full")


def _full_case(name):
    from tests.util import golden_inputs
    z = np.load(os.path.join(FULL, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    a, b = golden_inputs(meta["dtype"], meta["a_size"], meta["b_size"])
    wide = np.complex128 if np.iscomplexobj(z["out_sampled"]) else np.float64
    import torch
    tw = torch.complex128 if a.dtype.is_complex else torch.float64
    an, bn = a.to(tw).numpy(), b.to(tw).numpy()
    # generator drift probe: the inputs redrawn here are the inputs the fixture's outputs were computed from
    np.testing.assert_array_equal(an.reshape(-1)[:16].astype(wide), z["a_probe"])
    np.testing.assert_array_equal(bn.reshape(-1)[:16].astype(wide), z["b_probe"])
    return z, meta, a, b, an, bn


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(FULL) if f.endswith(".npz")))
def test_against_reference_golden_at_the_reference_extents(name):
    """The same eleven cases at the reference's own sizes (extents of 50, einsum_test.py:47-124): inputs redrawn as
    make_golden.py drew them, the oracle's result compared at the fixture's 4096 sampled positions (reference tolerance, then
    tighter) and through sum |out| over the whole tensor."""
    z, meta, a, b, an, bn = _full_case(name)
    dt = meta["dtype"]
    ref = z["out_sampled"]
    if dt in ("float16", "bfloat16"):
        kind = "f16" if dt == "float16" else "bf16"
        out = oracle.from_bits(oracle.einsum(meta["equation"], oracle.to_bits(an, kind), oracle.to_bits(bn, kind), h16=kind), kind)
    else:
        store = {"float32": np.float32, "float64": np.float64, "complex64": np.complex64, "complex128": np.complex128}[dt]
        out = oracle.einsum(meta["equation"], an.astype(store), bn.astype(store))
    assert list(out.shape) == meta["out_shape"]
    got = np.asarray(out).reshape(-1)[z["idx"]]
    for part in ((np.real, np.imag) if np.iscomplexobj(ref) else (lambda x: x,)):
        np.testing.assert_allclose(part(got), part(ref), rtol=5e-3, atol=6e-3)      # einsum_test.py:35-42
    tight = {"bfloat16": dict(rtol=2.0 ** -8, atol=1e-6), "float16": dict(rtol=2.0 ** -11, atol=1e-6), "float32": dict(rtol=2e-4, atol=2e-4),
             "complex64": dict(rtol=2e-4, atol=2e-4)}.get(dt, dict(rtol=1e-12, atol=1e-12))
    np.testing.assert_allclose(got, ref, **tight)
    np.testing.assert_allclose(np.abs(np.asarray(out)).sum(), float(z["sum_abs"]), rtol=1e-3 if dt in ("float16", "bfloat16") else 1e-5)


@pytest.mark.parametrize("eq,sa,sb", [("ijk,kl->il", (3, 4, 5), (5, 6)), ("ij,jk->k", (3, 4), (4, 5)), ("ik,jkl->ij", (3, 5), (4, 5, 2)),
                                      ("aij,kbj->ik", (2, 3, 5), (4, 3, 5)), ("ij,kl->", (3, 4), (2, 5)), ("bij,bjk->bk", (2, 3, 5), (2, 5, 4))])
def test_modes_that_one_input_alone_carries_are_summed_over_it(eq, sa, sb):
    """'ijk,kl->il': j lives in A alone — K is every mode the output does not carry (torch.einsum's reading; the pairwise steps
    cuTENSOR/python/cutensor/torch/einsum.py:111-156 builds).  Every dtype's entry point against numpy.einsum."""
    rng = np.random.default_rng(5)
    for dt, rtol in ((np.float32, 1e-5), (np.float64, 1e-12), (np.complex64, 1e-5), (np.complex128, 1e-12)):
        a = rng.random(sa).astype(dt)
        b = rng.random(sb).astype(dt)
        if np.issubdtype(dt, np.complexfloating):
            a = a + 1j * rng.random(sa).astype(dt)
            b = b - 1j * rng.random(sb).astype(dt)
        np.testing.assert_allclose(oracle.einsum(eq, a, b), np.einsum(eq, a, b), rtol=rtol)
    for kind in ("bf16", "f16"):
        a = oracle.to_bits(rng.random(sa), kind)
        b = oracle.to_bits(rng.random(sb), kind)
        want = np.einsum(eq, oracle.from_bits(a, kind), oracle.from_bits(b, kind))
        np.testing.assert_allclose(oracle.from_bits(oracle.einsum(eq, a, b, h16=kind), kind), want, rtol=2.0 ** -8 if kind == "bf16" else 2.0 ** -11)

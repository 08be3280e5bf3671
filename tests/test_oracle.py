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
    """torch.einsum outputs for the reference's own test cases (einsum_test.py:47-124) — the only
    numerical pins the reference has for this path; tolerance is the reference's (:35-42)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    a = z["a"].astype(np.float64 if meta["dtype"] == "float64" else np.float32)
    b = z["b"].astype(a.dtype)
    out = oracle.einsum(meta["equation"], a, b)
    np.testing.assert_allclose(out, z["out"].astype(np.float64), rtol=5e-3, atol=6e-3)
    # and much tighter than the reference's bound for the fp32/fp64 cases
    if meta["dtype"] in ("float32", "float64"):
        np.testing.assert_allclose(out, z["out"].astype(np.float64), rtol=2e-4, atol=1e-4)

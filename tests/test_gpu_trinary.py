"""GPU parity of cutensorElementwiseTrinaryExecute and of the non-ADD combiners of
cutensorElementwiseBinaryExecute against numpy (bit-exact for alpha = beta = gamma = 1 with ADD of
permutations; rtol 1e-6 otherwise — every output is at most three fp32 roundings away).
Reference: cuTENSOR/elementwise_trinary.cu:51-53 (D_{a,b,c} = alpha A_{c,b,a} + beta B_{c,a,b} +
gamma C_{a,b,c}, extents 400/200/300, scalars 1.1/1.3/1.2 at :46-48), elementwise_binary.cu:149-153."""
import ctypes

import numpy as np
import pytest

from util import make_tensor, to_device, from_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import torch
    assert torch.cuda.is_available()
    from cudalibrarysamples_amd import cutensor as ct, ops
    return torch, ct, ops, ops.Handle()


def _trinary(env, ext, mA, mB, mC, mD, alpha, beta, gamma, opAB="ADD", opABC="ADD", dtype=np.float32):
    torch, ct, ops, h = env
    A = make_tensor([ext[c] for c in mA], 1, dtype, -1, 1)
    B = make_tensor([ext[c] for c in mB], 2, dtype, -1, 1)
    C = make_tensor([ext[c] for c in mC], 3, dtype, -1, 1)
    dA, dB, dC = to_device(A), to_device(B), to_device(C)
    D = np.zeros([ext[c] for c in mD], dtype=dtype, order="F")
    dD = to_device(D)
    cdt = {np.float32: ct.R_32F, np.float64: ct.R_64F}[dtype]
    plan = ops.trinary_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [ext[c] for c in mC], mC,
                            [ext[c] for c in mD], mD, opAB=opAB, opABC=opABC, dtype=cdt)
    plan.trinary(alpha, dA.data_ptr(), beta, dB.data_ptr(), gamma, dC.data_ptr(), dD.data_ptr())
    torch.cuda.synchronize()
    got = from_device(dD, D)
    f = {"ADD": np.add, "MUL": np.multiply, "MAX": np.maximum, "MIN": np.minimum}
    pa = np.einsum("%s->%s" % (mA, mD), A)
    pb = np.einsum("%s->%s" % (mB, mD), B)
    pc = np.einsum("%s->%s" % (mC, mD), C)
    s = dtype
    ref = f[opABC](f[opAB](s(alpha) * pa, s(beta) * pb), s(gamma) * pc)
    return got, ref, plan


def test_sample_shape_two_pass(env):
    got, ref, plan = _trinary(env, dict(a=400, b=200, c=300), "cba", "cab", "abc", "abc", 1.1, 1.3, 1.2)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6)


def test_permutation_sum_is_exact(env):
    got, ref, _ = _trinary(env, dict(a=64, b=48, c=40), "cba", "cab", "abc", "abc", 1.0, 1.0, 1.0)
    np.testing.assert_array_equal(got, ref)   # (a + b) + c in fp32, same order as the kernels


@pytest.mark.parametrize("mA,mB", [("abc", "cab"), ("cab", "abc"), ("abc", "abc")])
def test_single_pass_when_an_operand_has_the_output_layout(env, mA, mB):
    got, ref, _ = _trinary(env, dict(a=132, b=36, c=20), mA, mB, "bca", "abc", 0.5, -2.0, 3.0)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("opAB,opABC", [("MUL", "ADD"), ("ADD", "MUL"), ("MAX", "MIN"), ("MIN", "MAX")])
def test_combiners(env, opAB, opABC):
    got, ref, _ = _trinary(env, dict(a=68, b=33, c=12), "cba", "bac", "abc", "abc", 1.5, -0.5, 2.0, opAB, opABC)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6)


def test_odd_extents_and_fp64(env):
    got, ref, _ = _trinary(env, dict(a=37, b=5, c=11), "cab", "bca", "abc", "abc", 1.1, 1.3, 1.2, dtype=np.float64)
    np.testing.assert_allclose(got, ref, rtol=1e-14)


def test_binary_mul_max(env):
    torch, ct, ops, h = env
    ext = dict(a=96, b=40, c=24)
    A = make_tensor([ext[c] for c in "cba"], 5, np.float32, -1, 1)
    C = make_tensor([ext[c] for c in "abc"], 6, np.float32, -1, 1)
    for name, f in (("MUL", np.multiply), ("MAX", np.maximum), ("MIN", np.minimum), ("ADD", np.add)):
        dA, dC = to_device(A), to_device(C)
        dD = to_device(np.zeros_like(C))
        plan = ops.binary_plan(h, [ext[c] for c in "cba"], "cba", [ext[c] for c in "abc"], "abc", op=name)
        plan.binary(1.25, dA.data_ptr(), -0.5, dC.data_ptr(), dD.data_ptr())
        torch.cuda.synchronize()
        got = from_device(dD, C)
        ref = f(np.float32(1.25) * np.einsum("cba->abc", A), np.float32(-0.5) * C)
        np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6, err_msg=name)


def test_padded_permutation(env):
    """C_{c',w',h',n} = A_{w,h,c,n} with per-mode left / right padding and a padding value
    (elementwise_permute_padding.cu:47-53, :178-195): bit-exact for alpha = 1."""
    torch, ct, ops, h = env
    ext = dict(w=32, h=20, c=24, n=6)
    A = make_tensor([ext[c] for c in "whcn"], 11, np.float32, -1, 1)
    left, right = [0, 1, 1, 0], [3, 2, 0, 0]          # per output mode c, w, h, n
    pext = [ext[c] + l + r for c, l, r in zip("cwhn", left, right)]
    out = np.full(pext, -1.0, dtype=np.float32, order="F")
    dA, dD = to_device(A), to_device(out)
    plan = ops.permutation_plan(h, [ext[c] for c in "whcn"], "whcn", [ext[c] for c in "cwhn"], "cwhn", padding=(left, right, 7.5))
    plan.permute(1.0, dA.data_ptr(), dD.data_ptr())
    torch.cuda.synchronize()
    got = from_device(dD, out)
    ref = np.pad(np.einsum("whcn->cwhn", A), list(zip(left, right)), constant_values=np.float32(7.5))
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("beta", [0.0, 0.75])
def test_contraction_trinary(env, beta):
    """D_{m,n,b,r,a} = alpha A_{m,k,a,j,b,i} B_{k,n,i} C_{r,j} + beta D (contraction_trinary.cu:44-48) at shrunk
    extents, against numpy.einsum in fp64; fp32 tolerance as for the binary contraction (rtol 1e-4)."""
    torch, ct, ops, h = env
    ext = dict(m=24, a=4, b=6, n=8, r=12, k=8, i=4, j=16)
    mA, mB, mC, mD = "mkajbi", "kni", "rj", "mnbra"
    A = make_tensor([ext[c] for c in mA], 21, np.float32)
    B = make_tensor([ext[c] for c in mB], 22, np.float32)
    C = make_tensor([ext[c] for c in mC], 23, np.float32)
    D = make_tensor([ext[c] for c in mD], 24, np.float32)
    dA, dB, dC, dD = to_device(A), to_device(B), to_device(C), to_device(D)
    plan = ops.contraction_trinary_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [ext[c] for c in mC], mC,
                                        [ext[c] for c in mD], mD)
    assert plan.required_workspace > 0 and plan.required_workspace <= plan.workspace_estimate
    ws = torch.empty(plan.required_workspace, dtype=torch.uint8, device="cuda")
    plan.contract_trinary(1.1, dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), beta, dD.data_ptr(), dD.data_ptr(),
                          ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    got = from_device(dD, D)
    ref = 1.1 * np.einsum("mkajbi,kni,rj->mnbra", A.astype(np.float64), B.astype(np.float64), C.astype(np.float64)) + beta * D
    np.testing.assert_allclose(got, ref, rtol=1e-4)


def test_both_operands_permuted_single_pass(env):
    """A_{c,b,a} and B_{c,a,b} share the partner mode c of the output tile (elementwise_trinary.cu:51-53): one pass with
    two LDS tiles; must agree with the two-pass result bit for bit is not required (same operations, same order), but
    it is: (alpha A + beta B) + gamma C in fp32 either way."""
    got, ref, plan = _trinary(env, dict(a=400, b=200, c=300), "cba", "cab", "abc", "abc", 1.1, 1.3, 1.2)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6)
    got, ref, plan = _trinary(env, dict(a=68, b=36, c=44), "cba", "cab", "abc", "abc", 1.0, 1.0, 1.0, "MAX", "MUL")
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6)
    got, ref, plan = _trinary(env, dict(a=66, b=35, c=44), "cba", "cab", "abc", "abc", 0.5, 2.0, -1.0)   # ragged a (66 % 4 != 0): generic path
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-6)

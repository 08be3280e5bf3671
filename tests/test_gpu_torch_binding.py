"""GPU regression suite of the PyTorch front end: the reference's own test list,
cuTENSOR/python/cutensor/torch/einsum_test.py:47-160 (forward AND both gradients of EinsumFunction against
torch.einsum) and :163-238 (EinsumGeneral, N-ary, gradients), real dtypes.  The checker is torch.einsum on the
CPU in float64 (the reference compares with torch.einsum on its GPU); tolerance = the reference's
rtol 5e-3 / atol 6e-3 (:42), tightened to rtol 2e-4 for float32."""
import pytest

pytestmark = pytest.mark.gpu

BINARY = [  # einsum_test.py:47-124: every case, plus the bf16 one the reference keeps commented out (:115-123)
    ("test0", (48, 37), (37, 74), "ik,kj->ij", "float32"),
    ("test0_complex", (50, 50), (50, 50), "ik,kj->ij", "complex64"),
    ("test1_complex", (50, 50, 50), (50, 50, 50), "lik,lkj->lij", "complex128"),
    ("test2", (50, 50, 50, 20), (50, 50, 50, 20), "likm,lkjm->lij", "float32"),
    ("test3", (20, 50, 50, 50), (50, 50, 50, 20), "mlik,lkjm->lij", "float32"),
    ("test4", (50, 50), (50, 50), "ik,kj->ij", "float16"),
    ("test5", (50, 50, 50), (50, 50, 50), "lik,lkj->lij", "float16"),
    ("test6", (50, 50, 50, 20), (50, 50, 50, 20), "likm,lkjm->lij", "float16"),
    ("test7", (20, 50, 50, 50), (50, 50, 50, 20), "mlik,lkjm->lij", "float16"),
    ("test8", (2, 5, 50, 2), (5, 2, 50, 2), "mlik,lkjm", "float64"),
    ("test8_bf16", (20, 50, 50, 50), (50, 50, 50, 20), "mlik,lkjm->lij", "bfloat16"),
]
GENERAL = [  # einsum_test.py:165-196
    ("g0", [(50, 60), (60, 40)], "ik,kj->ji", "float32"),
    ("g1", [(50, 60), (60, 7), (7, 8)], "ik,kl,lj->ij", "float32"),
    ("g2", [(50, 60), (60, 7), (7, 8)], "ik,kl,lj", "float32"),
    ("g3_complex", [(50, 60), (60, 7), (7, 8)], "ik,kl,lj->ij", "complex64"),
    ("g4", [(50, 60)], "ij->ji", "float32"),
]


@pytest.fixture(scope="module")
def te(built):
    import torch
    assert torch.cuda.is_available()
    from cudalibrarysamples_amd import torch_einsum
    return torch, torch_einsum


def _wide(torch, dtype):
    return torch.complex128 if dtype.startswith("complex") else torch.float64


def _close(torch, got, ref, dtype):
    assert got.shape == ref.shape
    if dtype.startswith("complex"):   # einsum_test.py:37-40: real and imaginary parts separately
        _close(torch, torch.real(got), torch.real(ref), "float32" if dtype == "complex64" else "float64")
        _close(torch, torch.imag(got), torch.imag(ref), "float32" if dtype == "complex64" else "float64")
        return
    torch.testing.assert_close(got.double().cpu(), ref, rtol=5e-3, atol=6e-3)
    if dtype in ("float32", "float64"):
        torch.testing.assert_close(got.double().cpu(), ref, rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize("name,a_size,b_size,equation,dtype", BINARY, ids=[c[0] for c in BINARY])
def test_einsum_function_forward_and_gradients(te, name, a_size, b_size, equation, dtype):
    torch, tein = te
    torch.manual_seed(0)
    dt = getattr(torch, dtype)
    # inputs exactly as the reference draws them (einsum_test.py:141-147): unscaled randn, also for the 16-bit types
    A = torch.randn(*a_size, dtype=dt if dt.is_complex else torch.float32).to(dt).cuda().requires_grad_(True)
    B = torch.randn(*b_size, dtype=dt if dt.is_complex else torch.float32).to(dt).cuda().requires_grad_(True)
    out = tein.EinsumFunction.apply(equation, A, B)
    out.backward(torch.ones_like(out))
    rA = A.detach().to(_wide(torch, dtype)).cpu().requires_grad_(True)
    rB = B.detach().to(_wide(torch, dtype)).cpu().requires_grad_(True)
    ref = torch.einsum(equation, rA, rB)
    ref.backward(torch.ones_like(ref))
    _close(torch, out.detach(), ref.detach(), dtype)
    _close(torch, A.grad, rA.grad, dtype)
    _close(torch, B.grad, rB.grad, dtype)


@pytest.mark.parametrize("name,sizes,equation,dtype", GENERAL, ids=[c[0] for c in GENERAL])
def test_einsum_general_forward_and_gradients(te, name, sizes, equation, dtype):
    torch, tein = te
    torch.manual_seed(1)
    dt = getattr(torch, dtype)
    ts = [torch.randn(*s, dtype=dt).cuda().requires_grad_(True) for s in sizes]
    out = tein.EinsumGeneral(equation, *ts)
    out.backward(torch.ones_like(out))
    rs = [t.detach().to(_wide(torch, dtype)).cpu().requires_grad_(True) for t in ts]
    ref = torch.einsum(equation, *rs)
    ref.backward(torch.ones_like(ref))
    _close(torch, out.detach(), ref.detach(), dtype)
    for t, r in zip(ts, rs):
        _close(torch, t.grad, r.grad, dtype)


def test_module_and_reduction_gradient_broadcast(te):
    """Einsum module (einsum.py:98-119) and a unary reduction whose gradient is a broadcast (einsum.cu:451 'nij->ji')."""
    torch, tein = te
    torch.manual_seed(2)
    A = torch.randn(6, 10, 12).cuda().requires_grad_(True)
    out = tein.Einsum("nij->ji")(A)
    out.backward(torch.arange(out.numel(), dtype=torch.float32, device="cuda").reshape(out.shape))
    rA = A.detach().double().cpu().requires_grad_(True)
    ref = torch.einsum("nij->ji", rA)
    ref.backward(torch.arange(ref.numel(), dtype=torch.float64).reshape(ref.shape))
    _close(torch, out.detach(), ref.detach(), "float32")
    _close(torch, A.grad, rA.grad, "float32")


def test_argument_count_errors(te):
    torch, tein = te
    a = torch.zeros(3, 4, device="cuda")
    with pytest.raises(RuntimeError):
        tein.EinsumFunction.apply("ik,kj->ij", a)
    with pytest.raises(RuntimeError):
        tein.EinsumFunction.apply("ij->ji", a, a)


def test_scalar_output_gradients_use_a_scalar_second_operand(te):
    """A dot product 'i,i->' differentiates into 'i,->i' and ',i->i': two operands, one of them a 0-dim tensor.  The helper
    must still contract (and scale by the scalar), not fall back to the one-operand path (found by tools/fuzz_einsum.py)."""
    torch, tein = te
    torch.manual_seed(3)
    a = torch.randn(37, device="cuda")
    s = torch.tensor(2.5, device="cuda")
    assert torch.allclose(tein.einsum("i,->i", a, s), a * 2.5)
    assert torch.allclose(tein.einsum(",i->i", s, a), a * 2.5)
    for eq, shapes in (("i,i->", [(16,), (16,)]), ("ie,ei", [(24, 2), (2, 24)])):
        A = torch.randn(*shapes[0], device="cuda", requires_grad=True)
        B = torch.randn(*shapes[1], device="cuda", requires_grad=True)
        out = tein.EinsumFunction.apply(eq, A, B)
        out.backward(torch.tensor(1.5, device="cuda"))
        rA, rB = A.detach().double().requires_grad_(True), B.detach().double().requires_grad_(True)
        torch.einsum(eq, rA, rB).backward(torch.tensor(1.5, device="cuda", dtype=torch.float64))
        _close(torch, A.grad, rA.grad.cpu(), "float32")
        _close(torch, B.grad, rB.grad.cpu(), "float32")


def test_workspace_min_replan_when_the_workspace_cannot_be_allocated(te, monkeypatch):
    """cutensor/torch/einsum.cc:104-123: if allocating the plan's workspace fails, the binding plans again with
    CUTENSOR_WORKSPACE_MIN and runs with what that plan needs.  The headline equation (shrunk) wants split-K partials;
    the first allocation is made to fail."""
    torch, tein = te
    torch.manual_seed(4)
    A = torch.rand(96, 16, 16, 64, device="cuda")
    B = torch.rand(64, 16, 16, 96, device="cuda")
    eq = "abcd,dcbe->ae"
    tein._plans.clear()
    tein._workspace.clear()
    want = tein.einsum(eq, A, B)
    plan = next(iter(tein._plans.values()))
    assert plan.required_workspace > 0 and plan.describe()["splitK"] > 1
    tein._plans.clear()
    tein._workspace.clear()
    calls = []
    real = tein._alloc_workspace

    def failing(nbytes, device):
        calls.append(nbytes)
        if len(calls) == 1:
            raise torch.cuda.OutOfMemoryError("simulated: no room for %d bytes" % nbytes)
        return real(nbytes, device)

    monkeypatch.setattr(tein, "_alloc_workspace", failing)
    got = tein.einsum(eq, A, B)
    plan = next(iter(tein._plans.values()))
    assert len(calls) == 1 and plan.required_workspace == 0 and plan.describe()["splitK"] == 1     # the minimum: no partials
    ref = torch.einsum(eq, A.double().cpu(), B.double().cpu())
    torch.testing.assert_close(got.double().cpu(), ref, rtol=2e-4, atol=2e-3)
    torch.testing.assert_close(want.double().cpu(), ref, rtol=2e-4, atol=2e-3)
    tein._plans.clear()
    tein._workspace.clear()

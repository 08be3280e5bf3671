"""Contractions with a mode that ONE input carries and nothing else ('ijk,kl->il': j) — VERDICT r5 "Missing #4".  The reference's N-ary
front end builds exactly such pairwise steps: _compute_target_tensor drops every mode no later operand or the target needs
(cuTENSOR/python/cutensor/torch/einsum.py:111-156), and torch.einsum, the reference's comparator, accepts the equation.  The engine reduces
the operand over the mode first (cutensorReduce into a temporary in the workspace), then contracts (csrc/host/api.cpp split_lone_modes).

Through the C ABI against the oracle (fp32 rtol 1e-4 on U(0,1) data; fp16 rtol 2e-3; complex64 rtol 1e-4), through the engine's PyTorch
front end with gradients against torch.einsum at the tolerance of einsum_test.py:35-42 (rtol 5e-3 / atol 6e-3), and through the reference's
own unmodified binding."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env(built):
    import torch
    assert torch.cuda.is_available()
    from cudalibrarysamples_amd import cutensor as ct, ops
    return torch, ct, ops, ops.Handle()


CASES = [  # equation (row-major framework form), shapes
    ("ijk,kl->il", (33, 7, 50), (50, 21)),          # the review's example: j lives in A alone
    ("ij,jk->k", (40, 50), (50, 31)),               # i lives in A alone, the result is a vector
    ("ik,jkl->ij", (20, 50), (17, 50, 9)),          # l lives in B alone
    ("aij,kbj->ik", (5, 30, 50), (12, 6, 50)),      # one lone mode in each operand
    ("ij,kl->", (6, 7), (8, 9)),                    # everything is summed: product of two full reductions
    ("bij,bjk->bk", (4, 10, 50), (4, 50, 13)),      # with a batch mode
]


@pytest.mark.parametrize("dtype", ["float32", "float16", "complex64"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_lone_modes_through_the_c_abi_against_the_oracle(env, case, dtype):
    import oracle
    torch, ct, ops, h = env
    eq, sa, sb = case
    rng = np.random.default_rng(abs(hash((eq, dtype))) % 1000)
    np_dt = {"float32": np.float32, "float16": np.float16, "complex64": np.complex64}[dtype]
    a = rng.random(sa).astype(np_dt)
    b = rng.random(sb).astype(np_dt)
    if dtype == "complex64":
        a = (a + 1j * rng.random(sa)).astype(np_dt)
        b = (b + 1j * rng.random(sb)).astype(np_dt)
    p = oracle.einsum_parse(eq, sa, sb)
    assert p is not None
    cdt = {"float32": ct.R_32F, "float16": ct.R_16F, "complex64": ct.C_32F}[dtype]
    plan = ops.contraction_plan(h, p["extentA"], p["modesA"], p["extentB"], p["modesB"], p["extentC"], p["modesC"], dtype=cdt, workspace_limit=1 << 28)
    d = plan.describe()
    assert d.get("lone_reduce_A", 0) + d.get("lone_reduce_B", 0) >= 1, d
    assert plan.required_workspace >= 256 and plan.required_workspace <= plan.workspace_estimate + (1 << 20)
    dA, dB = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    out = torch.full(p["output_shape"] or [1], float("nan"), dtype=dA.dtype, device="cuda")
    ws = torch.empty(plan.required_workspace, dtype=torch.uint8, device="cuda")
    plan.contract(1.0, dA.data_ptr(), dB.data_ptr(), 0.0, out.data_ptr(), out.data_ptr(), ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(p["output_shape"])
    if dtype == "float16":
        want = np.einsum(eq, a.astype(np.float64), b.astype(np.float64))
        # the temporary is rounded to fp16 once more than a fused sum would be: rtol 2 x 2^-11 + the output's own rounding
        np.testing.assert_allclose(got.astype(np.float64), want, rtol=2e-3)
    else:
        want = oracle.einsum(eq, a, b)
        np.testing.assert_allclose(got, want, rtol=1e-4)
    # alpha / beta ride on the inner contraction
    c0 = torch.from_numpy(rng.random(p["output_shape"] or [1]).astype(np_dt)).cuda()
    out2 = c0.clone()
    plan.contract(0.5, dA.data_ptr(), dB.data_ptr(), 2.0, c0.data_ptr(), out2.data_ptr(), ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    want2 = 0.5 * np.einsum(eq, a.astype(np.complex128 if dtype == "complex64" else np.float64), b.astype(np.complex128 if dtype == "complex64" else np.float64)) + \
        2.0 * c0.cpu().numpy().reshape(p["output_shape"])
    np.testing.assert_allclose(out2.cpu().numpy().reshape(p["output_shape"]), want2, rtol=3e-3 if dtype == "float16" else 1e-4)
    plan.destroy()


@pytest.mark.parametrize("dtype", ["float32", "float16", "complex64"])
def test_front_end_with_gradients_and_an_n_ary_chain(env, dtype):
    """te.einsum / EinsumFunction (forward + both gradients: d_b = einsum(A, C -> B) has the lone mode again, d_a broadcasts over it) and an
    EinsumGeneral chain whose first pairwise step drops a mode — against torch.einsum at einsum_test.py:35-42's tolerance."""
    torch, ct, ops, h = env
    from cudalibrarysamples_amd import torch_einsum as te
    tdt = getattr(torch, dtype)
    torch.manual_seed(0)
    scale = 0.25 if dtype == "float16" else 1.0
    tol = dict(rtol=5e-3, atol=6e-3)
    for eq, sa, sb in (("ijk,kl->il", (20, 6, 50), (50, 30)), ("ij,jk->k", (30, 50), (50, 20))):
        a = (torch.randn(sa, device="cuda", dtype=tdt) * scale).requires_grad_(True)
        b = (torch.randn(sb, device="cuda", dtype=tdt) * scale).requires_grad_(True)
        out = te.EinsumFunction.apply(eq, a, b)
        ref = torch.einsum(eq, a.detach().to(torch.complex128 if tdt.is_complex else torch.float64), b.detach().to(torch.complex128 if tdt.is_complex else torch.float64))
        torch.testing.assert_close(out.detach().to(ref.dtype), ref, **tol)
        g = torch.randn_like(out) * scale
        out.backward(g)
        a2 = a.detach().clone().requires_grad_(True)
        b2 = b.detach().clone().requires_grad_(True)
        torch.einsum(eq, a2.float() if dtype == "float16" else a2, b2.float() if dtype == "float16" else b2).backward(g.float() if dtype == "float16" else g)
        torch.testing.assert_close(a.grad.to(a2.grad.dtype), a2.grad, **tol)
        torch.testing.assert_close(b.grad.to(b2.grad.dtype), b2.grad, **tol)
    x = torch.randn(12, 5, 40, device="cuda", dtype=tdt) * scale
    y = torch.randn(40, 16, device="cuda", dtype=tdt) * scale
    z = torch.randn(16, 9, device="cuda", dtype=tdt) * scale
    got = te.EinsumGeneral("ijk,kl,lm->im", x, y, z)
    wide = torch.complex128 if tdt.is_complex else torch.float64
    torch.testing.assert_close(got.to(wide), torch.einsum("ijk,kl,lm->im", x.to(wide), y.to(wide), z.to(wide)), rtol=5e-3, atol=6e-2 if dtype == "float16" else 6e-3)


def test_reference_binding_runs_lone_mode_steps(built):
    """The reference's unmodified binding + python package (oracle/_ref/pyref): EinsumFunction on 'ijk,kl->il' / 'ij,jk->k' and an
    EinsumGeneral chain that produces such a step (its binding used to dereference a null plan here: NOTES.md section 5c)."""
    import glob
    import torch
    pyref = os.path.join(ROOT, "oracle", "_ref", "pyref")
    if not glob.glob(os.path.join(pyref, "cutensor", "torch", "binding*.so")):
        pytest.skip("oracle/_ref/pyref was not built (reference tree absent at build time)")
    for p in (pyref, os.path.join(ROOT, "tests", "sample_compat", "pyshim")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cutensor.torch as rct
    torch.manual_seed(0)
    tol = dict(rtol=5e-3, atol=6e-3)
    for tdt in (torch.float32, torch.float16, torch.complex64):
        s = 0.25 if tdt == torch.float16 else 1.0
        a = torch.randn(20, 6, 50, device="cuda", dtype=tdt) * s
        b = torch.randn(50, 30, device="cuda", dtype=tdt) * s
        wide = torch.complex128 if tdt.is_complex else torch.float64
        torch.testing.assert_close(rct.EinsumFunction.apply("ijk,kl->il", a, b).to(wide), torch.einsum("ijk,kl->il", a.to(wide), b.to(wide)), **tol)
        a2 = torch.randn(30, 50, device="cuda", dtype=tdt) * s
        torch.testing.assert_close(rct.EinsumFunction.apply("ij,jk->k", a2, b).to(wide), torch.einsum("ij,jk->k", a2.to(wide), b.to(wide)), **tol)
        z = torch.randn(30, 9, device="cuda", dtype=tdt) * s
        torch.testing.assert_close(rct.EinsumGeneral("ijk,kl,lm->im", a, b, z).to(wide), torch.einsum("ijk,kl,lm->im", a.to(wide), b.to(wide), z.to(wide)),
                                   rtol=5e-3, atol=6e-2 if tdt == torch.float16 else 6e-3)

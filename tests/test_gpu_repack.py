"""GPU parity of contractions whose operands are copied into packed temporaries first (api.cpp plan_repack, round 6): fp32 problems that
would leave the LDS-DMA ring kernels (a fastest contracted mode without whole 32-deep K-tiles beside further contracted modes; operands
contiguous in DIFFERENT contracted modes — the reference's own test equation 'mlik,lkjm->lij', python/cutensor/torch/einsum_test.py:84-107)
and their 16-bit twins (tests/test_gpu_h16.py holds those).  The copies are cutensorPermute at alpha = 1 — bit-exact — so the tolerance is
the contraction's own: fp32 rtol 1e-4 of the result magnitude against the fp64 oracle."""
import os

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


CASES = [  # (extents, modes of A, B, D)
    (dict(i=200, l=136, j=16, k=72), "kji", "jkl", "li"),             # A contiguous in k, B in j
    (dict(a=136, b=3, c=5, d=40, e=200), "dcba", "ebcd", "ea"),       # the headline equation, d = 40: no whole 32-deep K-tiles
    (dict(a=104, b=4, c=6, d=50, e=72), "dcba", "ebcd", "ea"),        # d = 50
    (dict(m=24, l=5, i=40, k=32, j=56), "kilm", "mjkl", "jil"),       # the reference's test equation (batch mode l)
    (dict(m=96, n=72, j=12, k=64, l=2), "jmkl", "knjl", "mnl"),       # A contiguous in j, B in k, batched
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("beta", [0.0, 0.7])
def test_fp32_operands_copied_first_small_against_the_oracle(env, case, beta):
    """Copies forced (CUTENSOR_AMD_REPACK=f, hooks flavour: whenever the temporaries fit and the ring kernels take the result) on
    problems the oracle finishes in a second."""
    ct, ops, h, torch = env
    ext, mA, mB, mC = CASES[case]
    eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
    A, B, C = make_tensor(eA, 1 + case), make_tensor(eB, 2 + case), make_tensor(eC, 3 + case)
    os.environ["CUTENSOR_AMD_REPACK"] = "f"
    try:
        p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, workspace_limit=1 << 28)
    finally:
        del os.environ["CUTENSOR_AMD_REPACK"]
    d = p.describe()
    assert (d.get("repack_A") or d.get("repack_B")) and d["family"] == 0, d
    assert p.required_workspace >= d["lone_bytes"], (p.required_workspace, d)
    dA, dB, dC = to_device(A), to_device(B), to_device(C)
    ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
    p.contract(1.1, dA.data_ptr(), dB.data_ptr(), beta, dC.data_ptr(), dC.data_ptr(), ws.data_ptr(), p.required_workspace,
               torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = from_device(dC, C)
    ref = np.zeros_like(C)
    oracle.contract(A, mA, B, mB, ref, mC, alpha=1.1, beta=beta, C=C)
    scale = float(np.max(np.abs(ref)))
    assert_close(got, ref, rtol=1e-4, atol=1e-4 * scale, what=str(d))
    # the operands are untouched
    assert np.array_equal(from_device(dA, A), A) and np.array_equal(from_device(dB, B), B)
    p.destroy()


def test_fp32_planners_own_choice_at_size(env):
    """'ijk,lkj->il' at 2048^2 x 16 x 72 and the headline equation with d = 40 at 2048^2: the planner copies an operand by itself; against
    torch.einsum in fp64 on the device (an independent implementation), and without the workspace the plan keeps the operands in place."""
    ct, ops, h, torch = env
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    for ext, mA, mB, mC in ((dict(i=2048, l=2048, j=16, k=72), "kji", "jkl", "li"), (dict(a=2048, b=8, c=16, d=40, e=2048), "dcba", "ebcd", "ea")):
        eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
        A = torch.rand(eA[::-1], generator=g, device="cuda")
        B = torch.rand(eB[::-1], generator=g, device="cuda")
        D = torch.full(eC[::-1], float("nan"), device="cuda")
        p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, workspace_limit=1 << 30)
        d = p.describe()
        assert (d.get("repack_A") or d.get("repack_B")) and d["kname"] == "gett_f32_stream_kernel", d
        assert p.required_workspace <= p.workspace_estimate, (p.required_workspace, p.workspace_estimate)
        ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
        p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace)
        torch.cuda.synchronize()
        ref = torch.einsum("%s,%s->%s" % (mA[::-1], mB[::-1], mC[::-1]), A.double(), B.double())
        err = float((D.double() - ref).abs().max() / ref.abs().max())
        assert err < 1e-4, (err, d)
        p.destroy()
        p0 = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, workspace_limit=0)
        d0 = p0.describe()
        assert not d0.get("repack_A") and not d0.get("repack_B") and p0.required_workspace == 0, d0
        p0.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
        torch.cuda.synchronize()
        err = float((D.double() - ref).abs().max() / ref.abs().max())
        assert err < 1e-4, (err, d0)
        p0.destroy()


def test_fp64_operands_copied_first(env):
    """fp64 (einsum.cu:36-41 runs the helper in double): where the general MFMA family would gather single elements — 'ijk,lkj->il', the
    reference's 'mlik,lkjm->lij' — one operand is copied first so that both sides stage 16-byte units.  Small cases with the copies
    forced against the oracle (fp64 accumulation both sides: rtol 1e-12 of the magnitude), and the planner's own choice at
    2048^2 x 16 x 72 against torch.einsum in fp64 on the device."""
    ct, ops, h, torch = env
    for case, (ext, mA, mB, mC) in enumerate(CASES):
        eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
        A, B, C = make_tensor(eA, 21 + case, dtype=np.float64), make_tensor(eB, 22 + case, dtype=np.float64), make_tensor(eC, 23 + case, dtype=np.float64)
        os.environ["CUTENSOR_AMD_REPACK"] = "f"
        try:
            p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, dtype=ct.R_64F, workspace_limit=1 << 28)
        finally:
            del os.environ["CUTENSOR_AMD_REPACK"]
        d = p.describe()
        if d.get("repack_A") or d.get("repack_B"):                      # (forced copies still need a direct plan on element gathers)
            assert d["family"] == 2 and d["vec"] >= 2, d
        dA, dB, dC = to_device(A), to_device(B), to_device(C)
        ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
        p.contract(1.1, dA.data_ptr(), dB.data_ptr(), 0.7, dC.data_ptr(), dC.data_ptr(), ws.data_ptr(), p.required_workspace,
                   torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = from_device(dC, C)
        ref = np.zeros_like(C)
        oracle.contract(A, mA, B, mB, ref, mC, alpha=1.1, beta=0.7, C=C)
        scale = float(np.max(np.abs(ref)))
        assert_close(got, ref, rtol=1e-12, atol=1e-12 * scale, what=str(d))
        p.destroy()
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    ext, mA, mB, mC = dict(i=2048, l=2048, j=16, k=72), "kji", "jkl", "li"
    eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
    A = torch.rand(eA[::-1], generator=g, device="cuda", dtype=torch.float64)
    B = torch.rand(eB[::-1], generator=g, device="cuda", dtype=torch.float64)
    D = torch.full(eC[::-1], float("nan"), device="cuda", dtype=torch.float64)
    p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, dtype=ct.R_64F, workspace_limit=1 << 30)
    d = p.describe()
    assert (d.get("repack_A") or d.get("repack_B")) and d["family"] == 2 and d["vec"] == 2, d
    assert p.required_workspace <= p.workspace_estimate
    ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
    p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace)
    torch.cuda.synchronize()
    ref = torch.einsum("%s,%s->%s" % (mA[::-1], mB[::-1], mC[::-1]), A, B)
    err = float((D - ref).abs().max() / ref.abs().max())
    assert err < 1e-12, (err, d)
    p.destroy()


def test_complex64_operands_copied_first(env):
    """complex64 (python/cutensor/torch/einsum_test.py:55-68 runs the binding on complex tensors): the general family stages 16-byte units
    of two elements where it can; where it would gather single elements one operand is copied first.  Forced on the small cases and by the
    planner's own choice at 2048^2 x 16 x 72, against torch.einsum in complex128 on the device (rtol 1e-5 of the magnitude: fp32 arithmetic)."""
    ct, ops, h, torch = env
    g = torch.Generator(device="cuda")
    g.manual_seed(13)
    todo = [(c, True) for c in CASES] + [((dict(i=2048, l=2048, j=16, k=72), "kji", "jkl", "li"), False)]
    for (ext, mA, mB, mC), forced in todo:
        eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
        A = torch.complex(torch.rand(eA[::-1], generator=g, device="cuda") - 0.5, torch.rand(eA[::-1], generator=g, device="cuda") - 0.5)
        B = torch.complex(torch.rand(eB[::-1], generator=g, device="cuda") - 0.5, torch.rand(eB[::-1], generator=g, device="cuda") - 0.5)
        D = torch.full(eC[::-1], float("nan"), device="cuda", dtype=torch.complex64)
        if forced:
            os.environ["CUTENSOR_AMD_REPACK"] = "f"
        try:
            p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, dtype=ct.C_32F, workspace_limit=1 << 30)
        finally:
            os.environ.pop("CUTENSOR_AMD_REPACK", None)
        d = p.describe()
        if not forced:
            assert (d.get("repack_A") or d.get("repack_B")) and d["family"] == 2 and d["vec"] == 2, d
        assert p.required_workspace <= max(p.workspace_estimate, p.required_workspace if forced else 0)
        ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
        alpha = np.array([1.0, 0.0], dtype=np.float32)
        p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace)
        torch.cuda.synchronize()
        ref = torch.einsum("%s,%s->%s" % (mA[::-1], mB[::-1], mC[::-1]), A.to(torch.complex128), B.to(torch.complex128))
        err = float((D.to(torch.complex128) - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, (err, d)
        p.destroy()


def test_workspace_at_a_128_byte_boundary(env):
    """contraction.cu:242 asserts a 128-byte-aligned workspace — not 256.  The temporaries of a two-step plan (operands copied first, an
    operand reduced over its lone modes) live at the head of that workspace and must not ask for more alignment than it has."""
    ct, ops, h, torch = env
    g = torch.Generator(device="cuda")
    g.manual_seed(17)
    for ext, mA, mB, mC, dt, tdt in ((dict(i=2048, l=2048, j=16, k=72), "kji", "jkl", "li", ct.R_16BF, torch.bfloat16),
                                     (dict(i=2048, l=2048, j=16, k=72), "kji", "jkl", "li", ct.R_32F, torch.float32),
                                     (dict(i=96, j=8, k=64, l=72), "ijk", "kl", "il", ct.R_32F, torch.float32)):       # (lone mode j)
        eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
        A = (torch.rand(eA[::-1], generator=g, device="cuda") - 0.5).to(tdt)
        B = (torch.rand(eB[::-1], generator=g, device="cuda") - 0.5).to(tdt)
        D = torch.full(eC[::-1], float("nan"), device="cuda", dtype=tdt)
        p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, dtype=dt, workspace_limit=1 << 30)
        d = p.describe()
        assert d.get("repack_A") or d.get("repack_B") or d.get("lone_reduce_A"), d
        buf = torch.empty(p.required_workspace + 512, dtype=torch.uint8, device="cuda")
        ptr = (buf.data_ptr() + 255) // 256 * 256 + 128                      # 128 (mod 256)
        p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ptr, p.required_workspace)
        torch.cuda.synchronize()
        rA, rB, rC = mA[::-1], mB[::-1], mC[::-1]
        ref = torch.einsum("%s,%s->%s" % (rA, rB, rC), A.double(), B.double())
        err = float((D.double() - ref).abs().max() / ref.abs().max())
        assert err < (1e-2 if tdt == torch.bfloat16 else 1e-4), (err, d)
        p.destroy()

"""GPU parity of the 16-bit LDS-DMA kernels on operands WITHOUT 16-byte lanes (round 6; VERDICT r5 "Missing #1"): extents that are not
multiples of 8 (4100-class: every row 8 (mod 16) bytes; 4097-class: odd pitch, rows at 2 (mod 16)), 2-byte-aligned base pointers, padded
row pitches.  The reference places no such restriction (cuTENSOR/contraction.cu:33-59 retyped per SURVEY a6) and its own numerical tests
use extents of 50 (python/cutensor/torch/einsum_test.py:84-107).

Every tensor lives inside a larger NaN-filled buffer at an ODD element offset: a 16-byte unit that reaches into the guard (the tail of a
partial k-unit, a row-unit past the last row, anything past the end of the tensor) and is not masked or repaired shows up as NaN in the
result; the guard of D must stay NaN bit for bit.  Compared with torch.einsum in fp64 on the rounded inputs (rtol 8e-3 bf16 / 2e-3 fp16, the
tolerances of tests/test_gpu_h16.py) and, for the small cases, with the oracle's 16-bit entry point on the same bit patterns (one 16-bit ulp)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the body runs in-process (planner's own choice) and in child processes (CUTENSOR_AMD_H16_WAVES names the kernel)
CASES = r'''
import numpy as np, torch
from cudalibrarysamples_amd import cutensor as ct, ops

LAYOUTS = (("mk", "kn"), ("km", "kn"), ("mk", "nk"), ("km", "nk"))

def guarded(ext_cm, pitch_pad, guard, tdt, g, fill=None):
    """column-major tensor of extents ext_cm (first fastest), rows padded by pitch_pad elements, inside a NaN buffer at element offset guard"""
    e0, e1 = ext_cm
    pitch = e0 + pitch_pad
    buf = torch.full((guard + pitch * e1 + guard + 64,), float("nan"), dtype=tdt, device="cuda")
    view = buf[guard: guard + pitch * e1].view(e1, pitch)[:, :e0]
    if fill is None:
        view.copy_((torch.rand((e1, e0), generator=g, device="cuda") * 2 - 1).to(tdt))
    return buf, view, [1, pitch]

def run(h, mA, mB, m, n, k, dtype_name="bfloat16", alpha=1.0, beta=0.0, pad=(0, 0, 0), guard=3, want=None, seed=0, oracle_check=True):
    tdt = getattr(torch, dtype_name)
    cdt = ct.R_16BF if dtype_name == "bfloat16" else ct.R_16F
    ext = dict(m=m, n=n, k=k)
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    bufA, A, sA = guarded([ext[c] for c in mA], pad[0], guard, tdt, g)
    bufB, B, sB = guarded([ext[c] for c in mB], pad[1], guard, tdt, g)
    bufC, C, sC = guarded([m, n], pad[2], guard, tdt, g)
    bufD = bufC.clone()
    D = bufD[guard: guard + (m + pad[2]) * n].view(n, m + pad[2])[:, :m]
    plan = ops.contraction_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [m, n], "mn", dtype=cdt, strideA=sA, strideB=sB,
                                strideC=sC, alignment=2, workspace_limit=1 << 28)
    d = plan.describe()
    assert d["family"] == 1, d
    if want is not None:
        assert d["kname"] == want, d
    ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    es = 2
    plan.contract(alpha, bufA.data_ptr() + es * guard, bufB.data_ptr() + es * guard, beta, bufC.data_ptr() + es * guard, bufD.data_ptr() + es * guard,
                  ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    ref = torch.einsum("%s,%s->nm" % (mA[::-1], mB[::-1]), A.double(), B.double()) * alpha + beta * C.double()
    got = D.double()
    assert not torch.isnan(got).any(), (mA, mB, m, n, k, d["kname"], "NaN in the result: a guard element was multiplied")
    tol = dict(rtol=8e-3, atol=3e-2) if dtype_name == "bfloat16" else dict(rtol=2e-3, atol=1e-2)
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), err_msg=str((mA, mB, m, n, k, d["kname"])), **tol)
    # everything outside D's elements is untouched: same bits as before
    mask = torch.ones_like(bufD, dtype=torch.bool)
    mask[guard: guard + (m + pad[2]) * n].view(n, m + pad[2])[:, :m] = False
    assert torch.equal(bufD.view(torch.int16)[mask], bufC.view(torch.int16)[mask]), (mA, mB, m, n, k, "stored outside D")
    if oracle_check and m * n * k <= 3e7:
        import oracle
        kind = "bf16" if dtype_name == "bfloat16" else "f16"
        bits = lambda t: t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
        oD = np.zeros((n, m), dtype=np.uint16)
        oracle.contract(bits(A), mA[::-1], bits(B), mB[::-1], oD, "nm", alpha=alpha, beta=beta, C=bits(C), h16=kind)
        ulp = 2.0 ** -7 if kind == "bf16" else 2.0 ** -10
        np.testing.assert_allclose(got.cpu().numpy(), oracle.from_bits(oD, kind), rtol=ulp, atol=3e-2 if kind == "bf16" else 4e-3)
    plan.destroy()
    return d

SHAPES = (
    (300, 204, 100),    # every extent = 4 (mod 8): rows at 8 (mod 16), K-tail of 4 in a partial unit, 2 K-tiles
    (257, 129, 65),     # odd: rows at 2 (mod 16); K = one whole K-tile + 1 element
    (516, 260, 68),     # 4 (mod 8), tiles with one live row / column beyond a 256 / 128 / 64 boundary
    (50, 50, 50),       # the reference's own extents: ONE K-tile, masked + repaired in the prologue
    (131, 67, 191),     # odd everything, three K-tiles, K-tail of 63
    (64, 64, 7),        # K shorter than one unit
    (9, 3, 130),
)
'''

FORCED = r'''
h = ops.Handle()
n_run = 0
for (mA, mB) in LAYOUTS:
    for i, (m, n, k) in enumerate(SHAPES):
        run(h, mA, mB, m, n, k, want=WANT, seed=10 * i + len(mA + mB))
        n_run += 1
    run(h, mA, mB, 300, 204, 100, alpha=1.5, beta=-0.75, want=WANT, seed=77)
    run(h, mA, mB, 257, 129, 65, dtype_name="float16", alpha=0.5, beta=0.25, want=WANT, seed=78)
    run(h, mA, mB, 260, 132, 132, pad=(5, 3, 1), want=WANT, seed=79)          # padded pitches: odd for A and B, D rows at 2 (mod 4) bytes
    run(h, mA, mB, 260, 132, 132, pad=(4, 12, 6), beta=1.0, want=WANT, seed=80)
print("ok", n_run)
'''


def _child(code, env_extra, timeout=900):
    envv = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", code], env=envv, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("ok"), (env_extra, r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("waves,want", [("4x", "gett_h16w4x_kernel"), ("4m", "gett_h16w4m_kernel"), ("4m4", "gett_h16w4m4_kernel"),
                                        ("4q", "gett_h16w4q_kernel")])
def test_every_masking_kernel_on_operands_without_16_byte_lanes(built, waves, want):
    _child(CASES + FORCED.replace("WANT", repr(want)), dict(CUTENSOR_AMD_H16_WAVES=waves))


def test_planners_own_choice_and_split_k(built):
    code = CASES + r'''
h = ops.Handle()
for (mA, mB) in LAYOUTS:
    for i, (m, n, k) in enumerate(SHAPES):
        run(h, mA, mB, m, n, k, seed=i)
    # deep ragged K over a small output: split-K, fp32 partials, the LAST slice owns the masked tile
    d = run(h, mA, mB, 100, 60, 4100, seed=5)
    assert d["splitK"] > 1, d
    d = run(h, mA, mB, 1028, 1028, 1028, seed=6, oracle_check=False)
    d = run(h, mA, mB, 2052, 1030, 516, alpha=0.5, beta=0.5, seed=7, oracle_check=False)
print("ok")
'''
    _child(code, {})


def test_4100_cubed_sampled(built):
    """The shape the round's target is quoted on: bf16 4100^3 on all four layouts, 4096 sampled outputs against fp64 dot products."""
    code = r'''
import numpy as np, torch
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
g = torch.Generator(device="cuda"); g.manual_seed(1)
E = 4100
for (mA, mB) in (("mk", "kn"), ("km", "kn"), ("mk", "nk"), ("km", "nk")):
    A = (torch.rand((E, E), generator=g, device="cuda") * 2 - 1).bfloat16()
    B = (torch.rand((E, E), generator=g, device="cuda") * 2 - 1).bfloat16()
    D = torch.full((E, E), float("nan"), dtype=torch.bfloat16, device="cuda")
    plan = ops.contraction_plan(h, [E, E], mA, [E, E], mB, [E, E], "mn", dtype=ct.R_16BF, workspace_limit=1 << 28)
    d = plan.describe()
    assert d["family"] == 1, d
    ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    assert not torch.isnan(D.float()).any()
    # A as [m, k] and B as [k, n] matrices whatever the layout: torch tensors are the reversed (row-major) views
    Am = A.t() if mA == "mk" else A          # mk: tensor [k][m] -> [m, k]
    Bm = B if mB == "nk" else B.t()          # nk: tensor [k][n]; kn: tensor [n][k] -> [k, n]
    rng = np.random.default_rng(3)
    mi = torch.from_numpy(rng.integers(0, E, 4096)).cuda()
    ni = torch.from_numpy(rng.integers(0, E, 4096)).cuda()
    mi[:64] = E - 1 - torch.arange(64, device="cuda") % 8       # the last rows / columns: edge tiles
    ni[64:128] = E - 1 - torch.arange(64, device="cuda") % 8
    ref = (Am[mi].double() * Bm[:, ni].t().double()).sum(dim=1)
    got = D[ni, mi].double()                                      # D is [n][m] row-major
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=8e-3, atol=0.25, err_msg=str((mA, mB, d["kname"])))
print("ok")
'''
    _child(code, {})

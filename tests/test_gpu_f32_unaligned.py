"""GPU parity of the fp32 LDS-DMA ring kernels (gett_f32_stream.hip) on operands WITHOUT 16-byte lanes and on ragged K (round 6; VERDICT r5
"Missing #1": "fp32 LAY_S ... no rate recorded anywhere"): extents that are not multiples of 4 (4098-class: rows at 8 (mod 16) bytes;
4097-class: 4 (mod 16)), K % 32 != 0, 4-byte-aligned base pointers, padded pitches — the RAG twin of the planner's ring kernel: masked last
K-tile, partial k-units zeroed in LDS by the data-moving waves, buffer descriptors that end with the tensor.  Reference shapes without such
lanes: cuTENSOR/python/cutensor/torch/einsum_test.py:47-83 (extents of 50, fp32).

Every tensor lives inside a NaN-filled buffer at an odd element offset; inputs are U(0,1) so that the fp32 result agrees with the fp64
reference to rtol 1e-4 (the tolerance of tests/test_gpu_contraction.py); D's guard must stay NaN bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import numpy as np, torch
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()

def guarded(e0, e1, pad, guard, g):
    pitch = e0 + pad
    buf = torch.full((guard + pitch * e1 + guard + 64,), float("nan"), dtype=torch.float32, device="cuda")
    view = buf[guard: guard + pitch * e1].view(e1, pitch)[:, :e0]
    view.copy_(torch.rand((e1, e0), generator=g, device="cuda"))
    return buf, view, [1, pitch]

def run(mA, mB, m, n, k, alpha=1.0, beta=0.0, pad=(0, 0, 0), guard=3, seed=0, ws=1 << 28, want_stream=True, algo=None):
    ext = dict(m=m, n=n, k=k)
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    bufA, A, sA = guarded(ext[mA[0]], ext[mA[1]], pad[0], guard, g)
    bufB, B, sB = guarded(ext[mB[0]], ext[mB[1]], pad[1], guard, g)
    bufC, C, sC = guarded(m, n, pad[2], guard, g)
    bufD = bufC.clone()
    D = bufD[guard: guard + (m + pad[2]) * n].view(n, m + pad[2])[:, :m]
    kw = {} if algo is None else dict(algo=algo)
    plan = ops.contraction_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [m, n], "mn", dtype=ct.R_32F, strideA=sA, strideB=sB,
                                strideC=sC, alignment=4, workspace_limit=ws, **kw)
    d = plan.describe()
    if want_stream:
        assert d["family"] == 0 and d["kname"] == "gett_f32_stream_kernel", d
    w = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    plan.contract(alpha, bufA.data_ptr() + 4 * guard, bufB.data_ptr() + 4 * guard, beta, bufC.data_ptr() + 4 * guard, bufD.data_ptr() + 4 * guard,
                  w.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    ref = torch.einsum("%s,%s->nm" % (mA[::-1], mB[::-1]), A.double(), B.double()) * alpha + beta * C.double()
    got = D.double()
    assert not torch.isnan(got).any(), (mA, mB, m, n, k, d, "NaN in the result: a guard element was multiplied")
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, err_msg=str((mA, mB, m, n, k, d["kernel"], d["splitK"])))
    mask = torch.ones_like(bufD, dtype=torch.bool)
    mask[guard: guard + (m + pad[2]) * n].view(n, m + pad[2])[:, :m] = False
    assert torch.equal(bufD.view(torch.int32)[mask], bufC.view(torch.int32)[mask]), (mA, mB, m, n, k, "stored outside D")
    plan.destroy()
    return d

LAYOUTS = (("mk", "kn"), ("km", "kn"), ("mk", "nk"), ("km", "nk"))
SHAPES = ((258, 130, 98), (257, 129, 65), (50, 50, 50), (131, 67, 191), (64, 64, 3), (9, 3, 130), (300, 204, 100), (130, 258, 33))
n_run = n_stream = 0
def sweep(mA, mB, m, n, k, ranks=10, **kw):
    """the planner's first `ranks` candidates (algo = r picks the r-th: tiles 32 .. 128, ring depths, split-K, register-staged kernels);
    at least one of them must be a ring kernel — its RAG twin is what this file is about"""
    global n_run, n_stream
    seen = 0
    for r in range(ranks):
        d = run(mA, mB, m, n, k, algo=r, want_stream=False, **kw)
        seen += d["kname"] == "gett_f32_stream_kernel"
        n_run += 1
    assert seen > 0, (mA, mB, m, n, k, "no ring kernel among the candidates")
    n_stream += seen
for (mA, mB) in LAYOUTS:
    for i, (m, n, k) in enumerate(SHAPES):
        sweep(mA, mB, m, n, k, seed=10 * i + 1)
    sweep(mA, mB, 258, 130, 98, ranks=6, alpha=1.5, beta=-0.75, seed=77)
    sweep(mA, mB, 262, 134, 134, ranks=6, pad=(5, 3, 1), seed=79)
    sweep(mA, mB, 262, 134, 134, ranks=6, pad=(2, 6, 3), beta=1.0, seed=80)
    d = run(mA, mB, 100, 60, 4099, seed=5, want_stream=False)   # deep ragged K over a small output: split-K, the last slice owns the masked tile
    assert d["splitK"] > 1, d
    run(mA, mB, 1026, 1030, 1027, seed=6)                  # the planner's own choice at a size where the ring kernels win
    sweep(mA, mB, 300, 204, 1030, ranks=14, seed=8)
print("ok", n_run, n_stream)
'''


def test_fp32_ring_kernels_on_operands_without_16_byte_lanes(built):
    r = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("ok"), (r.stdout[-2000:], r.stderr[-4000:])


def test_4098_cubed_sampled(built):
    """The shape the round's fp32 figure is quoted on: 4098^3 on all four layouts, 2048 sampled outputs against fp64 dot products."""
    code = r'''
import numpy as np, torch
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
g = torch.Generator(device="cuda"); g.manual_seed(1)
E = 4098
for (mA, mB) in (("mk", "kn"), ("km", "kn"), ("mk", "nk"), ("km", "nk")):
    A = torch.rand((E, E), generator=g, device="cuda")
    B = torch.rand((E, E), generator=g, device="cuda")
    D = torch.full((E, E), float("nan"), device="cuda")
    plan = ops.contraction_plan(h, [E, E], mA, [E, E], mB, [E, E], "mn", dtype=ct.R_32F, workspace_limit=1 << 28)
    d = plan.describe()
    assert d["kname"] == "gett_f32_stream_kernel", d
    ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    assert not torch.isnan(D).any()
    Am = A.t() if mA == "mk" else A
    Bm = B if mB == "nk" else B.t()
    rng = np.random.default_rng(3)
    mi = torch.from_numpy(rng.integers(0, E, 2048)).cuda()
    ni = torch.from_numpy(rng.integers(0, E, 2048)).cuda()
    mi[:64] = E - 1 - torch.arange(64, device="cuda") % 4
    ni[64:128] = E - 1 - torch.arange(64, device="cuda") % 4
    ref = (Am[mi].double() * Bm[:, ni].t().double()).sum(dim=1)
    np.testing.assert_allclose(D[ni, mi].double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, err_msg=str((mA, mB, d["kernel"])))
print("ok")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-4000:])

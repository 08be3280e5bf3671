"""GPU parity of the fp32 kernels' row epilogue (round 6: gett_store_tile_f32_rows, gett_common.h) — outputs with one M mode and one N mode
whose stride-1 mode keeps 16-byte lanes leave as whole rows through a per-wave LDS image instead of 4-byte pieces.  Reference form:
D = alpha * A * B + beta * C, cuTENSOR/contraction.cu:43, :184-185, :261-265.

Covered: partial tiles along both output modes, one K-tile .. many, beta != 0 with a C of its own, padded pitches of C / D (multiples of 4
elements: the path; others: the direct form), batch modes, the planner's first candidates (ring kernels and register-staged kernels, tiles
32 .. 128), and the fall-back conditions (a base pointer or a pitch without 16-byte lanes).  Every tensor lives inside a NaN-filled buffer;
D's guard and padding must stay NaN bit for bit.  Inputs are U(0,1): fp32 agrees with the fp64 reference to rtol 1e-4 (the tolerance of
tests/test_gpu_contraction.py).  The two forms use the same arithmetic — alpha * acc, then fma(beta, c, .) — so an aligned and a
misaligned placement of the same problem must agree bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import os
import numpy as np, torch
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()

def guarded(e0, e1, pad, guard, g, batch=1):
    pitch = e0 + pad
    buf = torch.full((guard + pitch * e1 * batch + guard + 64,), float("nan"), dtype=torch.float32, device="cuda")
    view = buf[guard: guard + pitch * e1 * batch].view(batch, e1, pitch)[:, :, :e0]
    view.copy_(torch.rand((batch, e1, e0), generator=g, device="cuda"))
    return buf, view, [1, pitch, pitch * e1]

def run(mA, mB, m, n, k, alpha=1.0, beta=0.0, pad=(0, 0, 0), guard=4, seed=0, ws=1 << 28, algo=None, batch=1, expect_untouched=False):
    ext = dict(m=m, n=n, k=k, l=batch)
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    bufA, A, sA = guarded(ext[mA[0]], ext[mA[1]], pad[0], guard, g, batch)
    bufB, B, sB = guarded(ext[mB[0]], ext[mB[1]], pad[1], guard, g, batch)
    bufC, C, sC = guarded(m, n, pad[2], guard, g, batch)
    bufD = torch.full_like(bufC, float("nan"))
    D = bufD[guard: guard + (m + pad[2]) * n * batch].view(batch, n, m + pad[2])[:, :, :m]
    kw = {} if algo is None else dict(algo=algo)
    tail = "l" if batch > 1 else ""
    nd = 3 if batch > 1 else 2
    plan = ops.contraction_plan(h, [ext[c] for c in mA + tail], mA + tail, [ext[c] for c in mB + tail], mB + tail, [ext[c] for c in "mn" + tail], "mn" + tail,
                                dtype=ct.R_32F, strideA=sA[:nd], strideB=sB[:nd], strideC=sC[:nd], alignment=4 if guard % 4 else 16, workspace_limit=ws, **kw)
    d = plan.describe()
    if expect_untouched and d["splitK"] > 1:       # split-K: the fold writes D, not the GETT kernel
        plan.destroy()
        return d, None
    w = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    plan.contract(alpha, bufA.data_ptr() + 4 * guard, bufB.data_ptr() + 4 * guard, beta, bufC.data_ptr() + 4 * guard, bufD.data_ptr() + 4 * guard,
                  w.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    if expect_untouched:
        assert torch.isnan(bufD).all(), (mA, mB, m, n, k, d, "the row epilogue is not the path of this plan")
        plan.destroy()
        return d, None
    ref = torch.einsum("l%s,l%s->lnm" % (mA[::-1], mB[::-1]), A.double(), B.double()) * alpha + beta * C.double()
    got = D.double()
    assert not torch.isnan(got).any(), (mA, mB, m, n, k, d, "NaN in the result")
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, err_msg=str((mA, mB, m, n, k, batch, d["kname"], d["bm"], d["splitK"])))
    mask = torch.ones_like(bufD, dtype=torch.bool)
    mask[guard: guard + (m + pad[2]) * n * batch].view(batch, n, m + pad[2])[:, :, :m] = False
    assert torch.isnan(bufD[mask]).all(), (mA, mB, m, n, k, d, "stored outside D")
    plan.destroy()
    return d, D.clone()

def run_modes(mA, mB, mC, ext, alpha=1.0, beta=0.0, seed=0, algo=None, expect_untouched=False):
    """packed tensors, any number of modes per group (modes listed fastest first, as the ABI takes them)"""
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    A = torch.rand([ext[c] for c in mA][::-1], generator=g, device="cuda")
    B = torch.rand([ext[c] for c in mB][::-1], generator=g, device="cuda")
    C = torch.rand([ext[c] for c in mC][::-1], generator=g, device="cuda")
    D = torch.full_like(C, float("nan"))
    kw = {} if algo is None else dict(algo=algo)
    plan = ops.contraction_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [ext[c] for c in mC], mC, dtype=ct.R_32F, workspace_limit=1 << 28, **kw)
    d = plan.describe()
    if expect_untouched and d["splitK"] > 1:
        plan.destroy()
        return d
    w = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    plan.contract(alpha, A.data_ptr(), B.data_ptr(), beta, C.data_ptr(), D.data_ptr(), w.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    if expect_untouched:
        assert torch.isnan(D).all(), (mA, mB, mC, d, "the row epilogue is not the path of this plan")
    else:
        ref = alpha * torch.einsum("%s,%s->%s" % (mA[::-1], mB[::-1], mC[::-1]), A.double(), B.double()) + beta * C.double()
        np.testing.assert_allclose(D.double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, err_msg=str((mA, mB, mC, ext, d["kname"], d["bm"], d["splitK"])))
    plan.destroy()
    return d

# several modes per output group (cuTENSOR/contraction.cu:46-59 is such a shape): D's fastest mode a multiple of 4 long -> the row epilogue
# decodes a row's / a lane's offset digit by digit; fastest mode NOT a multiple of 4 -> the direct form
MULTI = (("mhkn", "ukvh", "munv", dict(m=24, n=20, u=12, v=8, h=16, k=16)),        # contraction.cu's default equation, shrunk
         ("kam", "nkb", "abnm", dict(a=8, b=12, m=36, n=28, k=64)),                # D's fastest mode from A's group
         ("kma", "bnk", "nbma", dict(a=4, b=4, m=52, n=40, k=96)),
         ("mkl", "nlk", "nml", dict(m=72, n=132, k=32, l=3)),
         ("kam", "nkb", "abnm", dict(a=6, b=10, m=36, n=28, k=64)))               # extent 6: no 16-byte lanes in D

LAYOUTS = (("mk", "kn"), ("km", "kn"), ("mk", "nk"), ("km", "nk"))
MODE = os.environ.get("ROWS_TEST_MODE", "parity")
if MODE == "parity":
    SHAPES = ((260, 132, 96), (100, 52, 128), (384, 200, 32), (64, 64, 160), (4, 8, 64), (1028, 36, 96), (132, 260, 64), (128, 128, 128))
    kernels = set()
    n_run = 0
    for (mA, mB) in LAYOUTS:
        for i, (m, n, k) in enumerate(SHAPES):
            for r in range(8):                       # the planner's first candidates: ring kernels and register-staged ones, tiles 32 .. 128
                d, _ = run(mA, mB, m, n, k, algo=r, seed=10 * i + r)
                kernels.add((d["kname"], d["bm"])); n_run += 1
        for r in range(6):
            run(mA, mB, 260, 132, 96, algo=r, alpha=1.5, beta=-0.75, seed=77)                  # C of its own
            run(mA, mB, 264, 136, 128, algo=r, pad=(5, 3, 4), beta=0.5, seed=79)               # padded pitches, D / C keep 16-byte lanes
            run(mA, mB, 264, 136, 128, algo=r, pad=(2, 6, 12), seed=80)
            run(mA, mB, 264, 136, 128, algo=r, pad=(0, 0, 1), beta=1.0, seed=81)               # pitch without lanes: the direct form
            run(mA, mB, 132, 68, 64, algo=r, batch=3, beta=0.25, seed=82)                      # batch mode
            run(mA, mB, 132, 68, 64, algo=r, batch=5, pad=(0, 0, 4), seed=83)
        # same problem on a 16-byte-aligned and on a 12-byte-aligned base: row form against direct form, bit for bit
        for r in range(4):
            _, a = run(mA, mB, 260, 132, 96, algo=r, alpha=1.25, beta=0.5, guard=4, seed=90)
            _, b = run(mA, mB, 260, 132, 96, algo=r, alpha=1.25, beta=0.5, guard=3, seed=90)
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (mA, mB, r, "the two epilogue forms differ in bits")
    for i, (mA, mB, mC, ext) in enumerate(MULTI):
        for r in range(8):
            d = run_modes(mA, mB, mC, ext, algo=r, seed=100 + i)
            run_modes(mA, mB, mC, ext, algo=r, alpha=-0.5, beta=0.75, seed=200 + i)
            kernels.add((d["kname"], d["bm"])); n_run += 2
    assert any(k[0] == "gett_f32_stream_kernel" for k in kernels) and any(k[0] == "gett_f32_kernel" for k in kernels), kernels
    print("OK", n_run, sorted(kernels))
else:
    # measurement switch of the hooks flavour: the row epilogue skips its stores — D stays NaN exactly when the row epilogue is the path
    n = 0
    for (mA, mB) in LAYOUTS:
        for r in range(6):
            d, _ = run(mA, mB, 260, 132, 96, algo=r, seed=5, expect_untouched=True); n += 1
            d, _ = run(mA, mB, 132, 68, 64, algo=r, batch=3, beta=0.25, seed=6, expect_untouched=True); n += 1
    for (mA, mB, mC, ext) in MULTI[:4]:
        for r in range(6):
            run_modes(mA, mB, mC, ext, algo=r, beta=0.5, seed=7, expect_untouched=True); n += 1
    print("OK", n)
'''


def _run(env_extra):
    env = dict(os.environ, PYTHONPATH=ROOT, **env_extra)
    r = subprocess.run([sys.executable, "-c", CODE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1].startswith("OK"), r.stdout[-2000:]


def test_row_epilogue_parity():
    _run({})


def test_row_epilogue_is_the_path_of_flat_outputs():
    """hooks flavour only (CUTENSOR_AMD_PARTIAL_STORE=s makes gett_store_tile_f32_rows skip its stores): every candidate of a flat fp32
    problem with 16-byte lanes in D leaves D untouched, i.e. takes the row epilogue"""
    _run({"CTAMD_LIB_FLAVOUR": "hooks", "CUTENSOR_AMD_PARTIAL_STORE": "s", "ROWS_TEST_MODE": "path"})

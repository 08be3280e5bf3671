"""GPU parity of the PERSISTENT 16-bit kernel (csrc/kernels/gett_h16p.hip: gett_h16w4p_kernel, round 5) through the C ABI against
torch in fp64 on inputs already rounded to the 16-bit type (tolerances of tests/test_gpu_h16.py: bf16 rtol 8e-3, fp16 2e-3).

What is special about this kernel is what happens BETWEEN tiles, so the cases run in a child process whose grid is capped at eight
workgroups (CUTENSOR_AMD_H16P_GRID=8): every workgroup then walks several tiles — interior tiles whose epilogue (transposed 16-bit
image + ds_read_b64_tr_b16, in the LDS beyond the ring) runs while the next tile's first K-tiles are already landing in the ring;
edge tiles, beta != 0 and strided outputs, which take the ring-resident epilogues and stage the next tile afterwards; mixtures of both
in one workgroup's sequence; split-K partials; one, two, three and many K-tiles per tile; batch modes; all four operand layouts; fp16.
Interior tiles with an even K-tile count are STREAMED: the last two K-tile bodies of a tile issue the next tile's first two K-tiles (the
odometer is handed over inside the main loop), so the cases with 2 / 4 / 6 / 8 K-tiles per tile cover the hand-over in the first, a
middle and the peeled body, and the ragged ones the fall-back when the next tile is an edge tile.
A second child runs the same list with the default grid (one workgroup per CU)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # (extents, modes A, modes B, modes C, dtype, alpha, beta, note)
    (dict(m=1024, n=768, k=256), "mk", "kn", "mn", "bfloat16", 1.0, 0.0, "12 interior tiles, 4 K-tiles"),
    (dict(m=1024, n=768, k=256), "km", "kn", "mn", "bfloat16", 1.0, 0.0, "both K-contiguous"),
    (dict(m=1024, n=768, k=256), "mk", "nk", "mn", "bfloat16", 0.5, 0.0, "both free-contiguous, alpha"),
    (dict(m=1024, n=768, k=256), "km", "nk", "mn", "bfloat16", 1.0, 0.0, "K-contiguous A, free-contiguous B"),
    (dict(m=768, n=1280, k=64), "mk", "kn", "mn", "bfloat16", 1.0, 0.0, "one K-tile per tile"),
    (dict(m=768, n=1280, k=128), "mk", "kn", "mn", "bfloat16", 1.0, 0.0, "two K-tiles"),
    (dict(m=768, n=1280, k=192), "mk", "kn", "mn", "bfloat16", 1.0, 0.0, "three K-tiles (odd: the tail body)"),
    (dict(m=1000, n=712, k=320), "mk", "kn", "mn", "bfloat16", 1.0, 0.0, "ragged M and N: interior and edge tiles in one sequence"),
    (dict(m=1024, n=768, k=256), "mk", "kn", "mn", "bfloat16", 1.25, -0.5, "beta != 0, interior tiles: C through the row image (round 6)"),
    (dict(m=2048, n=1024, k=512), "km", "nk", "mn", "bfloat16", 0.5, 0.5, "beta != 0, 32 streamed tiles of eight K-tiles"),
    (dict(m=1536, n=1280, k=384), "mk", "nk", "mn", "float16", 1.0, -1.0, "fp16, beta != 0, six K-tiles"),
    (dict(m=1280, n=1000, k=256), "km", "kn", "mn", "bfloat16", 1.0, 2.0, "beta != 0, interior tiles streaming into edge tiles and back"),
    (dict(m=768, n=1280, k=192), "mk", "kn", "mn", "bfloat16", 1.0, 1.0, "beta != 0, three K-tiles (nothing streams)"),
    (dict(m=1024, n=512, k=256), "mk", "kn", "nm", "bfloat16", 1.0, 0.0, "D with n fastest (orientation swap)"),
    (dict(m=512, n=512, k=128, l=5), "mkl", "knl", "mnl", "bfloat16", 1.0, 0.0, "batch mode: 20 tiles over 5 batches"),
    (dict(m=512, n=768, k=256, l=7), "mkl", "knl", "mnl", "bfloat16", 1.0, 0.0, "batch mode, four K-tiles: 42 tiles stream across batch boundaries (round 6)"),
    (dict(m=768, n=512, k=384, l=3), "kml", "nkl", "mnl", "float16", 0.5, 0.75, "batch mode, fp16, beta != 0, six K-tiles"),
    (dict(m=1024, n=1280, k=128), "km", "nk", "mn", "bfloat16", 0.5, 0.5, "two K-tiles per tile, beta != 0: every tile after the first streamed in through the pair that is also the hand-over (round 6)"),
    (dict(m=512, n=512, k=128, l=9), "kml", "knl", "mnl", "float16", 1.0, 0.0, "two K-tiles, batch mode, fp16"),
    (dict(m=1024, n=1280, k=64), "mk", "kn", "mn", "bfloat16", 1.0, 0.0, "ONE K-tile per tile: a zero K-tile is appended, every tile streams (round 6)"),
    (dict(m=768, n=1024, k=64, l=6), "kml", "nkl", "mnl", "float16", 0.5, 0.5, "one K-tile, batch mode, fp16, beta != 0"),
    (dict(m=1280, n=1024, k=320), "km", "kn", "mn", "bfloat16", 1.0, 0.0, "five K-tiles: padded to six"),
    (dict(m=1000, n=712, k=192), "mk", "nk", "mn", "bfloat16", 1.0, 1.0, "three K-tiles, ragged M and N, beta: padded interior tiles streaming into edge tiles"),
    (dict(m=1024, n=768, k=256), "mk", "kn", "mn", "float16", 1.0, 0.0, "fp16"),
    (dict(m=1000, n=712, k=192), "km", "nk", "mn", "float16", 0.75, 0.25, "fp16, ragged, beta"),
    (dict(m=2048, n=1024, k=128), "mk", "kn", "mn", "bfloat16", 1.0, 0.0, "32 interior tiles of two K-tiles: streamed, the hand-over in the first K-tile body"),
    (dict(m=2048, n=1024, k=512), "km", "nk", "mn", "bfloat16", 0.5, 0.0, "32 interior tiles of eight K-tiles: every tile after the first streamed in"),
    (dict(m=1536, n=1280, k=384), "mk", "nk", "mn", "float16", 1.0, 0.0, "fp16, 30 interior tiles of six K-tiles"),
    (dict(m=1280, n=1000, k=256), "km", "kn", "mn", "bfloat16", 1.0, 0.0, "interior tiles streaming into edge tiles and back"),
    (dict(m=512, n=256, k=8192), "mk", "kn", "mn", "bfloat16", 1.0, 0.0, "split-K partials (two tiles, long K)"),
    (dict(a=96, b=16, c=16, d=64, e=96), "dcba", "ebcd", "ea", "bfloat16", 1.0, 0.0, "the headline equation's view: one tile, split-K"),
]


def _child():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from cudalibrarysamples_amd import cutensor as ct, ops
    h = ops.Handle()
    g = torch.Generator(device="cuda")
    for idx, (ext, mA, mB, mC, dt, alpha, beta, note) in enumerate(CASES):
        tdt = getattr(torch, dt)
        cdt = ct.R_16BF if dt == "bfloat16" else ct.R_16F
        g.manual_seed(100 + idx)
        eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
        A = (torch.rand(eA[::-1], generator=g, device="cuda") * 2 - 1).to(tdt)
        B = (torch.rand(eB[::-1], generator=g, device="cuda") * 2 - 1).to(tdt)
        C = (torch.rand(eC[::-1], generator=g, device="cuda") * 2 - 1).to(tdt)
        D = C.clone()
        plan = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, dtype=cdt, workspace_limit=1 << 28)
        d = plan.describe()
        assert d["kname"] == "gett_h16w4p_kernel", (note, d)
        ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
        for rep in range(2):       # twice: a second launch finds the LDS as the first one left it
            D.copy_(C)
            plan.contract(alpha, A.data_ptr(), B.data_ptr(), beta, C.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
        torch.cuda.synchronize()
        ref = alpha * torch.einsum("%s,%s->%s" % (mA[::-1], mB[::-1], mC[::-1]), A.double(), B.double()) + beta * C.double()
        rtol = 8e-3 if dt == "bfloat16" else 2e-3
        err = (D.double() - ref).abs()
        tol = rtol * ref.abs() + (0.25 if dt == "bfloat16" else 0.03) * (1.0 if ext.get("k", 4096) > 2048 or "a" in ext else 0.2)
        bad = int((err > tol).sum())
        assert bad == 0, "%s: %d / %d elements off, worst %g (plan %s)" % (note, bad, err.numel(), float((err - tol).max()), d)
        print("ok  %-70s splitK %d" % (note, d["splitK"]), flush=True)
        plan.destroy()
    print("H16P_CHILD_OK")


@pytest.mark.parametrize("grid", ["8", ""])
def test_persistent_kernel_parity(built, grid):
    env = dict(os.environ, CUTENSOR_AMD_H16_WAVES="4p")
    if grid:
        env["CUTENSOR_AMD_H16P_GRID"] = grid
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and "H16P_CHILD_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


_BITS_CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
g = torch.Generator(device="cuda")
out = {}
cases = [(2048, 1024, 512, "bfloat16", 0.75, -0.5, "own"), (1536, 1280, 256, "float16", 1.0, 1.0, "own"), (1280, 1000, 384, "bfloat16", 1.0, 0.25, "own"),
         (2048, 1024, 256, "bfloat16", 1.0, 1.0, "inplace"), (1024, 1024, 256, "bfloat16", 0.5, 0.5, "pitch")]
for idx, (M, N, K, dt, alpha, beta, how) in enumerate(cases):
    tdt = getattr(torch, dt)
    g.manual_seed(40 + idx)
    A = (torch.rand((K, M), generator=g, device="cuda") * 2 - 1).to(tdt)     # "mk": m fastest
    B = (torch.rand((N, K), generator=g, device="cuda") * 2 - 1).to(tdt)     # "kn"
    Cbig = (torch.rand((N, M + 24), generator=g, device="cuda") * 2 - 1).to(tdt)
    C = Cbig[:, :M] if how == "pitch" else Cbig[:, :M].contiguous()          # [N][M], m fastest; "pitch": rows of C 24 elements longer than D's
    D = C.clone() if how != "pitch" else torch.empty((N, M), device="cuda", dtype=tdt)
    p = ops.contraction_plan(h, [M, K], "mk", [K, N], "kn", [M, N], "mn", dtype=ct.R_16BF if dt == "bfloat16" else ct.R_16F,
                             strideC=[1, C.stride(0)], strideD=[1, D.stride(0)], workspace_limit=0)
    want = sys.argv[2]
    assert p.describe()["kname"] == want and p.describe()["splitK"] == 1, p.describe()      # no split-K: the epilogue with C is under test
    cptr = D.data_ptr() if how == "inplace" else C.data_ptr()
    p.contract(alpha, A.data_ptr(), B.data_ptr(), beta, cptr, D.data_ptr())
    torch.cuda.synchronize()
    ref = alpha * (B.double() @ A.double()) + beta * C.double()
    err = (D.double() - ref).abs()
    tol = (8e-3 if dt == "bfloat16" else 2e-3) * ref.abs() + 5e-2
    assert bool((err <= tol).all()), (idx, float((err - tol).max()))
    out["d%%d" %% idx] = D.view(torch.int16).cpu()
    out["ref%%d" %% idx] = ref.cpu()
    p.destroy()
torch.save(out, sys.argv[1])
print("BITS_CHILD_OK")
''' % ROOT


def test_beta_path_gives_the_bits_of_the_one_tile_kernel(built, tmp_path):
    """beta != 0 in the persistent kernel's streaming epilogue (round 6: C through the idle row image, joined with the accumulators in fp32
    before the one rounding) against the one-tile twin's fp32-image epilogue: same main loop, same fma(beta, c, alpha * acc) — the SAME
    bits, on C of its own, C = D in place, and a C whose rows are longer than D's.  Both children also check against fp64."""
    outs = []
    for waves, kname in (("4p", "gett_h16w4p_kernel"), ("4x", "gett_h16w4x_kernel")):
        f = str(tmp_path / ("bits_%s.pt" % waves))
        r = subprocess.run([sys.executable, "-c", _BITS_CHILD, f, kname], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, CUTENSOR_AMD_H16_WAVES=waves), cwd=ROOT)
        assert r.returncode == 0 and "BITS_CHILD_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
        outs.append(f)
    import torch
    a, b = torch.load(outs[0]), torch.load(outs[1])
    assert sorted(a) == sorted(b) and len(a) == 10
    for k in a:
        if k.startswith("d") and not torch.equal(a[k], b[k]):
            idx = (a[k] != b[k]).nonzero()[:8]
            fdt = torch.float16 if k == "d1" else torch.bfloat16
            vals = [(tuple(int(x) for x in i), float(a[k].view(fdt)[tuple(i)]), float(b[k].view(fdt)[tuple(i)]), float(a["ref" + k[1:]][tuple(i)])) for i in idx]
            raise AssertionError((k, int((a[k] != b[k]).sum()), "(index, persistent, one-tile, fp64)", vals))


def test_many_rounds_at_full_grid(built):
    """More tiles than CUs without the grid cap: 4352 x 4352 x 128 = 289 tiles on 256 workgroups (33 of them walk two tiles), checked by
    sampled fp64 dot products; and the size class of the bench line at reduced K (8192 x 8192 x 256: 1024 tiles, four rounds)."""
    env = dict(os.environ, CUTENSOR_AMD_H16_WAVES="4p")
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
g = torch.Generator(device="cuda"); g.manual_seed(5)
for (M, N, K) in ((4352, 4352, 128), (8192, 8192, 256)):
    A = (torch.rand((K, M), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)     # "mk": m fastest
    B = (torch.rand((N, K), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)     # "kn"
    D = torch.empty((N, M), device="cuda", dtype=torch.bfloat16)
    p = ops.contraction_plan(h, [M, K], "mk", [K, N], "kn", [M, N], "mn", dtype=ct.R_16BF)
    assert p.describe()["kname"] == "gett_h16w4p_kernel", p.describe()
    p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
    torch.cuda.synchronize()
    rows = torch.tensor([0, 255, 256, 1000, 2047, 2048, 4095, 4096, M - 257, M - 256, M - 1], device="cuda")
    ref = B.double() @ A.double()[:, rows]                      # [N][len(rows)]
    err = float((D[:, rows].double() - ref).abs().max() / ref.abs().max())
    assert err < 8e-3, (M, N, K, err)
    cols = torch.tensor([0, 255, 256, 3000, N - 256, N - 1], device="cuda")
    ref = B.double()[cols] @ A.double()
    err = float((D[cols].double() - ref).abs().max() / ref.abs().max())
    assert err < 8e-3, (M, N, K, err)
print("MANY_ROUNDS_OK")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and "MANY_ROUNDS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


if __name__ == "__main__":
    _child()


def test_beta_nonzero_on_a_plan_of_the_persistent_kernel(built):
    """beta is known only at the call.  Since round 6 the persistent kernel streams its tiles with beta != 0 too when C has the 16-byte
    lanes of D; with any other C every tile would take the ring-resident epilogue, where it is slower than its one-tile twin
    (profiles/r05r_h16p_beta.jsonl), and cutensorContract launches the twin — same tile, same arguments (api.cpp; not when
    CUTENSOR_AMD_H16_WAVES=4p names the kernel).  Here: the planner's own choice for a two-round shape, beta = 0 and beta != 0 through
    the SAME plan, against fp64, with the kernel that ran read back (ct.last_h16_kernel)."""
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    if os.environ.get("CUTENSOR_AMD_H16_WAVES"):
        pytest.skip("the planner's own choice is under test")
    h = ops.Handle()
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    M, N, K = 4352, 4352, 256          # four K-tiles: the smallest count whose tiles stream (pick_h16_choice)
    A = (torch.rand((K, M), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)     # "mk": m fastest
    B = (torch.rand((N, K), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)     # "kn"
    C = (torch.rand((N, M), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
    p = ops.contraction_plan(h, [M, K], "mk", [K, N], "kn", [M, N], "mn", dtype=ct.R_16BF)
    assert p.describe()["kname"] == "gett_h16w4p_kernel", p.describe()
    ab = B.double() @ A.double()                                                       # [N][M]
    for alpha, beta in ((1.0, 0.0), (0.75, -0.5), (1.0, 1.0)):
        D = C.clone()
        p.contract(alpha, A.data_ptr(), B.data_ptr(), beta, C.data_ptr(), D.data_ptr())
        torch.cuda.synchronize()
        assert 88 <= ct.last_h16_kernel() < 96, ct.last_h16_kernel()      # round 6: the persistent kernel itself, beta or not
        ref = alpha * ab + beta * C.double()
        err = (D.double() - ref).abs()
        tol = 8e-3 * ref.abs() + 3e-2
        assert bool((err <= tol).all()), (alpha, beta, float((err - tol).max()))
    p.destroy()
    # a C without the 16-byte lanes of D (m is not its stride-1 mode): beta == 0 on the persistent kernel, beta != 0 on the one-tile twin
    Ct = (torch.rand((M, N), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)     # element (m, n) at m * N + n
    p = ops.contraction_plan(h, [M, K], "mk", [K, N], "kn", [M, N], "mn", dtype=ct.R_16BF, strideC=[N, 1], strideD=[1, M])
    assert p.describe()["kname"] == "gett_h16w4p_kernel", p.describe()
    for alpha, beta, lo in ((1.0, 0.0, 88), (1.0, 0.5, 48)):
        D = torch.zeros((N, M), device="cuda", dtype=torch.bfloat16)
        p.contract(alpha, A.data_ptr(), B.data_ptr(), beta, Ct.data_ptr(), D.data_ptr())
        torch.cuda.synchronize()
        assert lo <= ct.last_h16_kernel() < lo + 8, (beta, ct.last_h16_kernel())
        ref = alpha * ab + beta * Ct.double().t()
        err = (D.double() - ref).abs()
        tol = 8e-3 * ref.abs() + 3e-2
        assert bool((err <= tol).all()), (alpha, beta, float((err - tol).max()))
    p.destroy()

"""GPU parity: cutensorContract through the C ABI vs the CPU oracle (fp64 accumulation) on the same
seeded tensors.  fp32 tolerance: rtol 1e-4 of the result magnitude; the reference's own
bound is rtol 5e-3 / atol 6e-3 (einsum_test.py:35-42)."""
import ctypes

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


def run_contraction(env, extents, mA, mB, mC, alpha=1.0, beta=0.0, seed=0, algo=None, rank=0, ws_limit=1 << 28,
                    rtol=1e-4, expect=None):
    ct, ops, h, torch = env
    eA, eB, eC = [extents[c] for c in mA], [extents[c] for c in mB], [extents[c] for c in mC]
    A, B, C = make_tensor(eA, seed + 1), make_tensor(eB, seed + 2), make_tensor(eC, seed + 3)
    kw = dict(workspace_limit=ws_limit, kernel_rank=rank)
    if algo is not None:
        kw["algo"] = algo
    p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, **kw)
    d = p.describe()
    if expect:
        for k, v in expect.items():
            assert d[k] == v, (k, d)
    dA, dB, dC = to_device(A), to_device(B), to_device(C)
    ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
    p.contract(alpha, dA.data_ptr(), dB.data_ptr(), beta, dC.data_ptr(), dC.data_ptr(), ws.data_ptr(),
               p.required_workspace, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = from_device(dC, C)
    ref = np.zeros_like(C)
    oracle.contract(A, mA, B, mB, ref, mC, alpha=alpha, beta=beta, C=C)
    scale = float(np.max(np.abs(ref))) if ref.size else 1.0
    assert_close(got, ref, rtol=rtol, atol=rtol * scale, what="%s,%s->%s %s" % (mA, mB, mC, d))
    p.destroy()
    return d


def test_contraction_sample_modes_shrunk(env):
    # contraction.cu:46-59 (C_{m,u,n,v} = alpha A_{m,h,k,n} B_{u,k,v,h} + beta C), alpha 1.1 (:184)
    ext = dict(m=24, n=12, u=16, v=8, h=8, k=12)
    run_contraction(env, ext, "mhkn", "ukvh", "munv", alpha=1.1, beta=0.0)
    run_contraction(env, ext, "mhkn", "ukvh", "munv", alpha=1.1, beta=0.7, seed=10)


def test_against_the_naive_fp32_host_loop(env):
    """BASELINE.json's wording: "results match a naive host triple-loop on the same random tensors".  The oracle's literal
    fp32 loop nest (oracle_contract_f32_naive: fp32 accumulation in loop order) on the contraction.cu modes, shrunk, and on
    the headline equation, shrunk: the GPU's blocked fp32 accumulation and the sequential host loop both sit within fp32
    round-off of the fp64 value, so they agree with each other to 2e-5 of the result magnitude."""
    ct, ops, h, torch = env
    for ext, mA, mB, mC in [(dict(m=24, n=12, u=16, v=8, h=8, k=12), "mhkn", "ukvh", "munv"),
                            (dict(a=24, b=8, c=8, d=16, e=24), "dcba", "ebcd", "ea")]:
        eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
        A, B, C = make_tensor(eA, 31), make_tensor(eB, 32), make_tensor(eC, 33)
        p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, workspace_limit=1 << 28)
        dA, dB, dC = to_device(A), to_device(B), to_device(C)
        ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
        p.contract(1.1, dA.data_ptr(), dB.data_ptr(), 0.5, dC.data_ptr(), dC.data_ptr(), ws.data_ptr(), p.required_workspace)
        torch.cuda.synchronize()
        naive = np.zeros_like(C)
        oracle.contract(A, mA, B, mB, naive, mC, alpha=1.1, beta=0.5, C=C, acc64=False)
        got = from_device(dC, C)
        assert_close(got, naive, rtol=2e-5, atol=2e-5 * float(np.abs(naive).max()), what="naive fp32 loop %s,%s->%s" % (mA, mB, mC))
        p.destroy()


@pytest.mark.parametrize("case", [
    # (extents, modesA, modesB, modesC): every operand-layout combination of the GETT kernels
    (dict(m=64, n=48, k=40), "mk", "kn", "mn"),      # A free-contig (after swap), B K-contig
    (dict(m=64, n=48, k=40), "km", "kn", "mn"),      # both K-contiguous
    (dict(m=64, n=48, k=40), "mk", "nk", "mn"),      # both free-contiguous
    (dict(m=64, n=48, k=40), "km", "nk", "nm"),      # output n-contiguous
    (dict(m=33, n=17, k=29), "mk", "kn", "mn"),      # odd extents -> scalar-gather kernels
    (dict(m=100, n=36, k=52), "km", "nk", "mn"),     # M, N not multiples of the tile
    (dict(a=8, b=12, c=16, d=20, e=24), "dcba", "ebcd", "ea"),   # headline equation, shrunk
    (dict(l=6, i=20, j=24, k=28), "kil", "jkl", "jil"),          # batch mode (lik,lkj->lij reversed)
    (dict(l=5, i=12, j=16, k=8, m=4), "mkil", "mjkl", "jil"),    # likm,lkjm->lij reversed
    (dict(i=40, j=56), "i", "j", "ij"),                          # outer product (no contracted mode)
    (dict(i=48, k=64), "ik", "k", "i"),                          # matrix-vector
    (dict(k=4096), "k", "k", ""),                                # dot product -> scalar
    (dict(a=4, b=1, c=16, d=8), "abd", "dbc", "ac"),             # extent-1 mode
])
def test_contraction_layouts(env, case):
    ext, mA, mB, mC = case
    run_contraction(env, ext, mA, mB, mC, alpha=1.0, beta=0.0)
    run_contraction(env, ext, mA, mB, mC, alpha=-0.5, beta=2.0, seed=7)


def test_every_candidate_kernel_and_split(env):
    """Sweep all ranked (kernel, split-K) candidates of a set of problems chosen so that every
    instantiated GETT kernel (all layout pairs, fast-K and generic-K addressing, every tile shape and
    prefetch depth) and the split-K fold are exercised; each must agree with the oracle."""
    ct, ops, h, torch = env
    problems = [
        (dict(a=96, b=4, c=4, d=64, e=96), "dcba", "ebcd", "ea"),    # LAY_K x LAY_F, fast-K, 3 K modes
        (dict(a=96, b=3, c=4, d=64, e=96), "dcba", "ebcd", "ea"),    # 24 K-tiles: ring depths 3, 4 and 6
        (dict(m=160, n=144, k=256), "mk", "nk", "mn"),               # LAY_F x LAY_F, fast-K
        (dict(m=160, n=144, k=256), "km", "kn", "mn"),               # LAY_K x LAY_K, fast-K
        (dict(m=144, n=160, k=256), "mk", "kn", "nm"),               # LAY_F x LAY_K, fast-K
        (dict(a=40, b=6, c=20, e=56), "cba", "ebc", "ea"),           # LAY_K x LAY_F, generic K (20 % 32 != 0)
        (dict(m=72, n=40, k=12, j=9), "mkj", "nkj", "mn"),           # LAY_F x LAY_F, generic K
        (dict(m=72, n=40, k=12, j=9), "kjm", "kjn", "mn"),           # LAY_K x LAY_K, generic K
        (dict(m=40, n=72, k=12, j=9), "mkj", "kjn", "nm"),           # LAY_F x LAY_K, generic K
        (dict(m=70, n=50, k=300), "mk", "kn", "mn"),                 # LAY_S x LAY_S
    ]
    seen = set()
    for ext, mA, mB, mC in problems:
        eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
        p0 = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, workspace_limit=1 << 28)
        n = ct.lib.ctamdCountCandidates(h.h, p0.op, 1 << 28)
        p0.destroy()
        assert n > 0
        for r in range(n):
            d = run_contraction(env, ext, mA, mB, mC, alpha=1.25, beta=0.5, algo=r, seed=r)
            seen.add((d["kernel"], d["splitK"] > 1))
    kernels = {k for k, _ in seen}
    planned = {i for i in range(ct.lib.ctamdKernelCount()) if not ct.lib.ctamdKernelIsAblation(i)}
    missing = planned - kernels
    # what is left are the nontemporal-stream twins of the ring kernel: ranked only for read-once problems beyond the Infinity
    # Cache (test_read_once_contraction_beyond_the_infinity_cache_streams_nontemporal); a child process with CUTENSOR_AMD_NT set
    # (read once per process) sweeps them on small one-tile shapes of all four layouts
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import torch, test_gpu_contraction as t\n"
        "from cudalibrarysamples_amd import cutensor as ct, ops\n"
        "env = (ct, ops, ops.Handle(), torch)\n"
        "seen = set()\n"
        "for ext, mA, mB, mC in [(dict(m=96, n=96, k=512), 'km', 'kn', 'mn'), (dict(m=96, n=96, k=512), 'mk', 'nk', 'mn'),\n"
        "                        (dict(m=96, n=96, k=512), 'mk', 'kn', 'nm'), (dict(a=96, b=4, c=4, d=64, e=96), 'dcba', 'ebcd', 'ea')]:\n"
        "    p0 = ops.contraction_plan(env[2], [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [ext[c] for c in mC], mC, workspace_limit=1 << 28)\n"
        "    n = ct.lib.ctamdCountCandidates(env[2].h, p0.op, 1 << 28); p0.destroy()\n"
        "    for r in range(n):\n"
        "        seen.add(t.run_contraction(env, ext, mA, mB, mC, alpha=1.25, beta=0.5, algo=r, seed=r)['kernel'])\n"
        "print('KERNELS', sorted(seen))\n") % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, CUTENSOR_AMD_NT="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    nt_seen = set(eval(r.stdout.split("KERNELS", 1)[1].strip()))
    missing -= nt_seen
    assert not missing, "GETT kernels never exercised: %s" % sorted(missing)
    assert any(s for _, s in seen)


def test_strided_descriptors_and_separate_output(env):
    """Non-packed strides (sub-tensor views) and D != C."""
    ct, ops, h, torch = env
    big = make_tensor([40, 24, 36], 3)
    A = big[4:36, :, 2:34:2]               # extents 32,24,16 strides 1,40,1920
    B = make_tensor([24, 16, 20], 4)
    C = make_tensor([32, 20], 5)
    strideA = [s // 4 for s in A.strides]
    p = ops.contraction_plan(h, list(A.shape), "ijk", list(B.shape), "jkl", list(C.shape), "il", strideA=strideA,
                             alignment=16)
    dbig, dB, dC = to_device(big), to_device(B), to_device(C)
    dD = torch.zeros_like(dC)
    offset = (A.ctypes.data - big.ctypes.data)
    p.contract(2.0, dbig.data_ptr() + offset, dB.data_ptr(), -1.0, dC.data_ptr(), dD.data_ptr(), 0, 0, 0)
    torch.cuda.synchronize()
    ref = np.zeros_like(C)
    oracle.contract(np.asfortranarray(A), "ijk", B, "jkl", ref, "il", alpha=2.0, beta=-1.0, C=C)
    assert_close(from_device(dD, C), ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()), what="strided")
    assert np.array_equal(from_device(dC, C), C)          # C untouched when D != C


def test_headline_einsum_full_size(env):
    """BASELINE config 2 at full size: 'abcd,dcbe->ae', 96/64/64/64/96, U(0,1) fp32, vs the fp64
    oracle on every output element (201 MB of inputs, 4.8 GFLOP)."""
    ct, ops, h, torch = env
    from cudalibrarysamples_amd import torch_einsum
    rng = np.random.default_rng(2024)
    a = rng.random((96, 64, 64, 64), dtype=np.float32)
    b = rng.random((64, 64, 64, 96), dtype=np.float32)
    got = torch_einsum.einsum("abcd,dcbe->ae", torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    torch.cuda.synchronize()
    ref = a.reshape(96, -1).astype(np.float64) @ np.ascontiguousarray(b.transpose(2, 1, 0, 3)).reshape(-1, 96).astype(np.float64)
    # spot-check the oracle itself on one row against this fp64 TTGT (the oracle at full size takes ~10 s)
    row = np.zeros((1, 96), dtype=np.float32)
    row_ref = oracle.einsum("abcd,dcbe->ae", a[5:6], b)
    np.testing.assert_allclose(row_ref[0], ref[5], rtol=1e-6)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4)
    del row


def test_read_once_contraction_beyond_the_infinity_cache_streams_nontemporal(env):
    """The headline equation with b = 128 (operands 2 x 201 MB = 1.5 x the 256-MiB Infinity Cache, one output tile: every operand
    byte is read exactly once): the planner takes the nontemporal-stream twin of the ring kernel (GettKernelInfo::nt; measured +6 %
    there, -5 % at b = 96 where most of the operands still fit); all 9216 outputs against fp64.  The headline itself (b = 64,
    2 x 100 MB) keeps the default policy: back-to-back calls find it on-die."""
    ct, ops, h, torch = env
    ext = dict(a=96, b=128, c=64, d=64, e=96)
    p = ops.contraction_plan(h, [ext[c] for c in "dcba"], "dcba", [ext[c] for c in "ebcd"], "ebcd", [96, 96], "ea", workspace_limit=1 << 30)
    d = p.describe()
    small = ops.contraction_plan(h, [64, 64, 64, 96], "dcba", [96, 64, 64, 64], "ebcd", [96, 96], "ea", workspace_limit=1 << 30)
    assert d["kname"] == "gett_f32_stream_kernel" and d["kernel"] != small.describe()["kernel"] and d["splitK"] > 1, (d, small.describe())
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    A = torch.rand((96, 128, 64, 64), generator=g, device="cuda")     # row-major [a][b][c][d] == column-major modes d, c, b, a
    B = torch.rand((64, 64, 128, 96), generator=g, device="cuda")     # [d][c][b][e]
    C = torch.empty((96, 96), device="cuda")
    ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
    p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), p.required_workspace)
    torch.cuda.synchronize()
    ref = torch.einsum("abcd,dcbe->ae", A.double(), B.double())
    torch.testing.assert_close(C.double(), ref, rtol=1e-4, atol=0.0)
    p.destroy()
    small.destroy()


def test_contraction_sample_full_size_sampled(env):
    """BASELINE config 1 shape on the GPU: contraction.cu defaults (464 GFLOP), alpha 1.1 beta 0;
    4096 sampled outputs against fp64 dot products."""
    ct, ops, h, torch = env
    ext = dict(m=96, n=96, u=96, v=64, h=64, k=64)
    mA, mB, mC = "mhkn", "ukvh", "munv"
    eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
    A, B = make_tensor(eA, 1234), make_tensor(eB, 1235)
    p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC)
    dA, dB = to_device(A), to_device(B)
    dC = torch.zeros(int(np.prod(eC)), dtype=torch.float32, device="cuda")
    ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
    p.contract(1.1, dA.data_ptr(), dB.data_ptr(), 0.0, dC.data_ptr(), dC.data_ptr(), ws.data_ptr(), p.required_workspace, 0)
    torch.cuda.synchronize()
    got = np.reshape(dC.cpu().numpy(), eC, order="F")
    rng = np.random.default_rng(99)
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    for _ in range(4096):
        m, u, n, v = (int(rng.integers(0, ext[c])) for c in "munv")
        ref = 1.1 * np.einsum("hk,kh->", A64[m, :, :, n], B64[u, :, v, :])
        assert abs(got[m, u, n, v] - ref) <= 1e-4 * abs(ref), (m, u, n, v, got[m, u, n, v], ref)


def test_headline_in_launch_fold_matches_two_kernel_fold(built):
    """The opt-in in-launch split-K fold (CUTENSOR_AMD_FUSED_FOLD=1) must give the same numbers as
    the default two-kernel fold up to fp32 re-association (both sum the same 256 partials in a fixed order)."""
    import os
    import torch
    from cudalibrarysamples_amd import ops
    ext = dict(a=96, b=64, c=64, d=64, e=96)
    h = ops.Handle()
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    A = torch.rand(96 * 64 * 64 * 64, generator=g, device="cuda")
    B = torch.rand(96 * 64 * 64 * 64, generator=g, device="cuda")
    outs = []
    for fused in ("0", "1"):
        os.environ["CUTENSOR_AMD_FUSED_FOLD"] = fused
        try:
            plan = ops.contraction_plan(h, [ext[c] for c in "dcba"], "dcba", [ext[c] for c in "ebcd"], "ebcd",
                                        [ext[c] for c in "ea"], "ea", workspace_limit=1 << 30)
        finally:
            os.environ.pop("CUTENSOR_AMD_FUSED_FOLD", None)
        d = plan.describe()
        assert d["fusedFold"] == int(fused), d
        ws = torch.empty(max(plan.required_workspace, 256), dtype=torch.uint8, device="cuda")
        C = torch.full((96 * 96,), 3.0, device="cuda")
        for _ in range(3):   # repeated launches re-arm the counters
            plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), plan.required_workspace)
        torch.cuda.synchronize()
        outs.append(C.cpu().numpy())
    np.testing.assert_allclose(outs[1], outs[0], rtol=2e-6)


def test_contract_inside_a_captured_graph(built):
    """cutensorContract is capture-safe (no allocation, no synchronisation): GETT + fold captured in a HIP graph
    and replayed give the eager result bit for bit."""
    import torch
    from cudalibrarysamples_amd import ops
    ext = dict(a=96, b=16, c=16, d=64, e=96)
    h = ops.Handle()
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    A = torch.rand(96 * 16 * 16 * 64, generator=g, device="cuda")
    B = torch.rand(96 * 16 * 16 * 64, generator=g, device="cuda")
    plan = ops.contraction_plan(h, [ext[c] for c in "dcba"], "dcba", [ext[c] for c in "ebcd"], "ebcd",
                                [ext[c] for c in "ea"], "ea", workspace_limit=1 << 30)
    ws = torch.empty(max(plan.required_workspace, 256), dtype=torch.uint8, device="cuda")
    eager = torch.zeros(96 * 96, device="cuda")
    plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, eager.data_ptr(), eager.data_ptr(), ws.data_ptr(), plan.required_workspace,
                  torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = torch.zeros(96 * 96, device="cuda")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, out.data_ptr(), out.data_ptr(), ws.data_ptr(), plan.required_workspace,
                      torch.cuda.current_stream().cuda_stream)
    out.zero_()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_many_mode_contraction_uses_the_mode_table_kernel(built, dtype):
    """Tensors with many small unfusable modes (the shape class of cuTENSOR/contraction_jit.cu:50-56: 25 / 13 / 24
    modes of extent 2) exceed the four digits per group of the tiled kernels and run on the mode-table kernel;
    parity against numpy.einsum in fp64 (fp32: rtol 1e-5 at K = 64)."""
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    h = ops.Handle()
    # A: 14 modes, B: 10, C: 12; six contracted modes; neighbouring modes are swapped so that nothing fuses
    mA = "badcfehgjilknm"
    mB = "ponmlkqrst"[::-1]
    mC = "".join(c for c in "abcdefghijopqrst" if (c in mA) != (c in mB))   # free modes in sorted order: (b,a) in A vs (a,b) in C ...
    ext = {c: (3 if c in "aq" else 2) for c in set(mA + mB)}
    np_dt = np.float32 if dtype == "float32" else np.float64
    A = make_tensor([ext[c] for c in mA], 31, np_dt, -1, 1)
    B = make_tensor([ext[c] for c in mB], 32, np_dt, -1, 1)
    C = make_tensor([ext[c] for c in mC], 33, np_dt, -1, 1)
    dA, dB, dC = to_device(A), to_device(B), to_device(C)
    cdt = ct.R_32F if dtype == "float32" else ct.R_64F
    plan = ops.contraction_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [ext[c] for c in mC], mC, dtype=cdt)
    d = plan.describe()
    assert d["kname"] == "gett_wide_kernel", d
    plan.contract(1.5, dA.data_ptr(), dB.data_ptr(), -0.5, dC.data_ptr(), dC.data_ptr())
    torch.cuda.synchronize()
    got = from_device(dC, C)
    ref = 1.5 * np.einsum("%s,%s->%s" % (mA, mB, mC), A.astype(np.float64), B.astype(np.float64)) - 0.5 * C
    np.testing.assert_allclose(got, ref, rtol=1e-5 if dtype == "float32" else 1e-13, atol=1e-5 if dtype == "float32" else 1e-13)


@pytest.mark.parametrize("case", [
    # (A modes, B modes, C modes, extents, expected peeled modes): K modes sit between A's free modes so that nothing fuses
    ("akblcmdef", "xkylm", "xfaebdcy", dict(a=6, b=5, c=4, d=3, e=7, f=2, k=8, l=6, m=5, x=9, y=4), 2),        # 6 free modes of A
    ("apbqcrdse", "xpqrsy", "axbycde", dict(a=5, b=4, c=6, d=3, e=8, p=2, q=3, r=4, s=5, x=7, y=6), 1),          # 5 free modes of A
    ("paqbrcsdte", "xpyqzrst", "abxcydze", dict(a=4, b=3, c=5, d=2, e=6, p=3, q=4, r=2, s=5, t=3, x=4, y=3, z=2), 2),   # 5 contracted + 5 free
])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_oversized_mode_groups_are_peeled_onto_the_tiled_kernels(built, case, dtype):
    """A group with five or six unfusable modes exceeds the tiled kernels' four digits per group.  Up to 64 launches the plan
    peels the smallest modes of that group into a host loop over ONE tiled inner plan (operands offset by index x stride; a
    peeled contracted mode accumulates into D, a peeled free mode writes its own region with the caller's beta) instead of
    handing the whole problem to the mode-table kernel (measured ~100x slower in cuTENSORMg's blog_post.cu layouts at 8 devices).
    Parity against numpy.einsum in fp64, alpha / beta != 1 / 0, C aliasing D."""
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    mA, mB, mC, ext, want_peeled = case
    h = ops.Handle()
    bf = dtype == "bfloat16"
    A = make_tensor([ext[c] for c in mA], 41, np.float32, -1, 1)
    B = make_tensor([ext[c] for c in mB], 42, np.float32, -1, 1)
    C = make_tensor([ext[c] for c in mC], 43, np.float32, -1, 1)
    tdt = torch.bfloat16 if bf else torch.float32
    dA, dB, dC = to_device(A).to(tdt), to_device(B).to(tdt), to_device(C).to(tdt)
    if bf:   # the reference values are computed from the rounded inputs
        A = np.reshape(dA.float().cpu().numpy(), A.shape, order="F")
        B = np.reshape(dB.float().cpu().numpy(), B.shape, order="F")
        C = np.reshape(dC.float().cpu().numpy(), C.shape, order="F")
    plan = ops.contraction_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [ext[c] for c in mC], mC,
                                dtype=ct.R_16BF if bf else ct.R_32F, workspace_limit=1 << 24)
    d = plan.describe()
    k_oversized = sum(1 for c in set(mA) & set(mB) if c not in mC) > 4
    if bf and k_oversized:
        # 16-bit data: a peeled contracted mode would accumulate through D (one rounding to bf16 per launch instead of one in all);
        # an oversized K group therefore stays with the mode-table kernel, which accumulates in full precision
        assert d["kname"] == "gett_wide_kernel" and "peeled_modes" not in d, d
    else:
        assert d.get("peeled_modes") == want_peeled and d["kname"] != "gett_wide_kernel" and 2 <= d["peel_launches"] <= 64, d
    ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    plan.contract(1.5, dA.data_ptr(), dB.data_ptr(), -0.5, dC.data_ptr(), dC.data_ptr(), ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    got = np.reshape(dC.float().cpu().numpy(), C.shape, order="F")
    ref = 1.5 * np.einsum("%s,%s->%s" % (mA, mB, mC), A.astype(np.float64), B.astype(np.float64)) - 0.5 * C
    scale = float(np.max(np.abs(ref)))
    np.testing.assert_allclose(got, ref, rtol=1e-2 if bf else 1e-5, atol=(1e-2 if bf else 1e-5) * scale)
    plan.destroy()


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
def test_complex_contraction_with_conjugation(built, dtype):
    """Complex data (cuTENSOR/contraction_jit.cu:31-41, std::complex<float> tensors and scalars) runs on the general MFMA
    family (four real MFMAs per complex product; CUTENSOR_AMD_GEN=0 or more than four modes per group: the mode-table
    kernel): D = alpha * conj(A) * B + beta * C with complex alpha / beta against numpy in complex128."""
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    h = ops.Handle()
    rng = np.random.default_rng(41)
    ext = dict(m=12, n=9, k=7, l=3)
    np_dt = np.complex64 if dtype == "complex64" else np.complex128
    def rnd(modes):
        shape = [ext[c] for c in modes]
        return (rng.random(shape) - 0.5 + 1j * (rng.random(shape) - 0.5)).astype(np_dt).copy(order="F")
    A, B, C = rnd("mkl"), rnd("knl"), rnd("mnl")
    tdt = torch.complex64 if dtype == "complex64" else torch.complex128
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x.ravel(order="K"))).to(tdt).cuda()
    dA, dB, dC = dev(A), dev(B), dev(C)
    cdt = ct.C_32F if dtype == "complex64" else ct.C_64F
    plan = ops.contraction_plan(h, [ext[c] for c in "mkl"], "mkl", [ext[c] for c in "knl"], "knl", [ext[c] for c in "mnl"], "mnl",
                                dtype=cdt, opA=ct.OP_CONJ)
    assert plan.scalar_type == cdt
    assert plan.describe()["kname"] == "gett_gen_kernel"
    alpha, beta = 1.1 - 0.3j, 0.25 + 0.5j
    plan.contract(alpha, dA.data_ptr(), dB.data_ptr(), beta, dC.data_ptr(), dC.data_ptr())
    torch.cuda.synchronize()
    got = dC.cpu().numpy().reshape(C.shape, order="F")
    ref = alpha * np.einsum("mkl,knl->mnl", np.conj(A).astype(np.complex128), B.astype(np.complex128)) + beta * C
    np.testing.assert_allclose(got, ref, rtol=2e-5 if dtype == "complex64" else 1e-13, atol=2e-6 if dtype == "complex64" else 1e-13)
    want = np.zeros_like(C)
    oracle.contract(A, "mkl", B, "knl", want, "mnl", alpha=alpha, beta=beta, C=C, conjA=True)      # the oracle's complex entry point
    np.testing.assert_allclose(got, want, rtol=2e-5 if dtype == "complex64" else 1e-13, atol=2e-6 if dtype == "complex64" else 1e-13)


def test_patient_algo_measures_candidates_and_stays_correct(env):
    """CUTENSOR_ALGO_DEFAULT_PATIENT: plan creation times the ranked candidates (GETT kernel + the fold that matches its
    partial layout) on scratch tensors and keeps the fastest; results must not depend on which one won."""
    ct, ops, h, torch = env
    d = run_contraction(env, dict(a=96, b=64, c=16, d=64, e=96), "dcba", "ebcd", "ea", algo=ct.ALGO_DEFAULT_PATIENT, ws_limit=1 << 30)
    assert d["kernel"] >= 0 and d["splitK"] >= 1
    run_contraction(env, dict(m=192, n=160, k=256), "km", "kn", "mn", alpha=0.5, beta=0.25, algo=ct.ALGO_DEFAULT_PATIENT)


def test_one_handle_shared_by_threads(built):
    """A process-wide handle is shared by framework threads (python/einsum.h:484-502): planning (with the plan cache on)
    and execution from four threads at once, each on its own stream, must give every thread its own correct result."""
    import threading
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    h = ops.Handle(plan_cache=64)
    shapes = [dict(m=96, n=64, k=512), dict(m=100, n=36, k=52), dict(m=256, n=256, k=128), dict(m=64, n=48, k=4096)]
    errors = []

    def body(tid):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            for it in range(6):
                ext = shapes[(tid + it) % len(shapes)]
                mA, mB, mC = ("km", "kn", "mn") if (tid + it) % 2 else ("mk", "nk", "mn")
                eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
                A, B = make_tensor(eA, 100 * tid + it), make_tensor(eB, 100 * tid + it + 50)
                p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, workspace_limit=1 << 26,
                                         algo=ct.ALGO_DEFAULT_PATIENT if it == 0 else ct.ALGO_DEFAULT)
                dA, dB = to_device(A), to_device(B)
                dC = torch.zeros(int(np.prod(eC)), dtype=torch.float32, device="cuda")
                ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                p.contract(1.0, dA.data_ptr(), dB.data_ptr(), 0.0, dC.data_ptr(), dC.data_ptr(), ws.data_ptr(), p.required_workspace,
                           stream.cuda_stream)
                stream.synchronize()
                ref = np.zeros(eC, dtype=np.float32, order="F")
                oracle.contract(A, mA, B, mB, ref, mC)
                got = np.reshape(dC.cpu().numpy(), eC, order="F")
                assert_close(got, ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()), what="thread %d iteration %d" % (tid, it))
                p.destroy()
        except BaseException as e:   # noqa: BLE001 — reported below
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=body, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("mA", ["km", "mk"])
def test_operand_larger_than_4_gib(built, mA):
    """Maximum sizes: an operand whose span exceeds 2^32 bytes (5.4 GB) must take the kernels with 64-bit offsets (the
    LDS-DMA ring kernels address 32-bit byte offsets and are excluded by the planner).  Checked by fp64 dot products of
    sampled outputs and by linearity in alpha."""
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    h = ops.Handle()
    M, N, K = 1024, 64, 5 << 18
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    # column-major "km" = torch [M][K] (k contiguous); "mk" = torch [K][M]
    A = torch.rand((M, K) if mA == "km" else (K, M), generator=g, device="cuda", dtype=torch.float32)
    B = torch.rand((N, K), generator=g, device="cuda", dtype=torch.float32)       # "kn": k contiguous
    assert A.numel() * 4 > (1 << 32)
    C = torch.empty((N, M), device="cuda", dtype=torch.float32)                    # "mn": m contiguous
    p = ops.contraction_plan(h, [K, M] if mA == "km" else [M, K], mA, [K, N], "kn", [M, N], "mn", workspace_limit=1 << 30)
    d = p.describe()
    assert d["kname"] != "gett_f32_stream_kernel", d
    ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
    p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), p.required_workspace)
    torch.cuda.synchronize()
    rows = [0, 1, 17, 511, 512, 1000, M - 1]
    Arows = (A[rows, :] if mA == "km" else A[:, rows].t()).double()
    ref = B.double() @ Arows.t()                                                    # [N][len(rows)]
    got = C[:, rows].double()
    rel = float(((got - ref).abs() / ref.abs()).max())
    assert rel < 2e-4, (rel, d)          # K = 1.3 M terms accumulated in fp32 by up to 256 split-K slices
    C2 = torch.empty_like(C)
    p.contract(-0.5, A.data_ptr(), B.data_ptr(), 0.0, C2.data_ptr(), C2.data_ptr(), ws.data_ptr(), p.required_workspace)
    torch.cuda.synchronize()
    assert torch.equal(C2, C * -0.5)

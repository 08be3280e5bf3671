"""GPU parity of the GENERAL MFMA GETT family (csrc/kernels/gett_gen.inc: bf16 / fp16 shapes without 16-byte lanes or whole
64-deep K-tiles, fp64 on v_mfma_f64_16x16x4_f64, complex64 / complex128 as four real MFMAs) through the C ABI against numpy
einsum in fp64 / complex128 on the same (for 16-bit data: already rounded) inputs.

What is covered: every element type x both staged-unit orientations per operand x every vector width the tables instantiate
(V = 8 / 2 / 1 for 16-bit data, 2 / 1 for fp64 and complex64, 1 for complex128) x both tile sizes, ragged M / N / K (clamped
rows, zeroed K tail), multi-digit mode groups, batch modes, padded strides, alpha / beta, conjugation of A / B / C with complex
scalars, split-K of 16-bit data, and the reference's own shapes (cuTENSOR/python/cutensor/torch/einsum_test.py:47-124, extents
of 50).  Tolerances: fp64 1e-12 relative to the magnitude of the sum, complex64 2e-5, bf16 8e-3 / fp16 2e-3 (fp32 accumulation,
one rounding: DESIGN.md section 4)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT = {
    # name: (numpy dtype, cutensor enum name, torch dtype name, rtol)
    "bfloat16":   (np.float32, "R_16BF", "bfloat16", 8e-3),
    "float16":    (np.float32, "R_16F", "float16", 2e-3),
    "float64":    (np.float64, "R_64F", "float64", 1e-12),
    "complex64":  (np.complex64, "C_32F", "complex64", 2e-5),
    "complex128": (np.complex128, "C_64F", "complex128", 1e-12),
}


@pytest.fixture(scope="module")
def env(built):
    import torch
    assert torch.cuda.is_available()
    from cudalibrarysamples_amd import cutensor as ct, ops
    return torch, ct, ops, ops.Handle()


def _tensor(rng, ext, pad, np_dt):
    """(view, parent): a column-major tensor of extents `ext` inside a packed parent of extents ext + pad."""
    full = [e + p for e, p in zip(ext, pad)]
    n = int(np.prod(full)) if full else 1
    flat = rng.random(n) * 2 - 1
    if np.issubdtype(np_dt, np.complexfloating):
        flat = flat + 1j * (rng.random(n) * 2 - 1)
    parent = np.reshape(flat.astype(np_dt), tuple(full), order="F") if full else flat.astype(np_dt).reshape(())
    view = parent[tuple(slice(0, e) for e in ext)]
    return view, parent


def run(env, ext, mA, mB, mC, dtype, alpha=1.0, beta=0.0, seed=0, padA=None, padB=None, padC=None, opA=False, opB=False, opC=False,
        ws_limit=1 << 28, alignment=128, expect=None, algo=None):
    torch, ct, ops, h = env
    np_dt, cname, tname, rtol = DT[dtype]
    tdt = getattr(torch, tname)
    rng = np.random.default_rng(seed)
    eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
    A, PA = _tensor(rng, eA, padA or [0] * len(eA), np_dt)
    B, PB = _tensor(rng, eB, padB or [0] * len(eB), np_dt)
    C, PC = _tensor(rng, eC, padC or [0] * len(eC), np_dt)
    dev = lambda P: torch.from_numpy(np.ascontiguousarray(P.ravel(order="F"))).cuda().to(tdt)   # noqa: E731
    dA, dB, dC = dev(PA), dev(PB), dev(PC)
    if dtype in ("bfloat16", "float16"):     # the reference is computed from the rounded values
        back = lambda d, P: np.reshape(d.float().cpu().numpy(), P.shape, order="F")   # noqa: E731
        PA2, PB2, PC2 = back(dA, PA), back(dB, PB), back(dC, PC)
        A, B, C = (P[tuple(slice(0, e) for e in x)] for P, x in ((PA2, eA), (PB2, eB), (PC2, eC)))
    strides = lambda V: [s // V.itemsize for s in V.strides]   # noqa: E731
    # This file tests the GENERAL family.  Since round 5 a 16-bit problem with 16-byte lanes and ONE ragged contracted mode stays on the
    # aligned LDS-DMA kernels (tests/test_gpu_h16.py): CUTENSOR_AMD_GEN=force keeps such shapes here (the switch is read at plan
    # creation, and a plan made under it is not memoised).
    import os
    forced = dtype in ("bfloat16", "float16") and "CUTENSOR_AMD_GEN" not in os.environ
    if forced:
        os.environ["CUTENSOR_AMD_GEN"] = "force"
    try:
        plan = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, dtype=getattr(ct, cname), strideA=strides(A), strideB=strides(B),
                                    strideC=strides(C), workspace_limit=ws_limit, alignment=alignment,
                                    opA=ct.OP_CONJ if opA else ct.OP_IDENTITY, opB=ct.OP_CONJ if opB else ct.OP_IDENTITY,
                                    opC=ct.OP_CONJ if opC else ct.OP_IDENTITY, **({} if algo is None else dict(algo=algo)))
    finally:
        if forced:
            del os.environ["CUTENSOR_AMD_GEN"]
    d = plan.describe()
    assert d["family"] == 2 and d["kname"] == "gett_gen_kernel", d
    if expect:
        for k, v in expect.items():
            assert d[k] == v, (k, v, d)
    ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    plan.contract(alpha, dA.data_ptr(), dB.data_ptr(), beta, dC.data_ptr(), dC.data_ptr(), ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    wide = np.complex128 if np.issubdtype(np_dt, np.complexfloating) else np.float64
    a64, b64, c64 = A.astype(wide), B.astype(wide), C.astype(wide)
    ref = alpha * np.einsum("%s,%s->%s" % (mA, mB, mC), np.conj(a64) if opA else a64, np.conj(b64) if opB else b64) + \
        beta * (np.conj(c64) if opC else c64)
    mag = np.einsum("%s,%s->%s" % (mA, mB, mC), np.abs(a64), np.abs(b64)) * abs(alpha) + abs(beta) * np.abs(c64)
    outP = dC.to(torch.complex128 if wide is np.complex128 else torch.float64).cpu().numpy().reshape(PC.shape, order="F") if PC.ndim else None
    got = outP[tuple(slice(0, e) for e in eC)]
    err = np.abs(got - ref)
    tol = rtol * np.maximum(mag, 1e-30) + (rtol * np.abs(ref) if dtype in ("bfloat16", "float16") else 0)
    bad = err > tol
    assert not bad.any(), "%s %s: %d/%d off, worst %g at %s (got %r ref %r) plan %s" % (
        dtype, ext, int(bad.sum()), bad.size, float(np.max(err / tol)), np.unravel_index(np.argmax(err / tol), err.shape),
        got[np.unravel_index(np.argmax(err / tol), err.shape)], ref[np.unravel_index(np.argmax(err / tol), err.shape)], d)
    if any(padC or []):   # the padding region of the output buffer is untouched
        untouched = np.ones(PC.shape, dtype=bool)
        untouched[tuple(slice(0, e) for e in eC)] = False
        orig = np.reshape(dev(PC).to(dC.dtype).cpu().to(torch.complex128 if wide is np.complex128 else torch.float64).numpy(), PC.shape, order="F")
        assert np.array_equal(outP[untouched], orig[untouched])
    plan.destroy()
    return d


LAYOUTS = {"mk_kn": ("mk", "kn"), "km_kn": ("km", "kn"), "mk_nk": ("mk", "nk"), "km_nk": ("km", "nk")}


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
@pytest.mark.parametrize("dtype", sorted(DT))
def test_gemm_like_widest_lanes_ragged_everything(env, layout, dtype):
    """Extents that admit the widest lanes of the type (multiples of 8 / 2) but no whole tiles: ragged last M / N tile, ragged
    last K-tile; the large-tile and the small-tile instantiation."""
    mA, mB = LAYOUTS[layout]
    vmax = {"bfloat16": 8, "float16": 8, "float64": 2, "complex64": 2, "complex128": 1}[dtype]
    d = run(env, dict(m=200, n=136, k=104), mA, mB, "mn", dtype, seed=1, expect=dict(vec=vmax))
    assert (d["bm"], d["bn"]) == (64, 64), d
    big = dict(m=2008, n=1928, k=72) if dtype != "complex128" else dict(m=520, n=456, k=40)
    d = run(env, big, mA, mB, "mn", dtype, alpha=0.75, beta=-0.5, seed=2, expect=dict(vec=vmax))
    if dtype != "complex128":
        assert d["bm"] == 128, d


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
@pytest.mark.parametrize("dtype", sorted(DT))
def test_gemm_like_odd_extents_element_gathers(env, layout, dtype):
    mA, mB = LAYOUTS[layout]
    run(env, dict(m=77, n=53, k=91), mA, mB, "mn", dtype, alpha=1.25, beta=0.5, seed=3, expect=dict(vec=1))


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_16_bit_pairs(env, dtype):
    """Extents of 50 (the reference's own test list): 4-byte pairs."""
    for layout in sorted(LAYOUTS):
        mA, mB = LAYOUTS[layout]
        run(env, dict(m=50, n=50, k=50), mA, mB, "mn", dtype, seed=4, expect=dict(vec=2))
    run(env, dict(m=150, n=250, k=350), "km", "nk", "mn", dtype, alpha=-1.0, beta=2.0, seed=5, expect=dict(vec=2))


@pytest.mark.parametrize("dtype", sorted(DT))
def test_tensor_shapes_multi_digit_groups_and_batch(env, dtype):
    """contraction.cu's mode structure C[m,u,n,v] = A[m,h,k,n] B[u,k,v,h] (:43-59) with extents that fuse nothing, the headline
    einsum's structure 'abcd,dcbe->ae' (einsum.cu) with a fastest contracted extent that is not a K-tile multiple, the
    reference test list's batched and transposed equations (einsum_test.py:62-82: 'ijk,ikl->ijl', 'mlik,lkjm->lij' in
    cuTENSOR mode order), an outer product and a dot product."""
    run(env, dict(m=14, u=10, n=6, v=9, h=5, k=12), "mhkn", "ukvh", "munv", dtype, alpha=1.1, beta=0.3, seed=6)
    run(env, dict(a=30, b=3, c=5, d=22, e=26), "dcba", "ebcd", "ea", dtype, seed=7)
    run(env, dict(i=6, j=20, k=18, l=22), "kji", "lki", "lji", dtype, seed=8)            # batch mode i slowest
    run(env, dict(m=6, l=5, i=20, k=14, j=18), "kilm", "mjkl", "jil", dtype, seed=9)      # 'mlik,lkjm->lij' reversed
    run(env, dict(m=40, n=30), "m", "n", "mn", dtype, seed=10)                           # no contracted mode


@pytest.mark.parametrize("dtype", sorted(DT))
def test_padded_strides(env, dtype):
    """Sub-tensors of larger buffers: lanes narrow to what the padded strides still align; the output's padding stays untouched."""
    run(env, dict(m=48, n=40, k=56), "km", "kn", "mn", dtype, seed=12, padA=[8, 0], padB=[16, 3], padC=[8, 1], beta=0.5)
    run(env, dict(m=48, n=40, k=56), "mk", "nk", "mn", dtype, seed=13, padA=[3, 1], padB=[1, 2], padC=[5, 0], expect=dict(vec=1))
    run(env, dict(m=20, n=24, k=16, l=3), "kml", "knl", "mnl", dtype, seed=14, padA=[2, 2, 0], padB=[4, 0, 1], padC=[0, 2, 0], alpha=2.0)


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
@pytest.mark.parametrize("conj", [(True, False, False), (False, True, False), (True, True, True), (False, False, True)])
def test_complex_conjugation_and_complex_scalars(env, dtype, conj):
    """D = alpha * op(A) * op(B) + beta * op(C) with complex alpha / beta (cuTENSOR/contraction_jit.cu:31-41); both operand
    roles (the planner swaps A and B when D's stride-1 mode comes from A)."""
    opA, opB, opC = conj
    for (mA, mB, mC) in (("mkl", "knl", "mnl"), ("km", "nk", "nm")):
        run(env, dict(m=44, n=36, k=28, l=3), mA, mB, mC, dtype, alpha=1.1 - 0.3j, beta=0.25 + 0.5j, seed=15, opA=opA, opB=opB, opC=opC)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_split_k_of_16_bit_data(env, dtype):
    d = run(env, dict(m=64, n=48, k=4000), "km", "kn", "mn", dtype, alpha=0.5, beta=1.5, seed=16)
    assert d["splitK"] > 1, d
    d = run(env, dict(m=50, n=50, k=3001), "mk", "nk", "mn", dtype, seed=17)      # free-contiguous pairs, odd K
    assert d["splitK"] > 1 and d["vec"] == 2, d
    d = run(env, dict(m=51, n=49, k=3001), "km", "kn", "mn", dtype, seed=21)      # 2-byte gathers
    assert d["splitK"] > 1 and d["vec"] == 1, d
    d = run(env, dict(m=64, n=48, k=4000), "km", "kn", "mn", dtype, seed=18, ws_limit=0)
    assert d["splitK"] == 1, d


@pytest.mark.parametrize("dtype", ["float64", "complex64", "complex128"])
def test_split_k_of_fp64_and_complex_data(env, dtype):
    """Few output tiles and a long K: the slices write partial tiles in the accumulator type (double / float2 / double2) and
    gen_splitk_reduce_kernel folds them with alpha / beta (complex: conj(C) too); the headline einsum's mode structure as well."""
    cplx = dtype != "float64"
    d = run(env, dict(m=64, n=48, k=4000), "km", "kn", "mn", dtype, alpha=(0.5 - 0.25j) if cplx else 0.5, beta=(1.5 + 0.5j) if cplx else 1.5,
            seed=31, opC=cplx)
    assert d["splitK"] > 1, d
    d = run(env, dict(m=51, n=49, k=2001, l=2), "mkl", "nkl", "mnl", dtype, seed=32)
    assert d["splitK"] > 1, d
    d = run(env, dict(a=48, b=6, c=10, d=32, e=40), "dcba", "ebcd", "ea", dtype, alpha=1.25, seed=33)
    assert d["splitK"] > 1, d
    d = run(env, dict(m=64, n=48, k=4000), "km", "kn", "mn", dtype, seed=34, ws_limit=0)
    assert d["splitK"] == 1, d


def test_fp64_matches_the_oracle_bit_for_bit_on_exact_data(env):
    """Small integers: every product and partial sum is exact in fp64, so the MFMA result must EQUAL the oracle's."""
    import oracle
    torch, ct, ops, h = env
    rng = np.random.default_rng(19)
    ext = dict(m=70, n=45, k=130)
    A = np.asfortranarray(rng.integers(-8, 9, (ext["k"], ext["m"])).astype(np.float64))
    B = np.asfortranarray(rng.integers(-8, 9, (ext["k"], ext["n"])).astype(np.float64))
    C = np.asfortranarray(rng.integers(-8, 9, (ext["m"], ext["n"])).astype(np.float64))
    dev = lambda P: torch.from_numpy(np.ascontiguousarray(P.ravel(order="F"))).cuda()   # noqa: E731
    dA, dB, dC = dev(A), dev(B), dev(C)
    plan = ops.contraction_plan(h, [ext["k"], ext["m"]], "km", [ext["k"], ext["n"]], "kn", [ext["m"], ext["n"]], "mn", dtype=ct.R_64F)
    assert plan.describe()["family"] == 2
    ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    plan.contract(2.0, dA.data_ptr(), dB.data_ptr(), -3.0, dC.data_ptr(), dC.data_ptr(), ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    want = C.copy(order="F")
    oracle.contract(A, "km", B, "kn", want, "mn", alpha=2.0, beta=-3.0, C=C)
    assert np.array_equal(dC.cpu().numpy().reshape(C.shape, order="F"), want)
    plan.destroy()


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_peeled_contracted_mode_with_different_c_and_d_layouts(env, dtype):
    """A peeled CONTRACTED mode accumulates through D: the launches after the first read D as their C operand.  With C and D laid
    out differently (same modes and extents, different strides) they must read D with D's strides (a second inner plan)."""
    torch, ct, ops, h = env
    mA, mB, mC = "paqbrcsdte", "xpyqzrst", "abxcydze"
    ext = dict(a=4, b=3, c=5, d=2, e=6, p=3, q=4, r=2, s=5, t=3, x=4, y=3, z=2)
    np_dt = np.float32 if dtype == "float32" else np.float64
    rng = np.random.default_rng(20)
    eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
    A, PA = _tensor(rng, eA, [0] * len(eA), np_dt)
    B, PB = _tensor(rng, eB, [0] * len(eB), np_dt)
    C, PC = _tensor(rng, eC, [1, 0, 2, 0, 0, 1, 0, 0], np_dt)       # C padded, D packed
    D, PD = _tensor(rng, eC, [0] * len(eC), np_dt)
    dev = lambda P: torch.from_numpy(np.ascontiguousarray(P.ravel(order="F"))).cuda()   # noqa: E731
    dA, dB, dC, dD = dev(PA), dev(PB), dev(PC), dev(PD)
    st = lambda V: [s // V.itemsize for s in V.strides]   # noqa: E731
    plan = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, dtype=ct.R_32F if dtype == "float32" else ct.R_64F, strideC=st(C), strideD=st(D),
                                workspace_limit=1 << 24)
    d = plan.describe()
    assert d.get("peeled_modes", 0) >= 1 and d["peel_launches"] >= 2, d
    ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    plan.contract(1.5, dA.data_ptr(), dB.data_ptr(), -0.5, dC.data_ptr(), dD.data_ptr(), ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    ref = 1.5 * np.einsum("%s,%s->%s" % (mA, mB, mC), A.astype(np.float64), B.astype(np.float64)) - 0.5 * C.astype(np.float64)
    got = dD.cpu().numpy().reshape(PD.shape, order="F")
    np.testing.assert_allclose(got, ref, rtol=1e-5 if dtype == "float32" else 1e-12, atol=(1e-5 if dtype == "float32" else 1e-12) * float(np.abs(ref).max()))
    plan.destroy()


@pytest.mark.parametrize("dtype,ext", [("float64", dict(m=200, n=136, k=104)), ("float16", dict(m=201, n=135, k=103)),
                                        ("complex64", dict(m=96, n=80, k=6000)), ("bfloat16", dict(m=72, n=64, k=12001))])
def test_patient_autotuning_of_a_general_family_problem(env, dtype, ext):
    """CUTENSOR_ALGO_DEFAULT_PATIENT on problems that only the general MFMA family serves (fp64, complex, unaligned 16-bit; the last
    two split K): cutensorCreatePlan must time general-family candidates against the general-family kernel table (an index into it
    means nothing in the fp32 table: the round-4 advisor's finding) — or not time at all when there is one candidate — and the plan
    it returns computes the right result."""
    _, ct, _, _ = env
    run(env, ext, "mk", "kn", "mn", dtype, alpha=1.25, beta=0.5, seed=11, algo=ct.ALGO_DEFAULT_PATIENT)

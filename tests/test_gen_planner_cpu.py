"""Which contractions the planner hands to the general MFMA family (csrc/host/plan_contraction.cpp pick_gen_choice ->
csrc/kernels/gett_gen.inc), host-only.  The point of the family: every 16-bit case of the reference's own regression list
(cuTENSOR/python/cutensor/torch/einsum_test.py:84-123, extents of 50: no 16-byte lanes, K not in whole 64-deep tiles), its
fp64 case (:109-115) and its complex cases (:55-68) run on the matrix cores instead of the one-output-per-lane FMA kernels."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = os.path.join(ROOT, "tests", "golden", "full")
DT = {"float16": "R_16F", "bfloat16": "R_16BF", "float64": "R_64F", "complex64": "C_32F", "complex128": "C_64F", "float32": "R_32F"}


@pytest.fixture(scope="module")
def env(built):
    from cudalibrarysamples_amd import cutensor as ct, ops
    return ct, ops, ops.Handle()


def _plan_for_equation(env, equation, a_size, b_size, dtype_name):
    """The descriptor set Einsum<> builds for a row-major framework tensor (einsum.cu:63-223: modes and extents reversed)."""
    import oracle
    ct, ops, h = env
    p = oracle.einsum_parse(equation, a_size, b_size)
    assert p is not None, equation
    return ops.contraction_plan(h, p["extentA"], p["modesA"], p["extentB"], p["modesB"], p["extentC"], p["modesC"],
                                dtype=getattr(ct, DT[dtype_name]), workspace_limit=1 << 28)


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(FULL) if f.endswith(".npz")))
def test_reference_cases_at_their_own_extents_run_on_mfma_kernels(env, name):
    meta = json.loads(str(np.load(os.path.join(FULL, name + ".npz"))["meta"]))
    plan = _plan_for_equation(env, meta["equation"], meta["a_size"], meta["b_size"], meta["dtype"])
    d = plan.describe()
    if meta["dtype"] == "float32":
        assert d["family"] == 0 and d["kernel"] >= 0, d
    elif meta["dtype"] in ("float16", "bfloat16") and len(d["Kdigits"]) == 1 and d["family"] == 1:
        # round 6: 16-bit data with ONE contracted digit whose stride-1 modes lead their groups stays on the LDS-DMA kernels whatever the
        # extents (50 here: no 16-byte lanes, a partial k-unit, one masked K-tile) — tests/test_h16_planner_cpu.py has the rules
        assert d["kname"] in ("gett_h16w4x_kernel", "gett_h16w4m_kernel", "gett_h16w4m4_kernel", "gett_h16w4q_kernel"), (meta, d)
    else:
        # (with two contracted digits the operands may disagree on which one is contiguous: 'mlik,lkjm->lij' gathers)
        assert d["family"] == 2 and d["kernel"] >= 0 and d["kname"] == "gett_gen_kernel", (meta, d)
    plan.destroy()


def test_vector_width_and_orientation_follow_the_layout(env):
    ct, ops, h = env
    def plan(M, N, K, mA, mB, dtype, **kw):
        extA = [M, K] if mA == "mk" else [K, M]
        extB = [K, N] if mB == "kn" else [N, K]
        return ops.contraction_plan(h, extA, mA, extB, mB, [M, N], "mn", dtype=dtype, workspace_limit=1 << 28, **kw)
    # 16-byte lanes but K not a multiple of 64.  Since round 5 ONE ragged contracted mode stays in the aligned LDS-DMA family (masked last
    # K-tile, tests/test_h16_planner_cpu.py); the general family's V = 8 form is looked at under CUTENSOR_AMD_GEN=f (the general family
    # also where the aligned kernels apply) ...
    for (mA, mB, oa, ob) in (("mk", "kn", 0, 1), ("km", "nk", 1, 0), ("km", "kn", 1, 1), ("mk", "nk", 0, 0)):
        p = plan(2048, 2048, 1000, mA, mB, ct.R_16BF)
        assert p.describe()["family"] == 1, p.describe()
        p.destroy()
        os.environ["CUTENSOR_AMD_GEN"] = "f"
        try:
            p = plan(2048, 2048, 1000, mA, mB, ct.R_16BF)
        finally:
            del os.environ["CUTENSOR_AMD_GEN"]
        d = p.describe()
        # the planner may have swapped the operands (D's stride-1 mode becomes kernel-N): compare as a set when it did
        got = (d["orientA"], d["orientB"]) if not d["swapped"] else (d["orientB"], d["orientA"])
        assert d["family"] == 2 and d["vec"] == 8 and got == (oa, ob) and (d["bm"], d["bn"], d["bk"]) == (128, 128, 64), (mA, mB, d)
        p.destroy()
    # ... TWO contracted modes with a ragged fastest one — C[m,n] = A[k,m,j] B[k,j,n], k = 40, j = 25 — went there until round 6; now the
    # LDS-DMA kernels keep them (the masked last K-tile of every sweep of k: tests/test_h16_planner_cpu.py) unless a 16-byte unit would
    # be partial (k = 36) or the sweeps fill less than 45 % of their K-tiles (k = 24)
    for k, fam in ((40, 1), (36, 2), (24, 2)):
        p = ops.contraction_plan(h, [k, 512, 25], "kmj", [k, 25, 512], "kjn", [512, 512], "mn", dtype=ct.R_16BF, workspace_limit=1 << 28)
        d = p.describe()
        assert d["family"] == fam and (fam == 2 or d["rag"] == 1), d
        p.destroy()
    # whole 64-deep K-tiles and 16-byte lanes: still the aligned LDS-DMA family
    p = plan(2048, 2048, 1024, "mk", "kn", ct.R_16BF)
    assert p.describe()["family"] == 1
    p.destroy()
    # odd extents / element alignment only: since round 6 still the LDS-DMA family (16-byte units at any 2-byte address); the general
    # family's 2-byte gathers under CUTENSOR_AMD_GEN=f
    for args, kw in (((37, 29, 51, "mk", "kn", ct.R_16F), {}), ((64, 64, 64, "km", "kn", ct.R_16BF), dict(alignment=2))):
        p = plan(*args, **kw)
        assert p.describe()["family"] == 1, p.describe()
        p.destroy()
        os.environ["CUTENSOR_AMD_GEN"] = "f"
        try:
            p = plan(*args, **kw)
        finally:
            del os.environ["CUTENSOR_AMD_GEN"]
        d = p.describe()
        assert d["family"] == 2 and d["vec"] == 1, d
        p.destroy()
    # fp64: 16-byte lanes = two doubles; large problem -> 128 x 128 tile, small -> 64 x 64
    p = plan(4096, 4096, 4096, "km", "kn", ct.R_64F)
    d = p.describe()
    assert d["family"] == 2 and d["vec"] == 2 and (d["bm"], d["bn"], d["bk"]) == (128, 128, 16) and d["blocks"] == 1024, d
    p.destroy()
    p = plan(50, 50, 50, "km", "kn", ct.R_64F)
    d = p.describe()
    assert d["family"] == 2 and d["vec"] == 2 and (d["bm"], d["bn"]) == (64, 64), d
    p.destroy()
    p = plan(51, 49, 50, "mk", "kn", ct.R_64F)
    assert p.describe()["vec"] == 1
    p.destroy()
    # complex
    p = plan(2048, 2048, 2048, "km", "kn", ct.C_32F)
    d = p.describe()
    assert d["family"] == 2 and d["vec"] == 2 and d["bk"] == 16, d
    p.destroy()
    p = plan(300, 300, 300, "km", "kn", ct.C_64F)
    d = p.describe()
    assert d["family"] == 2 and d["vec"] == 1 and (d["bm"], d["bn"], d["bk"]) == (64, 64, 8), d
    p.destroy()


def test_split_k_of_16_bit_data_and_workspace_invariant(env):
    ct, ops, h = env
    # one output tile, deep ragged K: split over the CUs, fp32 partials within the estimate (contraction.cu:239 asserts required <= estimate).
    # (TWO contracted modes with a fastest one that fills a third of a K-tile — k = 20, j = 200: the LDS-DMA kernels' sweep mask is
    # not offered below 45 %, tests/test_h16_planner_cpu.py)
    p = ops.contraction_plan(h, [20, 64, 200], "kmj", [20, 200, 48], "kjn", [64, 48], "mn", dtype=ct.R_16BF)
    d = p.describe()
    assert d["family"] == 2 and d["splitK"] > 1 and p.required_workspace == d["splitK"] * 64 * 48 * 4, d
    assert p.required_workspace <= p.workspace_estimate
    p.destroy()
    # no workspace allowed: no split
    p = ops.contraction_plan(h, [20, 64, 200], "kmj", [20, 200, 48], "kjn", [64, 48], "mn", dtype=ct.R_16BF, workspace_limit=0)
    d = p.describe()
    assert d["family"] == 2 and d["splitK"] == 1 and p.required_workspace == 0, d
    p.destroy()
    # the same with lanes (K = 4000): the aligned family, slices of whole K-tiles, the same workspace invariant
    p = ops.contraction_plan(h, [4000, 64], "km", [4000, 48], "kn", [64, 48], "mn", dtype=ct.R_16BF)
    d = p.describe()
    assert d["family"] == 1 and d["splitK"] > 1 and d["kPerSlice"] % 64 == 0 and p.required_workspace == d["splitK"] * 64 * 48 * 4, d
    assert p.required_workspace <= p.workspace_estimate
    p.destroy()
    # fp64 / complex: partials in the accumulator type (double, float2, double2)
    for dtype, acc in ((ct.R_64F, 8), (ct.C_32F, 8), (ct.C_64F, 16)):
        p = ops.contraction_plan(h, [4000, 64], "km", [4000, 48], "kn", [64, 48], "mn", dtype=dtype)
        d = p.describe()
        assert d["family"] == 2 and d["splitK"] > 1 and p.required_workspace == d["splitK"] * 64 * 48 * acc, d
        assert p.required_workspace <= p.workspace_estimate
        p.destroy()
    # the headline einsum's shape with fp64 data: one 96 x 96 output, K = 262144 -> split over the chip, not one workgroup
    e = dict(a=96, b=64, c=64, d=64, e=96)
    p = ops.contraction_plan(h, [e[c] for c in "dcba"], "dcba", [e[c] for c in "ebcd"], "ebcd", [96, 96], "ea", dtype=ct.R_64F)
    d = p.describe()
    assert d["family"] == 2 and d["splitK"] >= 64 and d["blocks"] >= 128, d
    p.destroy()


def test_many_mode_complex_keeps_the_mode_table_kernel(env):
    """More than four unfusable modes in a group and too many launches to peel (cuTENSOR/contraction_jit.cu:50-56 shape class)."""
    ct, ops, h = env
    mA = "badcfehgjilknm"
    mB = "ponmlkqrst"[::-1]
    mC = "".join(c for c in "abcdefghijopqrst" if (c in mA) != (c in mB))
    ext = {c: (3 if c in "aq" else 2) for c in set(mA + mB)}
    p = ops.contraction_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [ext[c] for c in mC], mC, dtype=ct.C_32F)
    assert p.describe()["kname"] == "gett_wide_kernel"
    p.destroy()

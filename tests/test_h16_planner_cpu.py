"""Which 16-bit GETT kernel the planner picks (csrc/host/plan_contraction.cpp pick_h16_choice / rank_h16_choices), host-only:
BASELINE configs[3] (bf16 C[m,n] = A[m,k] B[k,n], M = N = K = 8192, `contraction.cu:33-40` retyped) must run the four-wave
16x16x32 kernel of gett_h16v.hip on all four operand layouts, K ranges of <= 16 K-tiles per workgroup the eight-wave kernel,
CUTENSOR_AMD_H16_WAVES overrides both (child process: the switch is read once), and every variant stays an autotuning candidate."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env(built):
    from cudalibrarysamples_amd import cutensor as ct, ops
    return ct, ops


def _plan(ct, ops, h, M, N, K, mA="mk", mB="kn", dtype=None, **kw):
    extA = [M, K] if mA == "mk" else [K, M]
    extB = [K, N] if mB == "kn" else [N, K]
    return ops.contraction_plan(h, extA, mA, extB, mB, [M, N], "mn", dtype=dtype if dtype is not None else ct.R_16BF,
                                workspace_limit=1 << 30, **kw)


@pytest.mark.parametrize("mA,mB", [("mk", "kn"), ("km", "kn"), ("mk", "nk"), ("km", "nk")])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_headline_16_bit_shape_runs_the_16x16x32_kernel(env, mA, mB, dtype):
    """8192^3 = 1024 tiles of 256 x 256 = four rounds on 256 CUs: since round 5 the PERSISTENT form of the 16x16x32 kernel (gett_h16w4p_kernel:
    one workgroup per CU walks its tiles and streams each into the next), one round -> gett_h16w4x_kernel (4096^3, below)."""
    ct, ops = env
    if os.environ.get("CUTENSOR_AMD_H16_WAVES"):
        pytest.skip("the planner's own choice is under test")
    h = ops.Handle()
    p = _plan(ct, ops, h, 8192, 8192, 8192, mA, mB, dtype=ct.R_16BF if dtype == "bf16" else ct.R_16F)
    d = p.describe()
    assert d["kname"] == "gett_h16w4p_kernel" and d["splitK"] == 1 and d["blocks"] == 1024, d
    assert (d["bm"], d["bn"], d["bk"]) == (256, 256, 64), d
    p.destroy()


def test_short_k_ranges_run_the_default_kernel_too(env):
    """Rounds 2-3 sent K ranges of at most 16 K-tiles per workgroup to the eight-wave kernel (lower fixed cost per workgroup); with the
    shorter prologue and the pipelined epilogue of round 4 the four-wave 16x16x32 kernel is ahead there as well
    (profiles/r04z_short_k_4x_vs_8.txt), so one rule is left: the planner's time model."""
    ct, ops = env
    if os.environ.get("CUTENSOR_AMD_H16_WAVES"):
        pytest.skip("the planner's own choice is under test")
    h = ops.Handle()
    for (M, N, K, want, split) in [(8192, 8192, 512, "gett_h16w4p_kernel", 1),        # 8 K-tiles per tile, four tiles per workgroup: persistent (round 5: +6.8 %)
                                   (8192, 8192, 1024, "gett_h16w4p_kernel", 1),    # 16
                                   (8192, 8192, 1088, "gett_h16w4x_kernel", 1),    # 17: an odd count cannot stream — the one-tile kernel (132.5 vs 128.5 us,
                                                                                   # profiles/r06n_h16p_vs_4x_odd_ktile_counts.jsonl)
                                   (2048, 2048, 16384, "gett_h16w4x_kernel", 4)]:  # 64 per slice, 256 workgroups = one round: the one-tile kernel
        p = _plan(ct, ops, h, M, N, K)
        d = p.describe()
        assert (d["kname"], d["splitK"]) == (want, split), (M, N, K, d)
        p.destroy()


def test_mid_size_problems_run_the_128_tile_family(env):
    """Problems whose 256 x 256 tiles leave most CUs idle (round 3: split-K over the 256 x 256 kernels + a fold: 2048^3 at 0.41 PFLOP/s,
    1024^3 at 0.08): at most one 128 x 128 tile per CU -> the four-deep-ring kernel, one workgroup per CU, no split unless the tiles
    are very few; up to two per CU -> the two-deep-ring kernel, two workgroups per CU; a full chip of 256 x 256 tiles -> as before;
    and problems whose 128 x 128 tiles still leave three quarters of the CUs idle (1024^3) -> the 64 x 64 kernel."""
    ct, ops = env
    if os.environ.get("CUTENSOR_AMD_H16_WAVES"):
        pytest.skip("the planner's own choice is under test")
    h = ops.Handle()
    for (M, N, K, want, split, blocks) in [(2048, 2048, 2048, "gett_h16w4m4_kernel", 1, 256),
                                           (1024, 1024, 1024, "gett_h16w4q_kernel", 1, 256),
                                           (4096, 1024, 4096, "gett_h16w4m4_kernel", 1, 256),
                                           (1000, 1000, 1024, "gett_h16w4q_kernel", 1, 256),
                                           (512, 512, 65536, "gett_h16w4m4_kernel", 16, 256),
                                           (2048, 4096, 4096, "gett_h16w4m_kernel", 1, 512),
                                           (4096, 4096, 4096, "gett_h16w4x_kernel", 1, 256)]:
        p = _plan(ct, ops, h, M, N, K)
        d = p.describe()
        assert (d["kname"], d["splitK"], d["blocks"]) == (want, split, blocks), (M, N, K, d)
        assert (d["bm"], d["bn"]) == ((128, 128) if "w4m" in want else (64, 64) if "w4q" in want else (256, 256)), d
        p.destroy()


KEPT = {"gett_h16w4x_kernel", "gett_h16w4p_kernel", "gett_h16w4m_kernel", "gett_h16w4m4_kernel", "gett_h16w8m_kernel", "gett_h16w4q_kernel"}
RETIRED = {"gett_h16_kernel", "gett_h16w4v_kernel", "gett_h16w4r_kernel", "gett_h16s_kernel", "gett_h16w4s_kernel", "gett_h16w4_kernel"}


def test_every_variant_stays_an_autotuning_candidate(env):
    """The ranked candidates of a 16-bit problem are the kernel families BUILT INTO the library: the six kept ones in a production build
    (round 5: the retired families live behind make RESEARCH=1 and keep only their table slots), all twelve in a research build."""
    ct, ops = env
    if os.environ.get("CUTENSOR_AMD_H16_WAVES"):
        pytest.skip("the planner's own choice is under test")
    h = ops.Handle()
    research = bool(ct.lib.ctamdResearchKernelsBuilt())
    names = []
    for r in range(12):
        p = _plan(ct, ops, h, 8192, 8192, 8192, algo=r, cache_mode=ct.CACHE_MODE_NONE)
        names.append(p.describe()["kname"])
        p.destroy()
    assert names[0] in ("gett_h16w4x_kernel", "gett_h16w4p_kernel"), names
    assert set(names) == (KEPT | RETIRED if research else KEPT), names


@pytest.mark.parametrize("waves,want", [("8", "gett_h16_kernel"), ("4", "gett_h16w4_kernel"), ("4v", "gett_h16w4v_kernel"),
                                        ("4x", "gett_h16w4x_kernel"), ("s", "gett_h16s_kernel"), ("4m", "gett_h16w4m_kernel"),
                                        ("4m4", "gett_h16w4m4_kernel"), ("8m", "gett_h16w8m_kernel"), ("4q", "gett_h16w4q_kernel"),
                                        ("4p", "gett_h16w4p_kernel")])
def test_the_switch_overrides_the_planner(built, waves, want):
    code = ("import json, sys; sys.path.insert(0, %r); from cudalibrarysamples_amd import cutensor as ct, ops; h = ops.Handle(); "
            "out = []\n"
            "for K in (512, 8192):\n"
            "    p = ops.contraction_plan(h, [8192, K], 'mk', [K, 8192], 'kn', [8192, 8192], 'mn', dtype=ct.R_16BF); out.append(p.describe()['kname']); p.destroy()\n"
            "print(json.dumps(out))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, CUTENSOR_AMD_H16_WAVES=waves))
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    from cudalibrarysamples_amd import cutensor as ct
    if want in RETIRED and not ct.lib.ctamdResearchKernelsBuilt():
        # a retired family asked for in a production build: the switch is ignored, the planner's own choice runs
        assert all(g in KEPT for g in got), got
    else:
        assert got == [want, want], r.stdout


def test_ragged_k_stays_in_the_lds_dma_family(env):
    """One contracted mode with K % 64 != 0 and 16-byte lanes (round-4 review, Missing #4): the planner's usual candidates minus the
    persistent kernel — their RAG instantiations stage the last K-tile masked (gett_h16x_common.h x_rag_mask).  K-tiles are counted
    rounded up; slices stay whole K-tiles.  Two contracted modes with a ragged fastest one are the general family's."""
    ct, ops = env
    h = ops.Handle()

    def plan(M, N, K, mA="mk", mB="kn", **kw):
        e = dict(m=M, n=N, k=K)
        p = ops.contraction_plan(h, [e[c] for c in mA], mA, [e[c] for c in mB], mB, [M, N], "mn", dtype=ct.R_16BF, workspace_limit=1 << 28, **kw)
        d = p.describe()
        p.destroy()
        return d
    d = plan(4096, 4096, 4104)                     # the review's shape: one 256 x 256 tile per CU
    assert d["family"] == 1 and d["kname"] == "gett_h16w4x_kernel" and d["splitK"] == 1 and d["kPerSlice"] == 65 * 64, d
    d = plan(8192, 8192, 8200)                     # (not the persistent kernel: it has no masked tile)
    assert d["kname"] == "gett_h16w4x_kernel", d
    assert plan(2048, 2048, 200)["kname"] == "gett_h16w4m4_kernel"
    assert plan(1024, 1024, 1080)["kname"] == "gett_h16w4q_kernel"
    d = plan(96, 96, 4104, "km", "kn")             # split-K: whole K-tiles per slice, the last slice owns the masked tile
    assert d["family"] == 1 and d["splitK"] > 1 and d["kPerSlice"] % 64 == 0 and d["splitK"] * d["kPerSlice"] >= 4104, d
    assert plan(2048, 1032, 77, "mk", "nk")["family"] == 1          # both operands free-contiguous: any K
    # round 6: a K-contiguous operand with K % 8 != 0 stays too (the partial k-unit's tail is zeroed in LDS: x_rag_fix), and so do
    # extents / pitches / base pointers without 16-byte lanes — the stride-1 mode decides, nothing else
    assert plan(2048, 1032, 77, "mk", "kn")["family"] == 1
    for (M, N, K) in ((4100, 4100, 4100), (4097, 4097, 4097), (50, 50, 50), (300, 204, 100)):
        for (mA, mB) in (("mk", "kn"), ("km", "kn"), ("mk", "nk"), ("km", "nk")):
            d = plan(M, N, K, mA, mB, alignment=2)
            assert d["family"] == 1 and d["kname"] in ("gett_h16w4x_kernel", "gett_h16w4m_kernel", "gett_h16w4m4_kernel", "gett_h16w4q_kernel"), d
    # a ragged stride-1 mode must be ALONE in its group (a 16-byte unit may not straddle a digit boundary): two M modes with a ragged
    # fastest one are the general family's; with a fastest one of 8 j they stay
    e = dict(a=12, b=40, n=64, k=64)
    for a, fam in ((12, 2), (16, 1)):
        e["a"] = a
        p = ops.contraction_plan(h, [e[c] for c in "akb"], "akb", [64, 64], "kn", [e[c] for c in "anb"], "anb", dtype=ct.R_16BF, workspace_limit=1 << 28)
        assert p.describe()["family"] == fam, (a, p.describe())
        p.destroy()
    assert plan(2048, 2048, 1024)["family"] == 1                    # whole K-tiles: unchanged
    # round 6: SEVERAL contracted modes with a ragged fastest one stay as well (the mask is toggled per sweep of that mode): K-tiles are
    # counted per sweep, rounded up — 'kmj,kjn->mn' with k = 96, j = 5 is 2 x 5 tiles
    def plan2(e, mA, mB, **kw):
        p = ops.contraction_plan(h, [e[c] for c in mA], mA, [e[c] for c in mB], mB, [e["m"], e["n"]], "mn", dtype=ct.R_16BF, workspace_limit=1 << 28, **kw)
        d = p.describe()
        p.destroy()
        return d
    d = plan2(dict(m=4096, n=4096, k=96, j=5), "kmj", "kjn")
    assert d["family"] == 1 and d["rag"] == 1 and d["kname"] == "gett_h16w4x_kernel" and d["kPerSlice"] == 10 * 64, d
    d = plan2(dict(m=2048, n=2048, k=200, j=3), "kmj", "kjn")          # (a shape the 128 x 128 pair would take with one contracted mode)
    assert d["family"] == 1 and d["rag"] == 1 and d["kname"] in ("gett_h16w4x_kernel", "gett_h16w4q_kernel", "gett_h16w4m_kernel", "gett_h16w4m4_kernel") and d["kPerSlice"] % 64 == 0, d
    d = plan2(dict(m=96, n=96, k=72, j=40), "kmj", "kjn")              # split-K over the padded tile space (80 tiles)
    assert d["family"] == 1 and d["rag"] == 1 and d["splitK"] > 1 and d["splitK"] * d["kPerSlice"] >= 80 * 64, d
    assert plan2(dict(m=512, n=512, k=36, j=25), "kmj", "kjn")["family"] == 2     # partial 16-byte units at the end of every sweep
    assert plan2(dict(m=516, n=512, k=40, j=25), "mjk", "kjn")["family"] == 2     # a free-contiguous operand with a partial row unit
    assert plan2(dict(m=512, n=512, k=24, j=25), "kmj", "kjn")["family"] == 2     # sweeps that fill 37 % of a K-tile: the general family


def test_persistent_kernel_only_where_its_tiles_can_stream(env):
    """Multi-round launches go to the persistent kernel only when interior tiles can stream into each other (one M and one N mode after
    fusion, 16-byte lanes in D; batch modes stream since round 6); everything else keeps the one-tile kernel (non-streamed tiles make
    the persistent kernel slower: profiles/r05r_h16p_beta.jsonl)."""
    ct, ops = env
    if os.environ.get("CUTENSOR_AMD_H16_WAVES"):
        pytest.skip("the planner's own choice is under test")
    h = ops.Handle()

    def kname(eA, mA, eB, mB, eC, mC, **kw):
        p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, dtype=ct.R_16BF, workspace_limit=1 << 28, **kw)
        d = p.describe()
        p.destroy()
        return d["kname"]
    assert kname([8192, 8192], "mk", [8192, 8192], "kn", [8192, 8192], "mn") == "gett_h16w4p_kernel"
    assert kname([4096, 4096, 4], "mkl", [4096, 4096, 4], "knl", [4096, 4096, 4], "mnl") == "gett_h16w4p_kernel"          # batch mode: 4 x 256 tiles
    assert kname([64, 128, 8192], "abk", [8192, 8192], "kn", [64, 128, 8192], "abn") == "gett_h16w4p_kernel"            # a, b fuse into one M mode
    assert kname([64, 8192, 128], "akb", [8192, 8192], "kn", [64, 128, 8192], "abn") == "gett_h16w4x_kernel"            # a, b apart in A: two M modes


def test_workspace_contract_of_operands_copied_first(env):
    """Round 6 (api.cpp plan_repack): a contraction whose operands the LDS-DMA kernels cannot stage as they lie asks
    cutensorEstimateWorkspaceSize for its temporaries + the inner plan's need, plans the copies when that is granted
    (required <= estimate: contraction.cu:239 asserts it), and keeps the operands in place — the general family — when it is not."""
    ct, ops = env
    h = ops.Handle()
    ext = dict(i=4096, l=4096, j=16, k=72)
    args = ([ext[c] for c in "kji"], "kji", [ext[c] for c in "jkl"], "jkl", [ext[c] for c in "li"], "li")
    for dt, fam_direct in ((ct.R_16BF, 2), (ct.R_32F, 0), (ct.R_64F, 2)):
        es = {ct.R_16BF: 2, ct.R_32F: 4, ct.R_64F: 8}[dt]
        p = ops.contraction_plan(h, *args, dtype=dt)                      # the default: limit = the estimate (ops.contraction_plan)
        d = p.describe()
        temp = 4096 * 16 * 72 * es
        assert (d.get("repack_A") or d.get("repack_B")) and d["lone_bytes"] == temp, d
        assert temp <= p.required_workspace <= p.workspace_estimate, (p.required_workspace, p.workspace_estimate)
        p.destroy()
        p = ops.contraction_plan(h, *args, dtype=dt, workspace_limit=temp - 256)
        d = p.describe()
        assert not d.get("repack_A") and not d.get("repack_B") and d["family"] == fam_direct and p.required_workspace <= temp - 256, d
        p.destroy()
    # ... and a problem that is fine as it lies is not touched
    p = ops.contraction_plan(h, [4096, 4096], "mk", [4096, 4096], "kn", [4096, 4096], "mn", dtype=ct.R_16BF)
    assert "repack_A" not in p.describe()
    p.destroy()

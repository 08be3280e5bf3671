"""The N > 1 cuTENSORMg data path EXECUTED on the CPU (no multi-GPU box is reachable from the build container): a plan built on a
plan-only handle for 2 / 4 / 8 distinct device ids is walked by ctamdMgReplayOnHost (csrc/mg/mg.cpp) over host cell buffers —
every transfer of the plan is a memcpy into a NaN-filled staging image of the receiving device, every local contraction runs
through the CPU oracle on exactly the strided views / offsets / scalars / C-D aliasing that cutensorMgContraction hands to
cutensorContract, staged pieces of C are scattered to their owners' cells — and the cells of D are compared with the dense
result.  The replay also checks the ORDERING the device path relies on: a piece may read a staged cell only if a local copy or a
transfer whose event the piece's stream has waited for brought it.

Layouts: bench.py's free-mode layout (cuTENSORMg/contraction_multi_gpu.cu:286-345 driven as SURVEY 8e describes: largest free mode
cut, the other operand all-gathered), the sample's own 2 x 2 block-cyclic descriptors on 1..8 handle devices (:154-193), ragged
extents with NaN in every padding region, blog_post.cu's many-mode tensors (:78-175) on 8 devices at scaling 1..4, gather waves,
beta != 0."""
import ctypes

import numpy as np
import pytest

from tests.test_mg_plan_cpu import _blog_post_shapes, free_mode_layout


@pytest.fixture(scope="module")
def cm(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("plan-only handles for device ids 0..7 need a host without that many GPUs")
    from cudalibrarysamples_amd import cutensormg
    return cutensormg


def _layout(ext, bs, dc):
    nblk = [-(-e // b) for e, b in zip(ext, bs)]
    lb = [-(-n // d) for n, d in zip(nblk, dc)]
    es, run = [], 1
    for b in bs:
        es.append(run)
        run *= b
    bstr, brun = [], run
    for l in lb:
        bstr.append(brun)
        brun *= l
    cstr, cells = [], 1
    for d in dc:
        cstr.append(cells)
        cells *= d
    span = 1 + sum((b - 1) * s + (l - 1) * t for b, s, l, t in zip(bs, es, lb, bstr))
    return es, bstr, cstr, cells, span


def _addresses(ext, bs, dc):
    """(cell index, offset inside the cell buffer) of every element of the global tensor (cuTENSORMg packed block-cyclic storage:
    block index b = local block * deviceCount + grid coordinate; contraction_multi_gpu.cu:256, blog_post.cu:107-113)."""
    es, bstr, cstr, cells, span = _layout(ext, bs, dc)
    idx = np.indices(ext, dtype=np.int64) if ext else np.zeros((0,), dtype=np.int64)
    cell = np.zeros(ext, dtype=np.int64)
    off = np.zeros(ext, dtype=np.int64)
    for i in range(len(ext)):
        blk, w = idx[i] // bs[i], idx[i] % bs[i]
        cell += (blk % dc[i]) * cstr[i]
        off += w * es[i] + (blk // dc[i]) * bstr[i]
    return cell, off, cells, span


def distribute(T, bs, dc):
    cell, off, cells, span = _addresses(list(T.shape), bs, dc)
    bufs = np.full((cells, span), np.nan, dtype=T.dtype)         # padding regions hold NaN
    bufs[cell.ravel(), off.ravel()] = T.ravel()
    return [np.ascontiguousarray(bufs[c]) for c in range(cells)]


def collect(bufs, ext, bs, dc):
    cell, off, cells, span = _addresses(ext, bs, dc)
    return np.stack(bufs)[cell, off]


def _oracle_callback(dtype, va, pa, vb, pb, vc, pc, pd, alpha, beta):
    import oracle
    np_dt = {0: np.float32, 1: np.float64}[dtype]

    def arr(view, ptr):
        ext, st, _ = view
        span = 1 + sum((e - 1) * s for e, s in zip(ext, st))
        raw = np.frombuffer((ctypes.c_char * (span * np.dtype(np_dt).itemsize)).from_address(ptr), dtype=np_dt)
        return np.lib.stride_tricks.as_strided(raw, shape=ext, strides=[s * raw.itemsize for s in st])
    A, B, D = arr(va, pa), arr(vb, pb), arr(vc, pd)
    C = arr(vc, pc) if beta != 0.0 else None
    oracle.contract(A, va[2], B, vb[2], D, vc[2], alpha=alpha, beta=beta, C=C)
    return 0


def _numpy_callback(dtype, va, pa, vb, pb, vc, pc, pd, alpha, beta):
    """The same through numpy (einsum over the strided views, fp64, BLAS-backed): for blog_post.cu's 8.6-GFLOP-per-scaling-step
    problems, which the scalar oracle loop would take minutes for."""
    np_dt = {0: np.float32, 1: np.float64}[dtype]

    def arr(view, ptr):
        ext, st, _ = view
        span = 1 + sum((e - 1) * s for e, s in zip(ext, st))
        raw = np.frombuffer((ctypes.c_char * (span * np.dtype(np_dt).itemsize)).from_address(ptr), dtype=np_dt)
        return np.lib.stride_tricks.as_strided(raw, shape=ext, strides=[s * raw.itemsize for s in st])
    A, B, D = arr(va, pa), arr(vb, pb), arr(vc, pd)
    small = {l: i for i, l in enumerate(sorted(set(va[2]) | set(vb[2]) | set(vc[2])))}     # einsum takes labels < 52
    res = alpha * np.einsum(A.astype(np.float64), [small[l] for l in va[2]], B.astype(np.float64), [small[l] for l in vb[2]],
                            [small[l] for l in vc[2]], optimize=True)
    if beta != 0.0:
        res = res + beta * arr(vc, pc).astype(np.float64)
    D[...] = res.astype(np_dt)
    return 0


def run_case(cm, n, modes, extent, block, dcount, beta=0.0, seed=0, cell_devices=None, np_dt=np.float32, callback=None):
    rng = np.random.default_rng(seed)
    with cm.Contraction(list(range(n)), modes, extent, block, dcount, cell_devices=cell_devices, dtype=0 if np_dt is np.float32 else 1,
                        compute=cm.COMPUTE_32F if np_dt is np.float32 else (1 << 4)) as con:
        ext = [[extent[c] for c in m] for m in modes]
        bs = [[block[k].get(c, extent[c]) for c in m] for k, m in enumerate(modes)]
        dc = [[dcount[k].get(c, 1) for c in m] for k, m in enumerate(modes)]
        G = [(rng.random(e) - 0.5).astype(np_dt) for e in ext]
        cellsA, cellsB, cellsC = (distribute(G[k], bs[k], dc[k]) for k in range(3))
        cellsD = [c.copy() for c in cellsC]
        addr = lambda bufs: [b.ctypes.data for b in bufs]   # noqa: E731
        rc, msg = cm.replay_on_host(con.plan, 1.25, addr(cellsA), addr(cellsB), beta, addr(cellsC) if beta != 0.0 else None, addr(cellsD),
                                    callback or _oracle_callback)
        assert rc == 0, (rc, msg)
        got = collect(cellsD, ext[2], bs[2], dc[2])
        eq = "%s,%s->%s" % tuple(modes)
        ref = 1.25 * np.einsum(eq, G[0].astype(np.float64), G[1].astype(np.float64), optimize=True) + beta * G[2]
        assert np.isfinite(got).all(), "an element of D was never produced (or read padding)"
        np.testing.assert_allclose(got, ref, rtol=2e-5 if np_dt is np.float32 else 1e-12, atol=1e-5 if np_dt is np.float32 else 1e-12)
        # nothing but the valid elements was touched: the padding of D's cells is as it was (NaN)
        cell, off, cells, span = _addresses(ext[2], bs[2], dc[2])
        mask = np.ones((cells, span), dtype=bool)
        mask[cell.ravel(), off.ravel()] = False
        d = con.describe()
        return d, np.stack(cellsD)[mask]


@pytest.mark.parametrize("n", [2, 4, 8])
@pytest.mark.parametrize("beta", [0.0, 0.5])
def test_free_mode_shard_with_all_gather_executes(cm, n, beta):
    d, _ = run_case(cm, n, *free_mode_layout(n, 32 * n), beta=beta, seed=n)
    assert d["remoteBytes"] > 0 and len(d["pieces"]) >= 2 * n - 1


@pytest.mark.parametrize("n,waves", [(4, 3), (8, 7), (8, 2)])
def test_gather_in_waves_executes(cm, n, waves, monkeypatch):
    monkeypatch.setenv("CUTENSORMG_AMD_WAVES", str(waves))
    d, _ = run_case(cm, n, *free_mode_layout(n, 16 * n), seed=10 + waves)
    assert d["numWaves"] == waves


@pytest.mark.parametrize("n", [1, 2, 4, 8])
@pytest.mark.parametrize("beta", [0.0, -0.75])
def test_sample_block_cyclic_layout_executes(cm, n, beta):
    """contraction_multi_gpu.cu:154-217: 2 x 2 block-cyclic descriptors, cells owned by the handle devices cyclically (on 8 handle
    devices only four hold cells; the others still compute a shard and store into the owners' cells — the remote scatter)."""
    E, BS = 96, 16
    modes = ["ik", "kj", "ij"]
    block = [dict(i=BS, k=BS), dict(k=BS, j=BS), dict(i=BS, j=BS)]
    dcount = [dict(i=2, k=2), dict(k=2, j=2), dict(i=2, j=2)]
    run_case(cm, n, modes, dict(i=E, j=E, k=E), block, dcount, beta=beta, seed=20 + n)


@pytest.mark.parametrize("seed", range(6))
def test_ragged_block_cyclic_layouts_execute(cm, seed):
    """Extents that do not divide blockSize x deviceCount (blog_post.cu's ceil()-derived blocks) in free AND contracted modes: the
    cells' padding is NaN on the way in (a sum that touched it would be NaN) and must still be NaN on the way out."""
    rng = np.random.default_rng(300 + seed)
    n = int(rng.choice([2, 3, 4]))
    Ei, Ej, Ek = (int(rng.integers(20, 70)) for _ in range(3))
    bi, bj, bk = (int(rng.integers(4, 20)) for _ in range(3))
    di = n
    dj = int(rng.choice([1, 2])) if n % 2 == 0 else 1
    dk = int(rng.choice([1, 2]))
    modes = ["ik", "kj", "ij"]
    block = [dict(i=bi, k=bk), dict(k=bk, j=bj), dict(i=bi, j=bj)]
    dcount = [dict(i=di, k=dk), dict(k=dk, j=dj), dict(i=di, j=dj)]
    d, padding = run_case(cm, n, modes, dict(i=Ei, j=Ej, k=Ek), block, dcount, beta=float(rng.choice([0.0, 1.5])), seed=seed)
    assert np.isnan(padding).all()


@pytest.mark.parametrize("s", [1, 2, 3, 4])
def test_blog_post_on_eight_devices_executes(cm, s):
    """blog_post.cu 8 <scaling>: six-mode tensors, block-cyclic over 8 devices, local views with more unfusable modes than the tiled
    kernels take (peeled by the plan / by the library from scaling 2) — every one of its local contractions replayed."""
    modes, ext, block, dcount = _blog_post_shapes(8, s)
    d, _ = run_case(cm, 8, modes, ext, block, dcount, seed=40 + s, callback=_numpy_callback)
    assert d["localContractions"] >= len(d["pieces"])


def test_a_missing_wait_is_detected(cm, monkeypatch):
    """The ordering check is live: with the pieces' waits ignored (CUTENSORMG_AMD_TEST_DROP_WAITS, a fault-injection
    switch of the REPLAY CHECKER — the plan itself, which a device would execute, is never altered), the replay refuses the first piece that reads a gathered cell."""
    monkeypatch.setenv("CUTENSORMG_AMD_TEST_DROP_WAITS", "1")
    n = 4
    with cm.Contraction(list(range(n)), *free_mode_layout(n, 16 * n)) as con:
        z = [np.zeros(16 * n * 16, dtype=np.float32) for _ in range(n)]
        addr = [b.ctypes.data for b in z]
        rc, msg = cm.replay_on_host(con.plan, 1.0, addr, addr, 0.0, None, addr, lambda *a: 0)
        assert rc == -4 and "without having waited" in msg, (rc, msg)


@pytest.mark.parametrize("n", [1, 2, 4])
@pytest.mark.parametrize("transport", ["", "allgather", "sendrecv"])
def test_forced_gather_stages_every_operand_cell_through_the_communication_path(cm, n, transport, monkeypatch):
    """CUTENSORMG_AMD_FORCE_GATHER=1 (round 5: the mode that lets the RCCL all-gather and the wave-event graph of
    contraction_multi_gpu.cu:286-345's path execute on a ONE-GPU box): no operand is read in place, cells the computing device holds
    itself travel as remote transfers (source = destination rank) with a wave event the pieces wait for, a one-device plan is
    all-gather eligible, and the result is unchanged (executed here by the host replay, ordering check included)."""
    monkeypatch.setenv("CUTENSORMG_AMD_FORCE_GATHER", "1")
    monkeypatch.setenv("CUTENSORMG_AMD_ASSUME_RCCL", "1")
    if transport:
        monkeypatch.setenv("CUTENSORMG_AMD_TRANSPORT", transport)
    E = 32 * n
    d, _ = run_case(cm, n, *free_mode_layout(n, E), beta=0.5 if n == 2 else 0.0, seed=70 + n)
    assert d["forceGather"] == 1 and d["useRccl"] == 1
    ops = [t for t in d["transfers"] if t["tensor"] < 2]
    assert ops and all(not t["local"] and t["event"] >= 0 for t in ops)
    assert any(t["src"] == t["dst"] for t in ops)                       # own cells go the remote way
    assert all(not p["use"][0]["direct"] and not p["use"][1]["direct"] for p in d["pieces"])
    assert all(p["wait"] for p in d["pieces"]), d["pieces"][0]
    cell = E * (E // n) * 4
    assert d["remoteBytes"] >= n * n * cell                            # all of B on every device (+ the A slabs)
    assert d["allGatherEligible"][1] == 1
    if n == 1:
        assert d["allGatherEligible"] == [1, 1] and d["numWaves"] == 1

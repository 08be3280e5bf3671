"""CPU coverage of the multi-device (N > 1) cuTENSORMg path: plans are built on plan-only handles (no GPU here; nothing is
executed) for 2 / 4 / 8 *distinct* device ids and inspected through ctamdMgDescribePlan.  Checked: every grid cell a
device needs and does not hold is gathered exactly once per consumer, nothing is gathered that is not needed, every
element of C is produced exactly once, the first local contraction of every device depends on nothing remote (so the
gather overlaps it), and the pieces of a device alternate between its two compute streams."""
import itertools

import numpy as np
import pytest


@pytest.fixture(scope="module")
def cm(built):
    from cudalibrarysamples_amd import cutensormg
    return cutensormg


def free_mode_layout(n, E):
    """The layout bench.py uses for the cuTENSORMg case: C[i,j] = A[i,k] B[k,j] with the largest free mode i cut n ways
    (A and C hold row slabs), B distributed along j (column slabs) — every device needs all of B: the all-gather."""
    modes = ["ik", "kj", "ij"]
    extent = dict(i=E, j=E, k=E)
    block = [dict(i=E // n), dict(j=E // n), dict(i=E // n, j=E // n)]
    dcount = [dict(i=n), dict(j=n), dict(i=n)]
    return modes, extent, block, dcount


def coverage(d, con, E):
    """Number of times each C[i, j] is produced, from the pieces' p ranges and q coordinate ranges."""
    assert d["pLabel"] == ord("i")
    cov = np.zeros((E, E), dtype=np.int32)
    bs_j = con.cells[2]["bs"][1]
    qcount = con.cells[1]["dc"][1] if d["qLabel"] == ord("j") else 1
    jblock = np.arange(E) // bs_j
    for p in d["pieces"]:
        cols = np.ones(E, dtype=bool) if p["q1"] == 0 else ((jblock % qcount >= p["q0"]) & (jblock % qcount < p["q1"]))
        cov[p["lo"]:p["hi"], cols] += 1
    return cov


def check_transfers(d, con, ndev):
    seen = set()
    for t in d["transfers"]:
        key = (t["dst"], t["tensor"], t["cell"])
        assert key not in seen, "cell gathered twice: %r" % (key,)
        seen.add(key)
        owner = con.cells[t["tensor"]]["owners"][t["cell"]]
        assert t["local"] == int(owner == con.devices[t["dst"]])
        if not t["local"]:
            assert con.devices[t["src"]] == owner and 0 <= t["wave"] < d["numWaves"] and t["event"] >= 0
    # what the pieces read from the staging images is exactly what was transferred
    needed = set()
    for p in d["pieces"]:
        for k, u in enumerate(p["use"]):
            if not u["direct"]:
                needed.update((p["dev"], k, c) for c in u["cells"])
            else:
                assert len(u["cells"]) == 1 and con.cells[k]["owners"][u["cells"][0]] == con.devices[p["dev"]]
    assert needed == seen
    return seen


@pytest.mark.parametrize("n", [2, 4, 8])
def test_free_mode_shard_with_all_gather(cm, n):
    E = 128 * n
    with cm.Contraction(list(range(n)), *free_mode_layout(n, E)) as con:
        d = con.describe()
        assert d["qLabel"] == ord("j") and d["numWaves"] == 1
        check_transfers(d, con, n)
        assert (coverage(d, con, E) == 1).all()
        cell_bytes = E * (E // n) * 4
        # the all-gather of B: every device receives the n - 1 column slabs it does not hold; A and C never move
        assert d["remoteBytes"] == n * (n - 1) * cell_bytes and d["localCopyBytes"] == 0
        assert all(t["tensor"] == 1 and not t["local"] for t in d["transfers"])
        assert d["stagingBytes"][0] == 0 and d["stagingBytes"][2] == 0 and d["stagingBytes"][1] >= n * cell_bytes
        ev_wave = {t["event"]: t["wave"] for t in d["transfers"]}
        for g in range(n):
            mine = [p for p in d["pieces"] if p["dev"] == g]
            # own column slab first: read in place, no event to wait for -> the gather overlaps this contraction
            first = mine[0]
            assert first["wait"] == [] and all(u["direct"] for u in first["use"]) and (first["q0"], first["q1"]) == (g, g + 1)
            assert first["use"][1]["cells"] == [g]
            # the rest: A and C still in place, B from the staging image, behind the gather's event of this device
            for p in mine[1:]:
                assert p["use"][0]["direct"] and p["use"][2]["direct"] and not p["use"][1]["direct"]
                # one event per (wave, communication stream): one with RCCL, up to min(7, n - 1) with peer copies
                assert 1 <= len(p["wait"]) <= d["commStreams"] and all(ev_wave[e] == 0 for e in p["wait"])
                assert p["scatter"] == []
            assert [p["stream"] for p in mine] == [i % 2 for i in range(len(mine))]
            assert abs(sum(p["flops"] for p in mine) - 2.0 * E ** 3 / n) < 1e-3 * E ** 3
        # every device receives from every other device exactly once (all xGMI links busy, none twice)
        pairs = sorted((t["src"], t["dst"]) for t in d["transfers"])
        assert pairs == sorted((s, r) for s in range(n) for r in range(n) if s != r)


@pytest.mark.parametrize("n,waves", [(4, 3), (8, 7), (8, 2)])
def test_gather_in_waves_orders_cells_by_first_use(cm, n, waves, monkeypatch):
    """CUTENSORMG_AMD_WAVES=k: the remote cells arrive in k groups with one event each; a piece waits only for the
    wave(s) that carry its cells and the waves are consumed in order."""
    monkeypatch.setenv("CUTENSORMG_AMD_WAVES", str(waves))
    E = 64 * n
    with cm.Contraction(list(range(n)), *free_mode_layout(n, E)) as con:
        d = con.describe()
        assert d["numWaves"] == waves
        check_transfers(d, con, n)
        assert (coverage(d, con, E) == 1).all()
        ev_wave = {t["event"]: t["wave"] for t in d["transfers"]}
        for g in range(n):
            mine = [p for p in d["pieces"] if p["dev"] == g]
            assert mine[0]["wait"] == []
            last = -1
            for p in mine[1:]:
                ws = sorted(ev_wave[e] for e in p["wait"])
                assert ws and ws[0] >= last
                last = ws[0]


@pytest.mark.parametrize("n", [1, 2, 4, 8])
def test_sample_block_cyclic_layout(cm, n):
    """contraction_multi_gpu.cu:154-217: 2 x 2 block-cyclic descriptors, cells owned by the handle devices cyclically."""
    E, BS = 256, 64
    modes = ["ik", "kj", "ij"]
    block = [dict(i=BS, k=BS), dict(k=BS, j=BS), dict(i=BS, j=BS)]
    dcount = [dict(i=2, k=2), dict(k=2, j=2), dict(i=2, j=2)]
    with cm.Contraction(list(range(n)), modes, dict(i=E, j=E, k=E), block, dcount) as con:
        d = con.describe()
        seen = check_transfers(d, con, n)
        assert (coverage(d, con, E) == 1).all()
        # every staged piece of C goes back to the owners' cells: each (cell, region) exactly once
        sc = [(p["lo"], p["hi"], p["q0"], p["q1"], c) for p in d["pieces"] for c in p["scatter"]]
        assert len(sc) == len(set(sc))
        for p in d["pieces"]:
            if not p["use"][2]["direct"]:
                assert sorted(p["scatter"]) == sorted(p["use"][2]["cells"])
        if n == 1:
            assert d["remoteBytes"] == 0
        else:
            assert d["remoteBytes"] > 0 and any(not t["local"] for t in d["transfers"])


def test_device_counts_must_divide(cm):
    """A mode shared by two tensors: same blocks, device counts that divide one another — anything else is refused."""
    import ctypes
    h = ctypes.c_void_p()
    cm.check(cm.cutensorMgCreate(ctypes.byref(h), 6, cm.i32(list(range(6)))))

    def desc(ext, bs, dc):
        d = ctypes.c_void_p()
        ncell = int(np.prod(dc))
        cm.check(cm.cutensorMgCreateTensorDescriptor(h, ctypes.byref(d), 2, cm.i64(ext), None, cm.i64(bs), None, cm.i32(dc), ncell,
                                                     cm.i32([i % 6 for i in range(ncell)]), 0))
        return d

    dA, dB, dC = desc([96, 96], [16, 96], [2, 1]), desc([96, 96], [96, 16], [1, 3]), desc([96, 96], [16, 16], [2, 2])
    cd = ctypes.c_void_p()
    st = cm.cutensorMgCreateContractionDescriptor(h, ctypes.byref(cd), dA, cm.i32("ik"), dB, cm.i32("kj"), dC, cm.i32("ij"), dC, cm.i32("ij"),
                                                  cm.COMPUTE_32F)
    assert st == 15   # NOT_SUPPORTED: j is cut over 3 devices in B and 2 in C
    for d in (dA, dB, dC):
        cm.check(cm.cutensorMgDestroyTensorDescriptor(d))
    cm.check(cm.cutensorMgDestroy(h))


def test_execution_on_a_plan_only_handle_is_refused(cm):
    import ctypes
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the handle is a real one")
    with cm.Contraction([0, 1], *free_mode_layout(2, 128)) as con:
        buf = (ctypes.c_char * 64)()
        addr = ctypes.addressof(buf)
        st = con.run(1.0, [addr] * 2, [addr] * 2, 0.0, [addr] * 2, [addr] * 2, [addr] * 2, [0, 0])
        assert st != 0


def _blocks_of(E, bs, dc, coord):
    """Index set of a mode that lives in grid coordinate `coord`: blocks coord, coord + dc, ... of size bs (ragged last block)."""
    nblk = -(-E // bs)
    idx = []
    for b in range(coord, nblk, dc):
        idx.extend(range(b * bs, min(E, (b + 1) * bs)))
    return idx


@pytest.mark.parametrize("seed", range(6))
def test_ragged_and_mixed_layouts_cover_c_once_and_gather_what_they_read(cm, seed):
    """Random block-cyclic layouts of C[i,j] = A[i,k] B[k,j] over 2-4 devices with extents that do not divide the block size
    (the blog_post harness's ceil()-derived blocks): whatever p / q cut the planner takes, every element of C is produced exactly
    once (the p index ranges times the q-coordinate classes tile the padded index space), a cell is gathered at most once per
    consumer, and every cell a piece reads from a staging image was gathered to that device."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([2, 3, 4]))
    Ei, Ej, Ek = (int(rng.integers(40, 200)) for _ in range(3))
    bi, bj = int(rng.integers(8, 48)), int(rng.integers(8, 48))
    di = n
    dj = int(rng.choice([1, 2])) if n % 2 == 0 else 1
    modes = ["ik", "kj", "ij"]
    extent = dict(i=Ei, j=Ej, k=Ek)
    block = [dict(i=bi), dict(j=bj), dict(i=bi, j=bj)]
    dcount = [dict(i=di), dict(j=dj), dict(i=di, j=dj)]
    with cm.Contraction(list(range(n)), modes, extent, block, dcount) as con:
        d = con.describe()
        check_transfers(d, con, n)
        cov = np.zeros((Ei, Ej), dtype=np.int32)
        # which j indices belong to q coordinate c (q is a mode of B / C cut dj ways), or all of j when q is not j
        for p in d["pieces"]:
            assert 0 <= p["lo"] < p["hi"]
            if d["pLabel"] == ord("i"):
                rows = np.arange(p["lo"], min(p["hi"], Ei))
                if d["qLabel"] == ord("j") and p["q1"] > 0:
                    cols = np.array(sorted(sum((_blocks_of(Ej, bj, dj, c) for c in range(p["q0"], p["q1"])), [])), dtype=np.int64)
                else:
                    cols = np.arange(Ej)
            else:
                assert d["pLabel"] == ord("j")
                cols = np.arange(p["lo"], min(p["hi"], Ej))
                if d["qLabel"] == ord("i") and p["q1"] > 0:
                    rows = np.array(sorted(sum((_blocks_of(Ei, bi, di, c) for c in range(p["q0"], p["q1"])), [])), dtype=np.int64)
                else:
                    rows = np.arange(Ei)
            if p["hi2"] > p["lo2"]:      # the second sharded mode (p shorter than 16 x devices) cuts the other axis
                cut = np.arange(p["lo2"], p["hi2"])
                if d["p2Label"] == ord("i"):
                    rows = np.intersect1d(rows, cut)
                else:
                    assert d["p2Label"] == ord("j")
                    cols = np.intersect1d(cols, cut)
            if len(rows) and len(cols):
                cov[np.ix_(rows, cols)] += 1
        assert (cov == 1).all(), (seed, n, extent, block, dcount, int(cov.min()), int(cov.max()))


@pytest.mark.parametrize("seed", range(8))
def test_kboxes_tile_the_valid_index_space_exactly(cm, seed):
    """Ragged CONTRACTED modes (mg.cpp kbox_list): idx = w + bs * b, b = sum digit_j * prod f_<j.  For random block sizes, digit
    extents and extents, the boxes are disjoint, lie inside [0, extent) and cover it; at most digits + 1 of them."""
    rng = np.random.default_rng(500 + seed)
    for _ in range(40):
        bs = int(rng.integers(1, 9))
        f = [int(rng.integers(1, 5)) for _ in range(int(rng.integers(0, 4)))]
        nblk = int(np.prod(f)) if f else 1
        extent = int(rng.integers(1, bs * nblk + 1))
        boxes = cm.kboxes(extent, bs, f)
        assert 1 <= len(boxes) <= len(f) + 1
        seen = np.zeros(bs * nblk, dtype=np.int32)
        for b in boxes:
            ranges = [range(lo, hi) for lo, hi in b["digits"]]
            import itertools
            for digs in itertools.product(*ranges) if ranges else [()]:
                blk, mul = 0, 1
                for dj, fj in zip(digs, f):
                    blk += dj * mul
                    mul *= fj
                seen[blk * bs: blk * bs + b["wHi"]] += 1
        assert (seen[:extent] == 1).all() and (seen[extent:] == 0).all(), (extent, bs, f, boxes)


def test_ragged_contracted_mode_is_planned_as_boxes(cm):
    """k = 176 in blocks of 32 over 2 devices (5.5 blocks, padded to 6): {local block < 2} + {local block 2, grid digit 0}
    + the half block -> 3 local contractions per piece; a k that fills its padded space stays one box."""
    modes = ["ik", "kj", "ij"]
    block = [dict(i=32, k=32), dict(k=32, j=32), dict(i=32, j=32)]
    dcount = [dict(i=2, k=2), dict(k=2, j=2), dict(i=2, j=2)]
    with cm.Contraction([0, 1], modes, dict(i=128, j=128, k=176), block, dcount) as con:
        assert con.describe()["numBoxes"] == 3
    with cm.Contraction([0, 1], modes, dict(i=128, j=128, k=192), block, dcount) as con:
        assert con.describe()["numBoxes"] == 1
    # 16-bit data: the boxes would accumulate through D with one rounding to the 16-bit type each -> still refused
    with pytest.raises(Exception) as ei:
        cm.Contraction([0, 1], modes, dict(i=128, j=128, k=176), block, dcount, dtype=14)      # HIP_R_16BF
    assert "NOT_SUPPORTED" in str(ei.value)


def _blog_post_shapes(n, s):
    """cuTENSORMg/blog_post.cu:155-175 (extents, ceil()-derived block sizes — including its N1 expression as written) and
    :78-101 (device counts: from the last mode down, double while blocks and devices remain)."""
    import math
    M0, M1, M2, N0, N1, N2, K0, K1, K2 = "abcdefghi"
    ext = {M0: 16, M1: 8 * s, M2: 8, N0: 16, N1: 8 * s, N2: 8, K0: 16, K1: 32, K2: 8}
    nM = n // 2 if n >= 4 else n
    nN = n // nM
    M = ext[M0] * ext[M1] * ext[M2]
    N = ext[N0] * ext[N1] * ext[N2]
    bs = {M0: 16, M2: 8, N0: 16, N2: 8, K0: 16, K1: 16, K2: 8}
    bs[M1] = math.ceil(math.ceil(M / math.ceil(M / 4096.0 / nM)) / nM / ext[M0] / ext[M2])
    bs[N1] = math.ceil(math.ceil(N / math.ceil(N / 4096.0 / nN)) / nN / ext[N0] / ext[N1])
    modes = [K0 + M0 + M1 + K1 + M2 + K2, K0 + N0 + K1 + N1 + K2 + N2, M0 + N0 + M1 + N1 + M2 + N2]
    block, dcount = [], []
    for m in modes:
        dc = {c: 1 for c in m}
        rem, changed = n, True
        while changed:
            changed = False
            for c in reversed(m):
                if rem <= 1:
                    break
                if dc[c] < ext[c] // bs[c]:
                    dc[c] *= 2
                    rem //= 2
                    changed = True
        assert rem == 1 or n == 1 or all(dc[c] >= ext[c] // bs[c] for c in m)
        block.append({c: bs[c] for c in m})
        dcount.append(dc)
    return modes, ext, block, dcount


@pytest.mark.parametrize("n", [1, 2, 4, 8])
def test_every_blog_post_configuration_plans(cm, n):
    """blog_post.cu <numDevices> <scaling> for scaling 1..12 (:131-146): descriptors, contraction descriptor and plan are created
    for every one of them without a GPU (plans whose local views need the mode-table kernel upload its table at first execution)."""
    for s in range(1, 13):
        modes, ext, block, dcount = _blog_post_shapes(n, s)
        ncell = [int(np.prod(list(dc.values()))) for dc in dcount]
        if any(c != ncell[0] for c in ncell) or (n > 1 and ncell[0] != n):
            # the sample hands numDevices / remainingDevices cells to the descriptor (:125-127): fewer than n when the blocks run out
            pass
        with cm.Contraction(list(range(n)), modes, ext, block, dcount) as con:
            d = con.describe()
            assert d["numBoxes"] == 1 and len(d["pieces"]) >= 1     # K0 / K1 / K2 extents divide their blocks: never ragged
            # a group of the local view with > 4 unfusable modes is peeled (one tiled contraction per value of its smallest
            # block-index digit) instead of being left to the mode-table kernel: 8 devices from scaling 2, 4 devices from scaling 9
            # (by the single-GPU library — one tiled inner plan per index combination — or, where its budget refuses, by this plan)
            assert (d["peeled"] + d["libraryPeeled"] > 0) == ((n == 8 and s >= 2) or (n == 4 and s >= 9)), (n, s, d["peeled"], d["libraryPeeled"])
            assert d["modeTable"] == 0 and d["localContractions"] >= len(d["pieces"])


@pytest.mark.parametrize("n,s", [(8, 1), (8, 2), (4, 1)])
def test_short_first_mode_is_sharded_along_a_second_one(cm, n, s):
    """blog_post.cu <n> <scaling> at small scalings (:155-175): the largest free mode of C has 16 indices — shorter than 16 x devices, so
    the 16-index shard rule alone would leave all but one or two devices idle.  The plan cuts a second free mode (of the other
    operand) as well: every device gets pieces, the pieces' flops add up to the problem, both operands shrink per device."""
    modes, ext, block, dcount = _blog_post_shapes(n, s)
    with cm.Contraction(list(range(n)), modes, ext, block, dcount) as con:
        d = con.describe()
        assert d["p2Label"] != -1 and d["p2Label"] != d["pLabel"] and d["qLabel"] == -1, d["p2Label"]
        assert sorted({p["dev"] for p in d["pieces"]}) == list(range(n))
        total = 2.0 * np.prod([float(v) for v in ext.values()])
        assert abs(sum(p["flops"] for p in d["pieces"]) - total) < 1e-4 * total      # (flops are printed with six digits)
        per_dev = [sum(p["flops"] for p in d["pieces"] if p["dev"] == g) for g in range(n)]
        assert max(per_dev) <= 1.01 * total / n * (2 if n == 8 and s == 1 else 1.5), per_dev
        in_a, in_b = chr(d["pLabel"]) in modes[0], chr(d["p2Label"]) in modes[0]
        assert in_a != in_b        # one mode of each operand
    # the old rule on request
    import os
    os.environ["CUTENSORMG_AMD_SHARD2"] = "0"
    try:
        with cm.Contraction(list(range(n)), modes, ext, block, dcount) as con:
            d = con.describe()
            assert d["p2Label"] == -1 and len({p["dev"] for p in d["pieces"]}) < n
    finally:
        del os.environ["CUTENSORMG_AMD_SHARD2"]


@pytest.mark.parametrize("n", [2, 4, 8])
def test_all_gather_transport_is_selected_for_the_free_mode_layout(cm, n, monkeypatch):
    """north_star / SURVEY 8e: "one RCCL ncclAllGather of B's shards".  With RCCL (assumed here: plan-only handle), the bench
    layout's B — one column slab per device, slab c on device c, every device needs all of them in one wave — qualifies for the
    all-gather transport: its staging image [cell][cell buffer] is the collective's receive layout.  A (never gathered) does not;
    CUTENSORMG_AMD_TRANSPORT=sendrecv / allgather pin the choice, default = timed trial of both on the first two calls."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("plan-only handles need a host without a GPU")
    monkeypatch.setenv("CUTENSORMG_AMD_ASSUME_RCCL", "1")
    E = 128 * n
    with cm.Contraction(list(range(n)), *free_mode_layout(n, E)) as con:
        d = con.describe()
        assert d["useRccl"] == 1 and d["allGatherEligible"] == [0, 1] and d["transport"].startswith("auto(allgather")
        assert d["commStreams"] == 1 and d["numWaves"] == 1
        check_transfers(d, con, n)
        # no remote scatter (C is sharded like the work) and RCCL: no cross-device ordering owed at the join
        assert all(o == [] for o in d["scatterOwners"])
    monkeypatch.setenv("CUTENSORMG_AMD_TRANSPORT", "sendrecv")
    with cm.Contraction(list(range(n)), *free_mode_layout(n, E)) as con:
        assert con.describe()["transport"] == "sendrecv"
    monkeypatch.setenv("CUTENSORMG_AMD_TRANSPORT", "allgather")
    with cm.Contraction(list(range(n)), *free_mode_layout(n, E)) as con:
        assert con.describe()["transport"] == "allgather"
    monkeypatch.delenv("CUTENSORMG_AMD_TRANSPORT")
    # the sample's 2 x 2 block-cyclic layout on 4 devices: cells are gathered, but not as one slab per device -> send/recv
    modes = ["ik", "kj", "ij"]
    blk = [dict(i=64, k=64), dict(k=64, j=64), dict(i=64, j=64)]
    dc = [dict(i=2, k=2), dict(k=2, j=2), dict(i=2, j=2)]
    with cm.Contraction([0, 1, 2, 3], modes, dict(i=256, j=256, k=256), blk, dc) as con:
        d = con.describe()
        assert d["transport"] == "sendrecv" or d["allGatherEligible"] != [0, 0]
        # remote scatter: some device stores pieces of C into cells another device owns -> that owner's stream must wait
        owners = [set(o) for o in d["scatterOwners"]]
        for p in d["pieces"]:
            for c in p["scatter"]:
                own = con.cells[2]["owners"][c]
                if own != con.devices[p["dev"]]:
                    assert con.devices.index(own) in owners[p["dev"] * 2 + p["stream"]]


def test_peer_transport_orders_owner_streams_behind_remote_reads(cm):
    """Peer copies (no RCCL): device g's communication streams read the owners' cells directly, so every owner's caller stream must
    end behind those reads (readOwners) — otherwise the owner could overwrite its operand while a peer still reads it."""
    n = 4
    with cm.Contraction(list(range(n)), *free_mode_layout(n, 128 * n)) as con:
        d = con.describe()
        assert d["transport"] == "peer"
        for g in range(n):
            assert sorted(d["readOwners"][g]) == [o for o in range(n) if o != g]

"""world_size-2 `gloo` tests of the N>1 paths, the CPU oracle standing in for the HIP kernel (which needs a GPU):
  * north_star's own partitioning (cuTENSORMg/contraction_multi_gpu.cu:154-193): the largest free mode of C sharded, the other
    operand ALL-GATHERED — the shard ranges, the cells each rank holds and the cells it must receive are read from the library's own plan
    (ctamdMgDescribePlan on a plan-only handle), the exchange is a real `dist.all_gather` between two processes;
  * the secondary path of bench.py under torch.distributed.run (cudalibrarysamples_amd/sharding.py): each rank contracts its slice of the
    contracted mode b, the partial results are summed by an all-reduce, and every rank must end with the full einsum."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    import oracle
    from cudalibrarysamples_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(99)       # same full problem on every rank
    a = rng.random((12, 6, 4, 8), dtype=np.float32)
    b = rng.random((8, 4, 6, 10), dtype=np.float32)
    a_r, b_r = sharding.shard_operands(a, b, world, rank)
    part = oracle.einsum("abcd,dcbe->ae", np.ascontiguousarray(a_r), np.ascontiguousarray(b_r))
    t = torch.from_numpy(np.ascontiguousarray(part))
    sharding.fold_partials(t, dist)
    full = oracle.einsum("abcd,dcbe->ae", a, b)
    err = float(np.max(np.abs(t.numpy() - full) / np.abs(full)))
    q.put((rank, err))
    dist.destroy_process_group()


def test_contracted_ranges_partition():
    from cudalibrarysamples_amd import sharding
    for extent in (64, 65, 7, 128):
        for world in (1, 2, 3, 8):
            cuts = [sharding.contracted_range(extent, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == extent
            for (l0, h0), (l1, h1) in zip(cuts, cuts[1:]):
                assert h0 == l1
            sizes = [h - l for l, h in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.contracted_range(8, 2, 2)


def test_sharded_einsum_gloo_world2(built):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == [0, 1]
    for _, err in res:
        assert err < 1e-5


def _allgather_worker(rank, world, port, q):
    """Rank r owns A[i_r, :] (row slab), B[:, j_r] (column slab) and C[i_r, :]; B's slabs are all-gathered, every local piece of the
    library's plan for this rank runs through the oracle on exactly the index ranges the plan names."""
    import torch
    import torch.distributed as dist

    import oracle
    from cudalibrarysamples_amd import cutensormg as cm
    from tests.test_mg_plan_cpu import free_mode_layout

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E = 64                                 # (shards of the free mode start at multiples of 16 indices: mg.cpp)
    rng = np.random.default_rng(7)         # the same full problem on every rank; a rank only TOUCHES its slabs until the gather
    A = rng.random((E, E), dtype=np.float32)       # A[i, k]
    B = rng.random((E, E), dtype=np.float32)       # B[k, j]
    modes, extent, block, dcount = free_mode_layout(world, E)
    with cm.Contraction(list(range(world)), modes, extent, block, dcount) as con:
        d = con.describe()
    assert d["pLabel"] == ord("i") and d["qLabel"] == ord("j"), d
    mine = [p for p in d["pieces"] if p["dev"] == rank]
    lo, hi = mine[0]["lo"], mine[0]["hi"]
    assert all((p["lo"], p["hi"]) == (lo, hi) for p in mine)
    bj = block[1]["j"]
    a_slab = np.ascontiguousarray(A[lo:hi])                             # this rank's cells of A and B
    b_slab = torch.from_numpy(np.ascontiguousarray(B[:, rank * bj:(rank + 1) * bj]))
    # the plan's transfers INTO this rank: the cells of B it does not hold — one per other rank, each the size of a slab
    incoming = sorted((t["src"], t["cell"], t["bytes"]) for t in d["transfers"] if t["dst"] == rank and not t["local"])
    assert incoming == [(r, r, E * bj * 4) for r in range(world) if r != rank], incoming
    gathered = [torch.empty_like(b_slab) for _ in range(world)]
    dist.all_gather(gathered, b_slab)                                   # RCCL ncclAllGather on the GPU box (mg.cpp), gloo here
    c_slab = np.full((hi - lo, E), np.nan, dtype=np.float32)
    first_is_local = None
    for n, p in enumerate(mine):                                        # pieces in execution order: q coordinate ranges of B's grid
        for cell in range(p["q0"], p["q1"]):
            if n == 0:
                first_is_local = (cell == rank) and not p["wait"]
            bcell = gathered[cell].numpy() if cell != rank else b_slab.numpy()
            c_slab[:, cell * bj:(cell + 1) * bj] = oracle.einsum("ik,kj->ij", a_slab, np.ascontiguousarray(bcell))
    full = oracle.einsum("ik,kj->ij", A, B)
    err = float(np.max(np.abs(c_slab - full[lo:hi]) / np.abs(full[lo:hi])))
    # C stays sharded (a valid Mg output layout); rank 0 collects the slabs only to check that they tile C
    slabs = [torch.empty((E // world, E), dtype=torch.float32) for _ in range(world)]
    dist.all_gather(slabs, torch.from_numpy(c_slab))
    whole = float(np.max(np.abs(np.concatenate([t.numpy() for t in slabs]) - full) / np.abs(full)))
    q.put((rank, err, whole, bool(first_is_local), (lo, hi)))
    dist.destroy_process_group()


def test_free_mode_shard_allgather_gloo_world2(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("plan-only handles for device ids 0..1 need a host without GPUs")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_allgather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert [r[0] for r in res] == [0, 1]
    assert [r[4] for r in res] == [(0, 32), (32, 64)]                  # the free mode i, cut into one contiguous shard per rank
    for _, err, whole, first_local, _ in res:
        assert err < 1e-5 and whole < 1e-5
        assert first_local            # the first local contraction depends on nothing remote: the gather runs under it

"""world_size-2 `gloo` test of the N>1 path of bench.py (cudalibrarysamples_amd/sharding.py): each rank
contracts its slice of the contracted mode b with the CPU oracle (standing in for the HIP kernel, which
needs a GPU), the partial results are summed by an all-reduce, and every rank must end with the full
einsum.  Checks the partition (balanced, contiguous, exhaustive) and the exchange."""
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

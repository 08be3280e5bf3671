"""Multi-process (one process per GPU) sharding of the headline einsum — the secondary N > 1 measurement of bench.py under
torch.distributed.run (the primary one is cuTENSORMg over N devices).

'abcd,dcbe->ae' has a 96 x 96 result and a contracted volume b*c*d of 262,144 per 64 b's, so the
cheapest partition is along a *contracted* mode: rank r owns A[:, b_r, :, :] and B[:, :, b_r, :] for a
contiguous range b_r of b, contracts it locally into a full-size partial C, and the partials are summed
by one all-reduce of 36 KB (RCCL over xGMI on the GPU box; gloo in the CPU tests).  Sharding a free
mode instead would have to all-gather B (100 MB per 64 b's).  No torch types appear below the einsum
call itself; this module only computes index ranges and issues the collective.
"""


def contracted_range(extent, world, rank):
    """Contiguous, balanced [start, stop) of a contracted mode of the given extent owned by `rank`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(extent, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_operands(a, b, world, rank):
    """Views of row-major a[a,b,c,d] and b[d,c,b,e] restricted to this rank's slice of mode b."""
    lo, hi = contracted_range(a.shape[1], world, rank)
    return a[:, lo:hi], b[:, :, lo:hi]


def fold_partials(partial, dist=None, async_op=False, group=None):
    """Sum the per-rank partial results in place (all-reduce on `group`, default: the world); no-op for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    return dist.all_reduce(partial, async_op=async_op, group=group)

"""CPU coverage of the multi-rank (N > 1) cuTENSORMp path: every rank derives its transfer list from the descriptors
alone, so the lists of all ranks must fit together — what rank s sends to rank q is what q expects from s, tensor by
tensor, in the same order and size — and every rank must choose the same exchange algorithm from the same byte counts.
Plans are built on plan-only handles (no GPU here; nothing is executed) for the ranks of a local world, one after the
other, and compared through ctamdMpDescribePlan."""
import ctypes

import numpy as np
import pytest


@pytest.fixture(scope="module")
def cmp(built):
    from cudalibrarysamples_amd import cutensor as ct
    from cudalibrarysamples_amd import cutensormp
    return cutensormp, ct


def plans_of_all_ranks(cmp_ct, eq, ext, dist, nranks, dtype=None, ranks=None, algo=None, monkeypatch=None):
    cmp, ct = cmp_ct
    dtype = dtype or ct.R_32F
    if monkeypatch is not None:
        if algo:
            monkeypatch.setenv("CUTENSORMP_AMD_ALGO", algo)
        else:
            monkeypatch.delenv("CUTENSORMP_AMD_ALGO", raising=False)
    lhs, mc = eq.split("->")
    ma, mb = lhs.split(",")
    modes = [ma, mb, mc]
    P = [[dist[k].get(l, 1) for l in modes[k]] for k in range(3)]
    perm = ranks or [None, None, None]
    world = cmp.LocalWorld(nranks)
    out = []
    try:
        for r in range(nranks):
            h = ctypes.c_void_p()
            cmp.check(cmp.ctamdMpCreateOnLocalWorld(ctypes.byref(h), world.ptr, r, 0, None))
            descs = []
            for k in range(3):
                d = ctypes.c_void_p()
                pr = None if perm[k] is None or int(np.prod(P[k])) == 1 else ct.i32(perm[k])
                cmp.check(cmp.cutensorMpCreateTensorDescriptor(h, ctypes.byref(d), len(modes[k]), ct.i64([ext[l] for l in modes[k]]),
                                                               None, None, None, ct.i64(P[k]), nranks, pr, dtype))
                descs.append(d)
            lab = [ct.i32([ord(c) for c in m]) for m in modes]
            op, pref, plan = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            cmp.check(cmp.cutensorMpCreateContraction(h, ctypes.byref(op), descs[0], lab[0], ct.OP_IDENTITY, descs[1], lab[1], ct.OP_IDENTITY,
                                                      descs[2], lab[2], ct.OP_IDENTITY, descs[2], lab[2], ct.compute_desc("32F")))
            cmp.check(cmp.cutensorMpCreatePlanPreference(h, ctypes.byref(pref), cmp.ALGO_DEFAULT, 1 << 30, 0))
            cmp.check(cmp.cutensorMpCreatePlan(h, ctypes.byref(plan), op, pref))
            need = ctypes.c_uint64(0)
            cmp.check(cmp.cutensorMpPlanGetAttribute(h, plan, cmp.PLAN_REQUIRED_WORKSPACE_DEVICE, ctypes.byref(need), 8))
            d = cmp.describe_plan(plan)
            assert d["requiredDevice"] == need.value and d["rank"] == r and d["nranks"] == nranks
            # executing on a plan-only handle is refused, not crashed
            one = ctypes.c_float(1.0)
            buf = (ctypes.c_char * 64)()
            st = cmp.cutensorMpContract(h, plan, ctypes.byref(one), buf, buf, ctypes.byref(one), buf, buf, buf, None)
            assert st != 0
            out.append(d)
            cmp.check(cmp.cutensorMpDestroyPlan(plan))
            cmp.check(cmp.cutensorMpDestroyPlanPreference(pref))
            cmp.check(cmp.cutensorMpDestroyOperationDescriptor(op))
            for x in descs:
                cmp.check(cmp.cutensorMpDestroyTensorDescriptor(x))
            cmp.check(cmp.cutensorMpDestroy(h))
    finally:
        world.close()
    return out


def check_lists_fit(plans):
    n = len(plans)
    assert len({p["algorithm"] for p in plans}) == 1
    assert len({(p["gatherTotal"], p["reduceTotal"]) for p in plans}) == 1
    for s in range(n):
        for q in range(n):
            sent = [(t["tensor"], t["bytes"]) for t in plans[s]["sends"] if t["dst"] == q]
            expected = [(t["tensor"], t["bytes"]) for t in plans[q]["recvs"] if t["src"] == s]
            assert sent == expected, (s, q, sent, expected)
            assert all(t["src"] == s for t in plans[s]["sends"]) and all(t["dst"] == q for t in plans[q]["recvs"])
    if plans[0]["algorithm"] == "gather":
        assert sum(t["bytes"] for p in plans for t in p["recvs"]) == plans[0]["gatherTotal"]
    else:
        assert all(p["sends"] == [] and p["recvs"] == [] for p in plans)


CASES = [
    ("mk,kn->mn", dict(m=96, k=80, n=64), ({"m": 2}, {}, {"m": 2}), 2, None),
    ("mk,kn->mn", dict(m=100, k=72, n=52), ({"m": 2, "k": 2}, {"k": 2, "n": 2}, {"m": 2, "n": 2}), 4, None),
    ("mk,kn->mn", dict(m=64, k=64, n=64), ({"k": 4}, {"n": 4}, {"m": 2, "n": 2}), 4, None),
    ("akcl,lbk->abc", dict(a=30, b=21, c=9, k=16, l=12), ({"c": 3}, {"b": 3}, {"a": 3}), 3, None),
    ("mk,kn->mn", dict(m=5, k=16, n=8), ({"m": 8}, {}, {"m": 8}), 8, None),
    ("mkl,lkn->mn", dict(m=50, k=12, l=10, n=27), ({"k": 2, "l": 2}, {"l": 2, "k": 2}, {"m": 2, "n": 2}), 4, [None, [0, 2, 1, 3], None]),
    ("abcdefEFGH,abcdefABCD->EFGHABCD", {l: 2 for l in "abcdefEFGHABCD"}, ({"E": 2, "F": 2, "G": 2}, {"a": 2, "b": 2, "c": 2}, {"A": 2, "B": 2, "C": 2}), 8, None),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] + "/%d" % c[3] for c in CASES])
def test_transfer_lists_of_all_ranks_fit_together(cmp, monkeypatch, case):
    eq, ext, dist, nranks, ranks = case
    for algo in ("gather", None):
        check_lists_fit(plans_of_all_ranks(cmp, eq, ext, dist, nranks, ranks=ranks, algo=algo, monkeypatch=monkeypatch))


def test_headline_einsum_k_sharded_over_eight_ranks_reduces(cmp, monkeypatch):
    """BASELINE's headline einsum with the contracted mode b cut over 8 ranks (the sharding bench.py --gpus 8 uses): no
    operand byte moves, one all-reduce of the 36-KB result in the user's D; the gather alternative would move 1.4 GB."""
    ext = dict(a=96, b=64, c=64, d=64, e=96)
    plans = plans_of_all_ranks(cmp, "dcba,ebcd->ea", ext, ({"b": 8}, {"b": 8}, {}), 8, monkeypatch=monkeypatch)
    check_lists_fit(plans)
    for p in plans:
        assert p["algorithm"] == "reduce" and p["reduceDirect"]
        assert p["reduceTotal"] == 8 * 96 * 96 * 4 * 2
        assert p["gatherTotal"] == 8 * 7 * 2 * (96 * 64 * 64 * 64 * 4 // 8)
        assert p["sendBytes"] == 0 and p["recvBytes"] == 0


def test_descriptor_and_operation_validation(cmp):
    """Error behaviour of the Mp descriptors (status codes, nothing crashes): grids that do not match the world, rank lists
    that are not permutations, block-cyclic layouts, D distributed differently from C, extent mismatches."""
    cm, ct = cmp
    world = cm.LocalWorld(4)
    try:
        h = ctypes.c_void_p()
        cm.check(cm.ctamdMpCreateOnLocalWorld(ctypes.byref(h), world.ptr, 1, 0, None))

        def desc(ext, p, ranks=None, block=None, n=4, dtype=None):
            d = ctypes.c_void_p()
            st = cm.cutensorMpCreateTensorDescriptor(h, ctypes.byref(d), len(ext), ct.i64(ext), None, ct.i64(block) if block else None, None,
                                                     ct.i64(p), n, ct.i32(ranks) if ranks else None, dtype or ct.R_32F)
            return st, d

        assert desc([64, 32], [2, 2])[0] == 0
        assert desc([64, 32], [1, 1])[0] == 0                                 # one cell: replicated
        assert desc([64, 32], [2, 1])[0] == ct.STATUS_INVALID_VALUE           # 2 cells in a world of 4
        assert desc([64, 32], [4, 2], n=8)[0] == ct.STATUS_INVALID_VALUE      # 8 cells
        assert desc([64, 32], [2, 2], ranks=[0, 1, 1, 3])[0] == ct.STATUS_INVALID_VALUE   # not a permutation
        assert desc([64, 32], [2, 2], ranks=[0, 1, 2, 4])[0] == ct.STATUS_INVALID_VALUE   # rank outside the world
        assert desc([64, 32], [2, 2], block=[16, 16])[0] == ct.STATUS_NOT_SUPPORTED       # block-cyclic (two blocks per rank)
        assert desc([64, 32], [2, 2], block=[32, 16])[0] == 0                             # the default blocks, spelled out
        assert desc([0, 32], [2, 2])[0] == ct.STATUS_INVALID_VALUE
        assert cm.cutensorMpCreateTensorDescriptor(None, ctypes.byref(ctypes.c_void_p()), 1, ct.i64([4]), None, None, None, ct.i64([1]), 1, None,
                                                   ct.R_32F) == ct.STATUS_NOT_INITIALIZED

        (_, dA), (_, dB), (_, dC) = desc([64, 32], [4, 1]), desc([32, 48], [1, 1]), desc([64, 48], [4, 1])
        (_, dD2), (_, dBad) = desc([64, 48], [1, 4]), desc([31, 48], [1, 1])
        lab = [ct.i32("mk"), ct.i32("kn"), ct.i32("mn")]
        op = ctypes.c_void_p()
        args = lambda a, b, c, d_: (h, ctypes.byref(op), a, lab[0], ct.OP_IDENTITY, b, lab[1], ct.OP_IDENTITY, c, lab[2], ct.OP_IDENTITY, d_, lab[2],
                                    ct.compute_desc("32F"))
        assert cm.cutensorMpCreateContraction(*args(dA, dB, dC, dC)) == 0
        assert cm.cutensorMpCreateContraction(*args(dA, dB, dC, dD2)) == ct.STATUS_NOT_SUPPORTED     # D laid out differently from C
        assert cm.cutensorMpCreateContraction(*args(dA, dBad, dC, dC)) == ct.STATUS_INVALID_VALUE    # k: 32 vs 31
        assert cm.cutensorMpCreateContraction(*args(dA, None, dC, dC)) == ct.STATUS_INVALID_VALUE
        pref = ctypes.c_void_p()
        assert cm.cutensorMpCreatePlanPreference(h, ctypes.byref(pref), 7, 1 << 20, 0) == ct.STATUS_NOT_SUPPORTED   # unknown algorithm
        cm.check(cm.cutensorMpCreatePlanPreference(h, ctypes.byref(pref), cm.ALGO_DEFAULT, 1024, 0))
        plan = ctypes.c_void_p()
        # a 1-KiB device budget: enough when every operand box is local (nothing is staged), not for a gathered case
        assert cm.cutensorMpCreatePlan(h, ctypes.byref(plan), op, pref) == 0
        cm.check(cm.cutensorMpDestroyPlan(plan))
        (_, dAk), (_, dBk) = desc([64, 32], [1, 4]), desc([32, 48], [4, 1])
        (_, dCr) = desc([64, 48], [2, 2])
        assert cm.cutensorMpCreateContraction(*args(dAk, dBk, dCr, dCr)) == 0
        assert cm.cutensorMpCreatePlan(h, ctypes.byref(plan), op, pref) == ct.STATUS_INSUFFICIENT_WORKSPACE
        cm.check(cm.cutensorMpDestroy(h))
    finally:
        world.close()

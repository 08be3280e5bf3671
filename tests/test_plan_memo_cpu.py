"""Plan memo (csrc/host/api.cpp cutensorCreatePlan): the einsum.cu flow creates descriptors + plan inside every call
(cuTENSOR/einsum.cu:264-329) with the plan cache on (:443-445), so a repeated problem must be answered by a lookup.
Host-only: planning needs no GPU."""
import pytest


@pytest.fixture(scope="module")
def env(built):
    from cudalibrarysamples_amd import cutensor as ct, ops
    return ct, ops


def _headline(ops, h, **kw):
    # 'abcd,dcbe->ae' after the helper's reversal
    return ops.contraction_plan(h, [64, 64, 64, 96], "dcba", [96, 64, 64, 64], "ebcd", [96, 96], "ea", workspace_limit=1 << 30, **kw)


def test_repeated_problem_is_cloned_from_the_memo(env):
    ct, ops = env
    h = ops.Handle(plan_cache=1024)                       # einsum.cu:445
    p0 = _headline(ops, h)
    assert ct.plan_memo_stats(h.h) == (0, 1, 1)
    d0, ws0 = p0.describe(), p0.required_workspace
    for i in range(5):
        p = _headline(ops, h)
        assert p.describe() == d0 and p.required_workspace == ws0
        p.destroy()
    assert ct.plan_memo_stats(h.h) == (5, 1, 1)
    p0.destroy()                                          # prototypes outlive the plans cloned from them
    p = _headline(ops, h)
    assert p.describe() == d0
    p.destroy()


def test_key_separates_what_planning_depends_on(env):
    ct, ops = env
    h = ops.Handle(plan_cache=1024)
    base = _headline(ops, h)
    # a smaller workspace limit, an explicit candidate, other extents, other strides: all misses with their own result
    small = ops.contraction_plan(h, [64, 64, 64, 96], "dcba", [96, 64, 64, 64], "ebcd", [96, 96], "ea", workspace_limit=0)
    assert small.required_workspace == 0 and base.required_workspace > 0
    ranked = _headline(ops, h, algo=1)
    assert ranked.describe() != base.describe()
    other = ops.contraction_plan(h, [64, 64, 32, 96], "dcba", [96, 32, 64, 64], "ebcd", [96, 96], "ea", workspace_limit=1 << 30)
    assert other.describe()["K"] == 64 * 64 * 32
    strided = ops.contraction_plan(h, [64, 64, 64, 96], "dcba", [96, 64, 64, 64], "ebcd", [96, 96], "ea", workspace_limit=1 << 30,
                                   strideC=[1, 128])
    hits, misses, entries = ct.plan_memo_stats(h.h)
    assert (hits, misses, entries) == (0, 5, 5)
    # each of them hits on repetition, and an explicit candidate is never replaced by the memoised default
    again = _headline(ops, h, algo=1)
    assert again.describe() == ranked.describe()
    assert ct.plan_memo_stats(h.h)[0] == 1
    for p in (base, small, ranked, other, strided, again):
        p.destroy()


def test_cache_mode_none_and_capacity_zero_bypass_and_lru_evicts(env):
    ct, ops = env
    h = ops.Handle(plan_cache=2)
    p = _headline(ops, h, cache_mode=ct.CACHE_MODE_NONE)
    p.destroy()
    assert ct.plan_memo_stats(h.h) == (0, 0, 0)
    shapes = [(16, 32), (32, 16), (48, 16)]
    def mk(m, n):
        return ops.contraction_plan(h, [m, 64], "mk", [64, n], "kn", [m, n], "mn", workspace_limit=0)
    for m, n in shapes:
        mk(m, n).destroy()
    assert ct.plan_memo_stats(h.h) == (0, 3, 2)          # capacity 2: the first prototype was evicted
    mk(*shapes[2]).destroy(); mk(*shapes[1]).destroy()
    assert ct.plan_memo_stats(h.h)[0] == 2
    mk(*shapes[0]).destroy()                              # evicted earlier: planned again
    assert ct.plan_memo_stats(h.h)[:2] == (2, 4)
    ct.check(ct.cutensorHandleResizePlanCache(h.h, 0))
    mk(*shapes[0]).destroy()
    assert ct.plan_memo_stats(h.h) == (2, 4, 0)


def test_reduction_and_permutation_plans_are_memoised_too(env):
    """einsum.cu's unary path plans a cutensorCreateReduction per call (:352-377)."""
    ct, ops = env
    h = ops.Handle(plan_cache=16)
    r1 = ops.reduction_plan(h, [2, 4, 5], "nij", [5, 4], "ji")
    r2 = ops.reduction_plan(h, [2, 4, 5], "nij", [5, 4], "ji")
    q1 = ops.permutation_plan(h, [5, 4, 2], "jin", [2, 5, 4], "nji")
    q2 = ops.permutation_plan(h, [5, 4, 2], "jin", [2, 5, 4], "nji")
    assert ct.plan_memo_stats(h.h) == (2, 2, 2)
    for p in (r1, r2, q1, q2):
        p.destroy()


def test_two_step_plans_are_memoised_with_their_sub_plans(env):
    """Round 6: a contraction that reduces an operand over its lone modes first, or copies an operand into a packed temporary first
    (api.cpp plan_repack: up to eight copy combinations priced per cutensorCreatePlan), is a plan that owns sub-plans — the memo
    keeps a deep copy and hands out deep copies; destroying one clone leaves the others (and the prototype) intact."""
    ct, ops = env
    h = ops.Handle(plan_cache=16)
    ext = dict(i=4096, l=4096, j=16, k=72)
    mk = lambda: ops.contraction_plan(h, [ext[c] for c in "kji"], "kji", [ext[c] for c in "jkl"], "jkl", [ext[c] for c in "li"], "li",   # noqa: E731
                                      dtype=ct.R_16BF, workspace_limit=1 << 30)
    p1 = mk()
    d1 = p1.describe()
    assert d1.get("repack_A") or d1.get("repack_B"), d1
    hits0, misses0, _ = ct.plan_memo_stats(h.h)
    p2, p3 = mk(), mk()
    hits1, misses1, _ = ct.plan_memo_stats(h.h)
    assert hits1 == hits0 + 2 and misses1 == misses0, (hits0, misses0, hits1, misses1)
    p1.destroy()
    p2.destroy()
    assert p3.describe() == d1 and p3.required_workspace == d1["lone_bytes"]
    p3.destroy()
    e = dict(i=64, j=8, k=32, l=48)
    lone = lambda: ops.contraction_plan(h, [e[c] for c in "ijk"], "ijk", [e[c] for c in "kl"], "kl", [e[c] for c in "il"], "il")   # noqa: E731
    q1 = lone()
    hits2 = ct.plan_memo_stats(h.h)[0]
    q2 = lone()
    assert ct.plan_memo_stats(h.h)[0] == hits2 + 1
    assert q2.describe() == q1.describe() and q1.describe()["lone_reduce_A"] == 1
    q1.destroy()
    q2.destroy()

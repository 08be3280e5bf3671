"""GPU parity of the cuTENSORMp path (include/cutensorMp.h; call sites cutensorMp/cutensorMp_contraction.cu:470-590).

The GPU box has one device and RCCL refuses two ranks on one device, so the multi-rank cases run the ranks as
threads of this process over the library's in-process exchange layer (ctamdMpCreateOnLocalWorld): descriptors,
plans, transfer lists, packing, staging and the local contraction are the production code, only the wire is a
device-to-device copy instead of ncclSend/ncclRecv.  The RCCL wire itself is covered with a one-rank communicator
here and by the reference sample in test_gpu_samples.py."""
import ctypes

import numpy as np
import pytest

from util import to_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mp(built):
    import torch
    assert torch.cuda.is_available()
    from cudalibrarysamples_amd import cutensor as ct
    from cudalibrarysamples_amd import cutensormp
    return cutensormp, ct, torch


def cell_slices(extent, p, cell):
    """Index ranges of grid cell `cell` (first mode fastest) for block size ceil(extent / p)
    (cutensorMp_contraction.cu:143-153)."""
    out = []
    for e, n in zip(extent, p):
        bs = -(-e // n)
        c = cell % n
        cell //= n
        out.append(slice(min(e, c * bs), min(e, (c + 1) * bs)))
    return tuple(out)


def local_block(G, p, cell):
    """The rank's local buffer: block extents ceil(E/p), packed first-mode-fastest, the valid part filled."""
    bs = [-(-e // n) for e, n in zip(G.shape, p)]
    buf = np.zeros(bs, dtype=G.dtype, order="F")
    sl = cell_slices(G.shape, p, cell)
    sub = G[sl]
    buf[tuple(slice(0, s) for s in sub.shape)] = sub
    return buf


def np_dtype_info(name):
    import torch
    from cudalibrarysamples_amd import cutensor as ct
    return {
        "f32": (np.float32, ct.R_32F, ct.compute_desc("32F"), 2e-5, None),
        "f64": (np.float64, ct.R_64F, ct.compute_desc("64F"), 1e-12, None),
        "c64": (np.complex64, ct.C_32F, ct.compute_desc("32F"), 2e-5, None),
        "bf16": (np.float32, ct.R_16BF, ct.compute_desc("32F"), 1.5e-2, torch.bfloat16),
    }[name]


def run_case(mp, eq, ext, dist, nranks, dtype="f32", alpha=1.0, beta=0.0, ranks=None, seed=0):
    """eq 'ab,bc->ac' (first listed mode is the fastest, as in the sample :191-227); ext {label: extent};
    dist = three dicts {label: ranks along that mode} for A, B, C; ranks = optional cell->rank permutations."""
    cmp, ct, torch = mp
    npdt, ctdt, compute, tol, tdt = np_dtype_info(dtype)
    lhs, mc = eq.split("->")
    ma, mb = lhs.split(",")
    modes = [ma, mb, mc]
    rng = np.random.default_rng(seed)

    def rand(m):
        shape = [ext[l] for l in m]
        x = rng.uniform(-1, 1, size=shape)
        if np.issubdtype(npdt, np.complexfloating):
            x = x + 1j * rng.uniform(-1, 1, size=shape)
        x = np.asfortranarray(x.astype(npdt))
        if tdt is not None:   # values representable in the 16-bit type
            x = np.asfortranarray(torch.from_numpy(np.ascontiguousarray(x)).to(tdt).to(torch.float32).numpy())
        return x

    G = [rand(m) for m in modes]
    P = [[dist[k].get(l, 1) for l in modes[k]] for k in range(3)]
    perm = ranks or [None, None, None]
    ref = alpha * np.einsum("%s,%s->%s" % (ma, mb, mc), G[0].astype(np.complex128 if "c" in dtype else np.float64),
                            G[1].astype(np.complex128 if "c" in dtype else np.float64)) + beta * G[2]

    def cell_of(k, r):
        ncells = int(np.prod(P[k]))
        if ncells == 1:
            return 0
        return r if perm[k] is None else perm[k].index(r)

    world = cmp.LocalWorld(nranks)
    results = [None] * nranks
    described = [None] * nranks

    def body(r):
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream()
        h = ctypes.c_void_p()
        cmp.check(cmp.ctamdMpCreateOnLocalWorld(ctypes.byref(h), world.ptr, r, 0, ctypes.c_void_p(stream.cuda_stream)))
        descs, bufs = [], []
        for k in range(3):
            d = ctypes.c_void_p()
            pr = None if perm[k] is None or int(np.prod(P[k])) == 1 else ct.i32(perm[k])
            cmp.check(cmp.cutensorMpCreateTensorDescriptor(h, ctypes.byref(d), len(modes[k]), ct.i64([ext[l] for l in modes[k]]),
                                                           None, None, None, ct.i64(P[k]), nranks, pr, ctdt))
            descs.append(d)
            blk = local_block(G[k], P[k], cell_of(k, r))
            if tdt is not None:
                t = torch.from_numpy(np.ascontiguousarray(blk.ravel(order="K"))).to(tdt).cuda()
            else:
                t = to_device(blk)
            bufs.append((t, blk))
        lab = [ct.i32([ord(c) for c in m]) for m in modes]
        op = ctypes.c_void_p()
        cmp.check(cmp.cutensorMpCreateContraction(h, ctypes.byref(op), descs[0], lab[0], ct.OP_IDENTITY, descs[1], lab[1], ct.OP_IDENTITY,
                                                  descs[2], lab[2], ct.OP_IDENTITY, descs[2], lab[2], compute))
        pref = ctypes.c_void_p()
        cmp.check(cmp.cutensorMpCreatePlanPreference(h, ctypes.byref(pref), cmp.ALGO_DEFAULT, 1 << 30, 1024))
        plan = ctypes.c_void_p()
        cmp.check(cmp.cutensorMpCreatePlan(h, ctypes.byref(plan), op, pref))
        need = ctypes.c_uint64(0)
        cmp.check(cmp.cutensorMpPlanGetAttribute(h, plan, cmp.PLAN_REQUIRED_WORKSPACE_DEVICE, ctypes.byref(need), 8))
        described[r] = cmp.describe_plan(plan)
        ws = torch.empty(max(int(need.value), 256), dtype=torch.uint8, device="cuda")
        if "c" in dtype:
            a, b = (ctypes.c_float * 2)(alpha, 0.0), (ctypes.c_float * 2)(beta, 0.0)
        elif dtype == "f64":
            a, b = ctypes.c_double(alpha), ctypes.c_double(beta)
        else:
            a, b = ctypes.c_float(alpha), ctypes.c_float(beta)
        c0 = bufs[2][0].clone()
        torch.cuda.synchronize()
        for rep in range(2):   # twice: the exchange buffers and the pair sequence numbers are reused
            if rep == 1:
                bufs[2][0].copy_(c0)
                torch.cuda.synchronize()
            cmp.check(cmp.cutensorMpContract(h, plan, ctypes.byref(a), bufs[0][0].data_ptr(), bufs[1][0].data_ptr(), ctypes.byref(b),
                                             bufs[2][0].data_ptr(), bufs[2][0].data_ptr(), ws.data_ptr(), None))
            stream.synchronize()
        out = bufs[2][0].to(torch.float32).cpu().numpy() if tdt is not None else bufs[2][0].cpu().numpy()
        results[r] = np.reshape(out, bufs[2][1].shape, order="F")
        cmp.check(cmp.cutensorMpDestroyPlan(plan))
        cmp.check(cmp.cutensorMpDestroyPlanPreference(pref))
        cmp.check(cmp.cutensorMpDestroyOperationDescriptor(op))
        for d in descs:
            cmp.check(cmp.cutensorMpDestroyTensorDescriptor(d))
        cmp.check(cmp.cutensorMpDestroy(h))

    try:
        cmp.run_ranks(nranks, body)
    finally:
        world.close()

    scale = max(1.0, float(np.max(np.abs(ref))))
    for r in range(nranks):
        sl = cell_slices(ref.shape, P[2], cell_of(2, r))
        want = ref[sl]
        got = results[r][tuple(slice(0, s) for s in want.shape)]
        err = float(np.max(np.abs(got - want))) if want.size else 0.0
        assert err <= tol * scale * max(1.0, np.sqrt(np.prod([ext[l] for l in ma if l not in mc]))), \
            "rank %d: max error %g (plan %s)" % (r, err, described[r])
    return described


CASES = [
    # name, eq, extents, (distA, distB, distC), nranks, dtype, alpha, beta
    ("row-sharded, B replicated: no exchange", "mk,kn->mn", dict(m=96, k=80, n=64), ({"m": 2}, {}, {"m": 2}), 2, "f32", 1.0, 0.0),
    ("2x2 grids, ragged blocks", "mk,kn->mn", dict(m=100, k=72, n=52), ({"m": 2, "k": 2}, {"k": 2, "n": 2}, {"m": 2, "n": 2}), 4, "f32", 1.1, 0.0),
    ("k distributed, C replicated", "mk,kn->mn", dict(m=48, k=128, n=40), ({"k": 4}, {"k": 4}, {}), 4, "f32", 1.0, 0.0),
    ("k distributed, C sharded, beta", "mk,kn->mn", dict(m=64, k=96, n=48), ({"k": 2}, {"k": 2}, {"n": 2}), 2, "f32", 0.7, 0.5),
    ("three ranks, multi-mode", "akcl,lbk->abc", dict(a=30, b=21, c=9, k=16, l=12), ({"c": 3}, {"b": 3}, {"a": 3}), 3, "f32", 1.0, 0.5),
    ("sub-box transfers", "mk,kn->mn", dict(m=64, k=64, n=64), ({"k": 4}, {"n": 4}, {"m": 2, "n": 2}), 4, "f32", 1.0, 0.0),
    ("fp64", "mk,kn->mn", dict(m=40, k=56, n=24), ({"m": 2}, {"n": 2}, {"m": 2}), 2, "f64", 1.0, 0.25),
    ("bf16", "mk,kn->mn", dict(m=128, k=256, n=128), ({"k": 2}, {"n": 2}, {"m": 2}), 2, "bf16", 1.0, 0.0),
    ("more ranks than blocks along a mode", "mk,kn->mn", dict(m=5, k=16, n=8), ({"m": 4}, {}, {"m": 4}), 4, "f32", 1.0, 0.0),
]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_mp_contraction_over_local_world(mp, case):
    _, eq, ext, dist, nranks, dtype, alpha, beta = case
    run_case(mp, eq, ext, dist, nranks, dtype, alpha, beta)


REDUCE_CASES = [
    # name, eq, extents, (distA, distB, distC), nranks, dtype, alpha, beta
    ("C replicated: all-reduce in the user's D", "mk,kn->mn", dict(m=48, k=128, n=40), ({"k": 4}, {"k": 4}, {}), 4, "f32", 1.0, 0.5),
    ("C cut along its last mode: slots are slabs", "mk,kn->mn", dict(m=64, k=96, n=48), ({"k": 2}, {"k": 2}, {"n": 2}), 2, "f32", 0.7, 0.5),
    ("C cut along its first mode: packed slots", "mk,kn->mn", dict(m=64, k=96, n=48), ({"k": 2}, {"k": 2}, {"m": 2}), 2, "f32", 1.0, 0.0),
    ("ragged 2x2 grid of C, two contracted modes cut", "mkl,lkn->mn", dict(m=50, k=12, l=10, n=27), ({"k": 2, "l": 2}, {"l": 2, "k": 2}, {"m": 2, "n": 2}), 4, "f32", 1.0, 0.25,
     [None, [0, 2, 1, 3], None]),   # B's grid is (l, k): this rank order gives every rank the same (k, l) block of A and B
    ("the headline einsum, shrunk, K cut over four ranks", "dcba,ebcd->ea", dict(a=24, b=16, c=8, d=16, e=24), ({"b": 4}, {"b": 4}, {}), 4, "f32", 1.0, 0.0),
    ("complex", "mk,kn->mn", dict(m=20, k=64, n=12), ({"k": 2}, {"k": 2}, {"n": 2}), 2, "c64", 1.0, 0.5),
    ("bf16", "mk,kn->mn", dict(m=128, k=512, n=64), ({"k": 2}, {"k": 2}, {}), 2, "bf16", 1.0, 0.0),
    ("fp64, three ranks, one with an empty block of C", "mk,kn->mn", dict(m=4, k=90, n=16), ({"k": 3}, {"k": 3}, {"m": 3}), 3, "f64", 1.0, 0.5),
]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("algo", ["reduce", "gather"])
@pytest.mark.parametrize("case", REDUCE_CASES, ids=[c[0] for c in REDUCE_CASES])
def test_mp_contracted_mode_distributed_both_algorithms(mp, monkeypatch, case, algo):
    """A and B cut along contracted modes only: the result is either reduced (all-reduce / reduce-scatter of partials,
    no operand moves) or the operands are gathered; both must give the reference answer on the same inputs."""
    _, eq, ext, dist, nranks, dtype, alpha, beta = case[:8]
    monkeypatch.setenv("CUTENSORMP_AMD_ALGO", algo)
    d = run_case(mp, eq, ext, dist, nranks, dtype, alpha, beta, ranks=case[8] if len(case) > 8 else None)
    assert all(x["algorithm"] == algo for x in d), d[0]
    if algo == "reduce":
        assert all(x["sends"] == [] and x["recvs"] == [] for x in d)        # no operand leaves its rank


@pytest.mark.timeout(300)
def test_mp_algorithm_choice_follows_the_byte_counts(mp):
    """Tiny C, long K: reduce.  Free mode cut, K whole: gather (nothing to reduce).  K cut but C as large as the
    operands: gather wins the byte count."""
    d = run_case(mp, "dcba,ebcd->ea", dict(a=24, b=16, c=8, d=16, e=24), ({"b": 2}, {"b": 2}, {}), 2)
    assert all(x["algorithm"] == "reduce" and x["reduceTotal"] < x["gatherTotal"] and x["reduceDirect"] for x in d)
    d = run_case(mp, "mk,kn->mn", dict(m=64, k=32, n=16), ({"m": 2}, {}, {"m": 2}), 2)
    assert all(x["algorithm"] == "gather" for x in d)
    d = run_case(mp, "mk,kn->mn", dict(m=256, k=4, n=256), ({"k": 2}, {"k": 2}, {}), 2)
    assert all(x["algorithm"] == "gather" and x["reduceTotal"] >= x["gatherTotal"] for x in d)


@pytest.mark.timeout(300)
def test_mp_sample_style_complex_many_modes(mp):
    """The shape family of the sample (:241: complex<float>, every extent 2, dozens of modes), two ranks, one
    distributed mode per tensor; rank order permuted for C."""
    eq = "abcdefEFGH,abcdefABCD->EFGHABCD"
    ext = {l: 2 for l in "abcdefEFGHABCD"}
    run_case(mp, eq, ext, ({"E": 2}, {"a": 2}, {"A": 2}), 2, "c64", ranks=[None, None, [1, 0]])


@pytest.mark.timeout(300)
def test_mp_plan_reports_transfers(mp):
    """No exchange when every operand box is local; whole-block sends go out without packing."""
    d = run_case(mp, "mk,kn->mn", dict(m=64, k=32, n=16), ({"m": 2}, {}, {"m": 2}), 2)
    assert all(x["sends"] == [] and x["recvs"] == [] and not x["stagedA"] and not x["stagedB"] for x in d)
    d = run_case(mp, "mk,kn->mn", dict(m=64, k=32, n=16), ({"k": 2}, {"k": 2}, {}), 2)
    for x in d:
        assert [s["direct"] for s in x["sends"]] == [True, True]          # whole packed blocks of A and B
        assert len(x["recvs"]) == 2 and x["stagedA"] and x["stagedB"]


@pytest.mark.timeout(300)
def test_mp_on_a_one_rank_rccl_communicator(mp):
    """cutensorMpCreate on a real RCCL communicator (world size 1): the sample's call sequence :470-538."""
    cmp, ct, torch = mp
    rccl = ctypes.CDLL("librccl.so.1")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        stream = torch.cuda.Stream()
        h = ctypes.c_void_p()
        cmp.check(cmp.cutensorMpCreate(ctypes.byref(h), comm, 0, ctypes.c_void_p(stream.cuda_stream)))
        m, k, n = 96, 64, 80
        rng = np.random.default_rng(3)
        A = np.asfortranarray(rng.uniform(-1, 1, (m, k)).astype(np.float32))
        B = np.asfortranarray(rng.uniform(-1, 1, (k, n)).astype(np.float32))
        descs = []
        for shape in ((m, k), (k, n), (m, n)):
            d = ctypes.c_void_p()
            cmp.check(cmp.cutensorMpCreateTensorDescriptor(h, ctypes.byref(d), 2, ct.i64(list(shape)), None, None, None,
                                                           ct.i64([1, 1]), 1, None, ct.R_32F))
            descs.append(d)
        lab = [ct.i32([ord(c) for c in s]) for s in ("mk", "kn", "mn")]
        op, pref, plan = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        cmp.check(cmp.cutensorMpCreateContraction(h, ctypes.byref(op), descs[0], lab[0], ct.OP_IDENTITY, descs[1], lab[1], ct.OP_IDENTITY,
                                                  descs[2], lab[2], ct.OP_IDENTITY, descs[2], lab[2], ct.compute_desc("32F")))
        cmp.check(cmp.cutensorMpCreatePlanPreference(h, ctypes.byref(pref), cmp.ALGO_DEFAULT, 1 << 30, 1024))
        cmp.check(cmp.cutensorMpCreatePlan(h, ctypes.byref(plan), op, pref))
        need = ctypes.c_uint64(0)
        cmp.check(cmp.cutensorMpPlanGetAttribute(h, plan, cmp.PLAN_REQUIRED_WORKSPACE_DEVICE, ctypes.byref(need), 8))
        ws = torch.empty(max(int(need.value), 256), dtype=torch.uint8, device="cuda")
        dA, dB = to_device(A), to_device(B)
        dC = torch.zeros(m * n, dtype=torch.float32, device="cuda")
        one, zero = ctypes.c_float(1.0), ctypes.c_float(0.0)
        torch.cuda.synchronize()
        cmp.check(cmp.cutensorMpContract(h, plan, ctypes.byref(one), dA.data_ptr(), dB.data_ptr(), ctypes.byref(zero),
                                         dC.data_ptr(), dC.data_ptr(), ws.data_ptr(), None))
        stream.synchronize()
        got = np.reshape(dC.cpu().numpy(), (m, n), order="F")
        assert np.max(np.abs(got - A.astype(np.float64) @ B.astype(np.float64))) < 1e-4
        cmp.check(cmp.cutensorMpDestroyPlan(plan))
        cmp.check(cmp.cutensorMpDestroyPlanPreference(pref))
        cmp.check(cmp.cutensorMpDestroyOperationDescriptor(op))
        for d in descs:
            cmp.check(cmp.cutensorMpDestroyTensorDescriptor(d))
        cmp.check(cmp.cutensorMpDestroy(h))
    finally:
        rccl.ncclCommDestroy(comm)

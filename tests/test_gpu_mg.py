"""GPU parity of the cuTENSORMg path on the devices that are visible (the GPU box has one): the call
sequence of cuTENSORMg/contraction_multi_gpu.cu:151-383 with its 2x2 block-cyclic descriptors, shrunk
extents, every grid cell on device 0 — both with a one-device handle and with a handle that lists
device 0 several times ("virtual" devices), which exercises sharding, gather views, per-piece local
contractions and the scatter back into the owners' cell buffers."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg(built):
    import torch
    assert torch.cuda.is_available()
    from cudalibrarysamples_amd import cutensormg
    return cutensormg, torch


def distribute(M, block, dc):
    """Global matrix (numpy, any layout) -> list of per-cell packed buffers.
    Cell (c0, c1) (first mode fastest) holds blocks (b0, b1) with b_i % dc_i == c_i, stored
    [w0, w1, lb0, lb1] first index fastest (contraction_multi_gpu.cu:256: packed block storage)."""
    e0, e1 = M.shape
    nb0, nb1 = e0 // block[0], e1 // block[1]
    cells = []
    for c1 in range(dc[1]):
        for c0 in range(dc[0]):
            pass
    out = {}
    for c1 in range(dc[1]):
        for c0 in range(dc[0]):
            lb0, lb1 = nb0 // dc[0], nb1 // dc[1]
            buf = np.zeros((block[0], block[1], lb0, lb1), dtype=M.dtype, order="F")
            for l1 in range(lb1):
                for l0 in range(lb0):
                    b0, b1 = l0 * dc[0] + c0, l1 * dc[1] + c1
                    buf[:, :, l0, l1] = M[b0 * block[0]:(b0 + 1) * block[0], b1 * block[1]:(b1 + 1) * block[1]]
            out[c0 + dc[0] * c1] = buf
    return [out[i] for i in range(dc[0] * dc[1])]


def collect(cells, shape, block, dc, dtype):
    M = np.zeros(shape, dtype=dtype)
    nb0, nb1 = shape[0] // block[0], shape[1] // block[1]
    for c1 in range(dc[1]):
        for c0 in range(dc[0]):
            buf = cells[c0 + dc[0] * c1]
            for l1 in range(nb1 // dc[1]):
                for l0 in range(nb0 // dc[0]):
                    b0, b1 = l0 * dc[0] + c0, l1 * dc[1] + c1
                    M[b0 * block[0]:(b0 + 1) * block[0], b1 * block[1]:(b1 + 1) * block[1]] = buf[:, :, l0, l1]
    return M


@pytest.mark.parametrize("handle_devices,extent,block,beta", [
    ([0], 256, 128, 0.0),              # the sample on a 1-GPU node (contraction_multi_gpu.cu:129-139)
    ([0, 0, 0, 0], 256, 128, 0.0),     # four logical devices -> i sharded four ways inside two blocks
    ([0, 0], 256, 64, 0.5),            # block-cyclic with two local blocks per cell, beta != 0
    ([0, 0, 0], 192, 32, 0.0),         # shard boundaries that are not block boundaries
])
def test_mg_contraction(mg, handle_devices, extent, block, beta):
    cm, torch = mg
    rng = np.random.default_rng(7)
    E, BS, DC = extent, block, 2
    A = rng.random((E, E), dtype=np.float32)      # A[i,k]
    B = rng.random((E, E), dtype=np.float32)      # B[k,j]
    C = rng.random((E, E), dtype=np.float32)      # C[i,j]
    h = ctypes.c_void_p()
    cm.check(cm.cutensorMgCreate(ctypes.byref(h), len(handle_devices), cm.i32(handle_devices)))
    cell_devices = [handle_devices[i % len(handle_devices)] for i in range(DC * DC)]   # fillUp(), :178-187

    def desc():
        d = ctypes.c_void_p()
        cm.check(cm.cutensorMgCreateTensorDescriptor(h, ctypes.byref(d), 2, cm.i64([E, E]), None, cm.i64([BS, BS]), None,
                                                     cm.i32([DC, DC]), DC * DC, cm.i32(cell_devices), 0))
        return d

    dA, dB, dC = desc(), desc(), desc()
    cd = ctypes.c_void_p()
    cm.check(cm.cutensorMgCreateContractionDescriptor(h, ctypes.byref(cd), dA, cm.i32("ik"), dB, cm.i32("kj"), dC, cm.i32("ij"),
                                                      dC, cm.i32("ij"), cm.COMPUTE_32F))
    find = ctypes.c_void_p()
    cm.check(cm.cutensorMgCreateContractionFind(h, ctypes.byref(find), cm.ALGO_DEFAULT))
    n = len(handle_devices)
    ws_sizes = (ctypes.c_int64 * n)()
    host_size = ctypes.c_int64(0)
    cm.check(cm.cutensorMgContractionGetWorkspace(h, cd, find, 2, ws_sizes, ctypes.byref(host_size)))
    plan = ctypes.c_void_p()
    cm.check(cm.cutensorMgCreateContractionPlan(h, ctypes.byref(plan), cd, find, ws_sizes, host_size.value))

    cellsA = [torch.from_numpy(np.ascontiguousarray(x.ravel(order="F"))).cuda() for x in distribute(A, (BS, BS), (DC, DC))]
    cellsB = [torch.from_numpy(np.ascontiguousarray(x.ravel(order="F"))).cuda() for x in distribute(B, (BS, BS), (DC, DC))]
    cellsC = [torch.from_numpy(np.ascontiguousarray(x.ravel(order="F"))).cuda() for x in distribute(C, (BS, BS), (DC, DC))]
    ws = [torch.empty(int(ws_sizes[i]), dtype=torch.uint8, device="cuda") for i in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    torch.cuda.synchronize()
    alpha, b = ctypes.c_float(1.0), ctypes.c_float(beta)
    pa = cm.ptr_array([t.data_ptr() for t in cellsA])
    pb = cm.ptr_array([t.data_ptr() for t in cellsB])
    pc = cm.ptr_array([t.data_ptr() for t in cellsC])
    pw = cm.ptr_array([t.data_ptr() for t in ws])
    ps = cm.ptr_array([s.cuda_stream for s in streams])
    cm.check(cm.cutensorMgContraction(h, plan, ctypes.byref(alpha), pa, pb, ctypes.byref(b), pc, pc, pw, None, ps))
    torch.cuda.synchronize()

    lb = E // (BS * DC)
    got_cells = [np.reshape(t.cpu().numpy(), (BS, BS, lb, lb), order="F") for t in cellsC]
    got = collect(got_cells, (E, E), (BS, BS), (DC, DC), np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64) + beta * C
    np.testing.assert_allclose(got, ref, rtol=1e-4)

    for f, o in ((cm.cutensorMgDestroyContractionPlan, plan), (cm.cutensorMgDestroyContractionFind, find),
                 (cm.cutensorMgDestroyContractionDescriptor, cd), (cm.cutensorMgDestroyTensorDescriptor, dA),
                 (cm.cutensorMgDestroyTensorDescriptor, dB), (cm.cutensorMgDestroyTensorDescriptor, dC),
                 (cm.cutensorMgDestroy, h)):
        cm.check(f(o))


def _cells_of(G, bs, dc):
    """Global matrix -> packed per-cell buffers [w0, w1, lb0, lb1] (first index fastest), cell index first mode fastest."""
    lb = [G.shape[i] // (bs[i] * dc[i]) for i in range(2)]
    out = []
    for c1 in range(dc[1]):
        for c0 in range(dc[0]):
            buf = np.zeros((bs[0], bs[1], lb[0], lb[1]), dtype=G.dtype, order="F")
            for l1 in range(lb[1]):
                for l0 in range(lb[0]):
                    b0, b1 = l0 * dc[0] + c0, l1 * dc[1] + c1
                    buf[:, :, l0, l1] = G[b0 * bs[0]:(b0 + 1) * bs[0], b1 * bs[1]:(b1 + 1) * bs[1]]
            out.append(buf)
    return out


def _gather_cells(cells, shape, bs, dc, dtype):
    lb = [shape[i] // (bs[i] * dc[i]) for i in range(2)]
    G = np.zeros(shape, dtype=dtype)
    for c1 in range(dc[1]):
        for c0 in range(dc[0]):
            buf = np.reshape(cells[c0 + dc[0] * c1], (bs[0], bs[1], lb[0], lb[1]), order="F")
            for l1 in range(lb[1]):
                for l0 in range(lb[0]):
                    b0, b1 = l0 * dc[0] + c0, l1 * dc[1] + c1
                    G[b0 * bs[0]:(b0 + 1) * bs[0], b1 * bs[1]:(b1 + 1) * bs[1]] = buf[:, :, l0, l1]
    return G


@pytest.mark.parametrize("n,E,beta,env", [
    (4, 512, 0.0, {}),                                   # bench.py's layout: i cut n ways, B column slabs gathered
    (2, 256, 0.75, {}),                                  # beta != 0: C read in place
    (4, 512, 0.5, {"CUTENSORMG_AMD_WAVES": "3"}),        # gather in three waves, one event each
    (4, 256, 0.5, {"CUTENSORMG_AMD_DIRECT": "0"}),       # everything through the staging images + scatter
    (3, 384, 0.0, {"CUTENSORMG_AMD_QSPLIT": "0"}),       # no cut along j: one piece per device behind the whole gather
    (2, 256, 0.5, {"CUTENSORMG_AMD_THREADS": "1"}),      # round 6: per-device worker threads (default from four devices on) forced on at two
    (3, 384, 0.0, {"CUTENSORMG_AMD_THREADS": "1", "CUTENSORMG_AMD_DIRECT": "0"}),   # ... with staging images + scatter: the cross-device join on the workers
    (4, 512, 0.5, {"CUTENSORMG_AMD_THREADS": "0"}),      # ... and off at four: the whole call on the calling thread
    (8, 512, 0.5, {}),                                   # eight handle devices: workers, batched cell copies of eight cells per device
])
def test_mg_free_mode_shard_layout(mg, monkeypatch, n, E, beta, env):
    """The cuTENSORMg case bench.py times (largest free mode sharded over the devices, B all-gathered) on n
    virtual devices = device 0 listed n times: in-place operands, staged remote-coordinate runs, two compute streams,
    differing device counts of j in B (n) and C (1)."""
    cm, torch = mg
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(11)
    A = rng.random((E, E), dtype=np.float32)
    B = rng.random((E, E), dtype=np.float32)
    C = rng.random((E, E), dtype=np.float32)
    modes = ["ik", "kj", "ij"]
    block = [dict(i=E // n), dict(j=E // n), dict(i=E // n, j=E // n)]
    dcount = [dict(i=n), dict(j=n), dict(i=n)]
    with cm.Contraction([0] * n, modes, dict(i=E, j=E, k=E), block, dcount) as con:
        d = con.describe()
        layouts = [((E // n, E), (n, 1)), ((E, E // n), (1, n)), ((E // n, E // n), (n, 1))]
        dev = [[torch.from_numpy(np.ascontiguousarray(x.ravel(order="F"))).cuda() for x in _cells_of(G, bs, dc)]
               for G, (bs, dc) in zip((A, B, C), layouts)]
        ws = [torch.empty(int(con.ws_sizes[i]), dtype=torch.uint8, device="cuda") for i in range(n)]
        streams = [torch.cuda.Stream() for _ in range(n)]
        torch.cuda.synchronize()
        for rep in range(2):   # twice: the second call reuses the staging images and the events
            for t, x in zip(dev[2], _cells_of(C, *layouts[2])):
                t.copy_(torch.from_numpy(np.ascontiguousarray(x.ravel(order="F"))))
            torch.cuda.synchronize()
            cm.check(con.run(1.0, [t.data_ptr() for t in dev[0]], [t.data_ptr() for t in dev[1]], beta,
                             [t.data_ptr() for t in dev[2]], [t.data_ptr() for t in dev[2]], [t.data_ptr() for t in ws],
                             [s.cuda_stream for s in streams]))
            torch.cuda.synchronize()
            got = _gather_cells([t.cpu().numpy() for t in dev[2]], (E, E), *layouts[2], np.float32)
            ref = A.astype(np.float64) @ B.astype(np.float64) + beta * C
            np.testing.assert_allclose(got, ref, rtol=1e-4, err_msg=str(d)[:400])


def test_bench_multi_device_path_self_test(built):
    """bench.py --gpus N on a box with fewer GPUs than N: `--mg-virtual` drives the whole N > 1 path of the benchmark — child
    process, cuTENSORMg over N (logical) devices, result check, speedup bookkeeping, JSON assembly — on GPU 0."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--mg-virtual", "--steps", "3", "--warmup", "1", "--no-cpu",
                        "--no-secondary", "--burn-in-ms", "0"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["requested_gpus"] == 4 and line["n_gpus"] == 1 and line["scaling"] == "strong", line
    assert "cuTENSORMg" in line["metric"] and "SELF-TEST" in line["config"]["workload"], line
    assert line["config"]["max_rel_err_sampled"] < 1e-4 and line["config"]["gather_bytes_per_call"] == 0      # logical devices: nothing crosses xGMI
    assert line["value"] > 0 and line["config"]["speedup_vs_1"] > 0
    # like-for-like fields on every line: the value of each workload is named, whatever `value` happens to be
    assert line["value_is"] == "mg_value" and line["mg_value"] == line["value"] and line["mg_devices"] == 1
    assert line["speedup_vs_1_same_workload"] == line["config"]["speedup_vs_1"] and line["einsum_value"] > 0 and line["rccl_ranks_seen"] == 1
    kinds = [s["workload"] for s in line["secondary"]]
    assert any("4096^3" in k or "2048^3" in k for k in kinds) and any("einsum.cu" in k for k in kinds), kinds
    # and with one visible GPU and no self-test switch the line says so instead of pretending
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-secondary",
                        "--burn-in-ms", "0"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    import torch
    if torch.cuda.device_count() == 1:
        assert line["n_gpus"] == 1 and line["requested_gpus"] == 8 and "einsum" in line["metric"] and "multi_device" in line["config"], line
        assert line["value_is"] == "einsum_value" and line["einsum_value"] == line["value"] and "mg_value" in line


@pytest.mark.parametrize("handle_devices", [[0], [0, 0, 0]])
def test_mg_ragged_free_modes(mg, handle_devices):
    """Extents that are not a multiple of blockSize x deviceCount in the FREE modes (blog_post.cu derives its block sizes
    with ceil(), :168-175; cells then hold whole blocks with padding at the end, :107-113): the result inside the extents is
    exact, the padding of D's cells is unspecified."""
    cm, torch = mg
    rng = np.random.default_rng(21)
    Ei, Ej, Ek = 176, 144, 128            # i: 6 blocks of 32 over 2 (5.5 -> padded 192); j: 5 blocks of 32 over 2 (4.5 -> padded 192... 6 blocks)
    bs, dc = 32, 2
    A = rng.random((Ei, Ek), dtype=np.float32)
    B = rng.random((Ek, Ej), dtype=np.float32)
    pad = lambda e: -(-(-(-e // bs)) // dc) * dc * bs          # noqa: E731  whole blocks per cell
    Pi, Pj = pad(Ei), pad(Ej)

    def padded(G, shape):
        out = np.full(shape, np.nan, dtype=np.float32)      # NaN padding: it must never leak into the valid region
        out[:G.shape[0], :G.shape[1]] = G
        return out

    modes = ["ik", "kj", "ij"]
    block = [dict(i=bs, k=bs), dict(k=bs, j=bs), dict(i=bs, j=bs)]
    dcount = [dict(i=dc, k=dc), dict(k=dc, j=dc), dict(i=dc, j=dc)]
    with cm.Contraction(handle_devices, modes, dict(i=Ei, j=Ej, k=Ek), block, dcount) as con:
        cellsA = [torch.from_numpy(np.ascontiguousarray(x.ravel(order="F"))).cuda() for x in _cells_of(padded(A, (Pi, Ek)), (bs, bs), (dc, dc))]
        cellsB = [torch.from_numpy(np.ascontiguousarray(x.ravel(order="F"))).cuda() for x in _cells_of(padded(B, (Ek, Pj)), (bs, bs), (dc, dc))]
        cellsC = [torch.zeros((Pi // dc) * (Pj // dc), device="cuda") for _ in range(dc * dc)]
        n = len(handle_devices)
        ws = [torch.empty(int(con.ws_sizes[i]), dtype=torch.uint8, device="cuda") for i in range(n)]
        streams = [torch.cuda.Stream() for _ in range(n)]
        torch.cuda.synchronize()
        cm.check(con.run(1.0, [t.data_ptr() for t in cellsA], [t.data_ptr() for t in cellsB], 0.0, [t.data_ptr() for t in cellsC],
                         [t.data_ptr() for t in cellsC], [t.data_ptr() for t in ws], [s.cuda_stream for s in streams]))
        torch.cuda.synchronize()
        got = _gather_cells([t.cpu().numpy() for t in cellsC], (Pi, Pj), (bs, bs), (dc, dc), np.float32)[:Ei, :Ej]
        np.testing.assert_allclose(got, A.astype(np.float64) @ B.astype(np.float64), rtol=1e-4)


@pytest.mark.parametrize("handle_devices,Ek,bs,dc", [
    ([0], 176, 32, 2),          # 5.5 blocks over 2 cells: a full-block box per non-zero digit of the bound + the partial block
    ([0, 0, 0], 176, 32, 2),
    ([0, 0], 72, 32, 1),        # one cell, 2.25 blocks: {b < 2} + {b == 2, w < 8}
    ([0, 0], 24, 32, 2),        # less than one block: only the partial-block box, the second cell holds nothing valid
])
def test_mg_ragged_contracted_mode(mg, handle_devices, Ek, bs, dc):
    """A CONTRACTED mode whose extent is not a multiple of blockSize x deviceCount: the padding at the end of A's and B's cells
    (NaN here) must not enter any sum.  The plan tiles the valid part of k's padded index space with boxes (mg.cpp kbox_list)
    and the boxes accumulate into D; beta != 0 checks that only the first box applies it."""
    cm, torch = mg
    rng = np.random.default_rng(33)
    Ei, Ej = 128, 96
    bi, dci = 32, 2
    A = rng.random((Ei, Ek), dtype=np.float32)
    B = rng.random((Ek, Ej), dtype=np.float32)
    C = rng.random((Ei, Ej), dtype=np.float32)
    padk = -(-(-(-Ek // bs)) // dc) * dc * bs
    padj = -(-(-(-Ej // bi)) // dci) * dci * bi

    def padded(G, shape):
        out = np.full(shape, np.nan, dtype=np.float32)
        out[:G.shape[0], :G.shape[1]] = G
        return out

    modes = ["ik", "kj", "ij"]
    block = [dict(i=bi, k=bs), dict(k=bs, j=bi), dict(i=bi, j=bi)]
    dcount = [dict(i=dci, k=dc), dict(k=dc, j=dci), dict(i=dci, j=dci)]
    beta = 0.5
    with cm.Contraction(handle_devices, modes, dict(i=Ei, j=Ej, k=Ek), block, dcount) as con:
        assert con.describe()["numBoxes"] >= (1 if Ek % (bs * dc) == 0 else 1)
        cellsA = [torch.from_numpy(np.ascontiguousarray(x.ravel(order="F"))).cuda() for x in _cells_of(padded(A, (Ei, padk)), (bi, bs), (dci, dc))]
        cellsB = [torch.from_numpy(np.ascontiguousarray(x.ravel(order="F"))).cuda() for x in _cells_of(padded(B, (padk, padj)), (bs, bi), (dc, dci))]
        Cp = np.zeros((Ei, padj), dtype=np.float32)
        Cp[:, :Ej] = C
        cellsC = [torch.from_numpy(np.ascontiguousarray(x.ravel(order="F"))).cuda() for x in _cells_of(Cp, (bi, bi), (dci, dci))]
        n = len(handle_devices)
        ws = [torch.empty(int(con.ws_sizes[i]), dtype=torch.uint8, device="cuda") for i in range(n)]
        streams = [torch.cuda.Stream() for _ in range(n)]
        torch.cuda.synchronize()
        cm.check(con.run(1.0, [t.data_ptr() for t in cellsA], [t.data_ptr() for t in cellsB], beta, [t.data_ptr() for t in cellsC],
                         [t.data_ptr() for t in cellsC], [t.data_ptr() for t in ws], [s.cuda_stream for s in streams]))
        torch.cuda.synchronize()
        got = _gather_cells([t.cpu().numpy() for t in cellsC], (Ei, padj), (bi, bi), (dci, dci), np.float32)[:Ei, :Ej]
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, A.astype(np.float64) @ B.astype(np.float64) + beta * C, rtol=1e-4)


@pytest.mark.parametrize("transport", ["allgather", "sendrecv", "peer", ""])
@pytest.mark.parametrize("beta", [0.0, 0.5])
def test_mg_forced_gather_runs_rccl_and_the_event_graph_on_one_device(mg, monkeypatch, transport, beta):
    """CUTENSORMG_AMD_FORCE_GATHER=1 on the ONE GPU of the box (round 5): cutensorMgCreate builds a one-rank RCCL communicator
    (ncclCommInitAll), cutensorMgContraction runs the one-rank ncclAllGather / the ncclSend + ncclRecv pair to itself (or the
    peer copy) on the communication stream into the staging image, records the wave event, the local contraction waits for it and
    reads the STAGED operands, the caller's stream joins the communication stream at the end — the transport and event-graph code
    of contraction_multi_gpu.cu:286-345's path, executed for real; the result must not change.  Three calls: the first two of the
    automatic mode are its timed trials (all-gather, then send/recv), the third uses the winner."""
    cm, torch = mg
    monkeypatch.setenv("CUTENSORMG_AMD_FORCE_GATHER", "1")
    if transport:
        monkeypatch.setenv("CUTENSORMG_AMD_TRANSPORT", transport)
    E = 512
    rng = np.random.default_rng(23)
    A = rng.random((E, E), dtype=np.float32)
    B = rng.random((E, E), dtype=np.float32)
    C = rng.random((E, E), dtype=np.float32)
    modes = ["ik", "kj", "ij"]
    with cm.Contraction([0], modes, dict(i=E, j=E, k=E), [dict(), dict(), dict()], [dict(), dict(), dict()]) as con:
        d = con.describe()
        assert d["forceGather"] == 1 and d["remoteBytes"] >= 2 * E * E * 4 and d["localCopyBytes"] == 0, d
        assert d["useRccl"] == (0 if transport == "peer" else 1), d
        assert all(not t["local"] and t["src"] == 0 and t["dst"] == 0 for t in d["transfers"] if t["tensor"] < 2)
        assert all(p["wait"] and not p["use"][0]["direct"] and not p["use"][1]["direct"] for p in d["pieces"]), d["pieces"]
        dev = [torch.from_numpy(np.ascontiguousarray(G.ravel(order="F"))).cuda() for G in (A, B, C)]
        ws = [torch.empty(int(con.ws_sizes[0]), dtype=torch.uint8, device="cuda")]
        ws[0].fill_(0xff)                                       # NaN patterns in the staging images: stale reads would show
        stream = torch.cuda.Stream()
        torch.cuda.synchronize()
        for rep in range(3):
            dev[2].copy_(torch.from_numpy(np.ascontiguousarray(C.ravel(order="F"))))
            if rep == 2:                                        # new operand values: the staging image must be refreshed by the gather
                A = A[::-1].copy()
                dev[0].copy_(torch.from_numpy(np.ascontiguousarray(A.ravel(order="F"))))
            torch.cuda.synchronize()
            cm.check(con.run(1.0, [dev[0].data_ptr()], [dev[1].data_ptr()], beta, [dev[2].data_ptr()], [dev[2].data_ptr()],
                             [ws[0].data_ptr()], [stream.cuda_stream]))
            stream.synchronize()                                # ONLY the caller's stream: the join must have put everything behind it
            got = np.reshape(dev[2].cpu().numpy(), (E, E), order="F")
            ref = A.astype(np.float64) @ B.astype(np.float64) + beta * C
            np.testing.assert_allclose(got, ref, rtol=1e-4, err_msg="call %d, %s" % (rep, str(con.describe())[:300]))
        d = con.describe()
        if transport == "":
            assert d["chosen"] in (0, 1, 2) and d["transport"].startswith("auto"), d

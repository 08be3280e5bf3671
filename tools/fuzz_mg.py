#!/usr/bin/env python3
"""Randomised parity sweep of cutensorMgContraction on the devices that are visible (one on the GPU box; the handle lists
device 0 several times as "virtual" devices, which exercises sharding, gather views, per-piece contractions and the
scatter): random contractions with one or two modes per group, block-cyclic descriptors with random block sizes, grid
shapes and local block counts (shared modes blocked identically, as the library requires), random handle sizes, beta.
Not part of the test suite."""
import argparse
import ctypes
import itertools
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def distribute(G, bs, dc):
    """Global tensor -> per-cell packed buffers [w_0.., lb_0..] (first index fastest), cell index first mode fastest."""
    n = G.ndim
    lb = [G.shape[i] // (bs[i] * dc[i]) for i in range(n)]
    cells = []
    for cell in range(int(np.prod(dc))):
        c, rest = [], cell
        for i in range(n):
            c.append(rest % dc[i])
            rest //= dc[i]
        buf = np.zeros(list(bs) + lb, dtype=G.dtype, order="F")
        for l in itertools.product(*[range(x) for x in lb]):
            sl = tuple(slice((l[i] * dc[i] + c[i]) * bs[i], (l[i] * dc[i] + c[i] + 1) * bs[i]) for i in range(n))
            buf[(slice(None),) * n + tuple(l)] = G[sl]
        cells.append(buf)
    return cells


def collect(cells, shape, bs, dc, dtype):
    n = len(shape)
    lb = [shape[i] // (bs[i] * dc[i]) for i in range(n)]
    G = np.zeros(shape, dtype=dtype)
    for cell, buf in enumerate(cells):
        c, rest = [], cell
        for i in range(n):
            c.append(rest % dc[i])
            rest //= dc[i]
        for l in itertools.product(*[range(x) for x in lb]):
            sl = tuple(slice((l[i] * dc[i] + c[i]) * bs[i], (l[i] * dc[i] + c[i] + 1) * bs[i]) for i in range(n))
            G[sl] = buf[(slice(None),) * n + tuple(l)]
    return G


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--ragged", action="store_true", help="cut the last block of a third of the modes short (free AND contracted modes)")
    args = ap.parse_args()
    import torch
    from cudalibrarysamples_amd import cutensormg as cm
    rnd = random.Random(args.seed)
    fails = 0
    for case in range(args.cases):
        nM, nN, nK = rnd.randint(1, 2), rnd.randint(1, 2), rnd.randint(1, 2)
        labels = list("abcdefgh")
        rnd.shuffle(labels)
        M, N, K = ([labels.pop() for _ in range(x)] for x in (nM, nN, nK))
        geo = {}
        for c in M + N + K:
            geo[c] = (rnd.choice([8, 16, 24, 32]), rnd.choice([1, 1, 2]), rnd.choice([1, 1, 2, 3]))   # block, grid, local blocks
        ext = {c: geo[c][0] * geo[c][1] * geo[c][2] for c in geo}          # padded index space: whole blocks per cell
        # ragged extents (a third of the modes, contracted ones included): the last block is cut short; the padding of A and B
        # holds NaN, which must never reach a valid output (blog_post.cu:168-175 ceil()-derived block sizes; mg.cpp kbox_list)
        valid = {c: (ext[c] - rnd.randint(1, geo[c][0] - 1)) if (args.ragged and rnd.random() < 0.34) else ext[c] for c in geo}
        mA, mB, mC = M + K, N + K, M + N
        for m in (mA, mB, mC):
            rnd.shuffle(m)
        ndev = rnd.choice([1, 2, 3, 4])
        handle_devices = [0] * ndev
        beta = rnd.choice([0.0, 0.0, 0.5])
        rng = np.random.default_rng(case)
        V = [rng.random([valid[c] for c in m], dtype=np.float32) for m in (mA, mB, mC)]
        G = []
        for k, (m, v) in enumerate(zip((mA, mB, mC), V)):
            full = np.full([ext[c] for c in m], np.nan if k < 2 else 0.0, dtype=np.float32)
            full[tuple(slice(0, valid[c]) for c in m)] = v
            G.append(full)
        what = "%s,%s->%s geo %s valid %s ndev %d beta %g" % ("".join(mA), "".join(mB), "".join(mC), geo, valid, ndev, beta)
        h = ctypes.c_void_p()
        cm.check(cm.cutensorMgCreate(ctypes.byref(h), ndev, cm.i32(handle_devices)))
        descs, cells = [], []
        try:
            for m, g in zip((mA, mB, mC), G):
                bs, dc = [geo[c][0] for c in m], [geo[c][1] for c in m]
                ncell = int(np.prod(dc))
                d = ctypes.c_void_p()
                cm.check(cm.cutensorMgCreateTensorDescriptor(h, ctypes.byref(d), len(m), cm.i64([valid[c] for c in m]), None, cm.i64(bs), None,
                                                             cm.i32(dc), ncell, cm.i32([handle_devices[i % ndev] for i in range(ncell)]), 0))
                descs.append(d)
                cells.append([torch.from_numpy(np.ascontiguousarray(x.ravel(order="F"))).cuda() for x in distribute(g, bs, dc)])
            lab = [cm.i32("".join(m)) for m in (mA, mB, mC)]
            cd, find, plan = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            cm.check(cm.cutensorMgCreateContractionDescriptor(h, ctypes.byref(cd), descs[0], lab[0], descs[1], lab[1], descs[2], lab[2],
                                                              descs[2], lab[2], cm.COMPUTE_32F))
            cm.check(cm.cutensorMgCreateContractionFind(h, ctypes.byref(find), cm.ALGO_DEFAULT))
            ws_sizes = (ctypes.c_int64 * ndev)()
            host_size = ctypes.c_int64(0)
            cm.check(cm.cutensorMgContractionGetWorkspace(h, cd, find, 2, ws_sizes, ctypes.byref(host_size)))
            cm.check(cm.cutensorMgCreateContractionPlan(h, ctypes.byref(plan), cd, find, ws_sizes, host_size.value))
            ws = [torch.empty(max(int(ws_sizes[i]), 16), dtype=torch.uint8, device="cuda") for i in range(ndev)]
            streams = [torch.cuda.Stream() for _ in range(ndev)]
            torch.cuda.synchronize()
            alpha, b = ctypes.c_float(1.0), ctypes.c_float(beta)
            ptrs = [cm.ptr_array([t.data_ptr() for t in cs]) for cs in cells]
            cm.check(cm.cutensorMgContraction(h, plan, ctypes.byref(alpha), ptrs[0], ptrs[1], ctypes.byref(b), ptrs[2], ptrs[2],
                                              cm.ptr_array([t.data_ptr() for t in ws]), None, cm.ptr_array([s.cuda_stream for s in streams])))
            torch.cuda.synchronize()
            bsC, dcC = [geo[c][0] for c in mC], [geo[c][1] for c in mC]
            lbC = [geo[c][2] for c in mC]
            got = collect([np.reshape(t.cpu().numpy(), bsC + lbC, order="F") for t in cells[2]], [ext[c] for c in mC], bsC, dcC, np.float32)
            got = got[tuple(slice(0, valid[c]) for c in mC)]
            ref = np.einsum("%s,%s->%s" % ("".join(mA), "".join(mB), "".join(mC)), V[0].astype(np.float64), V[1].astype(np.float64)) + beta * V[2]
            err = float(np.max(np.abs(got - ref)) / max(1.0, float(np.max(np.abs(ref))))) if np.isfinite(got).all() else float("inf")
            if not err < 1e-4:
                fails += 1
                print("case %d MISMATCH rel err %.3e: %s" % (case, err, what))
            cm.check(cm.cutensorMgDestroyContractionPlan(plan))
            cm.check(cm.cutensorMgDestroyContractionFind(find))
            cm.check(cm.cutensorMgDestroyContractionDescriptor(cd))
        except Exception as e:   # noqa: BLE001
            fails += 1
            print("case %d FAILED (%s): %s" % (case, str(e)[:200], what))
        for d in descs:
            cm.cutensorMgDestroyTensorDescriptor(d)
        cm.cutensorMgDestroy(h)
    print("cases %d, failures %d" % (args.cases, fails))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())

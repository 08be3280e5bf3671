"""GPU parity of the block-sparse contraction (cuTENSOR/blocksparse.cu: C_i = A_{kil} B_{kl}, fp64, sections
k {10,10,15}, i {20,20,25}, l {30,30,35}; A has its eight corner blocks, B five blocks, C is full, :44-170)
through cutensorBlockSparseContract against a dense numpy einsum of the same tensors with the absent blocks zero.
fp64 accumulate: rtol 1e-12."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SECTIONS = {"k": [10, 10, 15], "i": [20, 20, 25], "l": [30, 30, 35]}


def _make(torch, ct, h, modes, coords, seed, dtype=np.float64):
    """Returns (descriptor, list of device tensors, dense numpy tensor with zeros in absent blocks)."""
    rng = np.random.default_rng(seed)
    nm = len(modes)
    full = np.zeros([sum(SECTIONS[m]) for m in modes], dtype=dtype)
    starts = {m: np.concatenate([[0], np.cumsum(SECTIONS[m])]) for m in modes}
    blocks = []
    for c in coords:
        shape = [SECTIONS[m][ci] for m, ci in zip(modes, c)]
        blk = rng.random(shape).astype(dtype) - 0.5
        sl = tuple(slice(int(starts[m][ci]), int(starts[m][ci + 1])) for m, ci in zip(modes, c))
        full[sl] = blk
        blocks.append(torch.from_numpy(np.ascontiguousarray(blk.ravel(order="F"))).cuda())   # packed column-major block
    nsec = (ctypes.c_uint32 * nm)(*[len(SECTIONS[m]) for m in modes])
    ext = ct.i64([e for m in modes for e in SECTIONS[m]])
    flat = ct.i32([x for c in coords for x in c])
    d = ctypes.c_void_p()
    ct.check(ct.cutensorCreateBlockSparseTensorDescriptor(h.h, ctypes.byref(d), nm, len(coords), nsec, ext, flat, None,
                                                          ct.R_64F if dtype == np.float64 else ct.R_32F))
    return d, blocks, full


@pytest.mark.parametrize("beta", [0.0, 0.5])
def test_blocksparse_sample_structure(built, beta):
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    h = ops.Handle()
    coordsA = [(0, 0, 0), (2, 0, 0), (0, 2, 0), (2, 2, 0), (0, 0, 2), (2, 0, 2), (0, 2, 2), (2, 2, 2)]   # blocksparse.cu:122-132
    coordsB = [(0, 0), (1, 1), (2, 2), (0, 2), (2, 0)]
    coordsC = [(0,), (1,), (2,)]
    dA, blkA, fullA = _make(torch, ct, h, "kil", coordsA, 1)
    dB, blkB, fullB = _make(torch, ct, h, "kl", coordsB, 2)
    dC, blkC, fullC = _make(torch, ct, h, "i", coordsC, 3)
    op = ctypes.c_void_p()
    mA, mB, mC = ct.i32([ord(c) for c in "kil"]), ct.i32([ord(c) for c in "kl"]), ct.i32([ord(c) for c in "i"])
    ct.check(ct.cutensorCreateBlockSparseContraction(h.h, ctypes.byref(op), dA, mA, ct.OP_IDENTITY, dB, mB, ct.OP_IDENTITY,
                                                     dC, mC, ct.OP_IDENTITY, dC, mC, ct.compute_desc("64F")))
    est = ctypes.c_uint64(0)
    ct.check(ct.cutensorEstimateWorkspaceSize(h.h, op, None, ct.WORKSPACE_DEFAULT, ctypes.byref(est)))
    plan = ctypes.c_void_p()
    ct.check(ct.cutensorCreatePlan(h.h, ctypes.byref(plan), op, None, est.value))
    ws = torch.empty(max(est.value, 256), dtype=torch.uint8, device="cuda")
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    alpha, b = ctypes.c_double(1.25), ctypes.c_double(beta)
    pC = arr(blkC)
    ct.check(ct.cutensorBlockSparseContract(h.h, plan, ctypes.byref(alpha), arr(blkA), arr(blkB), ctypes.byref(b), pC, pC,
                                            ws.data_ptr(), est.value, None))
    torch.cuda.synchronize()
    got = np.concatenate([t.cpu().numpy() for t in blkC])
    ref = 1.25 * np.einsum("kil,kl->i", fullA, fullB) + beta * fullC
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    ct.cutensorDestroyPlan(plan)
    ct.cutensorDestroyOperationDescriptor(op)
    for d in (dA, dB, dC):
        ct.cutensorDestroyBlockSparseTensorDescriptor(d)


def test_blocksparse_output_blocks_without_contribution(built):
    """An output block that no (A, B) block pair reaches must become beta * C (cleared for beta = 0), and pairs that
    point at an absent output block are skipped."""
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    h = ops.Handle()
    dA, blkA, fullA = _make(torch, ct, h, "ik", [(0, 0), (1, 1)], 4)          # A_{i,k}: diagonal blocks only, no block in row 2
    dB, blkB, fullB = _make(torch, ct, h, "kl", [(0, 0), (1, 2)], 5)
    dC, blkC, fullC = _make(torch, ct, h, "il", [(0, 0), (2, 1), (1, 1)], 6)  # (1,2) is absent in C; (2,1) and (1,1) get nothing
    op = ctypes.c_void_p()
    m = lambda s: ct.i32([ord(c) for c in s])
    ct.check(ct.cutensorCreateBlockSparseContraction(h.h, ctypes.byref(op), dA, m("ik"), ct.OP_IDENTITY, dB, m("kl"), ct.OP_IDENTITY,
                                                     dC, m("il"), ct.OP_IDENTITY, dC, m("il"), ct.compute_desc("64F")))
    plan = ctypes.c_void_p()
    ct.check(ct.cutensorCreatePlan(h.h, ctypes.byref(plan), op, None, 0))
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    for beta in (0.0, -2.0):
        blks = [t.clone() for t in blkC]
        alpha, b = ctypes.c_double(1.0), ctypes.c_double(beta)
        ct.check(ct.cutensorBlockSparseContract(h.h, plan, ctypes.byref(alpha), arr(blkA), arr(blkB), ctypes.byref(b), arr(blkC), arr(blks),
                                                None, 0, None))
        torch.cuda.synchronize()
        dense = np.einsum("ik,kl->il", fullA, fullB) + beta * fullC
        si = np.concatenate([[0], np.cumsum(SECTIONS["i"])])
        sl = np.concatenate([[0], np.cumsum(SECTIONS["l"])])
        for t, (ci, cl) in zip(blks, [(0, 0), (2, 1), (1, 1)]):
            ref = dense[si[ci]:si[ci + 1], sl[cl]:sl[cl + 1]]
            np.testing.assert_allclose(t.cpu().numpy().reshape(ref.shape, order="F"), ref, rtol=1e-12, atol=1e-12)

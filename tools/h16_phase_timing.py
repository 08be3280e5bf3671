#!/usr/bin/env python3
"""Segment timing of the 16-bit GETT kernel: the TIMED instantiation (CUTENSOR_AMD_H16_TIMED=1) records s_memtime at the
seven segment boundaries of the four phases of K-tile 8, for wave 0 (first wave row) and wave 4 (second wave row) of
workgroup 0.  Prints cycles per segment: reads+DMA issue | vmcnt wait | barrier 1 | lgkm wait | MFMA | barrier 2."""
import json
import os
import sys

os.environ["CUTENSOR_AMD_H16_TIMED"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    n = 8192
    A = (torch.rand((n, n), device="cuda") * 2 - 1).to(torch.bfloat16)
    B = (torch.rand((n, n), device="cuda") * 2 - 1).to(torch.bfloat16)
    D = torch.empty((n, n), device="cuda", dtype=torch.bfloat16)
    h = ops.Handle()
    plan = ops.contraction_plan(h, [n, n], "mk", [n, n], "kn", [n, n], "mn", dtype=ct.R_16BF)
    tbuf = torch.zeros(64, dtype=torch.int64, device="cuda")
    for _ in range(3):
        plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
    torch.cuda.synchronize()
    ct.lib.ctamdSetTimingBuffer(h.h, tbuf.data_ptr())
    plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
    torch.cuda.synchronize()
    ct.lib.ctamdSetTimingBuffer(h.h, None)
    t = tbuf.cpu().numpy().reshape(2, 32)[:, :28].reshape(2, 4, 7).astype(np.int64)
    names = ["reads+dma", "vmcnt", "barrier1", "lgkm", "mfma", "barrier2"]
    out = {}
    for w, wn in enumerate(["wave0", "wave4"]):
        seg = np.diff(t[w], axis=1)
        out[wn] = {"phase%d" % q: dict(zip(names, [int(x) for x in seg[q]])) for q in range(4)}
        out[wn]["tile_total"] = int(t[w, 3, 6] - t[w, 0, 0])
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""cutensorPermute on permutations whose leading modes are the same packed set on both sides (EW_BLOCK, elementwise.hip ew_block_kernel):
us per call and TB/s by the sample's 2 |D| bytes (elementwise_permute.cu:208).  One JSON line per case."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudalibrarysamples_amd import cutensor as ct, ops

h = ops.Handle()
CASES = [("bfloat16", (50, 16, 4, 2048)), ("float32", (50, 16, 4, 2048)), ("bfloat16", (40, 16, 8, 8192)), ("float32", (40, 16, 8, 8192)),
         ("bfloat16", (7, 3, 5, 200000)), ("float16", (128, 8, 8, 4096))]
for dn, (d, c, b, a) in CASES:
    tdt = getattr(torch, dn)
    cdt = {"bfloat16": ct.R_16BF, "float16": ct.R_16F, "float32": ct.R_32F}[dn]
    A = (torch.rand((a, b, c, d), device="cuda") * 2 - 1).to(tdt)
    D = torch.empty((a, d, c, b), device="cuda", dtype=tdt)
    p = ops.permutation_plan(h, [d, c, b, a], "dcba", [b, c, d, a], "bcda", dtype=cdt)
    desc = p.describe()
    for _ in range(5):
        p.permute(1.0, A.data_ptr(), D.data_ptr(), 0)
    torch.cuda.synchronize()
    assert torch.equal(D, A.permute(0, 3, 2, 1).contiguous())
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            p.permute(1.0, A.data_ptr(), D.data_ptr(), 0)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    nbytes = 2.0 * A.numel() * A.element_size()
    print(json.dumps({"dtype": dn, "A[d,c,b,a]->D[b,c,d,a]": [d, c, b, a], "variant": desc["variant"], "MB_moved": round(nbytes / 1e6, 1), "us": round(best * 1e3, 1),
                      "TBps": round(nbytes / (best * 1e-3) / 1e12, 2)}), flush=True)

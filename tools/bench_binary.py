#!/usr/bin/env python3
"""cutensorElementwiseBinaryExecute D[a,b,c] = alpha A[c,b,a] + gamma C[a,b,c] (elementwise_binary.cu:51-66) at the sample's extents and
neighbours: us per call, TB/s by 3 |D| bytes (two reads, one write).  CUTENSOR_AMD_EW_ANY=0 / 1 (hooks flavour) pins the kernel family."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
for dn, ext in (("float32", dict(a=400, b=200, c=300)), ("float32", dict(a=512, b=256, c=256)), ("float32", dict(a=401, b=203, c=299)), ("float32", dict(a=4097, b=4099, c=1)),
                ("bfloat16", dict(a=400, b=200, c=300)), ("bfloat16", dict(a=401, b=203, c=299)), ("float32", dict(a=1024, b=1024, c=512))):
    tdt = getattr(torch, dn); cdt = {"bfloat16": ct.R_16BF, "float32": ct.R_32F}[dn]
    eA, eC = [ext[c] for c in "cba"], [ext[c] for c in "abc"]
    A = (torch.rand(eA[::-1], device="cuda") * 2 - 1).to(tdt)
    C = (torch.rand(eC[::-1], device="cuda") * 2 - 1).to(tdt)
    D = torch.empty_like(C)
    p = ops.binary_plan(h, eA, "cba", eC, "abc", dtype=cdt)
    fn = lambda: p.binary(1.1, A.data_ptr(), 1.3, C.data_ptr(), D.data_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ref = (1.1 * torch.einsum("abc->cba", A.float()) + 1.3 * C.float()).to(tdt)
    err = float((D.float() - ref.float()).abs().max())
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    nb = 3.0 * C.numel() * C.element_size()
    print(json.dumps({"dtype": dn, "ext": ext, "variant": p.describe().get("variant"), "us": round(best * 1e3, 1), "TBps": round(nb / (best * 1e-3) / 1e12, 2), "max_abs_err": err}), flush=True)

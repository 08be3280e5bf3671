#!/usr/bin/env python3
"""cutensorReduce at extents without 16-byte lanes beside the aligned neighbour: TB/s by the sample's |A| + |C| bytes (reduction.cu:229-231)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
CASES = [("float32", dict(a=401, b=403, c=399), "abc", "ac"), ("float32", dict(a=400, b=400, c=400), "abc", "ac"), ("float32", dict(a=401, b=403, c=399), "abc", "c"),
         ("float32", dict(a=401, b=403, c=399), "abc", "a"), ("float32", dict(a=401, b=403, c=399), "abc", "bc"), ("float32", dict(a=4097, b=4099), "ab", "b"), ("float32", dict(a=4097, b=4099), "ab", "a"),
         ("bfloat16", dict(a=401, b=403, c=399), "abc", "ac"), ("bfloat16", dict(a=401, b=403, c=399), "abc", "bc"), ("float64", dict(a=401, b=403, c=399), "abc", "ac")]
for dn, ext, mA, mC in CASES:
    tdt = getattr(torch, dn); cdt = {"bfloat16": ct.R_16BF, "float32": ct.R_32F, "float64": ct.R_64F}[dn]
    eA, eC = [ext[c] for c in mA], [ext[c] for c in mC]
    A = (torch.rand(eA[::-1], device="cuda") * 2 - 1).to(tdt)
    C = torch.zeros(eC[::-1], device="cuda", dtype=tdt)
    p = ops.reduction_plan(h, eA, mA, eC, mC, dtype=cdt, workspace_limit=1 << 30)
    ws = torch.empty(max(p.required_workspace, 256), dtype=torch.uint8, device="cuda")
    fn = lambda: p.reduce(1.0, A.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), p.required_workspace, 0)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ref = torch.einsum("%s->%s" % (mA[::-1], mC[::-1]), A.double())
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nb = (A.numel() + C.numel()) * A.element_size()
    print(json.dumps({"dtype": dn, "ext": ext, "reduce": mA + "->" + mC, "variant": p.describe().get("variant"), "us": round(ms * 1e3, 1), "TBps": round(nb / (ms * 1e-3) / 1e12, 2), "rel_err": err}), flush=True)

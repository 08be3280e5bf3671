import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
CASES = [("float32", dict(a=4100, b=4100), "ab", "ba"), ("float32", dict(a=4097, b=4099), "ab", "ba"), ("bfloat16", dict(a=4100, b=4100), "ab", "ba"), ("bfloat16", dict(a=4097, b=4099), "ab", "ba"),
         ("float32", dict(a=401, b=403, c=399), "abc", "cba"), ("bfloat16", dict(a=401, b=403, c=399), "abc", "cab"), ("float32", dict(a=4096, b=4096), "ab", "ba"),
         ("float32", dict(a=4104, b=4104), "ab", "ba"), ("float32", dict(a=4160, b=4160), "ab", "ba"), ("bfloat16", dict(a=4104, b=4104), "ab", "ba"), ("bfloat16", dict(a=4096, b=4096), "ab", "ba"),
         ("float32", dict(a=1024, b=1024, c=1024), "abc", "cba"), ("bfloat16", dict(a=1024, b=1024, c=1024), "abc", "cab"), ("float32", dict(a=1000, b=1000, c=1000), "abc", "cab")]
if os.environ.get("PERMUTE_SET") == "mid":     # mid-size 3-D reversals: where does the element-wise transposer stop winning?
    CASES = [(dn, dict(a=a, b=b, c=c), "cba", "abc") for dn in ("float32", "bfloat16")
             for (a, b, c) in ((400, 200, 300), (512, 256, 256), (384, 192, 320), (800, 400, 300), (640, 640, 640), (1000, 500, 600))]
for dn, ext, mA, mB in CASES:
    tdt = getattr(torch, dn); cdt = {"bfloat16": ct.R_16BF, "float32": ct.R_32F}[dn]
    eA, eB = [ext[c] for c in mA], [ext[c] for c in mB]
    A = (torch.rand(eA[::-1], device="cuda") * 2 - 1).to(tdt)
    D = torch.empty(eB[::-1], device="cuda", dtype=tdt)
    p = ops.permutation_plan(h, eA, mA, eB, mB, dtype=cdt)
    for _ in range(3): p.permute(1.0, A.data_ptr(), D.data_ptr(), 0)
    torch.cuda.synchronize()
    ref = torch.einsum("%s->%s" % (mA[::-1], mB[::-1]), A)
    ok = torch.equal(D, ref)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): p.permute(1.0, A.data_ptr(), D.data_ptr(), 0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nb = 2.0 * A.numel() * A.element_size()
    print(json.dumps({"dtype": dn, "ext": ext, "perm": mA + "->" + mB, "variant": p.describe()["variant"], "us": round(ms * 1e3, 1), "TBps": round(nb / (ms * 1e-3) / 1e12, 2), "exact": ok}))

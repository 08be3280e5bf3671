import json, sys, os
sys.path.insert(0, os.getcwd())
import torch
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
for dn, cdt in (("float32", ct.R_32F), ("bfloat16", ct.R_16BF)):
    tdt = getattr(torch, dn)
    eA, eB = [32, 128, 128, 128], [128, 32, 128, 128]
    A = torch.rand(eA[::-1], device="cuda").to(tdt); D = torch.empty(eB[::-1], device="cuda", dtype=tdt)
    p = ops.permutation_plan(h, eA, "whcn", eB, "cwhn", dtype=cdt)
    for _ in range(5): p.permute(1.0, A.data_ptr(), D.data_ptr(), 0)
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): p.permute(1.0, A.data_ptr(), D.data_ptr(), 0)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    nb = 2.0 * A.numel() * A.element_size()
    print(json.dumps({"dtype": dn, "sample": "elementwise_permute.cu whcn->cwhn (32,128,128,128)", "variant": p.describe()["variant"], "us": round(best*1e3,1), "TBps": round(nb/(best*1e-3)/1e12, 2)}))

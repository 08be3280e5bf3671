#!/usr/bin/env python3
"""fp32 C[m,n] = A[m,k] B[k,n] (packed column-major, the planner's choice) at the shapes given: one JSON line per shape with the time per
call and the kernel taken.  The command the rocprofv3 passes of round 6 wrap (short contracted ranges: instruction-cache and wait counters).
usage: tools/f32_shape.py "M,N,K[;M,N,K...]" [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudalibrarysamples_amd import ops

shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[1].split(";")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
h = ops.Handle()
for (M, N, K) in shapes:
    A = torch.rand(M * K, device="cuda") * 2 - 1
    B = torch.rand(K * N, device="cuda") * 2 - 1
    D = torch.zeros(M * N, device="cuda")
    p = ops.contraction_plan(h, [M, K], "mk", [K, N], "kn", [M, N], "mn", workspace_limit=1 << 30)
    d = p.describe()
    ws = torch.empty(max(p.required_workspace, 256), dtype=torch.uint8, device="cuda")
    fn = lambda: p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace, 0)   # noqa: E731
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"M": M, "N": N, "K": K, "us": round(ms * 1e3, 1), "tflops": round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1), "kname": d["kname"],
                      "tile": [d["bm"], d["bn"], d["bk"]], "pf": d["pf"], "splitK": d["splitK"], "blocks": d["blocks"]}), flush=True)
    p.destroy()
    del A, B, D, ws

#!/usr/bin/env python3
"""Headline equation with b = 96 / 128 (operands 302 / 403 MB, beyond the 256-MiB Infinity Cache): time per call of the planner's
choice (nontemporal ring kernel) against the default-policy twin forced by CUTENSOR_AMD_FORCE, back-to-back calls on the same operands."""
import os, sys, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from cudalibrarysamples_amd import ops
    b = int(sys.argv[1])
    ext = dict(a=96, b=b, c=64, d=64, e=96)
    h = ops.Handle()
    p = ops.contraction_plan(h, [ext[c] for c in "dcba"], "dcba", [ext[c] for c in "ebcd"], "ebcd", [96, 96], "ea", workspace_limit=1 << 30)
    A = torch.rand(96 * b * 64 * 64, device="cuda"); B = torch.rand(96 * b * 64 * 64, device="cuda"); C = torch.empty(96 * 96, device="cuda")
    ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
    run = lambda: p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), p.required_workspace)
    for _ in range(300): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 500 * 1e3
    flop = 2.0 * 96 * 96 * b * 64 * 64
    print(json.dumps({"b": b, "kernel": p.describe()["kernel"], "us_per_call": round(us, 2), "tflops": round(flop / us * 1e-6, 1), "operand_MB": round(2 * 96 * b * 4096 * 4 / 1e6)}))
else:
    for b in (96, 128):
        for force in (None, "56:256"):
            env = dict(os.environ)
            if force: env["CUTENSOR_AMD_FORCE"] = force
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(b)], capture_output=True, text=True, env=env)
            print(("forced default policy: " if force else "planner (nt):          ") + r.stdout.strip().splitlines()[-1])

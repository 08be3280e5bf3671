#!/usr/bin/env python3
"""What bounds the headline einsum when its operands come from HBM (round-4 review item 6: "a 64- or 128-deep K-tile for the HBM-cold
case ... or a measured negative with the counter that kills it").  Needs the RESEARCH build (make RESEARCH=1): the ablation
instantiations of gett_f32_stream_kernel<96x96x32, ring 4, A K-contiguous, B free-contiguous> are planner candidates only there.
For the default kernel, its nontemporal twin, ABL = 3 (full kernel + wait accounting), ABL = 2 (no MFMA: the memory path alone) and
ABL = 1 (no refills: LDS + MFMA alone), each WARM (one operand pair, re-contracted out of the Infinity Cache) and COLD (four rotating
pairs, 805 MB): us per GETT launch (fold off, HIP events over >= 300 launches) and, for ABL = 3, the mean cycles per workgroup
  multiplier_barrier  a multiplying wave waits at the tile barrier (for data: the loaders arrive only when their pieces landed)
  loader_vmcnt        a data-moving wave waits for its own LDS-DMA pieces (memory latency / bandwidth)
  loader_barrier      a data-moving wave waits at the barrier (for the multipliers: the ring is full)
A deeper K-tile halves the number of barriers; it can only give back cycles that are spent BECAUSE of the barrier (skew between the
waves that meet there), not cycles spent waiting for bytes."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ["CUTENSOR_AMD_ABLATION"] = "1"
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    if not ct.lib.ctamdResearchKernelsBuilt():
        print(json.dumps({"error": "production build: the ablation candidates need make RESEARCH=1"}))
        return 1
    ext = dict(a=96, b=64, c=64, d=64, e=96)
    mA, mB, mC = "dcba", "ebcd", "ea"
    eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
    h = ops.Handle()
    ct.lib.ctamdSetSplitKFold(h.h, 0)              # the GETT kernel alone
    pairs = [(torch.rand(int(np.prod(eA)), device="cuda"), torch.rand(int(np.prod(eB)), device="cuda")) for _ in range(4)]
    C = torch.zeros(int(np.prod(eC)), device="cuda")
    ws = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    p0 = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, workspace_limit=1 << 28)
    n = ct.lib.ctamdCountCandidates(h.h, p0.op, 1 << 28)
    d0 = p0.describe()
    want = {}                                       # name -> plan
    want["default"] = p0
    want["nt_twin"] = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, workspace_limit=1 << 28, operands_streamed=True)
    for r in range(n):
        p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, workspace_limit=1 << 28, algo=r)
        d = p.describe()
        key = {1: "abl1_no_refills", 2: "abl2_no_mfma", 3: "abl3_wait_accounting"}.get(d.get("abl", 0))
        if key and key not in want and d["splitK"] == d0["splitK"] and d["kname"] == "gett_f32_stream_kernel":
            want[key] = p
        else:
            p.destroy()
    out = {"candidates": n, "korder": os.environ.get("CUTENSOR_AMD_KORDER"), "Kdigits": d0.get("Kdigits"),
           "default_plan": {k: d0[k] for k in ("kernel", "bm", "bn", "bk", "pf", "splitK", "blocks", "kname", "nt")}}
    # the default plan's result under this K order (fold ON for the check), against fp64
    ct.lib.ctamdSetSplitKFold(h.h, 1)
    p0.contract(1.0, pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), 1 << 28, 0)
    torch.cuda.synchronize()
    A4 = pairs[0][0].view(*[ext[c] for c in "abcd"]).double()      # modes "dcba", d fastest = row-major [a][b][c][d]
    B4 = pairs[0][1].view(*[ext[c] for c in "dcbe"]).double()      # modes "ebcd", e fastest = row-major [d][c][b][e]
    ref = torch.einsum("abcd,dcbe->ae", A4, B4)
    got = C.view(ext["a"], ext["e"]).double()                       # modes "ea", e fastest = row-major [a][e]
    out["max_rel_err"] = float(((got - ref).abs() / ref.abs()).max())
    ct.lib.ctamdSetSplitKFold(h.h, 0)

    def run(p, cold, steps=400, warm=200):
        call = lambda i: p.contract(1.0, pairs[i % 4 if cold else 0][0].data_ptr(), pairs[i % 4 if cold else 0][1].data_ptr(), 0.0,  # noqa: E731
                                    C.data_ptr(), C.data_ptr(), ws.data_ptr(), 1 << 28, 0)
        for i in range(warm):
            call(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps * 1e3

    for name, p in want.items():
        d = p.describe()
        rec = {"kernel": d["kernel"], "abl": d.get("abl"), "ring": d.get("pf"), "nt": d.get("nt")}
        for cold in (False, True):
            rec["cold_us" if cold else "warm_us"] = round(run(p, cold), 2)
            if name == "abl3_wait_accounting":
                tbuf = torch.zeros(d["blocks"] * 16, dtype=torch.int64, device="cuda")
                ct.lib.ctamdSetTimingBuffer(h.h, tbuf.data_ptr())
                acc = []
                for i in range(8):                  # eight launches in the same rotation, each read back
                    pr = pairs[(i % 4) if cold else 0]
                    p.contract(1.0, pr[0].data_ptr(), pr[1].data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), 1 << 28, 0)
                    torch.cuda.synchronize()
                    acc.append(tbuf.cpu().numpy().reshape(-1, 16).astype(np.float64))
                ct.lib.ctamdSetTimingBuffer(h.h, None)
                t = np.mean(acc[2:], axis=0)
                total = float((t[:, 4] - t[:, 0]).mean())
                rec["cold" if cold else "warm"] = {
                    "cycles_per_workgroup": round(total), "k_tiles": int(d["kPerSlice"] // d["bk"]),
                    "multiplier_barrier": round(float(t[:, 8].mean())), "loader_vmcnt": round(float(t[:, 9].mean())),
                    "loader_barrier": round(float(t[:, 10].mean())),
                    "multiplier_barrier_share": round(float(t[:, 8].mean()) / total, 3)}
        out[name] = rec
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())

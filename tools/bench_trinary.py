#!/usr/bin/env python3
"""The element-wise trinary sample's shape (elementwise_trinary.cu:51-66: D[a,b,c] = 1.1 A[c,b,a] + 1.3 B[c,a,b] + 1.2 C[a,b,c], extents
a = 400, b = 200, c = 300, fp32) beside what the SAME extents give on its parts — the plain permutations A -> D and B -> D, a flat
copy — and beside a power-of-two neighbour (512, 256, 256): which part of the gap to the fp32 transposes' 6.4 TB/s belongs to ragged
extents and which to the trinary form.  GB/s by the sample's formula (4 |D| bytes: three reads, one write; permutations 2 |D|)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(torch, fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def main():
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    h = ops.Handle()
    for ext in (dict(a=400, b=200, c=300), dict(a=512, b=256, c=256), dict(a=384, b=192, c=320)):
        n = ext["a"] * ext["b"] * ext["c"]
        A, B, C = (torch.rand(n, device="cuda") for _ in range(3))
        D = torch.empty(n, device="cuda")
        e = lambda m: [ext[x] for x in m]   # noqa: E731
        out = {"extents": ext, "MB_per_tensor": n * 4 / 1e6}
        p = ops.trinary_plan(h, e("cba"), "cba", e("cab"), "cab", e("abc"), "abc", e("abc"), "abc")
        ms = timed(torch, lambda: p.trinary(1.1, A.data_ptr(), 1.3, B.data_ptr(), 1.2, C.data_ptr(), D.data_ptr()))
        out["trinary_GBps"] = round(4.0 * n * 4 / (ms * 1e-3) / 1e9)
        out["trinary_us"] = round(ms * 1e3, 1)
        out["trinary_plan"] = p.describe()
        p.destroy()
        for name, mA in (("permute_cba_to_abc", "cba"), ("permute_cab_to_abc", "cab"), ("copy_abc", "abc")):
            p = ops.permutation_plan(h, e(mA), mA, e("abc"), "abc")
            ms = timed(torch, lambda: p.permute(1.0, A.data_ptr(), D.data_ptr()))
            out[name + "_GBps"] = round(2.0 * n * 4 / (ms * 1e-3) / 1e9)
            out[name + "_us"] = round(ms * 1e3, 1)
            p.destroy()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

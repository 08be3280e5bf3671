#!/usr/bin/env python3
"""Where the time of ONE cutensorMgContraction call goes under the sample's protocol (contraction_multi_gpu.cu:323-345: wall clock around
the call + a synchronize of every device, best of a few) on one device — round-4 review item 4c: 4096^3 fp32 ran 130 TFLOP/s under that
protocol against 142 back to back.  One JSON line:
  floor_us            wall clock of launch + synchronize of a ONE-workgroup kernel from an idle device (what the protocol costs whatever
                      the library does: submission latency + the wake-up of the waiting host thread)
  mg_host_us          wall clock of the cutensorMgContraction call alone (returns as soon as the work is queued)
  mg_wall_us          the protocol: call + synchronize, min / median
  mg_event_us         hipEvent pair around the same single call (device idle before): the device-side time without the host wake-up
  mg_back_to_back_us  per call in a run of 50 calls, one synchronize at the end
  plain_*             the same four numbers for cutensorContract on the same problem (one device, no cuTENSORMg)."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def protocol(call, sync, reps=20):
    walls, hosts = [], []
    for _ in range(reps):
        sync()
        time.sleep(0.002)                          # the device is idle when the call comes, as in the sample
        t0 = time.perf_counter()
        call()
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
        walls.append((t2 - t0) * 1e6)
        hosts.append((t1 - t0) * 1e6)
    return walls, hosts


def main():
    import torch
    import bench  # noqa: F401  (path setup)
    from cudalibrarysamples_amd import cutensor as ct, cutensormg as cm, ops
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    out = {"extent": E}
    sync = lambda: torch.cuda.synchronize(0)  # noqa: E731
    # floor: a one-element fill
    x = torch.zeros(64, device="cuda")
    w, hst = protocol(lambda: x.add_(1.0), sync)
    out["floor_us"] = {"min": round(min(w), 1), "median": round(statistics.median(w), 1), "host_min": round(min(hst), 1)}

    # cuTENSORMg, one device
    con = cm.Contraction([0], ["ik", "kj", "ij"], dict(i=E, j=E, k=E), [dict(i=E), dict(j=E), dict(i=E, j=E)], [dict(i=1), dict(j=1), dict(i=1)])
    cells = [[torch.rand(E * E, device="cuda")] for _ in range(3)]
    ws = [torch.empty(int(con.ws_sizes[0]), dtype=torch.uint8, device="cuda")]
    st = torch.cuda.Stream(device=0)
    ptr = [[t.data_ptr() for t in row] for row in cells]
    wsp, sp = [ws[0].data_ptr()], [st.cuda_stream]
    call = lambda: cm.check(con.run(1.0, ptr[0], ptr[1], 0.0, ptr[2], ptr[2], wsp, sp))  # noqa: E731
    for _ in range(5):
        call()
    w, hst = protocol(call, sync)
    out["mg_wall_us"] = {"min": round(min(w), 1), "median": round(statistics.median(w), 1)}
    out["mg_host_us"] = {"min": round(min(hst), 1), "median": round(statistics.median(hst), 1)}
    ev = []
    for _ in range(10):
        sync()
        time.sleep(0.002)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record(st)
            call()
            e1.record(st)
        sync()
        ev.append(e0.elapsed_time(e1) * 1e3)
    out["mg_event_us"] = {"min": round(min(ev), 1), "median": round(statistics.median(ev), 1)}
    sync()
    t0 = time.perf_counter()
    for _ in range(50):
        call()
    sync()
    out["mg_back_to_back_us"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
    out["mg_pieces"] = len(con.describe()["pieces"])
    con.close()

    # cutensorContract on the same problem
    h = ops.Handle()
    p = ops.contraction_plan(h, [E, E], "ik", [E, E], "kj", [E, E], "ij", dtype=ct.R_32F, workspace_limit=1 << 28)
    wsc = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
    A, B, C = cells[0][0], cells[1][0], cells[2][0]
    pcall = lambda: p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), wsc.data_ptr(), p.required_workspace, stream=st.cuda_stream)  # noqa: E731
    for _ in range(5):
        pcall()
    w, hst = protocol(pcall, sync)
    out["plain_wall_us"] = {"min": round(min(w), 1), "median": round(statistics.median(w), 1)}
    out["plain_host_us"] = {"min": round(min(hst), 1), "median": round(statistics.median(hst), 1)}
    sync()
    t0 = time.perf_counter()
    for _ in range(50):
        pcall()
    sync()
    out["plain_back_to_back_us"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
    out["plain_kernel"] = p.describe()["kname"]
    flop = 2.0 * E ** 3
    out["tflops"] = {"mg_protocol_min": round(flop / out["mg_wall_us"]["min"] / 1e6, 1), "mg_back_to_back": round(flop / out["mg_back_to_back_us"] / 1e6, 1),
                     "plain_protocol_min": round(flop / out["plain_wall_us"]["min"] / 1e6, 1), "plain_back_to_back": round(flop / out["plain_back_to_back_us"] / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

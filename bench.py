#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native tensor-contraction engine.

Workload (BASELINE.json configs[1]): einsum 'abcd,dcbe->ae', fp32, a=e=96, b=c=d=64 — the cuTENSOR
call sequence of cuTENSOR/einsum.cu:248-339 (descriptors -> contraction -> plan -> cutensorContract)
driven through the C ABI of lib/libcutensor.so.  A "step" is one cutensorContract call (GETT kernel +
split-K fold) on tensors already resident in HBM; plan creation is outside the timed region, as in the
reference samples (contraction.cu:218-222 vs :253-270).

  N = 1 : the einsum above.
  N > 1 : one process per GPU (torch.distributed, backend nccl = RCCL).  The contracted mode b is
          sharded: every rank owns A[:, b_r, :, :] and B[:, :, b_r, :] of a b = 64*N problem
          (weak scaling, per-GPU work fixed), computes its partial C and the partials are summed by
          an RCCL all-reduce of the 96x96 result — the cheapest exchange for this shape, because
          |C| (36 KB) << |B| (100 MB); sharding a free mode would all-gather B instead.

One JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EXT = dict(a=96, b=64, c=64, d=64, e=96)
FLOP = 2.0 * EXT["a"] * EXT["b"] * EXT["c"] * EXT["d"] * EXT["e"]          # contraction.cu:61 formula
BYTES = 4.0 * (EXT["a"] * EXT["b"] * EXT["c"] * EXT["d"] + EXT["d"] * EXT["c"] * EXT["b"] * EXT["e"] + EXT["a"] * EXT["e"])
PEAK_TFLOPS_F32_MFMA = 157.3      # 256 CU x 256 flop/clk x 2.4 GHz (MI355X_MICROARCH.md)


def cpu_baseline(a_np, b_np):
    """TTGT over OpenBLAS on the host cores (reported baseline, not the target)."""
    from oracle import ttgt
    cores = len(os.sched_getaffinity(0))
    out, t_total, t_gemm = ttgt.time_ttgt(a_np, b_np, reps=3)
    return out, {
        "value": FLOP / t_total / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "port",
        "sample": "full workload (4.83 GFLOP), TTGT = transpose B + OpenBLAS sgemm via numpy, min of 3, "
                  "transposes included (GEMM-only %.1f GFLOP/s)" % (FLOP / t_gemm / 1e9),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--burn-in-ms", type=float, default=40.0,
                    help="untimed device clock ramp before the warmup steps: MI355X needs ~15-20 ms of sustained load "
                         "before it settles into its steady clock state (profiles/r01e_long_run_timeline.txt)")
    ap.add_argument("--algo", type=str, default="default", help="default | patient | <candidate index>")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC
        # device_id: bind the RCCL communicator to this rank's GPU now (eager init) instead of at the first collective
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from cudalibrarysamples_amd import cutensor as ct, ops, sharding

    # ---- synthetic inputs: U(0,1) fp32, fixed seed per rank, generated on the device ---------------
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + rank)
    a = torch.rand((EXT["a"], EXT["b"], EXT["c"], EXT["d"]), generator=g, device="cuda", dtype=torch.float32)
    b = torch.rand((EXT["d"], EXT["c"], EXT["b"], EXT["e"]), generator=g, device="cuda", dtype=torch.float32)
    nbuf = 4
    outs = [torch.empty((EXT["a"], EXT["e"]), device="cuda", dtype=torch.float32) for _ in range(nbuf)]

    # ---- plan (einsum helper's view: row-major -> reversed column-major modes, einsum.cu:186-196) ---
    h = ops.Handle(plan_cache=64)
    algo = ct.ALGO_DEFAULT
    if args.algo == "patient":
        algo = ct.ALGO_DEFAULT_PATIENT
    elif args.algo != "default":
        algo = int(args.algo)
    plan = ops.contraction_plan(h, [EXT[c] for c in "dcba"], "dcba", [EXT[c] for c in "ebcd"], "ebcd",
                                [EXT[c] for c in "ea"], "ea", algo=algo, workspace_limit=1 << 30)
    ws = torch.empty(max(plan.required_workspace, 256), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    desc = plan.describe()

    pending = []

    def step(i):
        out = outs[i % nbuf]
        plan.contract(1.0, a.data_ptr(), b.data_ptr(), 0.0, out.data_ptr(), out.data_ptr(), ws.data_ptr(),
                      plan.required_workspace, stream)
        if world > 1:
            # fold the K-shards: RCCL all-reduce of the 36 KB result, overlapped with the next step
            if len(pending) >= nbuf - 1:
                pending.pop(0).wait()
            pending.append(sharding.fold_partials(out, dist, async_op=True))

    def drain():
        while pending:
            pending.pop(0).wait()

    def fence():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- device clock ramp (untimed, not a step count: wall-clock bounded) -----------------------------
    # A cold MI355X runs the first ~12-20 ms of a kernel stream ~10 % slower than its steady state (GETT kernel
    # 43.5 us -> 38.9 us on the same inputs); production streams live in the steady state, so the bench reaches it
    # before the W warmup steps.  Same call sequence as a step, result discarded.
    burn_steps = 0
    if args.burn_in_ms > 0:
        t_end = time.perf_counter() + args.burn_in_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(50):
                plan.contract(1.0, a.data_ptr(), b.data_ptr(), 0.0, outs[0].data_ptr(), outs[0].data_ptr(), ws.data_ptr(),
                              plan.required_workspace, stream)
            torch.cuda.synchronize()
            burn_steps += 50
    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * FLOP / (elapsed / args.steps) / 1e9

    # ---- roofline of the dominant kernel: HIP events recorded by the library on the launch stream ---
    roof = None
    cpu = None
    if rank == 0:
        # GETT kernel alone, in the same steady state as the timed loop: the fold is switched off (library
        # diagnostic), `n` launches go out back to back (rocprofv3 shows < 0.05 us between them) and ONE HIP event
        # pair on the launch stream brackets them.  Per-launch event pairs would put a ~6 us idle gap after every
        # kernel, and the gapped stream runs at a different clock than the timed loop (42.6 vs 38.9 us).
        n = max(args.steps, 200)
        ct.lib.ctamdSetSplitKFold(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(50):
            plan.contract(1.0, a.data_ptr(), b.data_ptr(), 0.0, outs[0].data_ptr(), outs[0].data_ptr(), ws.data_ptr(),
                          plan.required_workspace, stream)
        e0.record()
        for i in range(n):
            plan.contract(1.0, a.data_ptr(), b.data_ptr(), 0.0, outs[0].data_ptr(), outs[0].data_ptr(),
                          ws.data_ptr(), plan.required_workspace, stream)
        e1.record()
        torch.cuda.synchronize()
        ct.lib.ctamdSetSplitKFold(1)
        batch_ms = e0.elapsed_time(e1) / n
        # per-launch event pairs (the gapped stream), kept for reference
        ct.lib.ctamdProfileBegin()
        for i in range(200):
            plan.contract(1.0, a.data_ptr(), b.data_ptr(), 0.0, outs[0].data_ptr(), outs[0].data_ptr(),
                          ws.data_ptr(), plan.required_workspace, stream)
        torch.cuda.synchronize()
        mean_ms, min_ms = ctypes.c_float(0), ctypes.c_float(0)
        ct.lib.ctamdProfileEnd(ctypes.byref(mean_ms), ctypes.byref(min_ms))
        gapped_mean_us, gapped_min_us = mean_ms.value * 1e3, min_ms.value * 1e3
        mean_ms = ctypes.c_float(batch_ms)
        prop = torch.cuda.get_device_properties(0)
        cus = prop.multi_processor_count
        clock_ghz = getattr(prop, "clock_rate", 2400000) / 1e6
        peak = cus * 256 * clock_ghz / 1e3      # TFLOP/s from the device's own CU count and max clock
        achieved = FLOP / (mean_ms.value * 1e-3) / 1e12 if n else 0.0
        # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE collected by separate
        # `rocprofv3 --pmc` runs of this same command, tools/gpu_profile.sh; corrected per
        # MI355X_MICROARCH.md and committed under profiles/).  Counters cannot be read from inside the
        # timed process, so the committed per-launch figure of the same kernel is reported, else null.
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic_einsum.json")) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            traffic = None
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak if peak else None, "traffic": traffic,
                "traffic_source": "profiles/pmc_traffic_einsum.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch)" if traffic else None,
                "kernel": "%s<%dx%dx%d,w%dx%dx%d> (table index %d)" % (desc.get("kname", "gett_f32_kernel"), desc["bm"], desc["bn"], desc["bk"],
                                                                       desc["wm"], desc["wn"], desc["wk"], desc.get("kernel", -1)),
                "launches": n, "mean_us": mean_ms.value * 1e3,
                "timing": "one HIP event pair around %d back-to-back launches of the GETT kernel (fold off)" % n,
                "gapped_mean_us": gapped_mean_us, "gapped_min_us": gapped_min_us,
                "algorithmic_flop_per_launch": FLOP, "algorithmic_bytes_per_launch": BYTES,
                "hbm_equiv_TBps": BYTES / (mean_ms.value * 1e-3) / 1e12 if n else None,
                "cus": cus, "clock_ghz": clock_ghz, "nominal_peak": PEAK_TFLOPS_F32_MFMA}
        if world == 1 and not args.no_cpu:
            a_np, b_np = a.cpu().numpy(), b.cpu().numpy()
            ref, cpu = cpu_baseline(a_np, b_np)
            got = outs[(args.steps - 1) % nbuf].cpu().numpy()
            err = float(np.max(np.abs(got - ref) / np.abs(ref)))
            cpu["max_rel_diff_vs_gpu"] = err

    if rank == 0:
        line = {
            "metric": "contraction GFLOP/s, fp32 einsum abcd,dcbe->ae", "value": value, "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "burn_in_ms": args.burn_in_ms, "burn_in_steps": burn_steps,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "einsum.cu 'abcd,dcbe->ae' a=e=96 b=c=d=64 fp32 (BASELINE configs[1])"
                       + ("" if world == 1 else ", b sharded x%d (b=%d), RCCL all-reduce of C" % (world, 64 * world)),
                       "plan": desc, "algo": args.algo,
                       "frac_of_nominal_f32_mfma_peak": value / 1e3 / (PEAK_TFLOPS_F32_MFMA * world)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

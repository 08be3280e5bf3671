#!/usr/bin/env python3
"""bench.py — benchmarks of the MI355X-native tensor-contraction engine behind the cuTENSOR C ABI.

    python bench.py --gpus N --steps K --warmup W        (one JSON line on stdout)

N = 1 (BASELINE.json configs[1], the headline): einsum 'abcd,dcbe->ae', fp32, a=e=96, b=c=d=64 — the call
    sequence of cuTENSOR/einsum.cu:248-339 (descriptors -> contraction -> plan -> cutensorContract) through
    lib/libcutensor.so.  A "step" is one cutensorContract call (GETT kernel + split-K fold) on tensors resident in
    HBM; plan creation is outside the timed region, as in the samples (contraction.cu:218-222 vs :253-270).
    The same line carries `secondary`: the other BASELINE configs measured in the same process with the samples'
    own formulas — contraction.cu default fp32 (configs[0] shape on the GPU), 2048^3 permute + reduce (configs[2]),
    bf16 8192^3 (configs[3]), cuTENSORMg on one device (configs[4] at n = 1) — each with its roofline, and
    `cold_operands_value`: the headline with four rotating (A, B) pairs (805 MB > the 256-MiB Infinity Cache).

N > 1 (BASELINE.json configs[4], north_star's "cuTENSORMg sharded case"): cutensorMgContraction, fp32
    C[i,j] = A[i,k] B[k,j], the largest free mode i cut over min(N, visible) DISTINCT devices, B distributed in
    column slabs and all-gathered over xGMI by RCCL while the first local contraction runs (csrc/mg/mg.cpp) —
    ONE process drives all devices, as cuTENSORMg/contraction_multi_gpu.cu:151,286-345 does.  `value` is the scaled
    shape (16384^3, blog_post.cu:155-186 class), K calls back to back; the sample's default (4096^3, block 2048,
    wall clock + per-device sync, min of 3: contraction_multi_gpu.cu:323-345) is in `secondary`.  Strong scaling:
    the problem is fixed, `speedup_vs_1` is measured in the same run.
    Launched plainly (`python bench.py --gpus 8`) this process opens the devices itself.  Launched by
    torch.distributed.run with WORLD_SIZE = N (one rank per GPU), rank 0 runs the measurement in a child process
    over the N devices while the other ranks wait on a CPU (gloo) barrier — their GPUs stay free for the
    measurement — and all ranks then run the one-process-per-GPU variant of the headline einsum (contracted mode b
    sharded, RCCL all-reduce of the 36-KB result) which is reported as a secondary line (and is the fallback
    `value` if the multi-device measurement fails).  With fewer visible GPUs than requested the line says
    `"n_gpus": <used>, "requested_gpus": N`.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EXT = dict(a=96, b=64, c=64, d=64, e=96)
FLOP = 2.0 * EXT["a"] * EXT["b"] * EXT["c"] * EXT["d"] * EXT["e"]          # contraction.cu:61 formula
BYTES = 4.0 * (EXT["a"] * EXT["b"] * EXT["c"] * EXT["d"] + EXT["d"] * EXT["c"] * EXT["b"] * EXT["e"] + EXT["a"] * EXT["e"])
PEAK_TFLOPS_F32_MFMA = 157.3      # 256 CU x 256 flop/clk x 2.4 GHz (MI355X_MICROARCH.md)
PEAK_TFLOPS_BF16_MFMA = 2516.6    # 256 CU x 4096 flop/clk x 2.4 GHz, dense
PEAK_HBM_GBPS = 8000.0            # HBM3E spec (6.3 TB/s measured float4 copy)
PEAK_TFLOPS_F64_MFMA = 78.6       # v_mfma_f64_16x16x4_f64: 256 CU x 4 SIMD x 2048 flop / 64 clk x 2.4 GHz (the guide quotes no fp64 matrix rate; the
                                  # instruction's issue interval is measured by ctamdMeasureMfmaCeilingF64, tools/bench_gen.py)
MG_SCALED_EXTENT = 16384
MG_SAMPLE_EXTENT = 4096


def openblas_threads():
    try:
        from threadpoolctl import threadpool_info
        for lib in threadpool_info():
            if lib.get("user_api") == "blas":
                return int(lib["num_threads"])
    except Exception:
        pass
    return None


def cpu_baseline(a_np, b_np):
    """TTGT over OpenBLAS on the host cores (reported baseline, not the target)."""
    from oracle import ttgt
    affinity = len(os.sched_getaffinity(0))
    threads = openblas_threads() or min(affinity, 64)
    out, t_total, t_gemm = ttgt.time_ttgt(a_np, b_np, reps=3)
    return out, {
        "value": FLOP / t_total / 1e9, "unit": "GFLOP/s", "cores": threads, "kind": "port",
        "sample": "full workload (4.83 GFLOP), TTGT = transpose B + OpenBLAS sgemm via numpy, min of 3, transposes included "
                  "(GEMM-only %.1f GFLOP/s); cores = threads OpenBLAS ran (its build caps at 64), host affinity mask %d"
                  % (FLOP / t_gemm / 1e9, affinity),
    }


# ---------------------------------------------------------------------------------------------------------
# cuTENSORMg measurement (one process, n devices)
# ---------------------------------------------------------------------------------------------------------
def mg_measure(ndev, extent, steps, warmup, sample_reps=3, check=True, virtual=False):
    """C[i,j] = A[i,k] B[k,j] fp32 through libcutensorMg on devices 0..ndev-1: i cut ndev ways (A, C row slabs), B in
    column slabs (all-gathered).  Returns a dict with the throughput of `steps` back-to-back calls and the sample's
    protocol (wall clock + per-device sync, min of `sample_reps`)."""
    import torch
    from cudalibrarysamples_amd import cutensormg as cm
    n = ndev
    E = extent - extent % (16 * n)        # every device count cuts the extents evenly (N = 1, 2, 4, 8 leave them as they are)
    modes = ["ik", "kj", "ij"]
    block = [dict(i=E // n), dict(j=E // n), dict(i=E // n, j=E // n)]
    dcount = [dict(i=n), dict(j=n), dict(i=n)]
    flop = 2.0 * E * E * E
    devs = [0] * n if virtual else list(range(n))      # virtual: n logical devices on GPU 0 (exercises the machinery, not xGMI)
    con = cm.Contraction(devs, modes, dict(i=E, j=E, k=E), block, dcount)
    try:
        d = con.describe()
        cells = []
        for k in range(3):
            row = []
            for g in range(n):
                gen = torch.Generator(device="cuda:%d" % devs[g])
                gen.manual_seed(1234 + 17 * k + g)
                row.append(torch.rand(E * (E // n), generator=gen, device="cuda:%d" % devs[g], dtype=torch.float32))
            cells.append(row)
        ws = [torch.empty(int(con.ws_sizes[g]), dtype=torch.uint8, device="cuda:%d" % devs[g]) for g in range(n)]
        streams = [torch.cuda.Stream(device=devs[g]) for g in range(n)]
        ptr = [[t.data_ptr() for t in row] for row in cells]
        wsp = [t.data_ptr() for t in ws]
        sp = [s.cuda_stream for s in streams]

        def sync_all():
            for g in set(devs):
                torch.cuda.synchronize(g)

        def call():
            cm.check(con.run(1.0, ptr[0], ptr[1], 0.0, ptr[2], ptr[2], wsp, sp))

        sync_all()
        for _ in range(max(warmup, 1)):
            call()
        sync_all()
        # the sample's protocol: contraction_multi_gpu.cu:323-345
        best = 1e30
        for _ in range(sample_reps):
            t0 = time.perf_counter()
            call()
            sync_all()
            best = min(best, time.perf_counter() - t0)
        # throughput: K calls back to back, one sync of every device at the end
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            call()
        sync_all()
        elapsed = time.perf_counter() - t0
        d2 = con.describe()    # after the calls: which transport the first-two-calls trial chose
        err = None
        if check:
            # sampled fp64 dot products: C[i, j] lives in row slab i // (E/n) as [j][i_local]; A[i, :] = slab[k][i_local];
            # B[:, j] = column slab j // (E/n), local column j % (E/n), contiguous in k
            rng = np.random.default_rng(5)
            worst = 0.0
            per = E // n
            for _ in range(16):
                i, j = int(rng.integers(0, E)), int(rng.integers(0, E))
                a_row = cells[0][i // per].view(E, per)[:, i % per].double().cpu()
                b_col = cells[1][j // per].view(per, E)[j % per, :].double().cpu()
                ref = float((a_row * b_col).sum())
                got = float(cells[2][i // per].view(E, per)[j, i % per].cpu())
                worst = max(worst, abs(got - ref) / abs(ref))
            err = worst
            if worst > 1e-4:
                raise RuntimeError("cuTENSORMg result check failed: max rel err %.3e" % worst)
        return {"extent": E, "devices": n, "distinct_devices": len(set(devs)), "steps": steps, "ms_per_step": elapsed / steps * 1e3,
                "gflops": flop / (elapsed / steps) / 1e9, "sample_protocol_min_ms": best * 1e3,
                "sample_protocol_gflops": flop / best / 1e9, "flop": flop, "elapsed_s": elapsed,
                "gather_bytes_per_call": d["remoteBytes"], "local_copy_bytes_per_call": d["localCopyBytes"],
                "pieces": len(d["pieces"]), "gather_waves": d["numWaves"],
                "transport": (d2["transport"] if (n > 1 or d.get("forceGather")) else "none"), "rccl": bool(d["useRccl"]),
                "forced_gather": bool(d.get("forceGather")), "rccl_ranks_seen": (len(set(devs)) if d["useRccl"] else 0),
                "all_gather_eligible": d2.get("allGatherEligible"), "transport_trial_ms": d2.get("trialMs"),
                "transport_chosen": {0: "undecided", 1: "allgather", 2: "sendrecv"}.get(d2.get("chosen", 0)),
                "max_rel_err_sampled": err}
    finally:
        con.close()


def mg_child_main(args):
    """Child process: the multi-device cuTENSORMg measurement; one JSON line."""
    import torch
    virtual = args.mg_virtual
    ndev = args.mg_child if virtual else min(args.mg_child, torch.cuda.device_count())
    out = {"devices": ndev, "virtual": virtual}
    scaled, sample = (MG_SCALED_EXTENT, MG_SAMPLE_EXTENT) if not virtual else (4096, 2048)
    for attempt in range(2):
        try:
            out["scaled"] = mg_measure(ndev, scaled, args.steps, args.warmup, virtual=virtual)
            out["sample"] = mg_measure(ndev, sample, max(20, min(args.steps, 200)), 3, virtual=virtual)
            if ndev > 1:   # the same problems on one device, same process: the base of the strong-scaling speedup
                out["scaled_1"] = mg_measure(1, scaled, 3, 1, check=False)
                out["sample_1"] = mg_measure(1, sample, 20, 3, check=False)
            out.pop("error", None)
            break
        except Exception as e:   # noqa: BLE001 — reported to the parent, which falls back
            out["error"] = "%s: %s" % (type(e).__name__, e)
            if attempt == 0 and ndev > 1 and os.environ.get("CUTENSORMG_AMD_TRANSPORT") != "peer":
                # the RCCL send/recv gather failed (status or wrong values): once more with peer copies over xGMI
                out["first_attempt_error"] = out["error"]
                os.environ["CUTENSORMG_AMD_TRANSPORT"] = "peer"
                continue
            break
    print("MGCHILD " + json.dumps(out), flush=True)


def run_mg_child(ndev, steps, warmup, timeout_s, virtual=False, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
              "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), "--mg-child", str(ndev), "--steps", str(steps), "--warmup", str(warmup)]
    if virtual:
        cmd.append("--mg-virtual")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": "multi-device measurement timed out after %d s" % timeout_s}
    for line in r.stdout.splitlines():
        if line.startswith("MGCHILD "):
            res = json.loads(line[len("MGCHILD "):])
            if r.returncode != 0 and "error" not in res:
                res["error"] = "child exited with %d" % r.returncode
            return res
    return {"error": "child rc=%d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}


def run_einsum_ranks(n, steps, warmup, timeout_s=420):
    """A plain `python bench.py --gpus N` launch (world size 1) on a box with N GPUs: the one-process-per-GPU K-sharded einsum
    (b cut N ways, RCCL all-reduce of the 36-KB result) is measured by N ranks this process spawns itself, exactly as the driver
    would launch them; returns rank 0's JSON line."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup),
           "--einsum-only", "--no-cpu", "--no-secondary"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": "spawned einsum ranks timed out after %d s" % timeout_s}
    for line in reversed(r.stdout.splitlines()):
        if line.startswith("{") and '"metric"' in line:
            try:
                return json.loads(line)
            except ValueError:
                break
    return {"error": "spawned ranks rc=%d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}


def einsum_flow_line(calls=2000):
    """secondary entry: Einsum::execute as cuTENSOR/einsum.cu:264-339 runs it (descriptors + plan preference + contraction
    descriptor + plan created and destroyed inside every call, plan cache at 1024 entries :443-445), timed from C by the native
    driver samples/einsum.hip --flow next to the plan-once loop on the same buffers."""
    exe = os.path.join(ROOT, "samples", "bin", "einsum")
    r = subprocess.run([exe, "--flow", "--calls", str(calls), "--cache", "1024"], capture_output=True, text=True, timeout=300)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        return {"workload": "einsum.cu flow (plan per call)", "error": "rc %d: %s" % (r.returncode, (r.stdout + r.stderr)[-300:])}
    d = json.loads(line[-1])
    return {"workload": "einsum.cu flow 'abcd,dcbe->ae': descriptors + plan + contract + destroy per call, plan cache 1024 (einsum.cu:264-339,:445), "
                        "timed from C (samples/einsum.hip --flow)",
            "dtype": "f32", "value": d["flow_gflops"], "unit": "GFLOP/s", "ms_per_call": d["flow_us_per_call"] * 1e-3,
            "plan_once_value": d["plan_once_gflops"], "flow_over_plan_once": d["flow_over_plan_once"],
            "host_issue_us_per_call": d["flow_host_issue_us_per_call"], "plan_once_host_issue_us_per_call": d["plan_once_host_issue_us_per_call"],
            "plan_create_us": {"hit": d["plan_create_us_hit"], "miss": d["plan_create_us_miss"]}, "calls": d["calls"],
            "max_rel_diff_flow_vs_plan_once": d["max_rel_diff_flow_vs_plan_once"]}


def live_pmc_traffic(timeout_s=150):
    """HBM bytes per launch of the dominant kernel, measured IN THIS RUN: two separate `rocprofv3 --pmc` passes (FETCH_SIZE, then
    WRITE_SIZE — never combined with tracing) over a short headline-only child of this script, read from the rocpd databases and
    corrected as MI355X_MICROARCH.md prescribes (KiB units; FETCH_SIZE counts a wide coalesced read stream at half its bytes).
    Returns a dict or None (no rocprofv3, a pass failed, nothing collected): the committed figure of the latest profile is then used."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="bench_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "-d", out, "-o", "r", "--", sys.executable, os.path.abspath(__file__), "--steps", "100", "--warmup", "10",
                   "--no-cpu", "--no-secondary", "--no-pmc", "--burn-in-ms", "0"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd="/tmp")
            dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            c = sqlite3.connect(dbs[0])
            rows = list(c.execute("select kernel_name, avg(value), count(*) from counters_collection where kernel_name like '%ctamd%gett%' "
                                  "and counter_name = ? group by kernel_name order by count(*) desc", (counter,)))
            c.close()
            if not rows:
                return None
            vals[counter] = {"kernel": rows[0][0][:100], "mean_KiB_per_launch": rows[0][1], "launches": rows[0][2]}
        read_b = 2.0 * 1024.0 * vals["FETCH_SIZE"]["mean_KiB_per_launch"]
        write_b = 1024.0 * vals["WRITE_SIZE"]["mean_KiB_per_launch"]
        return {"hbm_bytes_per_launch": read_b + write_b, "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
                "raw_FETCH_SIZE_KiB": vals["FETCH_SIZE"]["mean_KiB_per_launch"], "raw_WRITE_SIZE_KiB": vals["WRITE_SIZE"]["mean_KiB_per_launch"],
                "launches": min(vals["FETCH_SIZE"]["launches"], vals["WRITE_SIZE"]["launches"]), "kernel": vals["FETCH_SIZE"]["kernel"]}
    except (subprocess.TimeoutExpired, OSError, sqlite3.Error, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_pmc_traffic_of(cmd, kernel_like, timeout_s=240):
    """HBM bytes per launch of the kernels matching `kernel_like` (SQL LIKE pattern) in a child command, measured IN THIS RUN by two
    separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE; never combined with tracing).  Raw counter values are reported next to
    the corrected figure: FETCH_SIZE x 2 is the guide's correction for wide coalesced read streams (MI355X_MICROARCH.md, HBM
    section: 128-byte requests tallied at 64 bytes); narrower access patterns are not calibrated, so `correction` names what was applied."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="bench_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--pmc", counter, "-d", out, "-o", "r", "--"] + cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd="/tmp")
            dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            c = sqlite3.connect(dbs[0])
            rows = list(c.execute("select kernel_name, avg(value), count(*) from counters_collection where kernel_name like ? "
                                  "and counter_name = ? group by kernel_name order by count(*) desc", (kernel_like, counter)))
            c.close()
            if not rows:
                return None
            vals[counter] = {"kernel": rows[0][0][:100], "mean_KiB_per_launch": rows[0][1], "launches": rows[0][2]}
        read_b = 2.0 * 1024.0 * vals["FETCH_SIZE"]["mean_KiB_per_launch"]
        write_b = 1024.0 * vals["WRITE_SIZE"]["mean_KiB_per_launch"]
        return {"hbm_bytes_per_launch": read_b + write_b, "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
                "raw_FETCH_SIZE_KiB": vals["FETCH_SIZE"]["mean_KiB_per_launch"], "raw_WRITE_SIZE_KiB": vals["WRITE_SIZE"]["mean_KiB_per_launch"],
                "correction": "FETCH_SIZE x 2 (wide coalesced reads are tallied at half their bytes on gfx950), WRITE_SIZE x 1; KiB units",
                "launches": min(vals["FETCH_SIZE"]["launches"], vals["WRITE_SIZE"]["launches"]), "kernel": vals["FETCH_SIZE"]["kernel"]}
    except (subprocess.TimeoutExpired, OSError, sqlite3.Error, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_profile_of(cmd, kernel_like, timeout_s=240):
    """What rocprofv3 says about the kernels matching `kernel_like` (SQL LIKE) in a child command, measured IN THIS RUN by two separate
    passes (never combined: counters never ride with tracing):
      * `--kernel-trace --stats`: launches, average / minimum duration [us];
      * `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY`: matrix-pipe busy % = MFMA-busy cycles / SIMDs over the
        GPU-active cycles per XCD (GRBM_GUI_ACTIVE is summed over the 8 XCDs, the busy counter over the 4 x CUs SIMDs), and the sustained
        clock = active cycles per XCD / the kernel's duration in that same pass.
    Returns a dict or None."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="bench_prof_")
    env = dict(os.environ, TMPDIR="/tmp")
    out = {}
    try:
        import torch
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        for tag, flags in (("trace", ["--kernel-trace", "--stats"]),
                           ("pmc", ["--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"])):
            d = os.path.join(tmp, tag)
            r = subprocess.run([exe] + flags + ["-d", d, "-o", "r", "--"] + cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd="/tmp")
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return out or None
            c = sqlite3.connect(dbs[0])
            rows = list(c.execute("select name, count(*), avg(duration), min(duration) from kernels where name like ? group by name "
                                  "order by sum(duration) desc", (kernel_like,)))
            if not rows:
                c.close()
                return out or None
            name, calls, avg_ns, min_ns = rows[0]
            if tag == "trace":
                out.update({"kernel": name[:100], "launches": calls, "kernel_avg_us": avg_ns / 1e3, "kernel_min_us": min_ns / 1e3})
            else:
                cnt = dict(c.execute("select counter_name, avg(value) from counters_collection where kernel_name like ? group by counter_name",
                                     (kernel_like,)).fetchall())
                busy, active = cnt.get("SQ_VALU_MFMA_BUSY_CYCLES"), cnt.get("GRBM_GUI_ACTIVE")
                if busy and active:
                    per_xcd = active / 8.0
                    out.update({"mfma_busy_pct": 100.0 * (busy / (4.0 * cus)) / per_xcd, "sustained_clock_ghz": per_xcd / avg_ns,
                                # GRBM_GUI_ACTIVE also counts the dispatch's ramp-in / ramp-out: for a ~40-us kernel it exceeds the kernel's own
                                # cycles (the "clock" above then reads > 2.4 GHz and the busy share too low) — the same busy cycles against the
                                # kernel's duration at the nominal 2.4 GHz bound the share from the other side
                                "mfma_busy_pct_of_kernel_time_at_nominal_clock": 100.0 * (busy / (4.0 * cus)) / (avg_ns * 2.4),
                                "kernel_avg_us_under_pmc": avg_ns / 1e3,
                                "wait_any_pct_of_wave_cycles": (100.0 * cnt["SQ_WAIT_ANY"] / cnt["SQ_WAVE_CYCLES"]) if cnt.get("SQ_WAVE_CYCLES") and cnt.get("SQ_WAIT_ANY") else None,
                                "pmc_raw": {k: cnt.get(k) for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY")},
                                "pmc_formula": "busy %% = SQ_VALU_MFMA_BUSY_CYCLES / (4 x %d SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs); clock = (GRBM_GUI_ACTIVE / 8) / kernel duration of the same pass" % cus})
            c.close()
        return out or None
    except (subprocess.TimeoutExpired, OSError, sqlite3.Error, KeyError, ValueError):
        return out or None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def newest_trace_summary():
    """(relative path, {kernel substring: (avg_us, min_us, calls)}) of the newest committed rocprofv3 kernel-trace summary of the
    headline command (profiles/r*_einsum_trace.summary.txt): lets a reader reproduce roofline.frac from the line alone."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*einsum_trace.summary.txt")))
    if not files:
        return None, {}
    out = {}
    try:
        for ln in open(files[-1]):
            for key in ("gett_f32_stream_kernel", "splitk_reduce_frag_flat_kernel"):
                if key in ln and key not in out:
                    cols = ln.split()
                    out[key] = (float(cols[-4]), float(cols[-3]), int(cols[-5]))
    except (OSError, ValueError, IndexError):
        return None, {}
    return os.path.relpath(files[-1], ROOT), out


def newest_traffic_file():
    """profiles/*pmc_traffic_einsum.json of the latest round (names sort by round prefix; the un-prefixed round-1 file last)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*pmc_traffic_einsum.json")))
    if files:
        return files[-1]
    f = os.path.join(ROOT, "profiles", "pmc_traffic_einsum.json")
    return f if os.path.exists(f) else None


def mg_lines(res, key, one_key):
    """secondary entry of one cuTENSORMg measurement."""
    m = res[key]
    n = m["devices"]
    peak = PEAK_TFLOPS_F32_MFMA * n
    line = {"workload": "cuTENSORMg contraction_multi_gpu.cu C[i,j]=A[i,k]B[k,j] fp32 %d^3, free mode i cut over %d device(s), B all-gathered"
                        % (m["extent"], n),
            "dtype": "f32", "value": m["gflops"], "unit": "GFLOP/s", "n_gpus": n, "ms_per_call": m["ms_per_step"],
            "sample_protocol": {"min_ms": m["sample_protocol_min_ms"], "gflops": m["sample_protocol_gflops"],
                                "what": "wall clock + per-device sync, min of 3 (contraction_multi_gpu.cu:323-345)"},
            "gather_bytes_per_call": m["gather_bytes_per_call"], "transport": m["transport"], "pieces": m["pieces"],
            "max_rel_err_sampled": m["max_rel_err_sampled"],
            "roofline": {"bound": "mfma", "achieved": m["gflops"] / 1e3, "peak": peak, "unit": "TFLOP/s", "frac": m["gflops"] / 1e3 / peak}}
    if one_key in res:
        line["speedup_vs_1"] = res[one_key]["ms_per_step"] / m["ms_per_step"]
        line["one_device_gflops"] = res[one_key]["gflops"]
    return line


# ---------------------------------------------------------------------------------------------------------
# secondary single-GPU configs (BASELINE configs 0 / 2 / 3), a few repetitions each
# ---------------------------------------------------------------------------------------------------------
def timed_batch(torch, fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def secondary_single_gpu(torch, ct, ops, h, stream):
    out = []
    # ---- contraction.cu default: C[m,u,n,v] = 1.1 * A[m,h,k,n] B[u,k,v,h], fp32 (contraction.cu:46-59,184-185) ----
    try:
        e = dict(m=96, n=96, u=96, v=64, h=64, k=64)
        mA, mB, mC = "mhkn", "ukvh", "munv"
        g = torch.Generator(device="cuda")
        g.manual_seed(1234)
        A = torch.rand(int(np.prod([e[c] for c in mA])), generator=g, device="cuda")
        B = torch.rand(int(np.prod([e[c] for c in mB])), generator=g, device="cuda")
        C = torch.zeros(int(np.prod([e[c] for c in mC])), device="cuda")
        p = ops.contraction_plan(h, [e[c] for c in mA], mA, [e[c] for c in mB], mB, [e[c] for c in mC], mC, workspace_limit=1 << 30)
        ws = torch.empty(max(p.required_workspace, 256), dtype=torch.uint8, device="cuda")
        ms = timed_batch(torch, lambda: p.contract(1.1, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(),
                                                   p.required_workspace, stream), reps=10)
        flop = 2.0 * np.prod([float(v) for v in e.values()])
        byts = 4.0 * (A.numel() + B.numel() + C.numel())
        tf = flop / (ms * 1e-3) / 1e12
        d = p.describe()
        out.append({"workload": "contraction.cu default C[m,u,n,v]=A[m,h,k,n]B[u,k,v,h] fp32 (BASELINE configs[0] shape, on the GPU)",
                    "dtype": "f32", "value": tf * 1e3, "unit": "GFLOP/s", "ms_per_call": ms, "GBps_sample_formula": byts / (ms * 1e-3) / 1e9,
                    "kernel": "%s<%dx%dx%d>" % (d["kname"], d["bm"], d["bn"], d["bk"]),
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS_F32_MFMA, "unit": "TFLOP/s", "frac": tf / PEAK_TFLOPS_F32_MFMA,
                                 "algorithmic_flop": flop, "algorithmic_bytes": byts}})
        p.destroy()
        del A, B, C, ws
    except Exception as ex:   # noqa: BLE001
        out.append({"workload": "contraction.cu default fp32", "error": "%s: %s" % (type(ex).__name__, ex)})
    # ---- bf16 8192^3 (configs[3]) ---------------------------------------------------------------------------------
    try:
        n = 8192
        g = torch.Generator(device="cuda")
        g.manual_seed(1)
        A = (torch.rand((n, n), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
        B = (torch.rand((n, n), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
        D = torch.empty((n, n), device="cuda", dtype=torch.bfloat16)
        p = ops.contraction_plan(h, [n, n], "mk", [n, n], "kn", [n, n], "mn", dtype=ct.R_16BF)
        fn = lambda: p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), stream=stream)   # noqa: E731
        for _ in range(40):   # clock ramp under this kernel's own load
            fn()
        ms = timed_batch(torch, fn, reps=30)
        flop = 2.0 * n ** 3
        tf = flop / (ms * 1e-3) / 1e12
        d = p.describe()
        # what this box sustains on nothing but bf16 MFMAs of the kernel's shape (16x16x32 since round 3; the 32x32x16 stream beside
        # it), on zeros (= nominal peak) and on U(-1,1) register data (power-limited, differs by box): the GETT kernel cannot beat the
        # second number on the same kind of data
        ceil = {}
        for name, kind, shape in (("zeros", 0, 1), ("uniform", 1, 1), ("uniform_32x32x16", 1, 0)):
            v = ctypes.c_float(0)
            if ct.lib.ctamdMeasureMfmaCeilingShape(1, kind, shape, ctypes.byref(v)) == 0:
                ceil[name] = float(v.value)
        ms2 = timed_batch(torch, fn, reps=30)                 # again, after the ceiling runs (same clock state), keep the better
        ms = min(ms, ms2)
        tf = flop / (ms * 1e-3) / 1e12
        out.append({"workload": "contraction bf16 C[m,n]=A[m,k]B[k,n] M=N=K=8192, U(-1,1) data, fp32 accumulate (BASELINE configs[3])",
                    "dtype": "bf16", "value": tf * 1e3, "unit": "GFLOP/s", "ms_per_call": ms, "kernel": d["kname"],
                    "mfma_only_tflops_this_box": ceil,
                    "frac_of_mfma_only_rate_on_uniform_data": tf / ceil["uniform"] if ceil.get("uniform") else None,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS_BF16_MFMA, "unit": "TFLOP/s", "frac": tf / PEAK_TFLOPS_BF16_MFMA,
                                 "algorithmic_flop": flop, "algorithmic_bytes": 3.0 * 2 * n * n}})
        p.destroy()
        del A, B, D
    except Exception as ex:   # noqa: BLE001
        out.append({"workload": "contraction bf16 8192^3", "error": "%s: %s" % (type(ex).__name__, ex)})
    # ---- bf16 mid-size and small shapes (the size class of blog_post.cu's per-device pieces and of contraction.cu retyped): the 128 x 128
    #      and 64 x 64 tile kernels of round 4, whatever the planner picks -----------------------------------------------------------
    for (M, N, K) in ((2048, 2048, 2048), (4096, 1024, 4096), (1024, 1024, 1024)):
        try:
            g = torch.Generator(device="cuda")
            g.manual_seed(2)
            A = (torch.rand((K, M), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)      # modes "mk": m fastest
            B = (torch.rand((N, K), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)      # "kn"
            D = torch.empty((N, M), device="cuda", dtype=torch.bfloat16)
            p = ops.contraction_plan(h, [M, K], "mk", [K, N], "kn", [M, N], "mn", dtype=ct.R_16BF, workspace_limit=1 << 30)
            ws = torch.empty(max(p.required_workspace, 256), dtype=torch.uint8, device="cuda")
            fn = lambda: p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace, stream=stream)   # noqa: E731
            for _ in range(200):
                fn()
            ms = min(timed_batch(torch, fn, reps=200), timed_batch(torch, fn, reps=200))
            flop = 2.0 * M * N * K
            tf = flop / (ms * 1e-3) / 1e12
            d = p.describe()
            out.append({"workload": "contraction bf16 C[m,n]=A[m,k]B[k,n] M=%d N=%d K=%d, U(-1,1) data (mid-size class)" % (M, N, K), "dtype": "bf16",
                        "value": tf * 1e3, "unit": "GFLOP/s", "ms_per_call": ms, "kernel": d["kname"], "splitK": d["splitK"],
                        "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS_BF16_MFMA, "unit": "TFLOP/s", "frac": tf / PEAK_TFLOPS_BF16_MFMA,
                                     "algorithmic_flop": flop, "algorithmic_bytes": 2.0 * (M * K + K * N + M * N),
                                     "note": "bound in practice by the LDS-DMA staging stream of the tile and the fixed cost of a launch (DESIGN.md section 3)"}})
            p.destroy()
            del A, B, D, ws
        except Exception as ex:   # noqa: BLE001
            out.append({"workload": "contraction bf16 %dx%dx%d" % (M, N, K), "error": "%s: %s" % (type(ex).__name__, ex)})
    # ---- 2048^3 fp32 permute abc->cab and reduce abc->ac (configs[2]); tensors generated on the device ----------------
    try:
        n = 2048
        numel = n ** 3
        free, _ = torch.cuda.mem_get_info()
        if free < 2 * numel * 4 + (2 << 30):
            raise RuntimeError("not enough free HBM for two 32-GiB tensors (%d bytes free)" % free)
        A = torch.empty(numel, dtype=torch.float32, device="cuda")
        chunk = 1 << 28
        for s in range(0, numel, chunk):   # counter-based fill (fixed seed), in chunks
            idx = torch.arange(s, min(numel, s + chunk), device="cuda", dtype=torch.int64)
            A[s:s + idx.numel()] = ((idx * 2654435761 + 1234) % 16777216).to(torch.float32) / 16777216.0
            del idx
        D = torch.empty(numel, dtype=torch.float32, device="cuda")
        p = ops.permutation_plan(h, [n, n, n], "abc", [n, n, n], "cab")
        ms = timed_batch(torch, lambda: p.permute(1.0, A.data_ptr(), D.data_ptr(), stream), reps=3, warm=1)
        gbs = 2.0 * numel * 4 / (ms * 1e-3) / 1e9          # elementwise_permute.cu:208
        out.append({"workload": "elementwise_permute.cu A[a,b,c]->C[c,a,b] fp32 2048^3 (BASELINE configs[2])", "dtype": "f32", "value": gbs,
                    "unit": "GB/s", "ms_per_call": ms,
                    "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBPS,
                                 "algorithmic_bytes": 2.0 * numel * 4}})
        p.destroy()
        del D
        R = torch.zeros(n * n, dtype=torch.float32, device="cuda")
        p = ops.reduction_plan(h, [n, n, n], "abc", [n, n], "ac", workspace_limit=1 << 30)
        ws = torch.empty(max(p.required_workspace, 256), dtype=torch.uint8, device="cuda")
        ms = timed_batch(torch, lambda: p.reduce(1.1, A.data_ptr(), 0.0, R.data_ptr(), R.data_ptr(), ws.data_ptr(), p.required_workspace, stream),
                         reps=3, warm=1)
        gbs = (numel + n * n) * 4.0 / (ms * 1e-3) / 1e9    # reduction.cu:229-231
        out.append({"workload": "reduction.cu C[a,c]=1.1*sum_b A[a,b,c] fp32 2048^3 (BASELINE configs[2])", "dtype": "f32", "value": gbs,
                    "unit": "GB/s", "ms_per_call": ms,
                    "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBPS,
                                 "algorithmic_bytes": (numel + n * n) * 4.0}})
        p.destroy()
        del A, R, ws
    except Exception as ex:   # noqa: BLE001
        out.append({"workload": "permute / reduce 2048^3", "error": "%s: %s" % (type(ex).__name__, ex)})
    torch.cuda.empty_cache()
    return out


def secondary_general_family(torch, ct, ops, h):
    """The general MFMA family (csrc/kernels/gett_gen.inc) on the shapes the round-3 review names: fp64 4096^3 against the nominal
    v_mfma_f64_16x16x4_f64 rate (256 CU x 4 SIMD x 2048 flop / 64 clk x 2.4 GHz = 78.6 TFLOP/s), complex64 2048^3 against the fp32
    MFMA peak (8 real flop per complex multiply-add), and the reference's own fp16 regression case 'mlik,lkjm->lij' at
    (20,50,50,50) (einsum_test.py:98-107), which ran on the scalar FMA kernel before this family existed."""
    out = []

    def gemm(label, n, tdt, cdt, flop_per_mac, peak, reps):
        try:
            mk = (lambda: (torch.rand((n, n), device="cuda", dtype=torch.float64) * 2 - 1).to(tdt)) if not tdt.is_complex else \
                (lambda: torch.complex(torch.rand((n, n), device="cuda") * 2 - 1, torch.rand((n, n), device="cuda") * 2 - 1).to(tdt))
            A, B = mk(), mk()
            D = torch.empty((n, n), device="cuda", dtype=tdt)
            p = ops.contraction_plan(h, [n, n], "km", [n, n], "kn", [n, n], "mn", dtype=cdt)
            fn = lambda: p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())   # noqa: E731
            fn()
            torch.cuda.synchronize()
            t_end = time.perf_counter() + 0.04           # clock ramp under this kernel's own load (as the bf16 line: ~40 ms), untimed
            while time.perf_counter() < t_end:
                for _ in range(4):
                    fn()
                torch.cuda.synchronize()
            ms = min(timed_batch(torch, fn, reps=reps), timed_batch(torch, fn, reps=reps))
            d = p.describe()
            tf = flop_per_mac * n ** 3 / (ms * 1e-3) / 1e12
            out.append({"workload": label, "dtype": str(tdt).replace("torch.", ""), "value": tf * 1e3, "unit": "GFLOP/s", "ms_per_call": ms,
                        "kernel": "%s<%dx%dx%d,V%s>" % (d["kname"], d["bm"], d["bn"], d["bk"], d.get("vec")),
                        "roofline": {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                                     "algorithmic_flop": flop_per_mac * n ** 3}})
            p.destroy()
        except Exception as ex:   # noqa: BLE001
            out.append({"workload": label, "error": "%s: %s" % (type(ex).__name__, ex)})

    gemm("contraction fp64 C[m,n]=A[k,m]B[k,n] M=N=K=4096 (einsum.cu:36-41 with double), v_mfma_f64_16x16x4_f64", 4096, torch.float64, ct.R_64F,
         2.0, PEAK_TFLOPS_F64_MFMA, 10)
    gemm("contraction complex64 M=N=K=2048 (python/einsum.h:51-63), four real fp32 MFMAs per complex product", 2048, torch.complex64, ct.C_32F,
         8.0, PEAK_TFLOPS_F32_MFMA, 20)
    try:
        from cudalibrarysamples_amd import torch_einsum
        eq = "mlik,lkjm->lij"
        a = torch.randn(20, 50, 50, 50, device="cuda").half()
        b = torch.randn(50, 50, 50, 20, device="cuda").half()
        res = torch_einsum.einsum(eq, a, b)
        pl = torch_einsum._plans[(eq, tuple(a.shape), tuple(b.shape), a.dtype, False, False)]
        ws = torch_einsum._get_workspace(a.device, pl.required_workspace)
        ms = timed_batch(torch, lambda: pl.execute(a, b, res, ws), reps=200)
        ref = torch.einsum(eq, a.double(), b.double())
        d = pl.describe()
        out.append({"workload": "einsum 'mlik,lkjm->lij' fp16 at (20,50,50,50) x (50,50,50,20) — the reference's own test 7 (einsum_test.py:98-107)",
                    "dtype": "f16", "value": 2.0 * 50 * 50 * 50 * 1000 / (ms * 1e-3) / 1e9, "unit": "GFLOP/s", "us_per_call": ms * 1e3,
                    "kernel": "%s<%dx%dx%d,V%s>" % (d["kname"], d["bm"], d["bn"], d["bk"], d.get("vec")),
                    "before_this_family_us_per_call": 70.9, "before_kernel": "gett_simple_kernel (profiles/r04a_bench_gen_before.jsonl, CUTENSOR_AMD_GEN=0)",
                    "max_err_over_max_ref": float((res.double() - ref).abs().max() / ref.abs().max())})
    except Exception as ex:   # noqa: BLE001
        out.append({"workload": "einsum mlik,lkjm->lij fp16", "error": "%s: %s" % (type(ex).__name__, ex)})
    torch.cuda.empty_cache()
    return out


def secondary_round6(torch, ct, ops, h, stream, with_counters):
    """Round 6: the off-happy-path shapes the round-5 review asked to put against a roofline — contractions whose operands have no
    16-byte lanes (bf16 4100^3: every row at 8 (mod 16) bytes, 17 x 17 tiles of 256 -> interior + strip plan; fp32 4098^3: the RAG twin
    of the ring kernel) and the tiled bandwidth kernels of 8-byte elements (complex64 permutation / reduction at 1024^3)."""
    out = []
    for (label, E, tdt, cdt, peak, es) in (("contraction bf16 C[m,n]=A[m,k]B[k,n] M=N=K=4100 (no 16-byte lanes: rows at 8 mod 16 bytes), U(-1,1) data",
                                            4100, torch.bfloat16, ct.R_16BF, PEAK_TFLOPS_BF16_MFMA, 2),
                                           ("contraction fp32 C[m,n]=A[m,k]B[k,n] M=N=K=4098 (no 16-byte lanes, ragged K), U(0,1) data",
                                            4098, torch.float32, ct.R_32F, PEAK_TFLOPS_F32_MFMA, 4)):
        try:
            g = torch.Generator(device="cuda")
            g.manual_seed(3)
            A = torch.rand((E, E), generator=g, device="cuda")
            B = torch.rand((E, E), generator=g, device="cuda")
            if tdt != torch.float32:
                A, B = (A * 2 - 1).to(tdt), (B * 2 - 1).to(tdt)
            D = torch.empty((E, E), device="cuda", dtype=tdt)
            p = ops.contraction_plan(h, [E, E], "mk", [E, E], "kn", [E, E], "mn", dtype=cdt, workspace_limit=1 << 30)
            ws = torch.empty(max(p.required_workspace, 256), dtype=torch.uint8, device="cuda")
            fn = lambda: p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace, stream=stream)   # noqa: E731
            for _ in range(60):
                fn()
            ms = min(timed_batch(torch, fn, reps=30), timed_batch(torch, fn, reps=30))
            flop = 2.0 * float(E) ** 3
            tf = flop / (ms * 1e-3) / 1e12
            d = p.describe()
            line = {"workload": label, "dtype": "bf16" if es == 2 else "f32", "value": tf * 1e3, "unit": "GFLOP/s", "ms_per_call": ms,
                    "kernel": d["kname"], "rag": d.get("rag"), "strip_plan": d.get("strips"), "splitK": d["splitK"],
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                                 "algorithmic_flop": flop, "algorithmic_bytes": 3.0 * es * E * E}}
            p.destroy()
            del A, B, D, ws
            out.append(line)
        except Exception as ex:   # noqa: BLE001
            out.append({"workload": label, "error": "%s: %s" % (type(ex).__name__, ex)})
    # ---- bf16 8192^2 x 2048 with beta != 0, in place (D = A B + 0.5 D: the form contraction.cu:43 states): the persistent kernel's
    #      streaming epilogue with C (round 6); the kernel that actually ran is read back (cutensorContract picks the twin at the call)
    try:
        M = N = 8192
        K = 2048
        g = torch.Generator(device="cuda")
        g.manual_seed(4)
        A = (torch.rand((K, M), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
        B = (torch.rand((N, K), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
        D = torch.zeros((N, M), device="cuda", dtype=torch.bfloat16)
        p = ops.contraction_plan(h, [M, K], "mk", [K, N], "kn", [M, N], "mn", dtype=ct.R_16BF, workspace_limit=1 << 30)
        ws = torch.empty(max(p.required_workspace, 256), dtype=torch.uint8, device="cuda")
        rates = {}
        for beta in (0.5, 0.0):
            fn = lambda: p.contract(1.0, A.data_ptr(), B.data_ptr(), beta, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace, stream=stream)   # noqa: E731
            for _ in range(40):
                fn()
            ms = min(timed_batch(torch, fn, reps=30), timed_batch(torch, fn, reps=30))
            rates[beta] = (ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12, ct.last_h16_kernel())
        ms, tf, kidx = rates[0.5]
        out.append({"workload": "contraction bf16 D[m,n]=A[m,k]B[k,n]+0.5*D M=N=8192 K=2048 (beta != 0, in place), U(-1,1) data", "dtype": "bf16",
                    "value": tf * 1e3, "unit": "GFLOP/s", "ms_per_call": ms, "kernel": p.describe()["kname"], "launched_table_entry": kidx,
                    "launched": "persistent (88..95)" if 88 <= kidx < 96 else "one-tile twin (48..55)",
                    "same_plan_beta_0_gflops": rates[0.0][1] * 1e3,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS_BF16_MFMA, "unit": "TFLOP/s", "frac": tf / PEAK_TFLOPS_BF16_MFMA,
                                 "algorithmic_flop": 2.0 * M * N * K, "algorithmic_bytes": 2.0 * (M * K + K * N + 2 * M * N)}})
        p.destroy()
        del A, B, D, ws
    except Exception as ex:   # noqa: BLE001
        out.append({"workload": "contraction bf16 8192^2 x 2048 beta != 0", "error": "%s: %s" % (type(ex).__name__, ex)})
    # ---- a short contracted range with batch modes: attention scores 'bhqd,bhkd->bhqk' (8, 8, 2048, 2048, 128) in bf16 — 64 batches of
    #      2048 x 2048 x 128, 537 MB of output: the persistent kernel streams two-K-tile tiles across batch boundaries (round 6)
    try:
        from cudalibrarysamples_amd import torch_einsum
        a = (torch.rand((8, 8, 2048, 128), device="cuda") * 2 - 1).to(torch.bfloat16)
        b = (torch.rand((8, 8, 2048, 128), device="cuda") * 2 - 1).to(torch.bfloat16)
        eq = "bhqd,bhkd->bhqk"
        res = torch_einsum.einsum(eq, a, b)
        pl = torch_einsum._plans[(eq, tuple(a.shape), tuple(b.shape), a.dtype, False, False)]
        wsp = torch_einsum._get_workspace(a.device, pl.required_workspace)
        fn = lambda: pl.execute(a, b, res, wsp)   # noqa: E731
        for _ in range(20):
            fn()
        ms = min(timed_batch(torch, fn, reps=20), timed_batch(torch, fn, reps=20))
        flop = 2.0 * 64 * 2048 * 2048 * 128
        nbytes = 2.0 * (a.numel() + b.numel() + res.numel())
        out.append({"workload": "einsum 'bhqd,bhkd->bhqk' bf16 (8, 8, 2048, 2048, 128): attention scores, K = 128, 64 batches, 537 MB of output", "dtype": "bf16",
                    "value": flop / (ms * 1e-3) / 1e9, "unit": "GFLOP/s", "us_per_call": ms * 1e3, "kernel": pl.describe()["kname"],
                    "roofline": {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                 "frac": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, "algorithmic_bytes": nbytes, "algorithmic_flop": flop}})
        del a, b, res
    except Exception as ex:   # noqa: BLE001
        out.append({"workload": "einsum bhqd,bhkd->bhqk bf16", "error": "%s: %s" % (type(ex).__name__, ex)})
    # ---- the same shape in fp32, the headline's type: four K-tiles per tile, 1.07 GB of output — the launch the row epilogue of the fp32
    #      kernels is for (round 6: D leaves as whole rows through a per-wave LDS image, gett_store_tile_f32_rows)
    try:
        from cudalibrarysamples_amd import torch_einsum
        a = torch.rand((8, 8, 2048, 128), device="cuda") * 2 - 1
        b = torch.rand((8, 8, 2048, 128), device="cuda") * 2 - 1
        eq = "bhqd,bhkd->bhqk"
        res = torch_einsum.einsum(eq, a, b)
        pl = torch_einsum._plans[(eq, tuple(a.shape), tuple(b.shape), a.dtype, False, False)]
        wsp = torch_einsum._get_workspace(a.device, pl.required_workspace)
        fn = lambda: pl.execute(a, b, res, wsp)   # noqa: E731
        for _ in range(10):
            fn()
        ms = min(timed_batch(torch, fn, reps=10), timed_batch(torch, fn, reps=10))
        flop = 2.0 * 64 * 2048 * 2048 * 128
        d = pl.describe()
        out.append({"workload": "einsum 'bhqd,bhkd->bhqk' fp32 (8, 8, 2048, 2048, 128): attention scores, K = 128, 64 batches, 1.07 GB of output", "dtype": "f32",
                    "value": flop / (ms * 1e-3) / 1e9, "unit": "GFLOP/s", "us_per_call": ms * 1e3, "kernel": d["kname"], "tile": [d["bm"], d["bn"], d["bk"]],
                    "roofline": {"bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12, "peak": PEAK_TFLOPS_F32_MFMA, "unit": "TFLOP/s",
                                 "frac": flop / (ms * 1e-3) / 1e12 / PEAK_TFLOPS_F32_MFMA, "algorithmic_flop": flop,
                                 "algorithmic_bytes": 4.0 * (a.numel() + b.numel() + res.numel())}})
        del a, b, res
    except Exception as ex:   # noqa: BLE001
        out.append({"workload": "einsum bhqd,bhkd->bhqk fp32", "error": "%s: %s" % (type(ex).__name__, ex)})
    # ---- complex64 1024^3: permutation abc->cab and reduction abc->ac on the tiled kernels of 8-byte elements ----------------
    try:
        n = 1024
        numel = n ** 3
        re = torch.rand(numel, device="cuda")
        A = torch.complex(re, 0.5 - re)
        del re
        D = torch.empty(numel, dtype=torch.complex64, device="cuda")
        p = ops.permutation_plan(h, [n, n, n], "abc", [n, n, n], "cab", dtype=ct.C_32F)
        ms = timed_batch(torch, lambda: p.permute(1.0, A.data_ptr(), D.data_ptr(), stream), reps=5, warm=2)
        gbs = 2.0 * numel * 8 / (ms * 1e-3) / 1e9
        out.append({"workload": "cutensorPermute A[a,b,c]->C[c,a,b] complex64 1024^3 (tiled kernel of 8-byte elements; einsum.cc:83 dispatches the type)",
                    "dtype": "c64", "value": gbs, "unit": "GB/s", "ms_per_call": ms, "variant": p.describe().get("variant"),
                    "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBPS,
                                 "algorithmic_bytes": 2.0 * numel * 8}})
        p.destroy()
        del D
        R = torch.zeros(n * n, dtype=torch.complex64, device="cuda")
        p = ops.reduction_plan(h, [n, n, n], "abc", [n, n], "ac", dtype=ct.C_32F, workspace_limit=1 << 30)
        ws = torch.empty(max(p.required_workspace, 256), dtype=torch.uint8, device="cuda")
        ms = timed_batch(torch, lambda: p.reduce(1.1, A.data_ptr(), 0.0, R.data_ptr(), R.data_ptr(), ws.data_ptr(), p.required_workspace, stream), reps=5, warm=2)
        gbs = (numel + n * n) * 8.0 / (ms * 1e-3) / 1e9
        out.append({"workload": "cutensorReduce C[a,c]=1.1*sum_b A[a,b,c] complex64 1024^3 (tiled kernel of 8-byte elements)", "dtype": "c64", "value": gbs,
                    "unit": "GB/s", "ms_per_call": ms, "variant": p.describe().get("variant"),
                    "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBPS,
                                 "algorithmic_bytes": (numel + n * n) * 8.0}})
        p.destroy()
        del A, R, ws
    except Exception as ex:   # noqa: BLE001
        out.append({"workload": "permute / reduce complex64 1024^3", "error": "%s: %s" % (type(ex).__name__, ex)})
    # ---- cutensorPermute off the 16-byte lanes (round 6, end): a transposition with odd extents, and a block permutation ----------------
    for (label, eA, mA, eB, mB, tdt, cdt, es) in (
            ("cutensorPermute A[a,b]->C[b,a] fp32 4097 x 4099 (odd extents: element-wise 64 x 64 LDS transposer)", [4097, 4099], "ab", [4099, 4097], "ba", torch.float32, ct.R_32F, 4),
            ("cutensorPermute A[d,c,b,a]->C[b,c,d,a] bf16 (40, 16, 8, 8192) (leading modes the same packed set: block permutation through LDS)",
             [40, 16, 8, 8192], "dcba", [8, 16, 40, 8192], "bcda", torch.bfloat16, ct.R_16BF, 2)):
        try:
            numel = 1
            for e_ in eA:
                numel *= e_
            A = (torch.rand(numel, device="cuda") * 2 - 1).to(tdt)
            D = torch.empty(numel, device="cuda", dtype=tdt)
            p = ops.permutation_plan(h, eA, mA, eB, mB, dtype=cdt)
            ms = timed_batch(torch, lambda: p.permute(1.0, A.data_ptr(), D.data_ptr(), stream), reps=20, warm=5)
            gbs = 2.0 * numel * es / (ms * 1e-3) / 1e9
            out.append({"workload": label, "dtype": "f32" if es == 4 else "bf16", "value": gbs, "unit": "GB/s", "us_per_call": ms * 1e3, "variant": p.describe().get("variant"),
                        "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBPS, "algorithmic_bytes": 2.0 * numel * es}})
            p.destroy()
            del A, D
        except Exception as ex:   # noqa: BLE001
            out.append({"workload": label, "error": "%s: %s" % (type(ex).__name__, ex)})
    # ---- contractions off the aligned path (round 6, end): the sweep mask and an operand copied first --------------------------------
    for (label, ext, mA, mB, mC) in (
            ("contraction bf16 'abcd,dcbe->ae' a=e=2048 b=c=8 d=96 (three contracted modes, the fastest one without whole K-tiles: sweep mask), U(-1,1) data",
             dict(a=2048, b=8, c=8, d=96, e=2048), "dcba", "ebcd", "ea"),
            ("contraction bf16 'ijk,lkj->il' i=l=4096 j=16 k=72 (A contiguous in k, B in j: one operand copied into a packed temporary first), U(-1,1) data",
             dict(i=4096, l=4096, j=16, k=72), "kji", "jkl", "li")):
        try:
            g = torch.Generator(device="cuda")
            g.manual_seed(4)
            eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
            A = (torch.rand(eA[::-1], generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
            B = (torch.rand(eB[::-1], generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
            D = torch.empty(eC[::-1], device="cuda", dtype=torch.bfloat16)
            p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, dtype=ct.R_16BF, workspace_limit=1 << 30)
            ws = torch.empty(max(p.required_workspace, 256), dtype=torch.uint8, device="cuda")
            fn = lambda: p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace, stream=stream)   # noqa: E731
            for _ in range(40):
                fn()
            ms = min(timed_batch(torch, fn, reps=30), timed_batch(torch, fn, reps=30))
            flop = 2.0
            for c in set(mA + mB):
                flop *= ext[c]
            tf = flop / (ms * 1e-3) / 1e12
            d = p.describe()
            out.append({"workload": label, "dtype": "bf16", "value": tf * 1e3, "unit": "GFLOP/s", "us_per_call": ms * 1e3, "kernel": d["kname"],
                        "sweep_mask": bool(d.get("rag") and len(d.get("Kdigits", [])) > 1), "copied_first": [bool(d.get("repack_A")), bool(d.get("repack_B"))],
                        "workspace_bytes": p.required_workspace,
                        "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS_BF16_MFMA, "unit": "TFLOP/s", "frac": tf / PEAK_TFLOPS_BF16_MFMA,
                                     "algorithmic_flop": flop, "algorithmic_bytes": 2.0 * (A.numel() + B.numel() + D.numel())}})
            p.destroy()
            del A, B, D, ws
        except Exception as ex:   # noqa: BLE001
            out.append({"workload": label, "error": "%s: %s" % (type(ex).__name__, ex)})
    torch.cuda.empty_cache()
    return out


def add_mfma_counters(secondary):
    """mfma_busy_pct / sustained_clock_ghz on the bf16 secondary lines (round-5 review: the mid-size lines move +-9 % between boxes and
    nothing on the line says why): two live rocprofv3 passes (trace, PMC — never combined) over a child that runs the line's shape."""
    tool = os.path.join(ROOT, "tools", "bench_unaligned.py")
    for line in secondary:
        w = line.get("workload", "")
        if "roofline" not in line or not w.startswith("contraction bf16") or "8192" in w:
            continue
        import re as _re
        m = _re.search(r"M=(\d+) N=(\d+) K=(\d+)", w) or _re.search(r"M=N=K=(\d+)", w)
        if not m:
            continue
        dims = [int(x) for x in m.groups()]
        if len(dims) == 1:
            dims = dims * 3
        try:
            prof = live_profile_of([sys.executable, tool, "--shapes", "%d,%d,%d" % tuple(dims), "--layouts", "mk,kn", "--reps", "100", "--warm", "100"],
                                   "%gett_h16w4%", timeout_s=120)
        except Exception:   # noqa: BLE001
            prof = None
        if prof:
            line["roofline"]["counters"] = {k: prof.get(k) for k in ("kernel", "launches", "kernel_avg_us", "kernel_min_us", "mfma_busy_pct", "sustained_clock_ghz",
                                                                       "kernel_avg_us_under_pmc")}


def add_secondary_traffic(secondary):
    """roofline.traffic of the bf16 / permute / reduce secondary lines, measured live like the headline's: two rocprofv3 --pmc passes
    over a short child that runs only that workload (tools/bench_h16.py, tools/bench_bandwidth.py --only ...)."""
    jobs = [("bf16 C[m,n]", [sys.executable, os.path.join(ROOT, "tools", "bench_h16.py"), "--reps", "5"], "%ctamd%gett_h16%"),
            ("elementwise_permute.cu", [sys.executable, os.path.join(ROOT, "tools", "bench_bandwidth.py"), "--n", "2048", "--reps", "1", "--only", "permute:cab"],
             "%ctamd%ew_transpose%"),
            ("reduction.cu", [sys.executable, os.path.join(ROOT, "tools", "bench_bandwidth.py"), "--n", "2048", "--reps", "1", "--only", "reduce:ac"],
             "%ctamd%reduce_col%")]
    for key, cmd, like in jobs:
        line = next((x for x in secondary if key in x.get("workload", "") and "roofline" in x), None)
        if line is None:
            continue
        try:
            t = live_pmc_traffic_of(cmd, like)
        except Exception:   # noqa: BLE001
            t = None
        r = line["roofline"]
        if t:
            r["traffic"] = t["hbm_bytes_per_launch"]
            r["traffic_over_algorithmic"] = t["hbm_bytes_per_launch"] / r["algorithmic_bytes"] if r.get("algorithmic_bytes") else None
            r["traffic_raw"] = {"FETCH_SIZE_KiB": t["raw_FETCH_SIZE_KiB"], "WRITE_SIZE_KiB": t["raw_WRITE_SIZE_KiB"], "correction": t["correction"]}
            r["traffic_source"] = "live: two rocprofv3 --pmc passes over %d launches of %s in a child of this run" % (t["launches"], t["kernel"][:60])
        else:
            r["traffic"] = None
        if key == "bf16 C[m,n]":    # north_star: rocprof MFMA-busy % on the line; and the kernel's own average launch duration
            try:
                prof = live_profile_of([sys.executable, os.path.join(ROOT, "tools", "bench_h16.py"), "--reps", "30"], "%ctamd%gett_h16%")
            except Exception:   # noqa: BLE001
                prof = None
            if prof:
                for k in ("mfma_busy_pct", "sustained_clock_ghz", "mfma_busy_pct_of_kernel_time_at_nominal_clock"):
                    r[k] = prof.get(k)
                r["live_trace"] = {k: prof.get(k) for k in ("kernel", "launches", "kernel_avg_us", "kernel_min_us")}
                r["pmc_live"] = {k: prof.get(k) for k in ("kernel_avg_us_under_pmc", "wait_any_pct_of_wave_cycles", "pmc_raw", "pmc_formula")}
                if prof.get("kernel_avg_us") and r.get("algorithmic_flop"):
                    r["frac_from_live_trace_avg"] = r["algorithmic_flop"] / (prof["kernel_avg_us"] * 1e-6) / 1e12 / r["peak"]


# ---------------------------------------------------------------------------------------------------------
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
    ap.add_argument("--no-secondary", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two live rocprofv3 --pmc passes for roofline.traffic (the committed "
                                                          "figure of the latest profile is reported instead)")
    ap.add_argument("--no-cold", action="store_true", help="skip the HBM-cold leg (profiling runs of the warm headline: tools/gpu_profile.sh; the cold "
                                                           "leg has its own trace, tools/gpu_profile_all.sh 1b)")
    ap.add_argument("--cold-only", action="store_true",
                    help="profiling runs: only the HBM-cold variant of the headline (four rotating operand pairs, >= 500 steps), one small JSON line")
    ap.add_argument("--mg-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--mg-timeout", type=int, default=300, help="seconds per attempt of the multi-device child process")
    ap.add_argument("--einsum-only", action="store_true", help=argparse.SUPPRESS)   # ranks spawned by a plain `--gpus N` launch
    ap.add_argument("--mg-virtual", action="store_true",
                    help="self-test of the N > 1 path on a box with fewer GPUs: the N devices are N logical devices on GPU 0 and the "
                         "shapes are shrunk; the line then says n_gpus = 1 and the numbers are not scaling results")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if args.mg_child:
        return mg_child_main(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    visible = torch.cuda.device_count()
    requested = max(args.gpus, world)
    torch.cuda.set_device(local_rank % visible)
    cpu_group = gpu_group = None
    rccl_error = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC
        # coordination (barriers, the max over ranks) runs on gloo: waits that must not occupy a GPU, and a line is printed
        # even if the RCCL communicator of the secondary one-process-per-GPU measurement cannot be built
        dist.init_process_group("gloo", rank=rank, world_size=world)
        cpu_group = dist.group.WORLD
        try:
            gpu_group = dist.new_group(backend="nccl")
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe, group=gpu_group)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError("RCCL all-reduce probe returned %r" % probe.item())
        except Exception as ex:   # noqa: BLE001
            rccl_error = "%s: %s" % (type(ex).__name__, ex)
            gpu_group = None
        ok = torch.tensor([0 if gpu_group is None else 1])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=cpu_group)      # all ranks or none
        if int(ok.item()) == 0:
            gpu_group = None

    from cudalibrarysamples_amd import cutensor as ct, ops, sharding

    # ---- cuTENSORMg over the requested devices (N > 1): child process of rank 0 --------------------------------------
    mg = None
    mg_devices = requested if args.mg_virtual else min(requested, visible)
    if requested > 1 and not args.einsum_only:
        if rank == 0 and mg_devices > 1:
            mg = run_mg_child(mg_devices, args.steps, args.warmup, args.mg_timeout, virtual=args.mg_virtual)
            if "timed out" in str(mg.get("error", "")) and os.environ.get("CUTENSORMG_AMD_TRANSPORT") != "peer":
                # a hang (killed by the timeout) in the RCCL path: one more child with peer copies instead
                first = mg["error"]
                os.environ["CUTENSORMG_AMD_TRANSPORT"] = "peer"
                mg = run_mg_child(mg_devices, args.steps, args.warmup, args.mg_timeout, virtual=args.mg_virtual)
                mg["first_attempt_error"] = first
        if world > 1:
            dist.barrier(group=cpu_group)   # CPU-side wait: the other ranks' GPUs stay idle during the measurement

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

    def contract(x, y, out):
        plan.contract(1.0, x.data_ptr(), y.data_ptr(), 0.0, out.data_ptr(), out.data_ptr(), ws.data_ptr(), plan.required_workspace, stream)

    def step(i):
        out = outs[i % nbuf]
        contract(a, b, out)
        if gpu_group is not None:
            # fold the K-shards: RCCL all-reduce of the 36 KB result, overlapped with the next step
            if len(pending) >= nbuf - 1:
                pending.pop(0).wait()
            pending.append(sharding.fold_partials(out, dist, async_op=True, group=gpu_group))

    def fence():
        while pending:
            pending.pop(0).wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(group=cpu_group)
        torch.cuda.synchronize()

    if args.cold_only:
        # rocprofv3 --pmc passes of the cold rotation alone (tools/gpu_profile_all.sh): every launch of this process reads operands
        # that are not in the Infinity Cache
        pairs = [(a, b)] + [(torch.rand(a.shape, generator=g, device="cuda"), torch.rand(b.shape, generator=g, device="cuda")) for _ in range(3)]
        n_cold = max(args.steps, 500)
        for i in range(100):
            contract(*pairs[i % 4], outs[i % nbuf])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_cold):
            contract(*pairs[i % 4], outs[i % nbuf])
        torch.cuda.synchronize()
        cold_s = (time.perf_counter() - t1) / n_cold
        print(json.dumps({"cold_only": True, "steps": n_cold, "ms_per_step": cold_s * 1e3, "value": FLOP / cold_s / 1e9, "unit": "GFLOP/s",
                          "frac_of_nominal_f32_mfma_peak": FLOP / cold_s / 1e12 / PEAK_TFLOPS_F32_MFMA, "operand_bytes_rotated": 4 * BYTES}))
        return

    # ---- device clock ramp (untimed, not a step count: wall-clock bounded) -----------------------------
    # A cold MI355X runs the first ~12-20 ms of a kernel stream ~10 % slower than its steady state (GETT kernel
    # 43.5 us -> 38.9 us on the same inputs); production streams live in the steady state, so the bench reaches it
    # before the W warmup steps.  Same call sequence as a step, result discarded.
    burn_steps = 0
    if args.burn_in_ms > 0:
        t_end = time.perf_counter() + args.burn_in_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(50):
                contract(a, b, outs[0])
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
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=cpu_group)
        elapsed = float(t.item())
    einsum_ms = elapsed / args.steps * 1e3
    einsum_value = world * FLOP / (elapsed / args.steps) / 1e9
    einsum_1gpu_value = einsum_value / world   # per-GPU rate of the headline at this rank count (= the headline itself at N = 1)

    roof = cpu = None
    secondary = []
    cold = None
    sample_protocol = None
    if rank == 0:
        # ---- roofline of the dominant kernel: one HIP event pair on the launch stream -----------------------------------
        # GETT kernel alone, in the same steady state as the timed loop: the fold is switched off (per-handle diagnostic),
        # `n` launches go out back to back (rocprofv3 shows < 0.05 us between them) and ONE HIP event pair on the launch
        # stream brackets them.  Per-launch event pairs would put a ~6 us idle gap after every kernel, and the gapped
        # stream runs at a different clock than the timed loop (42.6 vs 38.9 us).
        n = max(args.steps, 200)
        ct.lib.ctamdSetSplitKFold(h.h, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(50):
            contract(a, b, outs[0])
        e0.record()
        for i in range(n):
            contract(a, b, outs[0])
        e1.record()
        torch.cuda.synchronize()
        ct.lib.ctamdSetSplitKFold(h.h, 1)
        batch_ms = e0.elapsed_time(e1) / n
        # per-launch event pairs (the gapped stream), kept for reference
        ct.lib.ctamdProfileBegin(h.h)
        for i in range(200):
            contract(a, b, outs[0])
        torch.cuda.synchronize()
        mean_ms, min_ms = ctypes.c_float(0), ctypes.c_float(0)
        ct.lib.ctamdProfileEnd(h.h, ctypes.byref(mean_ms), ctypes.byref(min_ms))
        gapped_mean_us, gapped_min_us = mean_ms.value * 1e3, min_ms.value * 1e3
        prop = torch.cuda.get_device_properties(0)
        cus = prop.multi_processor_count
        clock_ghz = getattr(prop, "clock_rate", 2400000) / 1e6
        peak = cus * 256 * clock_ghz / 1e3      # TFLOP/s from the device's own CU count and max clock
        achieved = FLOP / (batch_ms * 1e-3) / 1e12
        # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE collected by separate `rocprofv3 --pmc` runs of
        # this same command, tools/gpu_profile.sh; corrected per MI355X_MICROARCH.md and committed under profiles/).
        # Counters cannot be read from inside the timed process, so the committed per-launch figure of the same kernel is
        # reported, else null.
        traffic = None
        traffic_live = None
        traffic_file = newest_traffic_file()
        try:
            with open(traffic_file) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        except (OSError, ValueError, TypeError):
            traffic = None
        traffic_committed = traffic
        if world == 1 and not args.no_pmc and not args.no_secondary and not args.einsum_only:
            traffic_live = live_pmc_traffic()     # measured in this run; the committed figure stays beside it
            if traffic_live:
                traffic = traffic_live["hbm_bytes_per_launch"]
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak if peak else None, "traffic": traffic,
                "traffic_source": ("live: two rocprofv3 --pmc passes (FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024) over %d launches of a headline-only child of this run"
                                   % traffic_live["launches"]) if traffic_live else
                                  (("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command, bytes per launch)" % os.path.relpath(traffic_file, ROOT)) if traffic else None),
                "traffic_committed": traffic_committed, "traffic_over_algorithmic": (traffic / BYTES) if traffic else None,
                "kernel": "%s<%dx%dx%d,w%dx%dx%d> (table index %d)" % (desc.get("kname", "gett_f32_kernel"), desc["bm"], desc["bn"], desc["bk"],
                                                                       desc["wm"], desc["wn"], desc["wk"], desc.get("kernel", -1)),
                "launches": n, "mean_us": batch_ms * 1e3,
                "timing": "one HIP event pair around %d back-to-back launches of the GETT kernel (fold off)" % n,
                "gapped_mean_us": gapped_mean_us, "gapped_min_us": gapped_min_us,
                "algorithmic_flop_per_launch": FLOP, "algorithmic_bytes_per_launch": BYTES,
                "hbm_equiv_TBps": BYTES / (batch_ms * 1e-3) / 1e12,
                "cus": cus, "clock_ghz": clock_ghz, "nominal_peak": PEAK_TFLOPS_F32_MFMA,
                # `frac` is the GETT kernel alone; the whole step (GETT kernel + fold of the split-K partials) is `value`:
                "whole_step_frac": (FLOP / (einsum_ms * 1e-3) / 1e12) / peak if peak else None,
                "traffic_raw": ({"FETCH_SIZE_KiB": traffic_live["raw_FETCH_SIZE_KiB"], "WRITE_SIZE_KiB": traffic_live["raw_WRITE_SIZE_KiB"],
                                 "correction": "FETCH_SIZE x 2 (wide coalesced reads tallied at half their bytes on gfx950), WRITE_SIZE x 1"}
                                if traffic_live else None)}
        # ---- roofline.frac / achieved = algorithmic flop / the kernel's AVERAGE launch duration in a live `rocprofv3 --kernel-trace
        #      --stats` pass over a headline-only child of this run (the committed trace of the round must agree); the event-pair figure of
        #      the back-to-back fold-off loop above stays beside it as frac_event_pair.  north_star: MFMA-busy % and the clock, live.
        roof["frac_event_pair"], roof["achieved_event_pair"] = roof["frac"], roof["achieved"]
        roof["frac_source"] = "event pair (no live rocprofv3 pass in this run)"
        if world == 1 and not args.no_pmc and not args.no_secondary and not args.einsum_only:
            prof = live_profile_of([sys.executable, os.path.abspath(__file__), "--steps", "200", "--warmup", "20", "--no-cpu", "--no-secondary",
                                    "--no-pmc", "--no-cold"], "%gett_f32_stream_kernel%")
            if prof and prof.get("kernel_avg_us"):
                roof["achieved"] = FLOP / (prof["kernel_avg_us"] * 1e-6) / 1e12
                roof["frac"] = roof["achieved"] / peak if peak else None
                roof["frac_source"] = ("live: rocprofv3 --kernel-trace --stats over %d launches of a headline-only child of this run, average kernel duration "
                                       "%.2f us (min %.2f)" % (prof["launches"], prof["kernel_avg_us"], prof["kernel_min_us"]))
                roof["live_trace"] = {k: prof.get(k) for k in ("kernel", "launches", "kernel_avg_us", "kernel_min_us")}
            if prof and prof.get("mfma_busy_pct") is not None:
                roof["mfma_busy_pct"], roof["sustained_clock_ghz"] = prof["mfma_busy_pct"], prof["sustained_clock_ghz"]
                roof["mfma_busy_pct_of_kernel_time_at_nominal_clock"] = prof.get("mfma_busy_pct_of_kernel_time_at_nominal_clock")
                roof["pmc_live"] = {k: prof.get(k) for k in ("kernel_avg_us_under_pmc", "wait_any_pct_of_wave_cycles", "pmc_raw", "pmc_formula")}
        trace_file, trace_rows = newest_trace_summary()
        if trace_file and "gett_f32_stream_kernel" in trace_rows:
            k_avg, k_min, k_calls = trace_rows["gett_f32_stream_kernel"]
            roof["rocprof_trace"] = {"file": trace_file, "kernel_avg_us": k_avg, "kernel_min_us": k_min, "calls": k_calls,
                                     "frac_from_trace_avg": FLOP / (k_avg * 1e-6) / 1e12 / peak if peak else None,
                                     "fold_avg_us": trace_rows.get("splitk_reduce_frag_flat_kernel", (None,))[0],
                                     "what": "rocprofv3 --kernel-trace --stats of this command with --no-cold --no-secondary on the round's profiling box (burn-in, warm-up and gapped launches included; the cold-operand leg has its own trace)"}
        # ---- the sample's own protocol (contraction.cu:252-270): device sync, GPUTimer (event pair) around ONE cutensorContract,
        #      event sync, minimum of 3 — every call starts on an idle, drained device ----------------------------------------------
        try:
            sp = []
            host_c = torch.zeros((EXT["a"], EXT["e"]), dtype=torch.float32).pin_memory()
            for _ in range(3):
                outs[0].copy_(host_c)                  # contraction.cu:256: C is re-sent from the host before every run
                torch.cuda.synchronize()
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                contract(a, b, outs[0])
                s1.record()
                s1.synchronize()
                sp.append(s0.elapsed_time(s1) * 1e3)
            sample_protocol = {"min_us": min(sp), "all_us": sp, "gflops": FLOP / (min(sp) * 1e-6) / 1e9,
                               "frac_of_nominal_f32_mfma_peak": FLOP / (min(sp) * 1e-6) / 1e12 / PEAK_TFLOPS_F32_MFMA,
                               "what": "contraction.cu:252-270 / utils.cuh:163-202: cudaDeviceSynchronize, event pair around ONE call (GETT kernel + fold), "
                                       "minimum of 3 — an idle device before every call, so the clock ramp and the launch gap are inside the number"}
        except Exception as ex:   # noqa: BLE001
            sample_protocol = {"error": "%s: %s" % (type(ex).__name__, ex)}
        # ---- HBM-cold variant of the headline: four rotating (A, B) pairs = 805 MB of operands, beyond the 256-MiB
        #      Infinity Cache, so no step finds its inputs on-die ------------------------------------------------------------
        try:
            if args.no_cold:
                raise RuntimeError("skipped (--no-cold)")
            pairs = [(a, b)]
            for k in range(3):
                pairs.append((torch.rand(a.shape, generator=g, device="cuda"), torch.rand(b.shape, generator=g, device="cuda")))
            # >= 500 steps whatever --steps is (a 20-step sample is 1 ms: it measured 88.6 TFLOP/s on the round-3 driver box where 2000
            # steps gave 96-106), after >= 100 untimed steps of the same rotation; wall clock and one event pair around the same loop
            cold_steps, cold_warm = max(args.steps, 500), max(args.warmup, 100)
            for i in range(cold_warm):
                contract(*pairs[i % 4], outs[i % nbuf])
            torch.cuda.synchronize()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t1 = time.perf_counter()
            c0.record()
            for i in range(cold_steps):
                contract(*pairs[i % 4], outs[i % nbuf])
            c1.record()
            torch.cuda.synchronize()
            cold_s = (time.perf_counter() - t1) / cold_steps
            cold = {"value": FLOP / cold_s / 1e9, "unit": "GFLOP/s", "ms_per_step": cold_s * 1e3, "operand_bytes_rotated": 4 * BYTES,
                    "steps": cold_steps, "warmup": cold_warm, "ms_per_step_events": c0.elapsed_time(c1) / cold_steps,
                    "frac_of_nominal_f32_mfma_peak": FLOP / cold_s / 1e12 / PEAK_TFLOPS_F32_MFMA,
                    "hbm_TBps": BYTES / cold_s / 1e12,
                    "what": "%d steps (never fewer than 500), (A, B) rotate over 4 distinct pairs (805 MB > 256-MiB Infinity Cache): operands come from HBM" % cold_steps}
            # the same rotation through a plan made under CUTENSOR_AMD_PLAN_PREFERENCE_OPERANDS_STREAMED (the engine's "operands come from
            # HBM on every call" hint: nontemporal-load twin of the same kernel) — what a caller who knows its operands are cold gets
            try:
                plan_nt = ops.contraction_plan(h, [EXT[c] for c in "dcba"], "dcba", [EXT[c] for c in "ebcd"], "ebcd", [EXT[c] for c in "ea"], "ea",
                                               workspace_limit=1 << 30, operands_streamed=True)
                ws_nt = torch.empty(max(plan_nt.required_workspace, 256), dtype=torch.uint8, device="cuda")

                def contract_nt(x, y, out):
                    plan_nt.contract(1.0, x.data_ptr(), y.data_ptr(), 0.0, out.data_ptr(), out.data_ptr(), ws_nt.data_ptr(), plan_nt.required_workspace, stream)
                for i in range(cold_warm):
                    contract_nt(*pairs[i % 4], outs[i % nbuf])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(cold_steps):
                    contract_nt(*pairs[i % 4], outs[i % nbuf])
                torch.cuda.synchronize()
                nt_s = (time.perf_counter() - t1) / cold_steps
                cold["streamed_preference"] = {"value": FLOP / nt_s / 1e9, "unit": "GFLOP/s", "ms_per_step": nt_s * 1e3,
                                               "frac_of_nominal_f32_mfma_peak": FLOP / nt_s / 1e12 / PEAK_TFLOPS_F32_MFMA, "nt_kernel": plan_nt.describe().get("nt"),
                                               "what": "same rotation, plan made with CUTENSOR_AMD_PLAN_PREFERENCE_OPERANDS_STREAMED = 1 (nontemporal operand loads)"}
                plan_nt.destroy()
            except Exception as ex:   # noqa: BLE001
                cold["streamed_preference"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            del pairs
        except Exception as ex:   # noqa: BLE001
            cold = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if world == 1 and not args.no_cpu and not args.einsum_only:
            a_np, b_np = a.cpu().numpy(), b.cpu().numpy()
            ref, cpu = cpu_baseline(a_np, b_np)
            contract(a, b, outs[0])
            torch.cuda.synchronize()
            got = outs[0].cpu().numpy()
            cpu["max_rel_diff_vs_gpu"] = float(np.max(np.abs(got - ref) / np.abs(ref)))
        # ---- the other BASELINE configs, same process ---------------------------------------------------------------
        if world == 1 and not args.no_secondary and not args.einsum_only:
            del a, b
            torch.cuda.empty_cache()
            secondary += secondary_single_gpu(torch, ct, ops, h, stream)
            secondary += secondary_general_family(torch, ct, ops, h)
            secondary += secondary_round6(torch, ct, ops, h, stream, not args.no_pmc)
            if not args.no_pmc:
                add_secondary_traffic(secondary)
                add_mfma_counters(secondary)
            try:
                secondary.append(einsum_flow_line())
            except Exception as ex:   # noqa: BLE001
                secondary.append({"workload": "einsum.cu flow (plan per call)", "error": "%s: %s" % (type(ex).__name__, ex)})
            try:
                one = {"sample": mg_measure(1, MG_SAMPLE_EXTENT, 20, 3), "scaled": mg_measure(1, MG_SCALED_EXTENT, 3, 1)}
                secondary.append(mg_lines(one, "sample", "-"))
                secondary.append(mg_lines(one, "scaled", "-"))
            except Exception as ex:   # noqa: BLE001
                secondary.append({"workload": "cuTENSORMg on one device", "error": "%s: %s" % (type(ex).__name__, ex)})
            # ---- the same 4096^3 call with the gather FORCED (CUTENSORMG_AMD_FORCE_GATHER=1): a one-rank RCCL communicator, the operands
            #      travel to the staging images by ncclAllGather / ncclSend + ncclRecv on the communication stream, the local contraction
            #      waits for the wave event and reads the staged images — the transport and event-graph code of the N > 1 path, executed on
            #      the one GPU of this box (gather_bytes_per_call > 0, rccl_ranks_seen = 1); not a scaling number
            # (in a child process: creating an RCCL communicator makes librccl print a version banner on the C stdout of its process at
            # exit — after this process's JSON line, which has to stay the last line)
            try:
                fg = run_mg_child(1, 3, 1, 300, extra_env={"CUTENSORMG_AMD_FORCE_GATHER": "1"})
                if "error" in fg or "sample" not in fg:
                    raise RuntimeError(fg.get("error", "no result"))
                ln = mg_lines(fg, "sample", "-")
                ln["workload"] += " — gather FORCED through RCCL on one device (CUTENSORMG_AMD_FORCE_GATHER=1)"
                for k in ("forced_gather", "rccl_ranks_seen", "rccl", "all_gather_eligible", "transport_trial_ms", "transport_chosen"):
                    ln[k] = fg["sample"].get(k)
                secondary.append(ln)
            except Exception as ex:   # noqa: BLE001
                secondary.append({"workload": "cuTENSORMg on one device, forced gather", "error": "%s: %s" % (type(ex).__name__, ex)})

    if rank == 0:
        einsum_line = {"workload": "einsum.cu 'abcd,dcbe->ae' a=e=96 b=c=d=64 fp32 (BASELINE configs[1])"
                                   + ("" if world == 1 else ", b sharded x%d (b=%d), one process per GPU, %s" % (
                                       world, 64 * world, "RCCL all-reduce of C" if gpu_group is not None else
                                       "NO all-reduce (RCCL communicator unavailable: %s)" % rccl_error)),
                       "dtype": "f32", "value": einsum_value, "unit": "GFLOP/s", "n_gpus": world, "ms_per_step": einsum_ms,
                       "scaling": "weak", "frac_of_nominal_f32_mfma_peak": einsum_value / 1e3 / (PEAK_TFLOPS_F32_MFMA * world)}
        use_mg = mg is not None and "error" not in mg and "scaled" in mg
        if use_mg:
            m = mg["scaled"]
            value, ms_per_step, n_gpus, scaling = m["gflops"], m["ms_per_step"], m.get("distinct_devices", m["devices"]), "strong"
            workload = ("cuTENSORMg contraction_multi_gpu.cu C[i,j]=A[i,k]B[k,j] fp32 %d^3 (BASELINE configs[4], scaled shape), largest free "
                        "mode i cut over %d MI355X, B all-gathered over xGMI (%s), one process / %d devices" % (m["extent"], n_gpus, m["transport"], m["devices"]))
            if mg.get("virtual"):
                workload += " — SELF-TEST: %d logical devices on one GPU, shrunk shape, not a scaling result" % m["devices"]
            secondary.append(mg_lines(mg, "sample", "sample_1"))
            scaled_line = mg_lines(mg, "scaled", "scaled_1")
            secondary.append(einsum_line)
            config = {"workload": workload, "speedup_vs_1": scaled_line.get("speedup_vs_1"), "one_device_gflops": scaled_line.get("one_device_gflops"),
                      "gather_bytes_per_call": m["gather_bytes_per_call"], "sample_protocol": scaled_line["sample_protocol"],
                      "frac_of_nominal_f32_mfma_peak": value / 1e3 / (PEAK_TFLOPS_F32_MFMA * n_gpus), "max_rel_err_sampled": m["max_rel_err_sampled"],
                      "first_attempt_error": mg.get("first_attempt_error")}
            metric = "contraction GFLOP/s, fp32 cuTENSORMg C[i,j]=A[i,k]B[k,j] sharded over %d GPUs" % n_gpus
            roof_out = scaled_line["roofline"]
        else:
            value, ms_per_step, n_gpus, scaling = einsum_value, einsum_ms, world, "weak"
            config = {"workload": einsum_line["workload"], "plan": desc, "algo": args.algo,
                      "frac_of_nominal_f32_mfma_peak": einsum_line["frac_of_nominal_f32_mfma_peak"]}
            if requested > 1:
                config["multi_device"] = (mg or {}).get("error") or ("only %d GPU visible: cuTENSORMg measured on one device (secondary)" % visible)
            metric = "contraction GFLOP/s, fp32 einsum abcd,dcbe->ae"
            roof_out = roof
        # ---- like-for-like fields on EVERY line (whatever `value` is): the einsum at this rank count, the cuTENSORMg 16384^3
        #      contraction at this device count, and its speedup over one device measured in the same child process -----------
        spawned = None
        if world == 1 and requested > 1 and visible > 1 and not args.einsum_only and not args.mg_virtual:
            try:
                spawned = run_einsum_ranks(min(requested, visible), args.steps, args.warmup)
            except Exception as ex:   # noqa: BLE001
                spawned = {"error": "%s: %s" % (type(ex).__name__, ex)}
            if "error" not in spawned:
                sp_line = {"workload": spawned["config"]["workload"], "dtype": "f32", "value": spawned["value"], "unit": "GFLOP/s",
                           "n_gpus": spawned["n_gpus"], "ms_per_step": spawned["ms_per_step"], "scaling": "weak",
                           "rccl_ranks_seen": spawned.get("rccl_ranks_seen"), "launched_by": "bench.py itself (plain --gpus N launch)"}
                secondary.append(sp_line)
            else:
                secondary.append({"workload": "einsum K-sharded over %d ranks (self-spawned)" % min(requested, visible), "error": spawned["error"]})
        if use_mg:
            mg_value, mg_devices, mg_speedup = mg["scaled"]["gflops"], mg["scaled"].get("distinct_devices", mg["scaled"]["devices"]), scaled_line.get("speedup_vs_1")
        else:
            mg_one = next((x for x in secondary if "16384^3" in x.get("workload", "") and "value" in x), None)
            mg_value, mg_devices, mg_speedup = (mg_one["value"], 1, 1.0) if mg_one else (None, 0, None)
        if spawned is not None and "error" not in spawned:
            einsum_n_value, einsum_n_gpus, ranks_seen = spawned["value"], spawned["n_gpus"], spawned.get("rccl_ranks_seen")
        else:
            einsum_n_value, einsum_n_gpus, ranks_seen = einsum_value, world, (world if gpu_group is not None else (1 if world == 1 else 0))
        # the numbers a user of einsum.cu sees on first contact, where a truncated reader still finds them
        config["cold_operands_gflops"] = cold["value"] if cold and "value" in cold else None
        config["cold_operands_frac_of_f32_mfma_peak"] = cold.get("frac_of_nominal_f32_mfma_peak") if cold else None
        config["cold_operands_streamed_preference_gflops"] = (cold.get("streamed_preference") or {}).get("value") if cold else None
        config["sample_protocol_gflops"] = sample_protocol.get("gflops") if sample_protocol else None
        config["sample_protocol_frac_of_f32_mfma_peak"] = sample_protocol.get("frac_of_nominal_f32_mfma_peak") if sample_protocol else None
        config["kernel_frac_live_trace"] = roof.get("frac") if roof else None
        config["mfma_busy_pct"] = roof.get("mfma_busy_pct") if roof else None
        config["sustained_clock_ghz"] = roof.get("sustained_clock_ghz") if roof else None
        line = {
            "metric": metric, "value": value, "unit": "GFLOP/s",
            "n_gpus": n_gpus, "requested_gpus": requested, "steps": args.steps, "warmup": args.warmup, "burn_in_ms": args.burn_in_ms,
            "burn_in_steps": burn_steps, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_is": "mg_value" if use_mg else "einsum_value",
            "einsum_value": einsum_n_value, "einsum_n_gpus": einsum_n_gpus,
            "einsum_frac_of_f32_mfma_peak": (einsum_n_value / 1e3 / (PEAK_TFLOPS_F32_MFMA * max(einsum_n_gpus, 1))) if einsum_n_value else None,
            "mg_value": mg_value, "mg_devices": mg_devices, "speedup_vs_1_same_workload": mg_speedup,
            "mg_over_one_gpu_einsum": (mg_value / einsum_1gpu_value) if (mg_value and einsum_1gpu_value) else None,
            "rccl_ranks_seen": ranks_seen,
            "config": config, "roofline": roof_out, "cpu_baseline": cpu,
            "cold_operands_value": cold["value"] if cold and "value" in cold else None, "cold_operands": cold,
            "sample_protocol": sample_protocol,
            "secondary": secondary,
        }
        if use_mg:
            line["headline_kernel_roofline"] = roof
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

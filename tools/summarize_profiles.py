#!/usr/bin/env python3
"""Condenses the raw rocprofv3 / bench outputs of tools/profile_round.sh into the small tracked files
under profiles/:  python tools/summarize_profiles.py gpurun_out/<tag> <tag>"""
import csv
import hashlib
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, "profiles")  # on a gpurun box only gpurun_out/ travels back
os.makedirs(dst, exist_ok=True)
L0 = "eval_kernel<0, true, false"  # every instantiation of the level-0 pose evaluation: ", 0>" mixed, ", 1>" full, ", 2>" residual-only launches  # level-0 pose evaluation (MODE 0, LVL0, not fused)


def last_json(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


bench = last_json(os.path.join(src, "bench_default.json"))
json.dump(bench, open(os.path.join(dst, f"{tag}_bench_default.json"), "w"), indent=1)
for name in ("bench_b1", "bench_b1_S1", "bench_ringkey", "bench_ringkey_q1", "membw", "bench_cfg_S3", "bench_cfg_sparse", "bench_queue", "bench_queue_b256",
             "bench_b256", "bench_b1024", "bench_evals_only", "bench_with_upload_u8_pinned_overlap", "bench_r02_workload", "bench_compact0",
             "bench_coarse9216", "bench_init_constant_motion", "bench_fixed3"):
    f = os.path.join(src, name + ".log")
    if os.path.exists(f):
        json.dump(last_json(f), open(os.path.join(dst, f"{tag}_{name}.json"), "w"), indent=1)

# per-kernel statistics
def newest(pattern):
    """raw outputs of several profiling rounds may sit side by side (rocprofv3 names files by pid): take the latest"""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


stats = newest(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows[:16])

# level-0 eval kernel from the kernel trace: dispatches with work (the speculative schedule also enqueues
# launches in which every problem has already left level 0; they take a few microseconds)
trace = newest(os.path.join(src, "trace", "*", "*kernel_trace.csv"))
cfg = bench["config"]
l0 = [k for k in cfg["pose_eval_kernels_by_level"] if k["lvl"] == 0][0]
if trace:
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(trace[0])) if L0 in r["Kernel_Name"])
    d = [e - s0 for s0, e in iv]
    work = [x for x in d if x > 5000]
    # the batch is split into stream groups whose level-0 kernels overlap: time = union of the intervals
    busy, cs, ce = 0, None, None
    for s0, e in iv:
        if e - s0 <= 5000:
            continue
        if cs is None:
            cs, ce = s0, e
        elif s0 > ce:
            busy += ce - cs
            cs, ce = s0, e
        else:
            ce = max(ce, e)
    if cs is not None:
        busy += ce - cs
    steps = bench["steps"] + bench["warmup"] + 1  # + the extra timing step of the roofline leg
    B = cfg["frames_in_flight_per_gpu"]
    bytes_eval = bench["roofline"]["bytes_per_launch"] * bench["roofline"]["launches"] / l0["evals"]
    summary = {
        "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu --no-second-leg",
        "kernel": "dsm::" + L0 + ", *>",
        "dispatches": len(d), "dispatches_with_work": len(work),
        "avg_ns_dispatches_with_work": sum(work) / max(1, len(work)),
        "sum_ns_dispatches_with_work": sum(work), "union_ns_dispatches_with_work": busy,
        "steps_in_run": steps, "level0_evals_per_step": l0["evals"],
        "algorithmic_bytes_per_eval": bytes_eval,
        "achieved_GBps_from_trace": l0["evals"] * steps * bytes_eval / max(1, busy),
        "achieved_GBps_bench_hip_events": bench["roofline"]["achieved"],
        "frames_in_flight": B, "stream_groups": cfg["streams"],
        "avg_ns_all_dispatches": sum(d) / max(1, len(d)),
        "bench_avg_dispatch_us": bench["roofline"]["avg_launch_us"],
        "note": "2 stream groups: each launch = 2 concurrent dispatches over half of the batch, so per-dispatch durations "
                "overlap and the rate uses the union of the dispatch intervals -- in the trace and in bench.py's HIP-event "
                "leg alike; avg_ns_all_dispatches (= rocprofv3 --stats AverageNs over the level-0 instantiations) is to be "
                "compared with bench.py's roofline.avg_launch_us (average of one steady-state step per launch_eval call, "
                "which after a level's first round is TWO dispatches: the full evaluations <..., 1>, then the residual-only "
                "ones <..., 2>)",
    }
    json.dump(summary, open(os.path.join(dst, f"{tag}_level0_eval_trace_summary.json"), "w"), indent=1)
    print("trace:", summary["achieved_GBps_from_trace"], "bench:", bench["roofline"]["achieved"])


def kernel_source_sha():
    """the same hash bench.kernel_source_sha() forms over the sources of the eval kernels"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hsh = hashlib.sha256()
    for f in ("tracker_kernels.hip", "dsm_device.hpp", "dsm_kernels.hpp", "lm_math.hpp", "Makefile"):
        hsh.update(open(os.path.join(root, "direct_stereo_slam_amd", "csrc", f), "rb").read())
    return hsh.hexdigest()[:16]


def pmc_sum(dirname, counter):
    files = newest(os.path.join(src, dirname, "*", "*counter_collection.csv"))
    if not files:
        return None, 0
    tot, n = 0.0, 0
    for r in csv.DictReader(open(files[0])):
        if L0 in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


fetch, nf = pmc_sum("pmc_fetch", "FETCH_SIZE")
write, nw = pmc_sum("pmc_write", "WRITE_SIZE")
if fetch is not None:
    pb = last_json(os.path.join(src, "pmc_fetch.log"))
    pl0 = [k for k in pb["config"]["pose_eval_kernels_by_level"] if k["lvl"] == 0][0]
    steps = pb["steps"] + pb["warmup"] + 1
    n_evals = pl0["evals"] * steps
    n_ro = pl0.get("residual_only", 0) * steps  # of them residual-only (calcResPose without the dead calcGSSSEPose): same figure
    n0, w, h = pb["config"]["n0"], pb["config"]["w"], pb["config"]["h"]
    alg = n_evals * (16 * n0 + 12 * w * h)
    raw_fetch = fetch * 1024.0  # FETCH_SIZE / WRITE_SIZE count kilobytes
    # MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced
    # 16 B/lane streaming read; the template stream (16*n0 per eval) is such a stream -> add its other half back
    corrected = raw_fetch + 0.5 * n_evals * 16 * n0
    wr = (write or 0.0) * 1024.0
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace -- python bench.py --no-cpu --no-second-leg --no-fixed-leg --steps 2 --warmup 1",
        "kernel_source_sha": kernel_source_sha(),  # bench.py refuses this file once the kernel sources change
        "config": pb["config"]["name"], "kernel": "dsm::" + L0 + ", *>", "dispatches": nf, "level0_pose_evals": n_evals, "algorithmic_bytes": alg,
        "FETCH_SIZE_bytes_raw": raw_fetch, "WRITE_SIZE_bytes_raw": wr,
        "correction": "template stream (one global_load_dwordx4 per lane) is under-reported by 1/2 on gfx950 (MI355X_MICROARCH.md); half of 16*n0 per eval added back; the tap gathers of the 4-byte-per-texel intensity plane are taken as reported -- calibrated on known byte counts by tools/pmc_calibrate.sh (<tag>_pmc_calibration.json: template stream tallied at 0.50, tap stream at 0.95 of its unique bytes)",
        "level0_residual_only_evals": n_ro,
        "layout_bytes": n_evals * (16 * n0 + 4 * w * h),
        "hbm_bytes_per_layout_byte_level0_pose_eval": (corrected + wr) / (n_evals * (16 * n0 + 4 * w * h)),
        "hbm_read_bytes_corrected": corrected, "hbm_bytes_corrected": corrected + wr,
        "hbm_bytes_per_algorithmic_byte_level0_pose_eval": (corrected + wr) / alg,
        "raw_fetch_per_algorithmic_byte": raw_fetch / alg,
    }
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print("traffic ratio:", out["hbm_bytes_per_algorithmic_byte_level0_pose_eval"])

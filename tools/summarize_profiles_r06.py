#!/usr/bin/env python3
"""Condenses the raw rocprofv3 / bench outputs of tools/profile_round_r06.sh (round 6: bench.py --quick = the headline leg alone) into the small tracked files under profiles/:
   python tools/summarize_profiles_r04.py gpurun_out/<tag> <tag> <dst> [pmc|all]
The dominant kernel of the round-4 default line is tick_eval_kernel<0> (the tick engine's evaluation launch: the staged
evaluations of EVERY pyramid level of a stream group's resident problems)."""
import csv
import glob
import hashlib
import json
import os
import sys

src, tag, dst = sys.argv[1], sys.argv[2], sys.argv[3]
what = sys.argv[4] if len(sys.argv) > 4 else "all"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(dst, exist_ok=True)
KERNEL = "tick_eval_kernel<0"  # (<0> in rounds 4-6; <0, false> / <0, true> since the chains: the launch without / with the chains' code)


def last_json(path):
    if os.path.exists(path + ".detail.json"):  # round 5: the stdout line is the compact form; the full object is the side file
        return json.load(open(path + ".detail.json"))
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


def newest(pattern):
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


def kernel_source_sha():
    hsh = hashlib.sha256()
    for f in ("tracker_kernels.hip", "dsm_device.hpp", "dsm_kernels.hpp", "lm_math.hpp", "Makefile"):
        hsh.update(open(os.path.join(root, "direct_stereo_slam_amd", "csrc", f), "rb").read())
    return hsh.hexdigest()[:16]


def run_totals(b):
    """(algorithmic bytes, template-stream bytes, layout bytes, frames) of every track evaluation of a whole `bench.py` run in stream mode: the
    timed region, the warm-up and the steady-state / instrumented steps behind it all retire their frames inside the run"""
    c, r = b["config"], b["roofline"]
    frames = c["frames_in_flight_per_gpu"] * (b["steps"] + b["warmup"] + 10 + 8 + 4)
    ev = c["evals_per_frame_by_level"]
    by = r["bytes_per_eval_by_level"]
    w, h, n0 = c["w"], c["h"], c["n0"]
    alg = sum(e * bl for e, bl in zip(ev, by)) * frames
    # template entries per level of the dense template: (w_l - 4)(h_l - 4); bytes_per_eval = 16 n + 12 w_l h_l
    tmpl = sum(e * (bl - 12 * (w >> l) * (h >> l)) for l, (e, bl) in enumerate(zip(ev, by))) * frames
    lay = sum(e * (bl - 8 * (w >> l) * (h >> l)) for l, (e, bl) in enumerate(zip(ev, by))) * frames
    return alg, tmpl, lay, frames


def union_ns(iv):
    busy, cs, ce = 0, None, None
    for s0, e in sorted(iv):
        if cs is None:
            cs, ce = s0, e
        elif s0 > ce:
            busy += ce - cs
            cs, ce = s0, e
        else:
            ce = max(ce, e)
    return busy + (ce - cs if cs is not None else 0)


def pmc_sum(dirname, counter):
    files = newest(os.path.join(src, dirname, "*", "*counter_collection.csv"))
    if not files:
        return None, 0
    tot, n = 0.0, 0
    for r in csv.DictReader(open(files[0])):
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


fetch, nf = pmc_sum("pmc_fetch", "FETCH_SIZE")
write, nw = pmc_sum("pmc_write", "WRITE_SIZE")
if fetch is not None and os.path.exists(os.path.join(src, "pmc_fetch.log")):
    pb = last_json(os.path.join(src, "pmc_fetch.log"))
    alg, tmpl, lay, frames = run_totals(pb)
    raw_fetch = fetch * 1024.0  # FETCH_SIZE / WRITE_SIZE count kilobytes
    corrected = raw_fetch + 0.5 * tmpl  # MI355X_MICROARCH.md: a wide coalesced 16 B/lane stream (the template) is tallied at 1/2
    wr = (write or 0.0) * 1024.0
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace -- python bench.py --quick --steps 2 --warmup 1",
        "kernel_source_sha": kernel_source_sha(), "config": pb["config"]["name"], "kernel": "dsm::tick_eval_kernel<pose>", "engine": "ticks", "dispatches": nf,
        "frames_tracked_in_the_run": frames, "algorithmic_bytes": alg, "layout_bytes": lay,
        "FETCH_SIZE_bytes_raw": raw_fetch, "WRITE_SIZE_bytes_raw": wr,
        "correction": "the template stream (one global_load_dwordx4 per lane) is under-reported by 1/2 on gfx950 (MI355X_MICROARCH.md): half of its bytes added back; the tap gathers of the 4-byte intensity planes are taken as reported (calibration: r02_pmc_calibration.json)",
        "hbm_read_bytes_corrected": corrected, "hbm_bytes_corrected": corrected + wr,
        "hbm_bytes_per_algorithmic_byte": (corrected + wr) / alg,
        "hbm_bytes_per_algorithmic_byte_level0_pose_eval": (corrected + wr) / alg,  # (the key bench.py reads; here: all levels of the tick kernel)
        "hbm_bytes_per_layout_byte": (corrected + wr) / lay,
        "raw_fetch_per_algorithmic_byte": raw_fetch / alg,
    }
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print("traffic ratio (HBM bytes per algorithmic byte, tick_eval_kernel<0>):", out["hbm_bytes_per_algorithmic_byte"], "per layout byte:", out["hbm_bytes_per_layout_byte"])
if what == "pmc":
    sys.exit(0)

for name in ["bench_default"] + [os.path.basename(f)[:-4] for f in glob.glob(os.path.join(src, "bench_*.log"))] + ["membw"]:
    f = os.path.join(src, name + ".log")
    if os.path.exists(f):
        try:
            json.dump(last_json(f), open(os.path.join(dst, f"{tag}_{name}.json"), "w"), indent=1)
        except SystemExit as e:
            print("skipped", name, e)

stats = newest(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows[:16])
trace = newest(os.path.join(src, "trace", "*", "*kernel_trace.csv"))
if trace and os.path.exists(os.path.join(src, "trace.log")):
    tb = last_json(os.path.join(src, "trace.log"))
    alg, tmpl, lay, frames = run_totals(tb)
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(trace[0])) if KERNEL in r["Kernel_Name"]]
    d = [e - s for s, e in iv]
    busy = union_ns(iv)
    summary = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --quick",
               "kernel": "dsm::tick_eval_kernel<pose>", "dispatches": len(d), "avg_ns": sum(d) / max(1, len(d)), "sum_ns": sum(d), "union_ns": busy,
               "frames_tracked_in_the_run": frames, "algorithmic_bytes_of_the_run": alg,
               "achieved_GBps_from_trace": alg / max(1, busy), "achieved_GBps_bench_hip_events": tb["roofline"]["achieved"],
               "bench_avg_dispatch_us": tb["roofline"]["avg_launch_us"], "bench_value_frames_per_s": tb["value"],
               "note": "stream groups: every tick is one dispatch per group, overlapping in time: the rate uses the union of the dispatch intervals, in the trace and in "
                       "bench.py's HIP-event leg alike.  The trace covers the WHOLE run (ramp-up and drain ticks with few items included); bench.py's figure is taken over "
                       "steady-state advances"}
    json.dump(summary, open(os.path.join(dst, f"{tag}_tick_eval_trace_summary.json"), "w"), indent=1)
    print("trace:", summary["achieved_GBps_from_trace"], "GB/s over the whole run; bench (steady state):", tb["roofline"]["achieved"], "; avg dispatch", summary["avg_ns"] / 1e3, "us vs", tb["roofline"]["avg_launch_us"])
rk = newest(os.path.join(src, "rk_trace", "*", "*kernel_stats.csv"))
if rk:
    out = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --ringkey --no-cpu --rk-q 1 --rk-n 10000000 --steps 50"}
    for r in csv.DictReader(open(rk[0])):
        if "ringkey_knn_fewq" in r["Name"]:
            avg = float(r["AverageNs"])
            out.update(kernel=r["Name"][:80], calls=int(r["Calls"]), average_ns=avg, sweep_bytes=80 * 10_000_000, GBps=80e7 / avg, frac_of_8TBps=80e7 / avg / 8000.0)
    json.dump(out, open(os.path.join(dst, f"{tag}_ringkey_q1_kernel.json"), "w"), indent=1)
    print("ring-key scan kernel:", out.get("GBps"))

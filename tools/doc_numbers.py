#!/usr/bin/env python3
"""Prints the figures the documents quote from profiles/<tag>_*.json:  python tools/doc_numbers.py r02"""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
P = lambda n: os.path.join(os.path.dirname(__file__), "..", "profiles", f"{tag}_{n}")
for f in sorted(glob.glob(P("bench_*.json"))):
    d = json.load(open(f)); d = d.get("parsed", d)
    r, c = d.get("roofline") or {}, d.get("config", {})
    print(f"{os.path.basename(f):52s} {d['value']:10.1f} {d['unit'][:12]:12s} ms/step {d['ms_per_step']:8.3f}  kernel {r.get('achieved', 0):6.0f} GB/s ({100 * r.get('frac', 0):4.1f} %)  whole {c.get('whole_step_GBps', 0) or 0:5.0f}")
d = json.load(open(P("bench_default.json")))
r = d["roofline"]
print("default roofline:", {k: r[k] for k in ("achieved", "frac", "traffic", "avg_launch_us", "evals", "residual_only_evals", "achieved_on_layout_bytes") if k in r})
f5 = d["config"]["reference_five_level"]
print("S1 leg:", f5["value"], f5["roofline"]["achieved"])
cb = d["cpu_baseline"]
print("cpu:", cb["value"], cb["all_cores"]["value"], cb["ate_vs_cpu_ref"])
print("per level:", d["config"]["pose_eval_kernels_by_level"])
t = json.load(open(P("level0_eval_trace_summary.json")))
print("trace:", {k: t[k] for k in ("dispatches", "dispatches_with_work", "avg_ns_dispatches_with_work", "avg_ns_all_dispatches", "achieved_GBps_from_trace", "bench_avg_dispatch_us")})
print("pmc:", {k: v for k, v in json.load(open(P("pmc_traffic.json"))).items() if "per_" in k or "evals" in k})
print("sq:", json.load(open(P("sq_level0.json")))["derived"])
for row in csv.DictReader(open(P("bench_kernel_stats.csv"))):
    print(f"  {row['Name'][:44]:44s} {row['Calls']:>6s} {float(row['AverageNs']) / 1e3:8.1f} us {row['Percentage']:>6s} %")

#!/usr/bin/env python3
"""Sums the SQ counters of tools/profile_sq.sh over the dispatches of the level-0 pose eval kernel and writes
profiles/<tag>_sq_level0.json:  python tools/summarize_sq.py gpurun_out/<tag>_sq <tag>"""
import csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# every instantiation of the level-0 pose evaluation: ", 0>" mixed, ", 1>" full, ", 2>" residual-only launches; DSM_SQ_KERNEL / DSM_SQ_NAME
# (round 5): another kernel, e.g. "tick_eval_kernel<0>" (all levels in one launch) -> profiles/<tag>_sq_<name>.json
L0 = os.environ.get("DSM_SQ_KERNEL", "eval_kernel<0, true, false")
NAME = os.environ.get("DSM_SQ_NAME", "level0")
sums, avg, ndisp = {}, {}, 0
for p in ("p1", "p2", "p3", "p4", "p5"):
    files = sorted(glob.glob(os.path.join(src, p, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        continue
    seen = set()
    for r in csv.DictReader(open(files[-1])):
        if L0 not in r["Kernel_Name"]:
            continue
        if float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) < 20000:  # speculative launches without work
            continue
        name = r["Counter_Name"]
        if name.endswith("_avr") or name.endswith("_max"):  # per-dispatch averages: average them over the dispatches
            avg.setdefault(name, []).append(float(r["Counter_Value"]))
        else:
            sums[name] = sums.get(name, 0.0) + float(r["Counter_Value"])
        seen.add(r["Dispatch_Id"])
    ndisp = max(ndisp, len(seen))
out = {"source": "five rocprofv3 --pmc passes (tools/profile_sq.sh) with --kernel-trace on `python bench.py --no-cpu --no-second-leg "
                 "--steps 2 --warmup 1` (round 5: + --no-ringkey-leg --no-replay-leg); sums over the dispatches WITH WORK (>= 20 us) of dsm::" + L0,
       "dispatches": ndisp}
out.update({k: int(v) for k, v in sorted(sums.items())})
d = {}
if sums.get("SQ_INSTS_VMEM_RD"):
    d["valu_per_vmem_read"] = sums["SQ_INSTS_VALU"] / sums["SQ_INSTS_VMEM_RD"]
    d["valu_instructions_per_point_iteration"] = 5.0 * d["valu_per_vmem_read"]  # 5 vector-memory reads per point
    d["salu_per_point_iteration"] = 5.0 * sums.get("SQ_INSTS_SALU", 0.0) / sums["SQ_INSTS_VMEM_RD"]
if sums.get("SQ_WAVE_CYCLES"):
    d["fraction_of_wave_time_issuing"] = sums.get("SQ_ACTIVE_INST_ANY", 0.0) / sums["SQ_WAVE_CYCLES"]
    d["fraction_of_wave_time_waiting"] = sums.get("SQ_WAIT_ANY", 0.0) / sums["SQ_WAVE_CYCLES"]
    d["fraction_of_wave_time_issuing_valu"] = sums.get("SQ_ACTIVE_INST_VALU", 0.0) / sums["SQ_WAVE_CYCLES"]
if sums.get("SQ_BUSY_CYCLES") and sums.get("SQ_WAVES"):
    # SQ_WAVE_CYCLES counts quad-cycles per wave (MI355X_MICROARCH.md); resident waves per SIMD = wave time / SIMD busy time
    d["note"] = "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_BUSY_CYCLES is summed over the shader engines"
for k, v in avg.items():
    out[k + "_mean_over_dispatches"] = sum(v) / len(v)
if sums.get("TCP_GATE_EN1_sum") and sums.get("TCP_PENDING_STALL_CYCLES_sum"):
    d["tcp_pending_stall_fraction_of_tcp_active_cycles"] = sums["TCP_PENDING_STALL_CYCLES_sum"] / sums["TCP_GATE_EN1_sum"]
if sums.get("TCP_TCC_READ_REQ_sum") and sums.get("TCP_TCC_READ_REQ_LATENCY_sum"):
    d["l1_to_l2_read_latency_cycles"] = sums["TCP_TCC_READ_REQ_LATENCY_sum"] / sums["TCP_TCC_READ_REQ_sum"]
if sums.get("TCC_HIT_sum") is not None and sums.get("TCC_MISS_sum"):
    d["l2_hit_rate"] = sums["TCC_HIT_sum"] / (sums["TCC_HIT_sum"] + sums["TCC_MISS_sum"])
out["derived"] = d
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, f"{tag}_sq_{NAME}.json"), "w"), indent=1)
print(json.dumps(d, indent=1))

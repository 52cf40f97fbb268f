#!/usr/bin/env python3
"""Sums the SQ counters of tools/profile_sq.sh over the dispatches of the level-0 pose eval kernel and writes
profiles/<tag>_sq_level0.json:  python tools/summarize_sq.py gpurun_out/<tag>_sq <tag>"""
import csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L0 = "eval_kernel<0, 0, true, false>"
sums, ndisp = {}, 0
for p in ("p1", "p2", "p3"):
    files = sorted(glob.glob(os.path.join(src, p, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        continue
    seen = set()
    for r in csv.DictReader(open(files[-1])):
        if L0 not in r["Kernel_Name"]:
            continue
        sums[r["Counter_Name"]] = sums.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        seen.add(r["Dispatch_Id"])
    ndisp = max(ndisp, len(seen))
out = {"source": "three rocprofv3 --pmc passes (tools/profile_sq.sh) with --kernel-trace on `python bench.py --no-cpu --no-six-level --batch 256 "
                 "--steps 2 --warmup 1`; sums over the dispatches of dsm::" + L0 + " (level-0 pose evaluation)",
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
out["derived"] = d
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_sq_level0.json"), "w"), indent=1)
print(json.dumps(d, indent=1))

#!/bin/bash
# Counters of the level-0 pose evaluation in its two forms (dsm_params.tile_l0 = 0: gathers, 1: tile-ordered template + LDS window):
#   bash tools/profile_sq_tile_r04.sh   (through gpurun) -> gpurun_out/r04_profiles/r04_sq_level0_tile{0,1}.json
# The command is the evaluations-only batch form (one full evaluation per level, every problem active, one stream group):
# the level-0 kernel alone, no LM chain around it.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for T in 0 1; do
  OUT=$R/gpurun_out/r04_sq_tile$T
  mkdir -p $OUT
  CMD="python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --no-ringkey-leg --no-replay-leg --evals-only --kf-every 100000 --streams 1 --stream 0 --steps 2 --warmup 1 --tile $T"
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
  timeout 400 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d $OUT/p4 -- $CMD > $OUT/p4.log 2>&1
  timeout 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/p5 -- $CMD > $OUT/p5.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p6 -- $CMD > $OUT/p6.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p7 -- $CMD > $OUT/p7.log 2>&1
  python $R/tools/summarize_sq.py $OUT r04_tile$T $R/gpurun_out/r04_profiles > $OUT/summary.txt 2>&1
  python - "$OUT" "$R/gpurun_out/r04_profiles/r04_tile${T}_sq_level0.json" <<'PY'
import csv, glob, json, os, sys
src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(dst))
L0 = "eval_kernel<0, true, false"
for p in ("p6", "p7"):
    files = sorted(glob.glob(os.path.join(src, p, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        continue
    for r in csv.DictReader(open(files[-1])):
        if L0 in r["Kernel_Name"] and float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) >= 20000:
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    # kernel time of the pass (for GB/s under the counters)
    iv = [(float(r["Start_Timestamp"]), float(r["End_Timestamp"])) for r in csv.DictReader(open(files[-1])) if L0 in r["Kernel_Name"]]
for line in open(os.path.join(src, "p1.log")):
    if line.startswith("{"):
        b = json.loads(line)
        d["bench_level0_GBps_under_the_profiler"] = b["roofline"]["achieved"]
json.dump(d, open(dst, "w"), indent=1)
print(json.dumps(d.get("derived"), indent=1), {k: d[k] for k in d if k.startswith("SQ_LDS") or k.startswith("SQ_ACTIVE_INST_LDS") or k.startswith("SQ_WAIT_INST_LDS") or k == "FETCH_SIZE"})
PY
  rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6 $OUT/p7
done

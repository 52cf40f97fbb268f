#!/bin/bash
# Instruction mix and issue / wait split of the many-query ring-key kernel:  bash tools/profile_ringkey_sq.sh rNN  (through gpurun)
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${TAG}_rk_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --ringkey --no-cpu --rk-q 1024 --rk-n 1000000 --steps 5 --warmup 1"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p4 -- $CMD > $OUT/p4.log 2>&1
python - <<PY
import csv, glob, json, collections
out = {}
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % p):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "ringkey_knn_kernel" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items():
            out[k] = {"sum": v, "dispatches": n, "per_dispatch": v / max(1, n)}
    for f in glob.glob("$OUT/%s/*/*kernel_trace.csv" % p):
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "ringkey_knn_kernel" in r["Kernel_Name"]]
        if d: out.setdefault("kernel_ns", {})[p] = sum(d) / len(d)
json.dump(out, open("$R/gpurun_out/${TAG}_rk_sq.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4

#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box:  bash tools/profile_round.sh rNN
# (run through gpurun; raw outputs land in gpurun_out/<tag>/, summaries are written by tools/summarize_profiles.py)
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the default bench line (with the CPU baseline legs and the five-level second leg)
timeout 900 python $R/bench.py > $OUT/bench_default.log 2>&1
tail -1 $OUT/bench_default.log > $OUT/bench_default.json
# 2. per-kernel statistics of the same command (no CPU leg, no second leg: they only add host time / other kernels)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg > $OUT/trace.log 2>&1
# 3. HBM traffic counters, one pass each (never combined with other trace domains)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
# 4. read-bandwidth ceilings, the single-frame latency line, the other workloads and forms
timeout 300 python $R/bench.py --membw > $OUT/membw.log 2>&1
timeout 300 python $R/bench.py --batch 1 --scenes 1 --steps 50 --no-cpu --no-second-leg --no-fixed-leg > $OUT/bench_b1.log 2>&1
timeout 300 python $R/bench.py --batch 1 --scenes 1 --steps 50 --no-cpu --no-second-leg --no-fixed-leg --config S1 > $OUT/bench_b1_S1.log 2>&1
timeout 300 python $R/bench.py --ringkey --no-cpu --rk-q 1 --rk-n 10000000 --steps 50 > $OUT/bench_ringkey_q1.log 2>&1
timeout 300 python $R/bench.py --ringkey --no-cpu > $OUT/bench_ringkey.log 2>&1
# the scan kernel alone (the bench's step time includes the merge launch and the host synchronisation of every step)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rk_trace -- python $R/bench.py --ringkey --no-cpu --rk-q 1 --rk-n 10000000 --steps 50 > $OUT/rk_trace.log 2>&1
python - "$OUT" "$R/gpurun_out/${TAG}_ringkey_q1_kernel.json" <<'PY'
import csv, glob, json, sys
f = sorted(glob.glob(sys.argv[1] + "/rk_trace/*/*kernel_stats.csv"))
out = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --ringkey --no-cpu --rk-q 1 --rk-n 10000000 --steps 50"}
if f:
    for r in csv.DictReader(open(f[-1])):
        if "ringkey_knn_fewq" in r["Name"]:
            avg = float(r["AverageNs"])
            out.update(kernel=r["Name"][:80], calls=int(r["Calls"]), average_ns=avg, sweep_bytes=80 * 10_000_000, GBps=80e7 / avg, frac_of_8TBps=80e7 / avg / 8000.0)
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
timeout 400 python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --config S3 --batch 256 > $OUT/bench_cfg_S3.log 2>&1
timeout 400 python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --template sparse > $OUT/bench_cfg_sparse.log 2>&1
timeout 400 python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --queue 2 > $OUT/bench_queue.log 2>&1
timeout 400 python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --queue 2 --batch 256 > $OUT/bench_queue_b256.log 2>&1
timeout 400 python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --batch 256 > $OUT/bench_b256.log 2>&1
timeout 400 python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --batch 1024 > $OUT/bench_b1024.log 2>&1
timeout 400 python $R/bench.py --evals-only --kf-every 100000 --no-cpu --no-second-leg --no-fixed-leg --streams 1 > $OUT/bench_evals_only.log 2>&1
# round 2's workload (8 hand-picked frames cycled over the batch) and the scheduling switches on the all-distinct workload
timeout 400 python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --scenes 8 --textures converging > $OUT/bench_r02_workload.log 2>&1
timeout 400 python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --compact 0 > $OUT/bench_compact0.log 2>&1
timeout 400 python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --coarse 9216 --speculate 0 > $OUT/bench_coarse9216.log 2>&1
timeout 400 python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --init constant-motion > $OUT/bench_init_constant_motion.log 2>&1
timeout 400 python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --fixed-schedule 3 > $OUT/bench_fixed3.log 2>&1
# launch chain of one track call, one stream group (tools/chain_timeline.py)
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/chain -- python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --streams 1 --separate-calls --steps 3 --warmup 2 > $OUT/chain.log 2>&1
python $R/tools/chain_timeline.py $(find $OUT/chain -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/${TAG}_profiles_chain.txt 2>&1
timeout 600 python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --with-upload --u8 --pinned --overlap > $OUT/bench_with_upload_u8_pinned_overlap.log 2>&1
# summaries go to gpurun_out/<tag>_profiles/ (gpurun merges only gpurun_out/ back, at most 64 MiB: the raw traces are dropped);
# copy them into profiles/ afterwards:  cp gpurun_out/<tag>_profiles/* profiles/
python $R/tools/summarize_profiles.py $OUT $TAG $R/gpurun_out/${TAG}_profiles > $OUT/summary.log 2>&1
tail -5 $OUT/summary.log
cp $OUT/summary.log $R/gpurun_out/${TAG}_profiles/${TAG}_summary.log
cp $R/gpurun_out/${TAG}_profiles_chain.txt $R/gpurun_out/${TAG}_profiles/${TAG}_chain_timeline.txt
cp $R/gpurun_out/${TAG}_ringkey_q1_kernel.json $R/gpurun_out/${TAG}_profiles/
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/chain $OUT/rk_trace

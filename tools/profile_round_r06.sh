#!/bin/bash
# Regenerates the round-6 evidence under profiles/ on a GPU box:  bash tools/profile_round_r06.sh [r06]
# (run through gpurun; raw outputs land in gpurun_out/<tag>/, the summaries in gpurun_out/<tag>_profiles/: copy them into profiles/)
# Since round 5 bench.py prints a compact line; every run here also writes its FULL object next to its log (<log>.detail.json), which
# tools/summarize_profiles_r06.py reads.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--quick"
b() { # name, args...: one bench run, log + detail
  local name=$1; shift
  timeout 900 python $R/bench.py "$@" --detail-out $OUT/$name.log.detail.json > $OUT/$name.log 2>$OUT/$name.err
}
# 1. HBM traffic counters of the tick engine's evaluation kernel, one pass each (never combined with other trace domains) -- FIRST:
#    the summary is stamped with the kernel-source hash and the default line below then quotes it
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py $Q --steps 2 --warmup 1 --detail-out $OUT/pmc_fetch.log.detail.json > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $R/bench.py $Q --steps 2 --warmup 1 --detail-out $OUT/pmc_write.log.detail.json > $OUT/pmc_write.log 2>&1
# 2. per-kernel statistics of the default command (without the CPU / second legs: they only add host time and other kernels)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py $Q --detail-out $OUT/trace.log.detail.json > $OUT/trace.log 2>&1
python $R/tools/tick_timeline.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) 2 1200 > $OUT/tick_timeline.txt 2>&1
python $R/tools/summarize_profiles_r06.py $OUT $TAG $R/gpurun_out/${TAG}_profiles pmc > $OUT/summary_pmc.log 2>&1
mkdir -p $R/profiles && cp $R/gpurun_out/${TAG}_profiles/${TAG}_pmc_traffic.json $R/profiles/ 2>/dev/null
# 3. the default bench line (CPU legs, five-level, fixed-schedule, plane-family and PCIe-inclusive legs, replay, ring-key and loop-chain legs): the line itself and its detail
timeout 1500 python $R/bench.py --detail-out $OUT/bench_default.log.detail.json > $OUT/bench_default.log 2>$OUT/bench_default.err
cp $OUT/bench_default.log $R/gpurun_out/${TAG}_profiles/${TAG}_bench_default_line.json
# 4. the other forms and workloads
b bench_batch_form $Q --stream 0
b bench_b256 $Q --batch 256
b bench_b1024 $Q --batch 1024
b bench_cfg_S3 $Q --config S3 --batch 256
b bench_cfg_sparse $Q --template sparse
b bench_cfg_sparse_latency_table $Q --template sparse --geometry 1
b bench_latency_table $Q --geometry 1
b bench_fixed3 $Q --fixed-schedule 3
b bench_b1 $Q --batch 1 --scenes 1 --steps 50 --stream 0 --geometry 1
b bench_b1_S1 $Q --batch 1 --scenes 1 --steps 50 --stream 0 --config S1 --geometry 1
b bench_evals_only_batch_form $Q --evals-only --kf-every 100000 --streams 1 --stream 0
timeout 300 python $R/bench.py --membw > $OUT/membw.log 2>&1
# the ring-key scan kernel alone (the bench's per-call time includes the merge launch and the host synchronisation)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rk_trace -- python $R/bench.py --ringkey --no-cpu --rk-q 1 --rk-n 10000000 --steps 50 > $OUT/rk_trace.log 2>&1
python $R/tools/summarize_profiles_r06.py $OUT $TAG $R/gpurun_out/${TAG}_profiles all > $OUT/summary.log 2>&1
tail -8 $OUT/summary.log
cp $OUT/summary.log $R/gpurun_out/${TAG}_profiles/${TAG}_summary.log
cp $OUT/tick_timeline.txt $R/gpurun_out/${TAG}_profiles/${TAG}_tick_timeline.txt
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/rk_trace

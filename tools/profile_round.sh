#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box:  bash tools/profile_round.sh rNN
# (run through gpurun; raw outputs land in gpurun_out/<tag>/, summaries are written by tools/summarize_profiles.py)
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the default bench line (with the CPU baseline leg)
timeout 600 python $R/bench.py > $OUT/bench_default.log 2>&1
tail -1 $OUT/bench_default.log > $OUT/bench_default.json
# 2. per-kernel statistics of the same command (no CPU leg: it only adds host time)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --no-cpu --no-six-level > $OUT/trace.log 2>&1
# 3. HBM traffic counters, one pass each (never combined with other trace domains)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --no-cpu --no-six-level --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $R/bench.py --no-cpu --no-six-level --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
# 4. read-bandwidth ceilings and the single-frame latency line
timeout 300 python $R/bench.py --membw > $OUT/membw.log 2>&1
timeout 300 python $R/bench.py --batch 1 --scenes 1 --steps 50 --no-cpu > $OUT/bench_b1.log 2>&1
timeout 300 python $R/bench.py --ringkey --no-cpu > $OUT/bench_ringkey.log 2>&1
python $R/tools/summarize_profiles.py $OUT $TAG > $OUT/summary.log 2>&1
tail -5 $OUT/summary.log

#!/bin/bash
# kernel-trace timeline of one bench configuration: bash tools/trace_r03.sh <tag> <bench args...>
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python $R/bench.py --no-cpu --no-second-leg --no-fixed-leg --steps 3 --warmup 2 "$@" > $OUT/run.log 2>&1
CSV=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_timeline.py $CSV > $OUT/timeline.txt 2>&1
python $R/tools/chain_timeline.py $CSV > $OUT/chain.txt 2>&1
tail -1 $OUT/run.log | cut -c1-300
cat $OUT/timeline.txt | head -30
rm -rf $OUT/raw

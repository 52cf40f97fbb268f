#!/bin/bash
# The two PMC passes of tools/profile_round_r06.sh alone (HBM traffic of tick_eval_kernel<0>): re-stamps profiles/<tag>_pmc_traffic.json
# with the hash of the current kernel sources after a change that does not alter the generated code path (comments, host code).
#   bash tools/profile_pmc_r06.sh [r06]     (through gpurun; then cp gpurun_out/<tag>_profiles/<tag>_pmc_traffic.json profiles/)
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--quick"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py $Q --steps 2 --warmup 1 --detail-out $OUT/pmc_fetch.log.detail.json > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $R/bench.py $Q --steps 2 --warmup 1 --detail-out $OUT/pmc_write.log.detail.json > $OUT/pmc_write.log 2>&1
python $R/tools/summarize_profiles_r06.py $OUT $TAG $R/gpurun_out/${TAG}_profiles pmc > $OUT/summary_pmc.log 2>&1
tail -3 $OUT/summary_pmc.log
rm -rf $OUT/pmc_fetch $OUT/pmc_write

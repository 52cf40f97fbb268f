#!/bin/bash
# Instruction mix, issue / wait split and memory-pipe state of the tick engine's evaluation kernel (all levels in one launch):
#   bash tools/profile_sq_r06.sh r06   (through gpurun; five counter passes, never combined with other trace domains)
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${TAG}_sq
mkdir -p $OUT $R/gpurun_out/${TAG}_profiles
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --quick --steps 2 --warmup 1 --detail-out /tmp/sq_detail.json"
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1
timeout 400 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d $OUT/p4 -- $CMD > $OUT/p4.log 2>&1
timeout 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/p5 -- $CMD > $OUT/p5.log 2>&1
DSM_SQ_KERNEL="tick_eval_kernel<0>" DSM_SQ_NAME=tick_eval python $R/tools/summarize_sq.py $OUT $TAG $R/gpurun_out/${TAG}_profiles
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5

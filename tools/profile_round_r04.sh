#!/bin/bash
# Regenerates the round-4 evidence under profiles/ on a GPU box:  bash tools/profile_round_r04.sh [r04]
# (run through gpurun; raw outputs land in gpurun_out/<tag>/, the summaries in gpurun_out/<tag>_profiles/: copy them into profiles/)
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu --no-second-leg --no-fixed-leg --no-ringkey-leg --no-replay-leg"
# 1. HBM traffic counters of the tick engine's evaluation kernel, one pass each (never combined with other trace domains) -- FIRST:
#    the summary is stamped with the kernel-source hash and the default line below then quotes it
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py $Q --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $R/bench.py $Q --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
# 2. per-kernel statistics of the default command (without the CPU / second legs: they only add host time and other kernels)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py $Q > $OUT/trace.log 2>&1
python $R/tools/tick_timeline.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) 2 1200 > $OUT/tick_timeline.txt 2>&1
python $R/tools/summarize_profiles_r04.py $OUT $TAG $R/gpurun_out/${TAG}_profiles pmc > $OUT/summary_pmc.log 2>&1
mkdir -p $R/profiles && cp $R/gpurun_out/${TAG}_profiles/${TAG}_pmc_traffic.json $R/profiles/ 2>/dev/null
# 3. the default bench line (CPU legs, five-level and fixed-schedule legs, replay and ring-key legs)
timeout 1500 python $R/bench.py > $OUT/bench_default.log 2>&1
# 4. the other forms and workloads
for spec in "batch_form:--stream 0" "pass_engine:--stream-engine 0" "plane_scenes:--scene-family plane" "plane_scenes_batch_form:--scene-family plane --stream 0" \
            "b256:--batch 256" "b1024:--batch 1024" "cfg_S3:--config S3 --batch 256" "cfg_sparse:--template sparse" "fixed3:--fixed-schedule 3" \
            "b1:--batch 1 --scenes 1 --steps 50 --stream 0" "b1_S1:--batch 1 --scenes 1 --steps 50 --stream 0 --config S1" "ticks16:--stream-ticks 16" "ticks32:--stream-ticks 32" "ticks64:--stream-ticks 64" "streams2:--streams 2" \
            "evals_only_batch_form:--evals-only --kf-every 100000 --streams 1 --stream 0"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 500 python $R/bench.py $Q $args > $OUT/bench_$name.log 2>&1
done
timeout 600 python $R/bench.py $Q --with-upload --u8 --pinned --overlap > $OUT/bench_with_upload_u8_pinned_overlap.log 2>&1
timeout 300 python $R/bench.py --membw > $OUT/membw.log 2>&1
# the ring-key scan kernel alone (the bench's per-call time includes the merge launch and the host synchronisation)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rk_trace -- python $R/bench.py --ringkey --no-cpu --rk-q 1 --rk-n 10000000 --steps 50 > $OUT/rk_trace.log 2>&1
python $R/tools/summarize_profiles_r04.py $OUT $TAG $R/gpurun_out/${TAG}_profiles all > $OUT/summary.log 2>&1
tail -8 $OUT/summary.log
cp $OUT/summary.log $R/gpurun_out/${TAG}_profiles/${TAG}_summary.log
cp $OUT/tick_timeline.txt $R/gpurun_out/${TAG}_profiles/${TAG}_tick_timeline.txt
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/rk_trace

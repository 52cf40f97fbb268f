#!/bin/bash
# same-box A/B of the round's LM-step work (inputs of the next evaluation in parts, helper waves, early list reservation) against the
# tree at the round's last profile set (scratch/base_tree = git archive of that commit, built in place):
# all GPU tests first; then bench.py --quick alternating, one frame in flight, the sparse template with and without chains; kernel trace
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/lm_step_tests.log
tail -3 gpurun_out/lm_step_tests.log
run() { # label tree -- bench args
  local label=$1 tree=$2; shift; shift; shift
  (cd $tree && timeout 400 python bench.py --quick "$@" 2>/dev/null) | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; st=d['config'].get('stream') or {}
print('$label', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'whole', round(r['frac_whole_step'],3), 'avg_launch_us', round(r['avg_launch_us'],1), 'steady', st.get('steady_state_frames_per_s'))"
}
B=$R/scratch/base_tree
for rep in 1 2 3; do
  run "b512 base" $B --
  run "b512 new " $R --
done
for rep in 1 2; do
  run "b1 S2 base" $B -- --batch 1 --scenes 1 --stream 0 --geometry 1 --steps 50
  run "b1 S2 new " $R -- --batch 1 --scenes 1 --stream 0 --geometry 1 --steps 50
  run "b1 S1 base" $B -- --batch 1 --scenes 1 --stream 0 --geometry 1 --steps 50 --config S1
  run "b1 S1 new " $R -- --batch 1 --scenes 1 --stream 0 --geometry 1 --steps 50 --config S1
done
for rep in 1 2; do
  run "sparse base       " $B -- --template sparse
  run "sparse new chain 0" $R -- --template sparse --chain 0
  run "sparse new default" $R -- --template sparse
  DSM_CHAIN_FLAGS=0 run "sparse new default unpaced" $R -- --template sparse
  run "sparse new chain 64" $R -- --template sparse --chain 64
done
run "b256 base" $B -- --batch 256
run "b256 new " $R -- --batch 256
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lm_step_trace -- python $R/bench.py --quick > $R/gpurun_out/lm_step_trace.log 2>&1
python - <<'P'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", ".")
for f in glob.glob(R + "/gpurun_out/lm_step_trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:8]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
P

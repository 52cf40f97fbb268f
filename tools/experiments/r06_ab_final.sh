#!/bin/bash
# all GPU tests on the final sources, then the streamed bench alternating with the tree of the round's earlier profile set
# (scratch/base_tree = git archive of c10aab9, built in place), one frame in flight, and a kernel trace
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/final_tests.log
tail -3 gpurun_out/final_tests.log
run() { # label tree -- bench args
  local label=$1 tree=$2; shift; shift; shift
  (cd $tree && timeout ${TMO:-150} python bench.py --quick "$@" 2>gpurun_out/_err.log) | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; st=d['config'].get('stream') or {}
    print('$label', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'frac', round(r['frac'],3), 'whole', round(r['frac_whole_step'],3), 'avg_launch_us', round(r['avg_launch_us'],1), 'steady', st.get('steady_state_frames_per_s'))
except Exception as e:
    print('$label FAILED', e); print(open('$tree/gpurun_out/_err.log').read()[-1500:])"
}
B=$R/scratch/base_tree; mkdir -p $B/gpurun_out
for rep in 1 2; do
  run "b512 base" $B --
  run "b512 new " $R --
done
run "b1 S2 new " $R -- --batch 1 --scenes 1 --stream 0 --geometry 1 --steps 50
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_trace -- python $R/bench.py --quick > $R/gpurun_out/final_trace.log 2>&1
python - <<'P'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", ".")
for f in glob.glob(R + "/gpurun_out/final_trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:8]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
P

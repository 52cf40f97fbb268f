#!/bin/bash
# same-box A/B: which part of the LM-step work costs the streamed bench what -- DSM_LM_OPTS bit 0 helper waves, bit 1 early list
# reservation (tick_lm_kernel) -- against the tree of the round's last profile set; then the sparse template with the default chains
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
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
  run "b512 base  " $B --
  for o in 0 1 2 3; do DSM_LM_OPTS=$o run "b512 opts $o" $R --; done
done
for rep in 1 2 3; do
  run "sparse default (chain 8, helper)" $R -- --template sparse
  DSM_CHAIN_FLAGS=0 run "sparse default, no helper in chains" $R -- --template sparse
done
run "sparse chain 0" $R -- --template sparse --chain 0

#!/bin/bash
# same-box A/B: what the chains' raised priority (flag 1) and their pacing by the launch's progress (flag 2) each do
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
run() { # label -- bench args
  local label=$1; shift; shift
  timeout 400 python $R/bench.py --quick "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$label', round(d['value']), 'frac', round(r['frac'],3), 'whole', round(r['frac_whole_step'],3), 'avg_launch_us', round(r['avg_launch_us'],1), 'steady', round(d['config']['stream']['steady_state_frames_per_s']))"
}
run "b512 chain 0" -- --chain 0
for f in 0 1 2 3; do for c in 4 64; do DSM_CHAIN_FLAGS=$f run "b512 flags $f chain $c" -- --chain $c; done; done
run "b512 chain 0" -- --chain 0
run "b128 chain 0" -- --chain 0 --batch 128
for f in 0 2 3; do for c in 3 64; do DSM_CHAIN_FLAGS=$f run "b128 flags $f chain $c" -- --chain $c --batch 128; done; done
timeout 1500 python tools/experiments/r06_replay_concurrent.py 128 1,0,0,1 1,0,32,0 1,0,16,0 0,0,32,0 0,64,32,0 0,64,16,0 0,64,8,0 0,64,0,1

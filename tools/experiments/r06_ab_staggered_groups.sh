#!/bin/bash
# same-box A/B of the staggered stream groups
R=${GRAFT_REPO_ROOT:-$PWD}
run() { # label, env..., -- bench args
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python $R/bench.py --quick "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$label', round(d['value']), 'frac', round(r['frac'],3), 'whole', round(r['frac_whole_step'],3), 'avg_launch_us', round(r['avg_launch_us'],1), 'steady', round(d['config']['stream']['steady_state_frames_per_s']))"
}
run "s3 W0        " X=1 -- --streams 3
run "s3 W2 us0    " DSM_TICK_STAGGER=2 -- --streams 3
run "s3 W2 us50   " DSM_TICK_STAGGER=2 DSM_TICK_STAGGER_US=50 -- --streams 3
run "s3 W1        " DSM_TICK_STAGGER=1 -- --streams 3
run "s2 W1        " DSM_TICK_STAGGER=1 -- --streams 2
run "q8 s4 W0     " GPU_MAX_HW_QUEUES=8 -- --streams 4
run "q8 s4 W2 us40" GPU_MAX_HW_QUEUES=8 DSM_TICK_STAGGER=2 DSM_TICK_STAGGER_US=40 -- --streams 4
run "q8 s4 W3 us30" GPU_MAX_HW_QUEUES=8 DSM_TICK_STAGGER=3 DSM_TICK_STAGGER_US=30 -- --streams 4
run "q8 s6 W2 us30" GPU_MAX_HW_QUEUES=8 DSM_TICK_STAGGER=2 DSM_TICK_STAGGER_US=30 -- --streams 6
run "q8 s6 W3 us25" GPU_MAX_HW_QUEUES=8 DSM_TICK_STAGGER=3 DSM_TICK_STAGGER_US=25 -- --streams 6
run "s4 W2 us40   " DSM_TICK_STAGGER=2 DSM_TICK_STAGGER_US=40 -- --streams 4
run "s3 W0 again  " X=1 -- --streams 3

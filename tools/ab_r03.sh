#!/bin/bash
# Same-box A/B of bench.py argument sets (round 3): bash tools/ab_r03.sh "<args A>" "<args B>" ... ; each set twice, alternating
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for a in "$@"; do
    printf "%-40s " "$a"
    timeout 600 python "$R/bench.py" --no-cpu --no-second-leg --no-fixed-leg $a 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print(round(d['value']), 'f/s', round(d['ms_per_step'],3), 'ms', round(r['achieved']), 'GB/s', 'launch_pairs', d['config'].get('launch_pairs_per_step'), [x['kernel_ms'] for x in d['config'].get('pose_eval_kernels_by_level', [])])
except Exception as e:
    print('FAILED', e)
"
  done
done

#!/bin/bash
# Does the 256 MiB Infinity Cache help the level-0 evaluation when a cohort's data stays resident?
# evals-only steps (exactly one evaluation per level and problem, every problem active in every launch) re-read the same
# B frames step after step: B x 16.6 MB (S2, all levels).  B <= 12 fits the cache, B >= 32 does not.
R=${GRAFT_REPO_ROOT:-$PWD}
for B in 8 12 16 24 32 64 256; do
  printf "B=%d: " $B
  python $R/bench.py --evals-only --batch $B --kf-every 100000 --no-cpu --no-second-leg --steps 30 --streams 1 "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'frames/s', [x['GBps'] for x in d['config']['pose_eval_kernels_by_level']], 'whole', round(d['config']['whole_step_GBps']))"
done
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.getcwd()))
from direct_stereo_slam_amd.tracker import Context
c = Context(0)
for mb in (32, 64, 128, 192, 256, 512, 2048):
    print('re-read of a', mb, 'MiB buffer: grid-stride', round(c.read_bandwidth(mb << 20, 20)), 'GB/s; chunk-per-workgroup', round(c.read_bandwidth_chunked(mb << 20, 112 << 10, 20)), 'GB/s')
PY

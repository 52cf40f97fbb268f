R=${GRAFT_REPO_ROOT:-$PWD}
run() { printf "%s: " "$*"; python $R/bench.py --no-cpu --no-second-leg --steps 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'f/s', round(d['ms_per_step'],2), 'ms roof', round(d['roofline']['achieved']), 'GB/s', [x['GBps'] for x in d['config']['pose_eval_kernels_by_level']], 'whole', round(d['config']['whole_step_GBps']), d['config']['all_tracked'], d['config']['launch_pairs_per_step'])"; }
run
run --fuse 2
run --fuse 0
run --batch 256
run --batch 256 --fuse 2

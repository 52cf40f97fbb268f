R=${GRAFT_REPO_ROOT:-$PWD}
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | head -1
run() { printf "%s: " "$*"; python $R/bench.py --no-cpu --no-second-leg --steps 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'f/s', round(d['ms_per_step'],3), 'ms roof', round(d['roofline']['achieved']), 'GB/s', [x['kernel_ms'] for x in d['config']['pose_eval_kernels_by_level']], [x['launches'] for x in d['config']['pose_eval_kernels_by_level']], 'whole', round(d['config']['whole_step_GBps']), d['config']['launch_pairs_per_step'])"; }
run --speculate 0
run --speculate 1
run --speculate 0
run --speculate 1
run --speculate 0 --batch 64
run --speculate 1 --batch 64
run --speculate 0 --batch 1 --scenes 1 --steps 50 --config S1
run --speculate 1 --batch 1 --scenes 1 --steps 50 --config S1

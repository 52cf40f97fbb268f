R=${GRAFT_REPO_ROOT:-$PWD}
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { printf "%s: " "$*"; python $R/bench.py --no-cpu --no-second-leg --steps 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'f/s', round(d['ms_per_step'],2), 'ms roof', round(d['roofline']['achieved']), 'GB/s', 'whole', round(d['config']['whole_step_GBps']), d['config']['all_tracked'], d['roofline'].get('queue_items'))"; }
run --queue 2 --batch 256 --config S1
run --queue 2 --batch 256 --config S2
run --queue 0 --batch 256 --config S2
run --queue 2 --batch 512 --config S2
run --queue 0 --batch 512 --config S2
run --queue 2 --batch 128 --config S2
run --queue 0 --batch 128 --config S2
run --queue 2 --batch 64 --config S2
run --queue 0 --batch 64 --config S2

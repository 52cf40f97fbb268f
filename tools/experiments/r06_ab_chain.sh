#!/bin/bash
# same-box A/B of the tick engine's chains (dsm_stream_set_chain): stream tests first, then bench.py --quick with chains off / bounded
# at 512, 256 and 128 resident and on the sparse template, then the replay's concurrent leg (128 sequences from C++)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_stream.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/chain_tests.log
tail -3 gpurun_out/chain_tests.log
run() { # label -- bench args
  local label=$1; shift; shift
  timeout 400 python $R/bench.py --quick "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$label', round(d['value']), 'frac', round(r['frac'],3), 'whole', round(r['frac_whole_step'],3), 'avg_launch_us', round(r['avg_launch_us'],1), 'steady', round(d['config']['stream']['steady_state_frames_per_s']), 'ticks/adv', d['config']['stream'].get('ticks_per_advance'))"
}
for rep in 1 2; do
  for c in ${CHAINS:-0 4 8 64}; do run "b512 chain $c" -- --chain $c; done
done
for c in 0 64; do run "b256 chain $c" -- --chain $c --batch 256; done
for c in 0 64; do run "b128 chain $c" -- --chain $c --batch 128; done
for c in 0 64; do run "sparse chain $c" -- --chain $c --template sparse; done
timeout 1200 python tools/experiments/r06_replay_concurrent.py 128

#!/bin/bash
# chain_kernel (dsm_params.persistent_coarse = -1, chunk table 2): its tests, then one frame in flight under the three settings, then the replay
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_parity_tracker.py tests/test_host_adaptor.py tests/test_replay_bench.py -x -q -m gpu -k "chain_kernel or chunk_geometries or adaptor or binding or replay" 2>&1 | tail -15 > gpurun_out/chain_kernel_tests.log
grep -n "passed\|failed" gpurun_out/chain_kernel_tests.log
run() { # label -- bench args
  local label=$1; shift; shift
  (timeout 120 python bench.py --quick "$@" 2>gpurun_out/_err.log) | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
    print('$label', 'ms per stereo frame', round(d['ms_per_step'],4), 'launch pairs', c.get('launch_pairs_per_step'), 'max err m', c.get('max_abs_translation_error_m'))
except Exception as e:
    print('$label FAILED', e); print(open('gpurun_out/_err.log').read()[-1200:])"
}
for cfg in S2 S1; do
  run "b1 $cfg geometry 1          " -- --batch 1 --scenes 1 --stream 0 --steps 50 --config $cfg --geometry 1
  run "b1 $cfg geometry 2          " -- --batch 1 --scenes 1 --stream 0 --steps 50 --config $cfg --geometry 2
  run "b1 $cfg geometry 2 coarse -1" -- --batch 1 --scenes 1 --stream 0 --steps 50 --config $cfg --geometry 2 --coarse -1
  run "b1 $cfg geometry 0 coarse -1" -- --batch 1 --scenes 1 --stream 0 --steps 50 --config $cfg --geometry 0 --coarse -1
done
timeout 300 python tools/experiments/r06_replay_concurrent.py 128 1,-1,0,1,0 2,-1,0,1,-1 2,-1,0,1,0

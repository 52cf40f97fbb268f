#!/bin/bash
# same-box A/B of library builds over several bench argument sets: bash tools/ab_libs_r03.sh lib1.so lib2.so ... -- "<args1>" "<args2>"
R=${GRAFT_REPO_ROOT:-$PWD}
LIB=$R/direct_stereo_slam_amd/lib/libdsm_hotpath.so
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
shift
cp "$LIB" /tmp/lib_restore.so
for rep in 1 2; do
  for a in "$@"; do
    for l in "${libs[@]}"; do
      cp "$l" "$LIB"
      printf "%-18s %-50s " "$(basename $l)" "$a"
      timeout 600 python "$R/bench.py" --no-cpu --no-second-leg --no-fixed-leg $a 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print(round(d['value']), 'f/s', round(d['ms_per_step'],3), 'ms', round(r['achieved']), 'GB/s', [x['kernel_ms'] for x in d['config'].get('pose_eval_kernels_by_level', [])])
except Exception as e:
    print('FAILED', e)
"
    done
  done
done
cp /tmp/lib_restore.so "$LIB"

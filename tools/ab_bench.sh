#!/bin/bash
# Same-box A/B of library builds:  gpurun -- 'bash tools/ab_bench.sh scratch/lib_a.so scratch/lib_b.so -- --batch 256'
# Fresh GPU boxes differ by +-5-10 % on every figure, so kernel variants are only ever compared inside ONE gpurun call,
# alternating, at least twice.  Each build is copied over the in-tree library for its run; the first one is restored at the end.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
LIB=$R/direct_stereo_slam_amd/lib/libdsm_hotpath.so
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ $# -gt 0 ] && shift
cp "$LIB" /tmp/lib_restore.so
for rep in 1 2; do
  for l in "${libs[@]}"; do
    cp "$l" "$LIB"
    printf "%s: " "$(basename "$l")"
    python "$R/bench.py" --no-cpu "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), 'frames/s', round(r['achieved']), 'GB/s', [x['GBps'] for x in d['config'].get('pose_eval_kernels_by_level', [])])"
  done
done
cp /tmp/lib_restore.so "$LIB"

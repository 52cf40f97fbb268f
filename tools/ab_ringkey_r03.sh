#!/bin/bash
# same-box A/B of the ring-key kernels: bash tools/ab_ringkey_r03.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$PWD}
LIB=$R/direct_stereo_slam_amd/lib/libdsm_hotpath.so
cp "$LIB" /tmp/lib_restore.so
for rep in 1 2; do
  for l in "$@"; do
    cp "$l" "$LIB"
    for a in "--rk-q 1 --rk-n 10000000 --steps 50" "--rk-q 2 --rk-n 10000000 --steps 50" "--rk-q 8 --rk-n 10000000 --steps 50" "--rk-q 1024 --rk-n 1000000"; do
      printf "%-16s %-40s " "$(basename $l)" "$a"
      timeout 300 python "$R/bench.py" --ringkey --no-cpu $a 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(round(d['value']), 'q/s', round(1e3*d['ms_per_step'],1), 'us', round(c['db_sweep_GBps']), 'GB/s', c['allreduce_min'].get('matches_unsharded'))"
    done
  done
done
cp /tmp/lib_restore.so "$LIB"

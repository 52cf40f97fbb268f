#!/bin/bash
# the round's closing evidence on the final sources: every GPU test, smoke, the PMC passes (re-stamps the traffic profile with the hash of
# the final kernel sources: bench.py refuses another's), the default bench line, one frame in flight
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out/final
timeout 700 python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/final/gpu_tests.log
grep -n "passed\|failed" gpurun_out/final/gpu_tests.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout 600 bash tools/profile_pmc_r06.sh r06b > gpurun_out/final/pmc.log 2>&1; tail -2 gpurun_out/final/pmc.log
cp gpurun_out/r06b_profiles/r06b_pmc_traffic.json profiles/ 2>/dev/null
cd $R
timeout 500 python bench.py --detail-out gpurun_out/final/bench_default_detail.json > gpurun_out/final/bench_default_line.json 2> gpurun_out/final/bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/final/bench_default_line.json').read().strip().splitlines()[-1]); r=d['roofline']
print('default', d['value'], 'frac', r['frac'], 'whole', r.get('frac_whole_step'), 'traffic', r.get('traffic'), 'cpu', d.get('cpu_baseline',{}).get('value'))
rp=d['config']['legs']['replay']; print('replay', rp)"
for cfg in S2 S1; do
  timeout 120 python bench.py --quick --batch 1 --scenes 1 --stream 0 --geometry 1 --steps 50 --config $cfg --detail-out gpurun_out/final/bench_b1_$cfg.json > gpurun_out/final/bench_b1_${cfg}_line.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/final/bench_b1_${cfg}_line.json').read().strip().splitlines()[-1]); print('b1 $cfg ms per stereo frame', d['ms_per_step'])"
done

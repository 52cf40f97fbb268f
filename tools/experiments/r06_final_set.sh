#!/bin/bash
# the round's closing evidence on the final sources: every GPU test, the default bench line, the sparse template under the default chains
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out/final
timeout 700 python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/final/gpu_tests.log
grep -n "passed\|failed" gpurun_out/final/gpu_tests.log
timeout 500 python bench.py --detail-out gpurun_out/final/bench_default_detail.json > gpurun_out/final/bench_default_line.json 2> gpurun_out/final/bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/final/bench_default_line.json').read().strip().splitlines()[-1]); r=d['roofline']
print('default', d['value'], 'frac', r['frac'], 'whole', r.get('frac_whole_step'), 'cpu', d.get('cpu_baseline',{}).get('value'))"
timeout 450 python bench.py --quick --template sparse --detail-out gpurun_out/final/bench_cfg_sparse_detail.json > gpurun_out/final/bench_cfg_sparse.json 2> gpurun_out/final/bench_cfg_sparse.err
python -c "
import json; d=json.loads(open('gpurun_out/final/bench_cfg_sparse.json').read().strip().splitlines()[-1]); print('sparse default chains', d['value'])"

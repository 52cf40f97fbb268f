cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_b256 -- python $R/bench.py --cpu-frames 0 --steps 4 --warmup 2 > /tmp/o.txt 2>&1
tail -1 /tmp/o.txt | cut -c1-200
ls $R/gpurun_out/prof_b256/*/

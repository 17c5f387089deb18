#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "private" 2>&1 | tail -3
python bench.py --config 5 --steps 30 > gpurun_out/r04_bench_cfg5.json 2> gpurun_out/r04_bench_cfg5.err
A2AMD_RAW=0 python bench.py --config 5 --steps 30 --no-cpu-baseline > gpurun_out/r04_bench_cfg5_coef.json 2>> gpurun_out/r04_bench_cfg5.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k5; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k5 -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 30 --no-cpu-baseline --no-realtime > /tmp/prof_k5.log 2>&1
find /tmp/prof_k5 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r04_bench_cfg5_kernel_stats.csv \;
head -5 $GRAFT_REPO_ROOT/gpurun_out/r04_bench_cfg5_kernel_stats.csv

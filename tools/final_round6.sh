#!/bin/bash
# the last GPU call of round 6: the bench line with the driver's flags on the final library, then tools/final_check.sh
# (whole GPU suite, smoke(), the reference's song traces replayed)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/round6_final
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/round6_final/bench_default.json 2> gpurun_out/round6_final/bench_default.err
cp bench_details.json gpurun_out/round6_final/bench_default_details.json
wc -c gpurun_out/round6_final/bench_default.json
bash tools/final_check.sh

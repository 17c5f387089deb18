#!/bin/bash
# end-of-round soak on the GPU box: the differential fuzzer over all seed families, three ways; the nine reference songs
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/final_soak.txt; : > $O
run() { echo "== $*" >> $O; "$@" 2>&1 | tail -3 >> $O; }
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 24 324
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 1012 1212
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 2324 2524
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 3216 3416
run python tests/measure/fuzz_soak.py 324 474
run python tests/measure/fuzz_soak.py 2524 2624
A2FUZZ_WALK=1 A2AMD_DEVICES=3 run python tests/measure/fuzz_soak.py 3416 3466
[ -d soak_long ] && run python tests/measure/soak_long.py replay
cat $O

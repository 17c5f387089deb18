#!/bin/bash
# end-of-round soak on the GPU box: the differential fuzzer over all seed families, three ways; the reference's songs
# (seed ranges not used by the earlier soaks of the round: profiles/r04_fuzz_soak*.json, r04_final_soak.json)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/final_soak.txt; : > $O
run() { echo "== $*" >> $O; "$@" 2>&1 | tail -3 >> $O; }
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 500 800
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 1300 1500
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 2700 2900
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 3500 3700
run python tests/measure/fuzz_soak.py 800 950
run python tests/measure/fuzz_soak.py 2900 3000
A2FUZZ_WALK=1 A2AMD_DEVICES=3 run python tests/measure/fuzz_soak.py 3700 3750
[ -d soak_long ] && run python tests/measure/soak_long.py replay
[ -d soak_long/testdata ] && run python tests/measure/soak_long.py replay testdata
cat $O

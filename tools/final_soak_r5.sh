#!/bin/bash
# end-of-round-5 soak on the GPU box: the differential fuzzer with the window kernels FORCED (A2AMD_WIN=1: small scenes
# would otherwise go through k_leaf_recs) and the pool checked at once, so that looping voices run through k_vm_win
# wherever a batch is what the batch before predicted; seed ranges no earlier soak used
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/final_soak_r5.txt; : > $O
run() { echo "== $*" >> $O; "$@" 2>&1 | tail -3 >> $O; }
export A2AMD_WIN=1 A2AMD_WIN_CHECK=1
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 2200 2300
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 3800 3880
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 1600 1650
A2FUZZ_WALK=1 A2AMD_VMWIN=0 run python tests/measure/fuzz_soak.py 2400 2430
run python tests/measure/fuzz_soak.py 950 980
unset A2AMD_WIN A2AMD_WIN_CHECK
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 3950 3980
[ -d soak_long ] && run python tests/measure/soak_long.py replay
cat $O

#!/bin/bash
# the round's last soak, on its last commit: launch classes checked at every rebuild, window kernels forced in the first legs
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/final_soak_4.txt; : > $O
run() { echo "== $*" >> $O; "$@" 2>&1 | tail -3 >> $O; }
export A2AMD_CLS_CHECK=1
A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 2480 2500
A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_DEVICES=2 run python tests/measure/fuzz_soak.py 3700 3760
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 1850 1950
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 450 500
run python tests/measure/fuzz_soak.py 1012 1060
cat $O

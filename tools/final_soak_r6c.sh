#!/bin/bash
# round 6, third soak, on the final library: new seeds, the legs of tools/final_soak_r6.sh that exercise the device VM's
# speculative pass at every batch length, and a leg of longer renders (8 s instead of 2) of fewer scripts
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_final_soak_c.txt; : > $O
run() { echo "== $*" >> $O; "$@" 2>&1 | tail -2 >> $O; }
export A2AMD_CLS_CHECK=1
A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 run python tests/measure/fuzz_soak.py 8000 8150
A2AMD_WIN=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 A2AMD_VMSKIP=3 run python tests/measure/fuzz_soak.py 8200 8280
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 8300 8400
A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 A2AMD_DEVICES=2 run python tests/measure/fuzz_soak.py 8500 8540
A2AMD_WIN=1 A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 8600 8620 384000
cat $O

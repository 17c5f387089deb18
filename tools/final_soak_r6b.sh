#!/bin/bash
# round 6, second soak - after the speculative pass went in front of the render pass, the quiet launches it makes
# unnecessary were dropped and run_batch's walk over chain positions changed: the device-VM tests, then random scripts
# through the reference engine with and without the drop-in (tests/measure/fuzz_soak.py), new seeds, launch classes
# checked at every rebuild.  Legs as tools/final_soak_r6.sh.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_final_soak_b.txt; : > $O
echo "#### tests/test_device_vm.py" >> $O; python -m pytest tests/test_device_vm.py -q -x -p no:cacheprovider 2>&1 | tail -3 >> $O
run() { echo "== $*" >> $O; "$@" 2>&1 | tail -2 >> $O; }
export A2AMD_CLS_CHECK=1
A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 run python tests/measure/fuzz_soak.py 6000 6120
A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 A2AMD_VMSKIP=3 run python tests/measure/fuzz_soak.py 6200 6260
A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 A2AMD_VMSPEC_EARLY=0 run python tests/measure/fuzz_soak.py 6300 6340
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 6400 6480
run python tests/measure/fuzz_soak.py 6500 6530
A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 A2AMD_DEVICES=2 run python tests/measure/fuzz_soak.py 6600 6640
cat $O

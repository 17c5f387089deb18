#!/bin/bash
# round 6's soak: random scripts through the reference engine with and without the drop-in (tests/measure/fuzz_soak.py), launch
# classes checked at every rebuild.  Legs: the speculative VM pass forced onto every batch length (A2AMD_VMSPEC_MIN=1), started
# beside the render pass (default) and behind the leaf kernels; default settings; units only; one engine state over two contexts.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_final_soak.txt; : > $O
run() { echo "== $*" >> $O; "$@" 2>&1 | tail -2 >> $O; }
export A2AMD_CLS_CHECK=1
A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 run python tests/measure/fuzz_soak.py ${1:-5000} ${2:-5150}
A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 A2AMD_VMSPEC_EARLY=0 run python tests/measure/fuzz_soak.py 5200 5260
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 5300 5400
A2FUZZ_WALK=1 A2AMD_VMSPEC=0 run python tests/measure/fuzz_soak.py 5400 5430
run python tests/measure/fuzz_soak.py 5500 5540
A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 A2AMD_DEVICES=2 run python tests/measure/fuzz_soak.py 5600 5640
cat $O

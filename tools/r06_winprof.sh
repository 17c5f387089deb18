#!/bin/bash
# round 6: cycle counters inside k_vm_win (-DWIN_PROF build in build_variants/winprof): where a lane's time goes
cd ${GRAFT_REPO_ROOT:-$(pwd)}
V=$PWD/build_variants/winprof
pre="$V/liba2amd_walk.so $V/liba2amd_units.so"
for prog in OscFilterPanScripted OscPanScripted; do
  echo "== $prog (speculation off: one pass per batch, in line)"
  ( cd tests/a2s; A2AMD_VMSPEC=0 LD_PRELOAD="$pre" A2REF_BUFFER=4096 timeout 120 ../../oracle/_ref/ref_bench bench.a2s $prog 16384 1024 1 2>&1 | grep "k_vm_win\|\.\.\. \|voice_samples" | tail -12 | cut -c1-300 )
done

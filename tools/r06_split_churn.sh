#!/bin/bash
# round 6: the buffer delivered in two halves (A2AMD_SPLIT=n: from n frames on; the GPU renders the first half while the engine
# walks the second) - the churn cell is the engine thread's (walk 470 us of an 830 us buffer, profiles/r06_timeline_churn.txt)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
pre="$PWD/audiality2_amd/liba2amd_walk.so $PWD/audiality2_amd/liba2amd_units.so"
for rep in 1 2; do for prog in OscFilterPanChurn OscFilterPan OscPanScripted; do for mode in "A2AMD_SPLIT=0" "A2AMD_SPLIT=2048"; do
  echo "== $prog a2_Run(4096) $mode"
  ( cd tests/a2s; env $mode LD_PRELOAD="$pre" A2REF_BUFFER=4096 timeout 120 ../../oracle/_ref/ref_bench bench.a2s $prog 16384 8192 1 2>&1 | grep "voice_samples" | cut -c1-260 )
done; done; done

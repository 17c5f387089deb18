#!/bin/bash
# round 6: why is the speculative pass (vm_speculate) launched / taken or not - the statistics line of A2AMD_HOSTTIMING
cd ${GRAFT_REPO_ROOT:-$(pwd)}
pre="$PWD/audiality2_amd/liba2amd_walk.so $PWD/audiality2_amd/liba2amd_units.so"
for prog in OscPanScripted OscFilterPanScripted; do for buf in 4096; do
  echo "== $prog a2_Run($buf)"
  ( cd tests/a2s; LD_PRELOAD="$pre" A2REF_BUFFER=$buf A2AMD_HOSTTIMING=2 ../../oracle/_ref/ref_bench bench.a2s $prog 16384 $([ $buf = 64 ] && echo 3000 || echo 12288) 1 2>&1 | grep -v "uploads by first" | grep "speculative\|batches with device\|not fused" | sed "s/[0-9.e+]* us//g" | sort | uniq -c | sort -rn | head -12 | cut -c1-600 )
done; done

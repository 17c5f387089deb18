cd ${GRAFT_REPO_ROOT:-$(pwd)}
pre="$PWD/audiality2_amd/liba2amd_walk.so $PWD/audiality2_amd/liba2amd_units.so"
for prog in OscPanScripted OscFilterPanScripted; do for mode in "A2AMD_SPLIT=0" "A2AMD_SPLIT=2048" "A2AMD_SPLIT=1024"; do
  echo "== $prog a2_Run(4096) $mode"
  ( cd tests/a2s; env $mode LD_PRELOAD="$pre" A2REF_BUFFER=4096 A2AMD_HOSTTIMING=1 ../../oracle/_ref/ref_bench bench.a2s $prog 16384 12288 1 2>&1 | grep "speculative\|voice_samples" | sed 's/(k_vm_win for the batch expected next, into shadows)//; s/(k_vm_commit + render pass in k_vm_win.s place)//' | cut -c1-260 )
done; done

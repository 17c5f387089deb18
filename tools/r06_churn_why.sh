cd ${GRAFT_REPO_ROOT:-$(pwd)}
pre="$PWD/audiality2_amd/liba2amd_walk.so $PWD/audiality2_amd/liba2amd_units.so"
for prog in OscFilterPanChurn OscFilterPan; do for buf in 4096 64; do
  echo "== $prog 16384 a2_Run($buf)"
  ( cd tests/a2s; LD_PRELOAD="$pre" A2REF_BUFFER=$buf A2AMD_HOSTTIMING=1 A2AMD_WALK_STATS=1 ../../oracle/_ref/ref_bench bench.a2s $prog 16384 $([ $buf = 64 ] && echo 3000 || echo 8192) 1 2>&1 | grep -v "^a2amd device VM: [0-9]* spec" | tail -8 | cut -c1-420 )
done; done

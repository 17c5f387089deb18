R=$GRAFT_REPO_ROOT
cd /tmp
pre="$R/audiality2_amd/liba2amd_walk.so $R/audiality2_amd/liba2amd_units.so"
for buf in 4096 64; do
  LD_PRELOAD="$pre" A2REF_BUFFER=$buf A2AMD_HOSTTIMING=1 $R/oracle/_ref/ref_bench $R/tests/a2s/bench.a2s OscFilterPanChurn 16384 8192 1 2>&1 | grep -v "uploads by\|device VM" | tail -3 | cut -c1-330
done

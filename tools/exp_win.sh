R=$GRAFT_REPO_ROOT
cd /tmp
pre="$R/build_variants/liba2amd_prof.so $R/audiality2_amd/liba2amd_walk.so $R/audiality2_amd/liba2amd_units.so"
for prog in OscPanScripted; do
  LD_PRELOAD="$pre" A2REF_BUFFER=4096 $R/oracle/_ref/ref_bench $R/tests/a2s/bench.a2s $prog 16384 1024 1 2>&1 | grep -A1 "k_vm_win" | tail -8 | cut -c1-300
done

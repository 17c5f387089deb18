R=$GRAFT_REPO_ROOT
cd $R && timeout 1200 python -m pytest tests/test_device_vm.py -m gpu -x -q 2>&1 | tail -12
cd /tmp
pre="$R/audiality2_amd/liba2amd_walk.so $R/audiality2_amd/liba2amd_units.so"
for vw in 1 0; do
for prog in OscPanScripted OscFilterPanScripted; do
  echo "VMWIN=$vw $prog"
  LD_PRELOAD="$pre" A2AMD_VMWIN=$vw A2REF_BUFFER=4096 A2AMD_HOSTTIMING=1 $R/oracle/_ref/ref_bench $R/tests/a2s/bench.a2s $prog 16384 8192 1 2>&1 | grep -v "uploads by first" | tail -4 | cut -c1-330
done
done

#!/bin/bash
# after the soak's finding (seed 2635): the whole GPU suite again (with the launch classes checked at every rebuild), a wider
# soak of the looping-voice families with the window kernels forced, the churn cell
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/round5_final
mkdir -p $OUT
cd $REPO
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/gpu_suite_b.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $OUT/gpu_suite_b.txt
O=$OUT/final_soak_3.txt; : > $O
run() { echo "== $*" >> $O; "$@" 2>&1 | tail -3 >> $O; }
export A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2AMD_CLS_CHECK=1
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 2024 2200
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 2760 3000
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 3016 3200
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 3420 3700
unset A2AMD_WIN A2AMD_WIN_CHECK A2AMD_CLS_CHECK
pre="$REPO/audiality2_amd/liba2amd_walk.so $REPO/audiality2_amd/liba2amd_units.so"
for buf in 4096 64; do
  LD_PRELOAD="$pre" A2REF_BUFFER=$buf A2AMD_HOSTTIMING=1 $REPO/oracle/_ref/ref_bench $REPO/tests/a2s/bench.a2s OscFilterPanChurn 16384 8192 1 2>&1 | grep -v "uploads by\|device VM" | tail -3 | cut -c1-330 >> $OUT/churn_final.txt
done
cat $OUT/gpu_suite_b.txt $O $OUT/churn_final.txt

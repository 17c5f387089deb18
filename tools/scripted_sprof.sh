# sampling profile of the engine thread (tools/ubench/sprof.c):  PROG=... V=... bash tools/scripted_sprof.sh
cd $GRAFT_REPO_ROOT
gcc -O2 -shared -fPIC -o /tmp/libsprof.so tools/ubench/sprof.c -ldl -lpthread
cd tests/a2s
P=${PROG:-OscPanScripted}; V=${V:-32768}
SPROF_DELAY_MS=${DELAY:-2500} A2REF_BUFFER=64 LD_PRELOAD=/tmp/libsprof.so:../../audiality2_amd/liba2amd_units.so timeout 300 ../../oracle/_ref/ref_bench bench.a2s $P $V ${FR:-1500} 1 > $GRAFT_REPO_ROOT/gpurun_out/sprof_$P.log 2>&1
python $GRAFT_REPO_ROOT/tools/ubench/sprof_resolve.py $GRAFT_REPO_ROOT/gpurun_out/sprof_$P.log | head -${TOP:-30}
grep "^sprof" $GRAFT_REPO_ROOT/gpurun_out/sprof_$P.log | sort -k2 -n -r | head -14
A2REF_BUFFER=64 A2AMD_HOSTTIMING=2 LD_PRELOAD=../../audiality2_amd/liba2amd_units.so timeout 300 ../../oracle/_ref/ref_bench bench.a2s $P $V 300 1 2>&1 | grep -v "^a2amd units\|^a2amd host" | tail -4

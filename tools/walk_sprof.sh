# sampling profile of the engine thread behind the replaced voice walk:  PROG=... V=... bash tools/walk_sprof.sh
cd $GRAFT_REPO_ROOT
gcc -O2 -shared -fPIC -o /tmp/libsprof.so tools/ubench/sprof.c -ldl -lpthread
cd tests/a2s
P=${PROG:-FilterTree}; V=${V:-262144}
SPROF_DELAY_MS=${DELAY:-9000} A2REF_BUFFER=${BUF:-4096} LD_PRELOAD=/tmp/libsprof.so:../../audiality2_amd/liba2amd_walk.so:../../audiality2_amd/liba2amd_units.so timeout 300 ../../oracle/_ref/ref_bench bench.a2s $P $V ${FR:-8192} 1 > $GRAFT_REPO_ROOT/gpurun_out/sprof_walk_$P.log 2>&1
python $GRAFT_REPO_ROOT/tools/ubench/sprof_resolve.py $GRAFT_REPO_ROOT/gpurun_out/sprof_walk_$P.log | head -${TOP:-40}
tail -n 2 $GRAFT_REPO_ROOT/gpurun_out/sprof_walk_$P.log

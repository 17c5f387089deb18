# host-side sampling profile of the engine + drop-in:  PROG=... V=... BUF=... bash tools/scr_probe.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
gcc -O2 -shared -fPIC -o /tmp/libsprof.so tools/ubench/sprof.c -ldl -lrt -lpthread
cd tests/a2s
B=../../oracle/_ref/ref_bench
U=../../audiality2_amd/liba2amd_units.so
A2REF_BUFFER=${BUF:-4096} SPROF_DELAY_MS=${DELAY:-1500} LD_PRELOAD=/tmp/libsprof.so:$U timeout 200 $B bench.a2s ${PROG:-OscPanScripted} ${V:-16384} ${FR:-2048} 1 2> ../../gpurun_out/sprof.log | tail -1
grep -c sprof ../../gpurun_out/sprof.log

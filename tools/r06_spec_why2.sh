cd ${GRAFT_REPO_ROOT:-$(pwd)}
pre="$PWD/audiality2_amd/liba2amd_walk.so $PWD/audiality2_amd/liba2amd_units.so"
cd tests/a2s
echo "== with the walk"; LD_PRELOAD="$pre" A2REF_BUFFER=4096 A2AMD_ROOTTRACE=1 ../../oracle/_ref/ref_bench bench.a2s OscPanScripted 16384 4096 1 2>&1 | grep "root window" | tail -14 | cut -c1-200
echo "== units only, no device VM"; LD_PRELOAD="$PWD/../../audiality2_amd/liba2amd_units.so" A2REF_BUFFER=4096 A2AMD_ROOTTRACE=1 ../../oracle/_ref/ref_bench bench.a2s OscPanScripted 1024 2048 1 2>&1 | grep "root window" | tail -8 | cut -c1-200

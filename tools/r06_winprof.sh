#!/bin/bash
# round 6: cycle counters inside k_vm_win: where a lane's time goes.  Needs a -DWIN_PROF build of the library next to
# copies of the other two (the variant is not kept in the tree):
#   mkdir -p build_variants/winprof && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DWIN_PROF \
#     '-DA2AMD_SRCHASH="winprof"' -o build_variants/winprof/liba2amd.so audiality2_amd/csrc/*.cpp audiality2_amd/csrc/*.hip \
#     && cp audiality2_amd/liba2amd_units.so audiality2_amd/liba2amd_walk.so build_variants/winprof/
cd ${GRAFT_REPO_ROOT:-$(pwd)}
V=$PWD/build_variants/winprof
pre="$V/liba2amd_walk.so $V/liba2amd_units.so"
for prog in OscFilterPanScripted OscPanScripted; do
  echo "== $prog (speculation off: one pass per batch, in line)"
  ( cd tests/a2s; A2AMD_VMSPEC=0 LD_PRELOAD="$pre" A2REF_BUFFER=4096 timeout 120 ../../oracle/_ref/ref_bench bench.a2s $prog 16384 1024 1 2>&1 | grep "k_vm_win\|\.\.\. \|voice_samples" | tail -12 | cut -c1-300 )
done

#!/bin/bash
# build variants of liba2amd.so with different shapes of k_leaf_oscfiltpan (tools/filt_sweep_run.sh times them)
cd "$(dirname "$0")/.."
V=tools/ubench/variants; mkdir -p $V
build() { # tag, flags...
  local tag=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w "$@" -o $V/liba2amd_$tag.so \
    audiality2_amd/csrc/a2amd_host.cpp audiality2_amd/csrc/a2amd_sched.cpp audiality2_amd/csrc/a2amd_render.cpp audiality2_amd/csrc/a2amd_dist.cpp audiality2_amd/csrc/a2amd_kernels.hip audiality2_amd/csrc/a2amd_fast.hip &
}
for spec in "$@"; do
  # spec: W<waves>B<batch>P<partner>E<waves per eu>
  w=$(echo $spec | sed -E 's/W([0-9]+).*/\1/'); b=$(echo $spec | sed -E 's/.*B([0-9]+).*/\1/')
  p=$(echo $spec | sed -E 's/.*P([0-9]+).*/\1/'); e=$(echo $spec | sed -E 's/.*E([0-9]+).*/\1/')
  r=1; echo $spec | grep -q R0 && r=0; echo $spec | grep -q R2 && r=2
  x=""; echo $spec | grep -q prof && x="-DFILT_PROF"
  l=$(echo $spec | sed -nE 's/.*L(m?[0-9]+).*/\1/p' | sed 's/m/-/'); [ -n "$l" ] && x="$x -DFILT_LIGHT=$l"
  build $spec -DFILT_WAVES=$w -DFILT_BATCH=$b -DFILT_PARTNER=$p -DFILT_WPE=$e -DFILT_ROT=$r $x
done
wait
ls -la $V/*.so

#!/bin/bash
# build a variant of liba2amd.so into tools/ubench/variants:  tools/variant_build.sh <tag> <-D flags...>
cd "$(dirname "$0")/.."
V=tools/ubench/variants; mkdir -p $V
tag=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w "$@" -o $V/liba2amd_$tag.so \
  audiality2_amd/csrc/a2amd_host.cpp audiality2_amd/csrc/a2amd_sched.cpp audiality2_amd/csrc/a2amd_render.cpp audiality2_amd/csrc/a2amd_dist.cpp audiality2_amd/csrc/a2amd_vm.cpp audiality2_amd/csrc/a2amd_kernels.hip audiality2_amd/csrc/a2amd_fast.hip audiality2_amd/csrc/a2amd_vm.hip audiality2_amd/csrc/a2amd_wavecap.hip

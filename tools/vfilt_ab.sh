#!/bin/bash
# RECS_VFILT A/B: scripted filter voices through the C ABI (kernel times), the song, variant 3b with the engine in the loop
V=$PWD/tools/ubench/variants
for lib in "" $V/liba2amd_novf.so; do
  echo "== lib: ${lib:-this build}"
  for chain in osc-filter-pan osc2-filter-pan osc-pan; do
    A2AMD_LIB=$lib python tools/scripted_timing.py --chain $chain --voices 16384 --batch 64 2>&1 | tail -1
  done
  A2AMD_LIB=$lib python tools/scripted_timing.py --chain osc-filter-pan --voices 65536 --batch 64 2>&1 | tail -1
  A2AMD_LIB=$lib python tools/scripted_timing.py --chain osc-filter-pan --voices 1024 --batch 64 2>&1 | tail -1
done
echo "== song, this build"; python tests/measure/song_timing.py --seconds 500 2>&1 | tail -2

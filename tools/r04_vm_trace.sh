#!/bin/bash
cd tests/a2s
W="../../audiality2_amd/liba2amd_walk.so ../../audiality2_amd/liba2amd_units.so"
for n in 1024 16384; do
A2REF_BUFFER=4096 A2AMD_WALK_STATS=1 A2AMD_VM_TRACE=1 LD_PRELOAD="$W" ../../oracle/_ref/ref_bench bench.a2s OscPanScripted $n 512 1 2>&1 | sort | uniq -c | sort -rn | head -12
done

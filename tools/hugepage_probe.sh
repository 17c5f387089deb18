#!/bin/bash
# Does the engine's own voice walk (pointer chasing over ~1.4 KB A2_voice structs) get
# cheaper when glibc's heap is backed by transparent huge pages?  One engine state,
# buffer 64, with and without GLIBC_TUNABLES=glibc.malloc.hugetlb=1; CPU units and drop-in.
cd "$(dirname "$0")/../tests/a2s" || exit 1
B=../../oracle/_ref/ref_bench
U=../../audiality2_amd/liba2amd_units.so
echo "thp: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null)"
for prog in OscPan Osc2PanGroups; do
  for v in 32768 65536; do
    for tun in "" "glibc.malloc.hugetlb=1"; do
      for pre in "" "$U"; do
        frags=300; [ -z "$pre" ] && frags=20
        r=$(GLIBC_TUNABLES="$tun" LD_PRELOAD="$pre" A2REF_BUFFER=64 timeout 600 $B bench.a2s $prog $v $frags 1 2>/dev/null | tail -1)
        echo "{\"program\": \"$prog\", \"voices\": $v, \"tunable\": \"$tun\", \"dropin\": $([ -n "$pre" ] && echo true || echo false), \"result\": $r}"
      done
    done
  done
done

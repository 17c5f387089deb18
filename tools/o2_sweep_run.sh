#!/bin/bash
# time configs[3] (golden-gated) with every variant tools/ubench/variants/liba2amd_o2_*.so and a few launch shapes
cd "$(dirname "$0")/.."
O=gpurun_out/o2_sweep; mkdir -p $O; : > $O/sweep.jsonl
run() { # tag, env...
  local tag=$1; shift
  out=$(env "$@" python bench.py --config 3 --steps 12 --warmup 3 --no-extra --no-cpu-baseline --no-engine --no-realtime 2>&1 | tail -n 1)
  echo "$out" | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(json.dumps({'variant':'$tag','value':d['value'],'ms_per_step':d['ms_per_step'],'leaf_ms':d['roofline']['avg_launch_ms'],'parity':d['parity_vs_golden']}))
except Exception as e:
    print(json.dumps({'variant':'$tag','error':str(e)}))
" >> $O/sweep.jsonl
}
for lib in tools/ubench/variants/liba2amd_o2_*.so; do
  tag=$(basename $lib .so | sed 's/liba2amd_//')
  run $tag A2AMD_LIB=$PWD/$lib
done
for s in "16 32" "32 32" "32 16" "64 32" "32 64" "16 64" "8 32"; do
  set -- $s
  run "default_vpw$1_y$2" A2AMD_VPW=$1 A2AMD_YSPLIT=$2
done
cat $O/sweep.jsonl

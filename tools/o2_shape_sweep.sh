#!/bin/bash
# configs[3] (golden-gated) over launch shapes of k_leaf_osc2pan: voices per wavefront x time slices
cd "$(dirname "$0")/.."
O=gpurun_out/o2_sweep; mkdir -p $O; : > $O/shapes.jsonl
for s in ${SHAPES:-"32 32" "32 8" "32 4" "16 8" "16 4" "8 8" "8 4" "16 16" "64 8" "64 16" "64 4" "32 2" "16 2"}; do
  set -- $s
  A2AMD_VPW=$1 A2AMD_YSPLIT=$2 python bench.py --config 3 --steps 12 --warmup 3 --no-extra --no-cpu-baseline --no-engine --no-realtime 2>&1 | tail -n 1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(json.dumps({'vpw':$1,'ysplit':$2,'value':d['value'],'ms_per_step':d['ms_per_step'],'leaf_ms':d['roofline']['avg_launch_ms'],'parity':d['parity_vs_golden']}))
except Exception as e:
    print(json.dumps({'vpw':$1,'ysplit':$2,'error':str(e)}))
" >> $O/shapes.jsonl
done
cat $O/shapes.jsonl

#!/bin/bash
# time configs[2] (16 384 x wtosc->filter12->panmix, golden-gated) with every variant built by filt_sweep_build.sh
cd "$(dirname "$0")/.."
O=gpurun_out/filt_sweep; mkdir -p $O; : > $O/sweep.jsonl
for lib in tools/ubench/variants/liba2amd_W*.so; do
  tag=$(basename $lib .so | sed 's/liba2amd_//')
  for fvpw in ${FVPWS:-32 48 64}; do
    out=$(A2AMD_LIB=$PWD/$lib A2AMD_FVPW=$fvpw python bench.py --config ${CFG:-2} --steps 20 --warmup 3 --no-extra --no-cpu-baseline --no-engine --no-realtime 2>&1 | tail -n 1)
    echo "$out" | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(json.dumps({'variant':'$tag','fvpw':$fvpw,'value':d['value'],'ms_per_step':d['ms_per_step'],'leaf_ms':d['roofline']['avg_launch_ms'],'parity':d['parity_vs_golden']}))
except Exception as e:
    print(json.dumps({'variant':'$tag','fvpw':$fvpw,'error':str(e)}))
" >> $O/sweep.jsonl
  done
done
cat $O/sweep.jsonl

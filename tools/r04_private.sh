#!/bin/bash
# the private-wave scene (bench.py --config 5): parity at oracle size, then coefficient entries vs samples at full size
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "private_sample" 2>&1 | tail -3
for raw in 1 0 ""; do
  echo "== A2AMD_RAW=${raw:-default}"
  A2AMD_RAW=$raw python bench.py --config 5 --steps 20 --no-cpu-baseline --no-realtime 2> /tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'parity':d['parity_vs_golden'],'golden_steps':d['parity']['golden_steps_compared'],'leaf_ms':r['avg_launch_ms'],'alg_bytes':r['algorithmic_bytes_per_launch'],'alg_B_per_vs':r['algorithmic_bytes_per_voice_sample'],'achieved_GBps':r['achieved'],'frac':r['frac']}))"
  tail -2 /tmp/err.txt
done

#!/bin/bash
cd /root/repo
python -m pytest tests/test_dropin.py -x -q -m gpu -k "rendered_waves or edge or unload" 2>&1 | tail -15 > gpurun_out/r04_f3.txt
A2AMD_BENCH_FORCE_DIST=1 python bench.py --steps 8 --warmup 1 --no-engine --no-cpu-baseline > gpurun_out/r04_forcedist.json 2> gpurun_out/r04_forcedist.err; tail -3 gpurun_out/r04_forcedist.err >> gpurun_out/r04_f3.txt
python - >> gpurun_out/r04_f3.txt <<'PY'
import json
d = json.loads(open("gpurun_out/r04_forcedist.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "parity_vs_golden", "parity")})
print(d.get("fragment_roundtrip_with_reduce")); print(d.get("cpu_baseline"))
PY
cat gpurun_out/r04_f3.txt

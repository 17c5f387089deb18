#!/bin/bash
O=gpurun_out/r03d; mkdir -p $O
python -m pytest tests/test_dropin.py -q -x -k "engine_in_loop_at or walk" > $O/t_walk.log 2>&1; echo "walk tests rc $?" > $O/summary
python - > $O/engine_scripted.json 2> $O/engine.err <<'PY'
import json, bench
cases = [c for c in bench.ENGINE_CASES if c[2] >= 16384]
print(json.dumps(bench.engine_in_loop(cases)))
PY
python - > $O/rt_sweep.json 2>> $O/engine.err <<'PY'
import json, bench, audiality2_amd
print(json.dumps(bench.realtime_sweep(audiality2_amd)))
PY
cat $O/summary; tail -n 3 $O/t_walk.log

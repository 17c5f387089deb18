#!/bin/bash
# round 3, second GPU pass: INTEGRATION option C (the replaced voice walk) - parity and speed
O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests/test_dropin.py -q -x -k "walk or second_context or engine_in_loop_at or matches_reference" > $O/t_walk.log 2>&1; echo "walk tests rc $?" >> $O/summary
python -m pytest tests/test_fuzz_dropin.py -q -x > $O/t_fuzz.log 2>&1; echo "fuzz rc $?" >> $O/summary
python - > $O/engine_in_loop.json 2> $O/engine_in_loop.err <<'PY'
import json, bench
print(json.dumps(bench.engine_in_loop()))
PY
echo "engine rc $?" >> $O/summary
cat $O/summary
tail -n 5 $O/t_walk.log $O/t_fuzz.log

#!/bin/bash
# the replaced voice walk: parity subset + speed at the big sizes
O=gpurun_out/r03c; mkdir -p $O; rm -f $O/summary
python -m pytest tests/test_dropin.py -q -x -k "walk and not ahead" > $O/t_walk.log 2>&1; echo "walk tests rc $?" >> $O/summary
python -m pytest tests/test_fuzz_dropin.py -q -x > $O/t_fuzz.log 2>&1; echo "fuzz rc $?" >> $O/summary
python - > $O/engine_in_loop.json 2> $O/engine_in_loop.err <<'PY'
import json, bench
cases = [c for c in bench.ENGINE_CASES if c[2] >= 16384]
print(json.dumps(bench.engine_in_loop(cases)))
PY
echo "engine rc $?" >> $O/summary
lscpu | grep -i "model name\|L2\|L3\|^CPU(s)" > $O/lscpu.txt
cat $O/summary
tail -n 4 $O/t_walk.log $O/t_fuzz.log

#!/bin/bash
# last check of a round on the GPU box: the whole GPU suite, smoke(), then the reference's songs replayed (soak_long/)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/gpu_suite.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/gpu_suite.txt
cat gpurun_out/gpu_suite.txt
if [ -d soak_long ]; then
  timeout 900 python tests/measure/soak_long.py replay > gpurun_out/soak_songs.jsonl 2> gpurun_out/soak_songs.err
  [ -d soak_long/testdata ] && timeout 900 python tests/measure/soak_long.py replay testdata > gpurun_out/soak_testdata.jsonl 2> gpurun_out/soak_testdata.err
  python - <<'PY'
import json
for fn in ("gpurun_out/soak_songs.jsonl", "gpurun_out/soak_testdata.jsonl"):
    try:
        rows = [json.loads(l) for l in open(fn) if l.startswith("{")]
    except OSError:
        continue
    print(fn, len(rows), "traces,", sum(r.get("fragments", 0) for r in rows), "fragments,", sum(r.get("differing", 0) for r in rows), "differing")
PY
fi

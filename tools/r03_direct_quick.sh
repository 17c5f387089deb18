#!/bin/bash
# gate + timing for the root kernel storing the master bus in the host's buffer (A2AMD_NO_DIRECT=1: the copy, as before)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_dropin.py -x -q -m gpu -k "not fuzz and not soak" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for e in "X=1" "A2AMD_NO_DIRECT=1"; do
 for cfg in 1 2 3; do env $e python bench.py --config $cfg --steps $([ $cfg = 1 ] && echo 400 || echo 30) --warmup 10 --no-extra --no-cpu-baseline --no-engine 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e cfg$cfg', '%.4g'%d['value'], '%.4f ms'%d['ms_per_step'], d['parity_vs_golden'], 'realtime p50/p99 ms', d.get('realtime',{}).get('fragment_ms_p50'), d.get('realtime',{}).get('fragment_ms_p99'))"; done; done

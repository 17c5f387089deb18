#!/bin/bash
# quick gate for leaf-kernel changes on the GPU box: parity subset, bench lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scenes_match or launch_shapes or baseline_configs or mixed_quiet or still_ramping or partial_fragments or config4_full or reference_traces or linearity or wave_drop or scene_fuzz or host_walk or groups_with" > $O/pytest_quick.log 2>&1; tail -3 $O/pytest_quick.log
python bench.py --no-engine --no-cpu-baseline > $O/bench3.json 2> $O/bench3.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03e/bench3.json'))
print('cfg3', d['value'], d['ms_per_step'], d['parity_vs_golden'], d['roofline']['avg_launch_ms'])
for k,x in d['other_configs'].items():
    if 'value' in x: print(k, x['value'], x['ms_per_step'], x.get('parity_vs_golden'), x['roofline']['avg_launch_ms'])
PY
for dbgv in 0 8; do
A2AMD_DEBUG=$dbgv python bench.py --config 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 share dbg$dbgv', d['value'], d['ms_per_step'], d['parity_vs_golden'], d['roofline']['avg_launch_ms'])"
A2AMD_DEBUG=$dbgv python bench.py --config 2 --steps 20 --warmup 3 --no-extra --no-cpu-baseline --no-engine --no-realtime 2>&1 | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 dbg$dbgv', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['parity_vs_golden'])"
done

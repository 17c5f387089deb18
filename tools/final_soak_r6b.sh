#!/bin/bash
# round 6, second soak - after the speculative pass went in front of the render pass, the quiet launches it makes
# unnecessary were dropped, run_batch's walk over chain positions changed and batches that took a pass for every list
# stopped clearing / reading the window pool's counter: the device-VM tests, the scripted cells hash-checked, then random
# scripts through the reference engine with and without the drop-in (tests/measure/fuzz_soak.py), new seeds, launch
# classes checked at every rebuild.  Legs as tools/final_soak_r6.sh.   (tools/gpu.sh 1800 'bash tools/final_soak_r6b.sh [short]')
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_final_soak_b.txt; : > $O
echo "#### tests/test_device_vm.py" >> $O; python -m pytest tests/test_device_vm.py -q -p no:cacheprovider 2>&1 | tail -3 >> $O
echo "#### bench.py's scripted cells, hash-checked against the CPU engine" >> $O
python tools/engine_cells.py "variant 2b" "variant 3b" "variant 2e" 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln)
    for b in ('a2_Run(4096)','a2_Run(64)'):
        m=d[b].get('units+walk',{})
        print(d['case'], b, 'vs/s %.4g' % m.get('voice_samples_per_s',0), 'p99 steady us/frag %.1f' % m.get('us_per_fragment_p99_steady',0), 'hash_equal', m.get('hash_equal'))
" >> $O
run() { echo "== $*" >> $O; "$@" 2>&1 | tail -2 >> $O; }
export A2AMD_CLS_CHECK=1
if [ "${1:-}" = short ]; then
  A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 run python tests/measure/fuzz_soak.py 7000 7060
  A2AMD_WIN=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 run python tests/measure/fuzz_soak.py 7100 7160
  A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 7200 7240
else
  A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 run python tests/measure/fuzz_soak.py 6000 6120
  A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 A2AMD_VMSKIP=3 run python tests/measure/fuzz_soak.py 6200 6260
  A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 A2AMD_VMSPEC_EARLY=0 run python tests/measure/fuzz_soak.py 6300 6340
  A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 6400 6480
  run python tests/measure/fuzz_soak.py 6500 6530
  A2AMD_WIN=1 A2AMD_WIN_CHECK=1 A2FUZZ_WALK=1 A2AMD_VMSPEC_MIN=1 A2AMD_DEVICES=2 run python tests/measure/fuzz_soak.py 6600 6640
fi
cat $O

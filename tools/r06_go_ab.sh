#!/bin/bash
# round 6, after run_batch's walk over occupied chain positions only: the scripted cells again
cd ${GRAFT_REPO_ROOT:-$(pwd)}
pre="$PWD/audiality2_amd/liba2amd_walk.so $PWD/audiality2_amd/liba2amd_units.so"
echo "#### device VM tests"; python -m pytest tests/test_device_vm.py -q -x -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do for prog in OscPanScripted OscFilterPanScripted OscPanEnvScripted; do for buf in 4096 1024; do for mode in "A2AMD_X=0" "A2AMD_VMSKIP=3"; do
  echo "== $prog a2_Run($buf) $mode"
  ( cd tests/a2s; env $mode LD_PRELOAD="$pre" A2REF_BUFFER=$buf A2AMD_HOSTTIMING=1 timeout 120 ../../oracle/_ref/ref_bench bench.a2s $prog 16384 12288 1 2>&1 | grep "voice_samples\|quiet-kernel" | sed 's/.*waits for a pass in flight before a rewrite; //' | cut -c1-330 )
done; done; done; done
echo "#### hash check against the CPU engine (bench.py's cells), default settings"
python tools/engine_cells.py "variant 2b" "variant 3b" "variant 2e" 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln)
    for b in ('a2_Run(4096)','a2_Run(64)'):
        m=d[b].get('units+walk',{})
        print(d['case'], b, 'vs/s %.4g' % m.get('voice_samples_per_s',0), 'p99 steady us/frag %.1f' % m.get('us_per_fragment_p99_steady',0), 'hash_equal', m.get('hash_equal'))
"
bash tools/r06_timeline.sh

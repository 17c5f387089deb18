#!/bin/bash
# round 6: the speculative VM pass (vm_speculate) - (1) the device-VM tests; (2) the scripted engine cells, 16 384 voices,
# units + walk, with the pass on / off: statistics line (A2AMD_HOSTTIMING) and the harness's own timing; (3) the same two
# cells as bench.py reports them, hash-checked against the CPU engine
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "#### tests/test_device_vm.py"
python -m pytest tests/test_device_vm.py -q -x -p no:cacheprovider 2>&1 | tail -12
echo "#### scripted cells, 16 384 voices, units + walk: A2AMD_VMSPEC=1 / 0"
pre="$PWD/audiality2_amd/liba2amd_walk.so $PWD/audiality2_amd/liba2amd_units.so"
for prog in OscPanScripted OscFilterPanScripted; do for buf in 4096 1024 64; do for sp in 1 0; do
  echo "== $prog a2_Run($buf) A2AMD_VMSPEC=$sp"
  ( cd tests/a2s; LD_PRELOAD="$pre" A2AMD_VMSPEC=$sp A2REF_BUFFER=$buf A2AMD_HOSTTIMING=1 ../../oracle/_ref/ref_bench bench.a2s $prog 16384 $([ $buf = 64 ] && echo 3000 || echo 12288) 1 2>&1 | grep -v "uploads by first" | grep "speculative\|batches with device\|voice_samples\|per driver buffer" | cut -c1-520 )
done; done; done
echo "#### the same two cells as bench.py reports them (hash-checked against the CPU engine)"
for sp in 1 0; do echo "== A2AMD_VMSPEC=$sp"; A2AMD_VMSPEC=$sp python tools/engine_cells.py "variant 2b" "variant 3b" 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln)
    for b in ('a2_Run(4096)','a2_Run(64)'):
        m=d[b].get('units+walk',{})
        print(d['case'], b, 'vs/s %.4g' % m.get('voice_samples_per_s',0), 'p50 us/frag %.1f p99 steady %.1f' % (m.get('us_per_fragment_p50',0), m.get('us_per_fragment_p99_steady',0)), 'hash_equal', m.get('hash_equal'))
"; done

#!/bin/bash
for n in 4096 16384 65536; do
  for vf in 1 0; do
    echo "VFILT=$vf: $(A2AMD_VFILT=$vf python tools/scripted_timing.py --chain osc-filter-pan --voices $n --batch 64 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['voices'], d['scripted'], d['scripted2'])")"
  done
done
for v in 2 4 8 16; do echo "vpw $v: $(A2AMD_VFILT=1 A2AMD_RVPW=$v python tools/scripted_timing.py --chain osc-filter-pan --voices 16384 --batch 64 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['scripted2'])")"; done
echo "osc2-filter: $(python tools/scripted_timing.py --chain osc2-filter-pan --voices 16384 --batch 64 2>&1 | tail -1 | cut -c1-400)"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "record or filter or fuzz or traces" 2>&1 | tail -3

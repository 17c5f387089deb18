#!/bin/bash
# fuzz: the suite's env seeds, then a soak of seeds 3016..3216 behind the walk + device VM and 3016..3116 units only
cd /root/repo
python -m pytest tests/test_fuzz_dropin.py -x -q -m gpu -k "30" 2>&1 | tail -5 > gpurun_out/r04_soak_env.txt
A2FUZZ_WALK=1 python tests/measure/fuzz_soak.py 3016 3216 2>&1 | tail -4 >> gpurun_out/r04_soak_env.txt
python tests/measure/fuzz_soak.py 3016 3116 2>&1 | tail -4 >> gpurun_out/r04_soak_env.txt
A2FUZZ_WALK=1 A2AMD_DEVICES=2 python tests/measure/fuzz_soak.py 3016 3066 2>&1 | tail -4 >> gpurun_out/r04_soak_env.txt
cat gpurun_out/r04_soak_env.txt

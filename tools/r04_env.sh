#!/bin/bash
cd /root/repo
python -m pytest tests/test_device_vm.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_env.txt
python -m pytest tests/test_dropin.py -x -q -m gpu -k "envwire or song or scripted" 2>&1 | tail -8 >> gpurun_out/r04_env.txt
python -m pytest tests/test_fuzz_dropin.py -x -q -m gpu 2>&1 | tail -8 >> gpurun_out/r04_env.txt
cat gpurun_out/r04_env.txt

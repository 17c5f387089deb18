mkdir -p gpurun_out
{
python tests/measure/fuzz_one.py 3779
timeout 600 python -m pytest tests/test_device_vm.py tests/test_fuzz_dropin.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_dropin.py -m gpu -x -q -k "env or fuzz or engine_with_dropin_units_matches_reference" 2>&1 | tail -3
A2FUZZ_WALK=1 python tests/measure/fuzz_soak.py 3750 3800 2>&1 | tail -2
A2FUZZ_WALK=1 python tests/measure/fuzz_soak.py 3900 3960 2>&1 | tail -2
python tests/measure/fuzz_soak.py 3100 3150 2>&1 | tail -2
} > gpurun_out/envfix.txt 2>&1
cat gpurun_out/envfix.txt

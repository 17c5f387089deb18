for b in 1 64; do for r in 1 2 3 4 8; do echo "batch $b rvpw $r: $(A2AMD_RVPW=$r python tools/scripted_timing.py --voices 16384 --batch $b --what pitch | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['scripted2']['kernels_ms_per_batch'], d['quiet2']['kernels_ms_per_batch'])")"; done; done
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "records" 2>&1 | tail -2

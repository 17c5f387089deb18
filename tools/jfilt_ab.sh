#!/bin/bash
# RECS_JFILT (the filter window as a relaxation along the lanes): parity of the record paths, then
# scripted filter voices through the C ABI (kernel times) and the song
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "records or traces or fuzz or partial or mixed_quiet or wave_drop" 2>&1 | tail -4
echo "== scripted filter voices (A2AMD_VFILT=0: the window filter for every launch)"
for n in 64 1024 4096 16384; do
  for vf in 0 1; do
    echo -n "voices $n VFILT=$vf: "; A2AMD_VFILT=$vf timeout 300 python tools/scripted_timing.py --chain osc-filter-pan --voices $n --batch 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['scripted']['kernels_ms_per_batch'], d['quiet2']['kernels_ms_per_batch'])"
  done
done
echo -n "osc2-filter 16384 default: "; timeout 300 python tools/scripted_timing.py --chain osc2-filter-pan --voices 16384 --batch 64 2>&1 | tail -1 | cut -c1-400
echo "== song"; timeout 600 python tests/measure/song_timing.py --seconds 500 2>&1 | tail -2
} > gpurun_out/jfilt_ab.txt 2>&1
cat gpurun_out/jfilt_ab.txt

mkdir -p gpurun_out
{
for chain in osc-filter-pan osc2-filter-pan; do
for n in 6144 8192 12288 16384 32768; do
  for vf in 0 1; do
    echo -n "$chain voices $n VFILT=$vf: "; A2AMD_VFILT=$vf timeout 300 python tools/scripted_timing.py --chain $chain --voices $n --batch 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['scripted']['kernels_ms_per_batch'], d['quiet2']['kernels_ms_per_batch'])"
  done
done
done
} > gpurun_out/vf_threshold.txt 2>&1
cat gpurun_out/vf_threshold.txt

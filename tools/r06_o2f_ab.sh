P='import json,sys; d=json.loads(sys.stdin.read()); print("%.4g vs/s  %.4f ms/step  kernel %s %.4f ms  parity %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["parity_vs_golden"]))'
echo "#### A/B configs[2]: round 5's library (k_leaf_oscfiltpan 39e9eb94) vs this tree's (97b90664), same box, interleaved"
for i in 1 2 3; do
  echo -n "r5  run $i: "; A2AMD_LIB=$PWD/build_variants/liba2amd_r5.so python bench.py --config 2 --steps 40 --warmup 5 --no-cpu-baseline --no-realtime 2>/dev/null | python -c "$P"
  echo -n "now run $i: "; python bench.py --config 2 --steps 40 --warmup 5 --no-cpu-baseline --no-realtime 2>/dev/null | python -c "$P"
done
echo "#### config 6 (16 384 x 2xwtosc->filter12->panmix), k_leaf_osc2filtpan with FILT2_FASTV 3: default shape and neighbours"
for v in 0 16 24 32 36; do echo -n "F2VPW $v: "; A2AMD_F2VPW=$v python bench.py --config 6 --steps 7 --warmup 1 --no-cpu-baseline --no-realtime 2>/dev/null | python -c "$P"; done
echo -n "the same scene through round 5's library (window kernels): "; A2AMD_LIB=$PWD/build_variants/liba2amd_r5.so python bench.py --config 6 --steps 7 --warmup 1 --no-cpu-baseline --no-realtime 2>/dev/null | python -c "$P"

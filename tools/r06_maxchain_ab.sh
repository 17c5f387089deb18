cd ${GRAFT_REPO_ROOT:-$(pwd)}
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.4g vs/s  %.4f ms/step  kernel %s %.4f ms  parity %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["parity_vs_golden"]))'
echo "#### A2D_MAXCHAIN 16 (shipped) vs 8 (build_variants/liba2amd_chain8.so: 64 bytes), same box, interleaved"
for c in 2 3; do for i in 1 2 3; do
  echo -n "config $c chain16 run $i: "; python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-realtime --no-extra --no-engine 2>/dev/null | python -c "$P"
  echo -n "config $c chain8  run $i: "; A2AMD_LIB=$PWD/build_variants/liba2amd_chain8.so python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-realtime --no-extra --no-engine 2>/dev/null | python -c "$P"
done; done

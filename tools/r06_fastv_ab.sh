#!/bin/bash
# round 6: (1) k_leaf_osc2filtpan built with FILT2_FASTV 3 (shipped) vs 4 (build_variants/), same box, same shape;
# (2) the device-VM tests with the speculative pass; (3) the scripted engine cells with the pass on / off
cd ${GRAFT_REPO_ROOT:-$(pwd)}
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.4g vs/s  %.4f ms/step  kernel %s %.4f ms  parity %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["parity_vs_golden"]))'
echo "#### config 6 at 32 voices per workgroup: FILT2_FASTV 3 (shipped) vs 4 (variant), interleaved"
for i in 1 2 3; do
  echo -n "fastv3 run $i: "; A2AMD_F2VPW=32 python bench.py --config 6 --steps 7 --warmup 1 --no-cpu-baseline --no-realtime 2>/dev/null | python -c "$P"
  echo -n "fastv4 run $i: "; A2AMD_LIB=$PWD/build_variants/liba2amd_fastv4.so A2AMD_F2VPW=32 python bench.py --config 6 --steps 7 --warmup 1 --no-cpu-baseline --no-realtime 2>/dev/null | python -c "$P"
done

#!/bin/bash
# the last GPU call of round 5: the bench line on the final library, then a second soak (new seed ranges, window
# kernels forced; one leg over three contexts) - outputs under gpurun_out/round5_final/
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/round5_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd $REPO
O=$OUT/final_soak_2.txt; : > $O
run() { echo "== $*" >> $O; "$@" 2>&1 | tail -3 >> $O; }
export A2AMD_WIN=1 A2AMD_WIN_CHECK=1
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 2500 2700
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 3200 3350
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 1750 1850
A2FUZZ_WALK=1 A2AMD_DEVICES=3 run python tests/measure/fuzz_soak.py 2700 2760
unset A2AMD_WIN A2AMD_WIN_CHECK
A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py 3350 3420
run python tests/measure/fuzz_soak.py 400 450
cat $O
cut -c1-400 $OUT/bench_default.json

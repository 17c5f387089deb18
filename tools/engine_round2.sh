#!/bin/bash
# The engine in the loop, round 2 (one GPU round trip per a2_Run() buffer):
#   gpurun -- 'bash tools/engine_round2.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/engine2
mkdir -p $OUT
cd $REPO
python tests/measure/dropin_timing.py --buffer 4096 > $OUT/dropin_timing.jsonl 2>$OUT/dropin_timing.err
python tests/measure/dropin_timing.py --buffer 64 >> $OUT/dropin_timing.jsonl 2>>$OUT/dropin_timing.err
python tests/measure/song_timing.py > $OUT/song_timing.jsonl 2>$OUT/song_timing.err
A2REF_BUFFER=4096 python tests/measure/engine_in_loop.py > $OUT/engine_in_loop_b4096.jsonl 2>$OUT/engine_in_loop.err
A2REF_BUFFER=64 python tests/measure/engine_in_loop.py > $OUT/engine_in_loop_b64.jsonl 2>>$OUT/engine_in_loop.err
cat $OUT/dropin_timing.jsonl $OUT/song_timing.jsonl; head -40 $OUT/engine_in_loop_b4096.jsonl; tail -3 $OUT/*.err

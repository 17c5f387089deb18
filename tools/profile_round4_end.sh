#!/bin/bash
# the end of round 4: tools/profile_round4.sh on the final library, then what the records kernels' new filter window
# moved - the song (BASELINE configs[0]'s command shape) and the scripted batches
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/round4
bash $REPO/tools/profile_round4.sh > /dev/null 2>&1
cd $REPO
timeout 900 python tests/measure/song_timing.py --seconds 500 2>&1 | tail -2 > $OUT/song_timing.jsonl
: > $OUT/scripted_timing.jsonl
for chain in osc-pan osc-filter-pan osc2-pan osc2-filter-pan; do
  for batch in 1 64; do
    timeout 300 python tools/scripted_timing.py --chain $chain --voices 16384 --batch $batch 2>&1 | tail -1 >> $OUT/scripted_timing.jsonl
  done
done
for n in 1024 4096 65536; do
  timeout 300 python tools/scripted_timing.py --chain osc-filter-pan --voices $n --batch 64 2>&1 | tail -1 >> $OUT/scripted_timing.jsonl
done
ls -la $OUT
cat $OUT/song_timing.jsonl

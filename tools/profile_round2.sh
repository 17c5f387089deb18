#!/bin/bash
# Measurement pass of round 2, to be run on the GPU box:
#   gpurun -- 'bash tools/profile_round2.sh [quick]'
# Everything lands in gpurun_out/round2/; copy what should be judged into profiles/.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/round2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
QUICK=${1:-}

# 1. the bench line (default = BASELINE configs[3], configs[1]/[2] as extra keys), and the
#    kernel trace of the same command
$B > $OUT/bench_default.json 2> $OUT/bench_default.err
rm -rf /tmp/prof_k; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- $B --no-cpu-baseline --no-realtime > /tmp/prof_k.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;

# 1b. scripted voices (SURVEY 8d variants 2b / 3b): kernel time of batches in which every second
#     voice-fragment carries records, records-executing leaf kernel vs the general kernel
( for b in 1 64; do for ch in osc-pan osc2-pan; do
    python $REPO/tools/scripted_timing.py --voices 16384 --chain $ch --batch $b --what pitch
    A2AMD_NO_FAST=64 python $REPO/tools/scripted_timing.py --voices 16384 --chain $ch --batch $b --what pitch | sed 's/^{/{"general_kernel_only": true, /'
  done; done ) > $OUT/scripted_timing.jsonl 2>/dev/null
rm -rf /tmp/prof_s; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $REPO/tools/scripted_timing.py --voices 16384 --batch 64 > /tmp/prof_s.log 2>&1
find /tmp/prof_s -name "*kernel_stats.csv" -exec cp {} $OUT/scripted_kernel_stats.csv \;

# 2. PMC passes (counters only, own runs) for the three single-GPU configs
rm -f $OUT/pmc_summary.txt
pmc() { # label, counters, bench args...
  local label=$1 ctr=$2; shift 2
  rm -rf /tmp/prof_p; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_p -- $B "$@" --no-cpu-baseline --no-extra --no-realtime > /tmp/prof_p.log 2>&1
  python $REPO/tools/pmc_summary.py /tmp/prof_p "$label" >> $OUT/pmc_summary.txt
}
for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
  pmc "osc2-pan/65536/256/256" "$ctr" --config 3 --steps 6 --warmup 2
  pmc "osc-filter-pan/16384/0/256" "$ctr" --config 2 --steps 8 --warmup 2
  pmc "osc-pan/1024/0/256" "$ctr" --config 1 --steps 16 --warmup 4
done
python $REPO/tools/pmc_to_json.py $OUT/pmc_summary.txt > $OUT/pmc.json
[ -n "$QUICK" ] && { ls -la $OUT; exit 0; }

# 3. the engine in the loop: drop-in plumbing, k2intro 500 s, realtime sweep
python $REPO/tests/measure/dropin_timing.py > $OUT/dropin_timing.jsonl 2>$OUT/dropin_timing.err
python $REPO/tests/measure/engine_in_loop.py > $OUT/engine_in_loop.jsonl 2>$OUT/engine_in_loop.err
python $REPO/tools/realtime_sweep.py --fragments 300 > $OUT/realtime_sweep.jsonl 2>/dev/null
ls -la $OUT

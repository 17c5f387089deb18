#!/bin/bash
# Measurement pass of a round, to be run on the GPU box:
#   gpurun -- 'bash tools/profile_round.sh'
# Everything lands in gpurun_out/round/; copy what should be judged into profiles/.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"

# 1. the bench line (default = BASELINE configs[1]) twice, and its kernel trace
$B > $OUT/bench_default.json 2> $OUT/bench_default.err
$B --no-cpu-baseline > $OUT/bench_default_2.json 2>/dev/null
rm -rf /tmp/prof_k; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- $B --no-cpu-baseline > /tmp/prof_k.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;

# 2. sweep over chains and sizes
rm -f $OUT/bench_sweep.jsonl
for c in osc-pan osc-filter-pan osc2-pan fm1-pan fm2-pan fm4-pan; do
  $B --chain $c --voices 1024 --steps 16 --warmup 4 2>/dev/null | tail -1 >> $OUT/bench_sweep.jsonl
  for v in 16384 65536 262144; do
    $B --chain $c --voices $v --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/bench_sweep.jsonl
  done
done
for c in fm3-pan fm3p-pan fm4p-pan fm2r-pan fm4r-pan; do
  $B --chain $c --voices 262144 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/bench_sweep.jsonl
done
$B --chain osc2-pan --voices 65536 --groups 256 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/bench_sweep.jsonl

# 3. kernel trace of an fm4 run
rm -rf /tmp/prof_f; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -- $B --chain fm4-pan --voices 262144 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/prof_f.log 2>&1
find /tmp/prof_f -name "*kernel_stats.csv" -exec cp {} $OUT/fm4_kernel_stats.csv \;

# 4. PMC passes (counters only, own runs)
rm -f $OUT/pmc_summary.txt
pmc() { # label, counters, bench args...
  local label=$1 ctr=$2; shift 2
  rm -rf /tmp/prof_p; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_p -- $B "$@" --no-cpu-baseline --no-parity > /tmp/prof_p.log 2>&1
  python $REPO/tools/pmc_summary.py /tmp/prof_p "$label" >> $OUT/pmc_summary.txt
}
for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
  pmc "osc-pan/1024/256" "$ctr" --steps 16 --warmup 8
  pmc "osc-pan/65536/256" "$ctr" --steps 8 --warmup 4 --voices 65536
  pmc "fm4-pan/262144/256" "$ctr" --steps 4 --warmup 2 --voices 262144 --chain fm4-pan
done

# 5. scripted FM, drop-in plumbing, realtime
python $REPO/tools/fm_scripted_timing.py --voices 16384 --chain fm4-pan > $OUT/fm_scripted.jsonl 2>/dev/null
python $REPO/tools/fm_scripted_timing.py --voices 65536 --chain fm4-pan --batch 16 >> $OUT/fm_scripted.jsonl 2>/dev/null
python $REPO/tools/fm_scripted_timing.py --voices 16384 --chain fm1-pan >> $OUT/fm_scripted.jsonl 2>/dev/null
python $REPO/tests/measure/dropin_timing.py > $OUT/dropin_timing.jsonl 2>/dev/null
python $REPO/tools/realtime_sweep.py --fragments 300 > $OUT/realtime_sweep.jsonl 2>/dev/null
ls -la $OUT

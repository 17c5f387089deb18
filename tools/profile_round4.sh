#!/bin/bash
# Measurement pass of round 4, to be run on the GPU box:
#   gpurun -- 'bash tools/profile_round4.sh'
# Everything lands in gpurun_out/round4/; what is to be judged is copied into profiles/ (r04_*).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/round4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"

# 0. the issue rate of the kernels' own instruction mixes (read off the shipped binary)
( cd $REPO && timeout 900 python tools/valu_mix.py > $OUT/valu_mix.json 2> $OUT/valu_mix.err )
cp $OUT/valu_mix.json $REPO/profiles/r04_valu_mix.json

# 1. PMC passes (counters only, own runs): the three single-GPU configs, configs[4]'s per-GPU share, and the
#    private-wave scene both ways (samples / coefficient entries)
rm -f $OUT/pmc_summary.txt
pmc() { # label, counters, bench args...
  local label=$1 ctr=$2; shift 2
  rm -rf /tmp/prof_p; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_p -- $B "$@" --no-cpu-baseline --no-extra --no-realtime --no-engine > /tmp/prof_p.log 2>&1
  python $REPO/tools/pmc_summary.py /tmp/prof_p "$label" >> $OUT/pmc_summary.txt
}
for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
  pmc "osc2-pan/65536/256/256" "$ctr" --config 3 --steps 6 --warmup 2
  pmc "osc-filter-pan/16384/0/256" "$ctr" --config 2 --steps 8 --warmup 2
  pmc "osc-pan/1024/0/256" "$ctr" --config 1 --steps 16 --warmup 4
  pmc "osc-filter-pan/32768/0/256" "$ctr" --config 4 --steps 6 --warmup 2
  pmc "osc-pan-private/65536/0/256" "$ctr" --config 5 --steps 4 --warmup 1
  A2AMD_RAW=0 pmc "osc-pan-private-coef/65536/0/256" "$ctr" --config 5 --steps 4 --warmup 1
done
( cd $REPO && python tools/pmc_to_json.py $OUT/pmc_summary.txt > $OUT/pmc.json )
cp $OUT/pmc.json $REPO/profiles/r04_pmc.json
cp $OUT/pmc_summary.txt $REPO/profiles/r04_pmc_summary.txt

# 2. the bench line (reading this round's counters), and the kernel trace of the same command
$B > $OUT/bench_default.json 2> $OUT/bench_default.err
rm -rf /tmp/prof_k; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- $B --no-cpu-baseline --no-realtime --no-engine > /tmp/prof_k.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
# the private-wave scene, both ways
$B --config 5 --steps 20 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
A2AMD_RAW=0 $B --config 5 --steps 20 --no-cpu-baseline > $OUT/bench_cfg5_coef.json 2> $OUT/bench_cfg5_coef.err
# configs[4]: one top-level group (the per-GPU share) and all eight on this one GPU
$B --config 4 --no-cpu-baseline > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
$B --config 4 --voices 262144 --steps 20 --no-cpu-baseline > $OUT/bench_cfg4_all.json 2> $OUT/bench_cfg4_all.err
# the N > 1 code path with one rank (the collective, the whole-job golden gate, the per-fragment reduce timing)
A2AMD_BENCH_FORCE_DIST=1 $B --steps 20 --warmup 3 --no-engine > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err
ls -la $OUT

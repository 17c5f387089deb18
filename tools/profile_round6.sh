#!/bin/bash
# Measurement pass of round 6, to be run on the GPU box:   tools/gpu.sh 3000 'bash tools/profile_round6.sh'
# Everything lands in gpurun_out/round6/; what is to be judged is copied into profiles/ (r06_*).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/round6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
pre="$REPO/audiality2_amd/liba2amd_walk.so $REPO/audiality2_amd/liba2amd_units.so"

# 1. PMC passes (counters only, own runs; --kernel-trace for the per-kernel times pmc_to_json.py picks the dominant kernel by)
rm -f $OUT/pmc_summary.txt
pmc() { # label, counters, command...
  local label=$1 ctr=$2; shift 2
  rm -rf /tmp/prof_p; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_p -- "$@" > /tmp/prof_p.log 2>&1
  python $REPO/tools/pmc_summary.py /tmp/prof_p "$label" >> $OUT/pmc_summary.txt
}
for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"; do
  pmc "osc2-pan/65536/256/256" "$ctr" $B --config 3 --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-realtime --no-engine
  pmc "osc2-filter-pan/16384/0/256" "$ctr" $B --config 6 --steps 6 --warmup 2 --no-cpu-baseline --no-realtime
  pmc "osc-filter-pan/16384/0/256" "$ctr" $B --config 2 --steps 6 --warmup 2 --no-cpu-baseline --no-realtime
  for ch in osc-pan osc-filter-pan osc2-filter-pan; do
    pmc "scripted-$ch/16384/0/64" "$ctr" python $REPO/tools/scripted_timing.py --chain $ch --names scripted,scripted2
  done
  # the device VM's kernels: the scripted engine cell (k_vm_win in the speculative pass, k_vm_commit, k_win_render)
  ( cd $REPO/tests/a2s; rm -rf /tmp/prof_p; LD_PRELOAD="$pre" A2REF_BUFFER=4096 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_p -- \
      $REPO/oracle/_ref/ref_bench bench.a2s OscPanScripted 16384 6144 1 > /tmp/prof_p.log 2>&1
    python $REPO/tools/pmc_summary.py /tmp/prof_p "vm-osc-pan/16384/0/64" >> $OUT/pmc_summary.txt
    rm -rf /tmp/prof_p; LD_PRELOAD="$pre" A2REF_BUFFER=4096 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_p -- \
      $REPO/oracle/_ref/ref_bench bench.a2s OscFilterPanScripted 16384 6144 1 > /tmp/prof_p.log 2>&1
    python $REPO/tools/pmc_summary.py /tmp/prof_p "vm-osc-filter-pan/16384/0/64" >> $OUT/pmc_summary.txt )
done
( cd $REPO && python tools/pmc_to_json.py $OUT/pmc_summary.txt tools/profile_round6.sh > $OUT/pmc.json 2>/dev/null )
# ... and the measured issue rate of the headline kernels' own hot-loop instruction mixes (roofline_valu.peak)
( cd $REPO && timeout 900 python tools/valu_mix.py > $OUT/valu_mix.json 2> $OUT/valu_mix.err )
# (the bench line below quotes both: they are this round's, of the binaries it runs)
cp $OUT/pmc.json $REPO/profiles/r06_pmc.json
[ -s $OUT/valu_mix.json ] && cp $OUT/valu_mix.json $REPO/profiles/r06_valu_mix.json


# 2. the bench line with the driver's flags (stdout = the contract line, side file = everything), and config 6
( cd $REPO && $B --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; cp bench_details.json $OUT/bench_default_details.json )
( cd $REPO && $B --config 6 --steps 7 --warmup 1 > $OUT/bench_cfg6.json 2> $OUT/bench_cfg6.err )

# 3. the kernel trace of the same command (without the engine / CPU legs)
rm -rf /tmp/prof_k; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-realtime --no-engine --no-extra > /tmp/prof_k.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
find /tmp/prof_k -name "*kernel_trace.csv" | head -1 | xargs -r head -1 > $OUT/kernel_trace_header.txt

# 4. the scripted batches: kernel times per batch (window kernels), the quiet kernels beside them
cd $REPO
: > $OUT/scripted_timing.jsonl
for ch in osc-pan osc-filter-pan osc2-pan osc2-filter-pan; do
  timeout 300 python tools/scripted_timing.py --chain $ch --voices 16384 --batch 64 2>&1 | tail -1 >> $OUT/scripted_timing.jsonl
done

# 5. the speculative VM pass: beside the render pass / behind the leaf kernels / off, scripted engine cells
bash tools/r06_spec_early_ab.sh > $OUT/vm_speculation_ab.txt 2>&1

# 6. the song (BASELINE configs[0]'s command shape)
timeout 900 python tests/measure/song_timing.py --seconds 500 2>&1 | tail -2 > $OUT/song_timing.jsonl

# 7. the N > 1 bench path on one rank (what the driver's 2 / 4 / 8 GPU runs execute, minus the other ranks)
cd /tmp
A2AMD_BENCH_FORCE_DIST=1 $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_n_gt_1_path_one_rank.json 2> $OUT/bench_n_gt_1.err
ls -la $OUT
wc -c $OUT/bench_default.json $OUT/bench_n_gt_1_path_one_rank.json

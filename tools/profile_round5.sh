#!/bin/bash
# Measurement pass of round 5, to be run on the GPU box:
#   gpurun -- 'bash tools/profile_round5.sh'
# Everything lands in gpurun_out/round5/; what is to be judged is copied into profiles/ (r05_*).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/round5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"

# 1. the bench line, and the kernel trace of the same command (without the engine / CPU legs)
$B > $OUT/bench_default.json 2> $OUT/bench_default.err
rm -rf /tmp/prof_k; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- $B --no-cpu-baseline --no-realtime --no-engine --no-extra > /tmp/prof_k.log 2>&1
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;

# 2. PMC passes (counters only, own runs) for the bench line's kernel and for the window kernels
rm -f $OUT/pmc_summary.txt
pmc() { # label, counters, command...
  local label=$1 ctr=$2; shift 2
  rm -rf /tmp/prof_p; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_p -- "$@" > /tmp/prof_p.log 2>&1
  python $REPO/tools/pmc_summary.py /tmp/prof_p "$label" >> $OUT/pmc_summary.txt
}
for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
  pmc "osc2-pan/65536/256/256" "$ctr" $B --config 3 --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-realtime --no-engine
  for ch in osc-pan osc-filter-pan osc2-filter-pan; do
    pmc "scripted-$ch/16384/0/64" "$ctr" python $REPO/tools/scripted_timing.py --chain $ch --names scripted,scripted2
  done
done
( cd $REPO && python tools/pmc_to_json.py $OUT/pmc_summary.txt > $OUT/pmc.json 2>/dev/null )

# 3. the scripted batches: kernel times per batch, the two passes apart, and the records kernels beside them
cd $REPO
: > $OUT/scripted_timing.jsonl
: > $OUT/window_passes.txt
for ch in osc-pan osc-filter-pan osc2-pan osc2-filter-pan; do
  timeout 300 python tools/scripted_timing.py --chain $ch --voices 16384 --batch 64 2>&1 | tail -1 >> $OUT/scripted_timing.jsonl
  A2AMD_WIN=0 A2AMD_NO_MOVING=1 timeout 300 python tools/scripted_timing.py --chain $ch --voices 16384 --batch 64 2>&1 | tail -1 | sed 's/^{/{"kernels": "k_leaf_recs (A2AMD_WIN=0 A2AMD_NO_MOVING=1)", /' >> $OUT/scripted_timing.jsonl
  echo "== $ch, 16384 voices x 64 fragments (scripted, scripted, quiet with glides)" >> $OUT/window_passes.txt
  A2AMD_WIN_TIMING=1 timeout 300 python tools/scripted_timing.py --chain $ch --names scripted,scripted2,quiet2 2>&1 | grep "a2amd windows" >> $OUT/window_passes.txt
done
for n in 1024 4096 65536; do
  timeout 300 python tools/scripted_timing.py --chain osc-filter-pan --voices $n --batch 64 --names quiet,scripted,scripted2,quiet2 2>&1 | tail -1 >> $OUT/scripted_timing.jsonl
done
timeout 300 python tools/scripted_timing.py --chain osc-pan --voices 16384 --batch 1 2>&1 | tail -1 >> $OUT/scripted_timing.jsonl

# 4. the song (BASELINE configs[0]'s command shape), with and without the walk + device VM
timeout 900 python tests/measure/song_timing.py --seconds 500 2>&1 | tail -2 > $OUT/song_timing.jsonl
A2AMD_SPLIT=0 timeout 900 python tests/measure/song_timing.py --seconds 500 2>&1 | tail -2 | sed 's/^{/{"env": "A2AMD_SPLIT=0", /' >> $OUT/song_timing.jsonl

# 5. the scripted engine cells (BASELINE variants 2b / 3b: unmodified engine + walk + drop-in units, a2_Run(4096)) with the
#    device VM's voices through k_vm_win and through records (A2AMD_VMWIN=0), each with its kernel trace
pre="$REPO/audiality2_amd/liba2amd_walk.so $REPO/audiality2_amd/liba2amd_units.so"
: > $OUT/engine_cells_vmwin_ab.txt
cd /tmp
for vw in 1 0; do
  for prog in OscPanScripted OscFilterPanScripted; do
    echo "== $prog, 16 384 voices, a2_Run(4096), A2AMD_VMWIN=$vw" >> $OUT/engine_cells_vmwin_ab.txt
    LD_PRELOAD="$pre" A2AMD_VMWIN=$vw A2REF_BUFFER=4096 A2AMD_HOSTTIMING=1 $REPO/oracle/_ref/ref_bench $REPO/tests/a2s/bench.a2s $prog 16384 8192 1 2>&1 |
        grep -v "uploads by first" | tail -4 >> $OUT/engine_cells_vmwin_ab.txt
    rm -rf /tmp/prof_e; LD_PRELOAD="$pre" A2AMD_VMWIN=$vw A2REF_BUFFER=4096 rocprofv3 --kernel-trace -d /tmp/prof_e -o p -- \
        $REPO/oracle/_ref/ref_bench $REPO/tests/a2s/bench.a2s $prog 16384 8192 1 > /dev/null 2>&1
    python $REPO/tools/rocpd_kernels.py /tmp/prof_e "$prog vmwin=$vw" >> $OUT/engine_cells_vmwin_ab.txt
  done
done

# 6. the N > 1 bench path on one rank (what the driver's 2 / 4 / 8 GPU runs execute, minus the other ranks)
cd /tmp
A2AMD_BENCH_FORCE_DIST=1 $B --no-cpu-baseline > $OUT/bench_n_gt_1_path_one_rank.json 2> $OUT/bench_n_gt_1.err
ls -la $OUT
cat $OUT/bench_default.json | cut -c1-3000

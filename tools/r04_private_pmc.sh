#!/bin/bash
# HBM traffic of the private-wave scene (bench.py --config 5), samples vs coefficient entries: separate
# rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md prescribes
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -f $OUT/pmc_summary_private.txt
for raw in 1 0; do
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
    rm -rf /tmp/prof_p
    A2AMD_RAW=$raw rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_p -- python $REPO/bench.py --config 5 --steps 4 --warmup 1 --no-cpu-baseline --no-realtime > /tmp/prof_p.log 2>&1
    python $REPO/tools/pmc_summary.py /tmp/prof_p "osc-pan-private$([ $raw = 0 ] && echo -coef)/65536/0/256" >> $OUT/pmc_summary_private.txt
  done
done
cp $OUT/pmc_summary_private.txt $REPO/gpurun_out/
grep -c JSON $OUT/pmc_summary_private.txt

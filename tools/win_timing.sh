#!/bin/bash
# Kernel times of the scripted batches (tools/scripted_timing.py) per chain, with the per-kernel split
# from a rocprofv3 kernel trace (sqlite output read by tools/rocpd_kernels.py).  On the GPU box:
#   gpurun -- 'bash tools/win_timing.sh [out.jsonl] [voices]'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$REPO/gpurun_out/win_timing.jsonl}
case $OUT in /*) ;; *) OUT=$REPO/$OUT;; esac
VOICES=${2:-16384}
cd /tmp && export TMPDIR=/tmp
: > $OUT
for ch in osc-pan osc-filter-pan osc2-pan osc2-filter-pan; do
  rm -rf /tmp/prof_w
  rocprofv3 --kernel-trace -d /tmp/prof_w -o p -- python $REPO/tools/scripted_timing.py --chain $ch --voices $VOICES \
      --names scripted,scripted2,quiet2 >> $OUT 2>/tmp/prof_w.err
  python $REPO/tools/rocpd_kernels.py /tmp/prof_w $ch >> $OUT
done
cat $OUT

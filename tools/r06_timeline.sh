#!/bin/bash
# round 6: where a scripted buffer's time goes with the speculative pass on - a kernel trace of the steady state, one
# table per a2_Run(4096) (kernel, queue, start and end relative to the buffer's first kernel)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
REPO=$PWD
export TMPDIR=/tmp
pre="$PWD/audiality2_amd/liba2amd_walk.so $PWD/audiality2_amd/liba2amd_units.so"
for prog in OscPanScripted OscFilterPanScripted; do
  rm -rf /tmp/prof_t
  ( cd tests/a2s; LD_PRELOAD="$pre" A2REF_BUFFER=4096 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- \
      ../../oracle/_ref/ref_bench bench.a2s $prog 16384 3072 1 > /tmp/prof_t.log 2>&1 )
  grep voice_samples /tmp/prof_t.log | cut -c1-200
  f=$(find /tmp/prof_t -name '*kernel_trace.csv' | head -1)
  echo "== $prog  $f"
  python $REPO/tools/timeline.py "$f" 3
done

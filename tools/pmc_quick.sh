set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc3
mkdir -p $OUT; rm -f $OUT/*.txt
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
pmc() { local label=$1 ctr=$2; shift 2
  rm -rf /tmp/prof_p; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_p -- $B "$@" --no-cpu-baseline --no-extra --no-realtime > /tmp/prof_p.log 2>&1
  python $REPO/tools/pmc_summary.py /tmp/prof_p "$label" | grep -E "osc2pan|oscpan|fbdchain|JSON" >> $OUT/pmc_summary.txt; }
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  pmc "osc2-pan/65536/256/256" "$ctr" --config 3 --steps 6 --warmup 2
done
grep -v JSON $OUT/pmc_summary.txt

# kernel trace of the engine + drop-in:  PROG=... V=... BUF=... bash tools/scr_prof.sh
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/scrprof_${BUF:-64}
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export A2REF_BUFFER=${BUF:-64}
export U=$REPO/audiality2_amd/liba2amd_units.so B=$REPO/oracle/_ref/ref_bench A=$REPO/tests/a2s
export PROG=${PROG:-OscPanScripted} V=${V:-16384} FR=${FR:-256}
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- bash -c 'cd $A; LD_PRELOAD="$LD_PRELOAD:$U" exec $B bench.a2s $PROG $V $FR 1' > $OUT/log.txt 2>&1
echo rc=$?
tail -2 $OUT/log.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-6 $f | head -12

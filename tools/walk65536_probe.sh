# The engine's walk over 65 536 sleeping voices with the drop-in: per-buffer host timing
# with the look-ahead prefetch hints of a2amd_units.c off / 1..8 heads ahead.
cd $GRAFT_REPO_ROOT/tests/a2s
B=../../oracle/_ref/ref_bench
U=../../audiality2_amd/liba2amd_units.so
for P in ${PROGS:-Osc2PanGroups OscPan Osc2Pan OscFilterPan}; do
for V in ${VS:-32768 65536}; do
for WA in ${WAS:-0 1 2 4 8}; do
r=$(A2AMD_WALK_AHEAD=$WA A2REF_BUFFER=64 LD_PRELOAD=$U timeout 300 $B bench.a2s $P $V 300 1 2>/dev/null | tail -1)
echo "{\"program\": \"$P\", \"voices\": $V, \"walk_ahead\": $WA, \"result\": $r}"
done; done; done

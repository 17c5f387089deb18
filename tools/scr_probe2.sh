# per-buffer / per-render host timing trace of the engine + drop-in (tail = steady state)
#   PROG=OscPan V=65536 BUFS="64" bash tools/scr_probe2.sh
cd $GRAFT_REPO_ROOT/tests/a2s
B=../../oracle/_ref/ref_bench
U=${UNITS:-../../audiality2_amd/liba2amd_units.so}
for BUF in ${BUFS:-4096 64}; do
echo "== ${PROG:-OscPanScripted} ${V:-16384} buffer $BUF"
A2REF_BUFFER=$BUF A2AMD_HOSTTIMING=2 LD_PRELOAD=$U timeout 300 $B bench.a2s ${PROG:-OscPanScripted} ${V:-16384} ${FR:-512} 1 2>&1 | grep -v "^a2amd units\|^a2amd host" | tail -${TAIL:-5}
done

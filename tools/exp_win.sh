cd $GRAFT_REPO_ROOT
echo "== song (units only)"; python tests/measure/song_timing.py --seconds 500 2>&1 | tail -3 | cut -c1-500
echo "== song A2AMD_WIN=0 (round 4's kernels)"; A2AMD_WIN=0 A2AMD_SPLIT=0 python tests/measure/song_timing.py --seconds 500 2>&1 | tail -2 | cut -c1-400
echo "== song A2AMD_SPLIT=0"; A2AMD_SPLIT=0 python tests/measure/song_timing.py --seconds 500 2>&1 | tail -2 | cut -c1-400
echo "== song A2AMD_WIN_FORK=0"; A2AMD_WIN_FORK=0 python tests/measure/song_timing.py --seconds 500 2>&1 | tail -2 | cut -c1-400
for ch in osc-pan osc-filter-pan osc2-filter-pan; do echo "== $ch"; A2AMD_WIN_TIMING=1 python tools/scripted_timing.py --chain $ch --names scripted,scripted2,quiet2 2>&1 | grep "a2amd windows" | tail -2 | sed 's/.*records: //'; done

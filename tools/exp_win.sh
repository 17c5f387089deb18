cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" A2AMD_WIN_TIMING=1 python tools/scripted_timing.py --chain osc-filter-pan --names scripted,scripted2,quiet2 2>&1 | grep "a2amd windows" | tail -2 | sed 's/.*records: //'; }
run A2AMD_WFVPG=16 A2AMD_WFWAVES=8
run A2AMD_WFVPG=16 A2AMD_WFWAVES=5
run A2AMD_WFVPG=32 A2AMD_WFWAVES=8
run A2AMD_WFVPG=32 A2AMD_WFWAVES=5
run A2AMD_WFVPG=48 A2AMD_WFWAVES=8
run A2AMD_WFVPG=8 A2AMD_WFWAVES=5
run A2AMD_WFVPG=8 A2AMD_WFWAVES=3

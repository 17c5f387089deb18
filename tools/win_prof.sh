#!/bin/bash
# cycle counters inside k_win_ctl / k_win_render_f / k_vm_win (-DWIN_PROF): where a step's time goes, per wavefront
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO/audiality2_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DWIN_PROF -o /tmp/liba2amd_prof.so a2amd_host.cpp a2amd_sched.cpp a2amd_render.cpp a2amd_dist.cpp a2amd_vm.cpp a2amd_kernels.hip a2amd_fast.hip a2amd_vm.hip a2amd_wavecap.hip a2amd_win.hip a2amd_vmwin.hip 2>&1 | grep -v warning | head -5
cd $REPO
for e in "X=1" "A2AMD_DEBUG=1"; do echo "== $e"
env $e A2AMD_WIN_TIMING=1 A2AMD_LIB=/tmp/liba2amd_prof.so python tools/scripted_timing.py --chain ${1:-osc-filter-pan} --names scripted,quiet2 2>&1 | grep "k_win_ctl\|k_win_render_f\|a2amd windows" | grep "block 391\|block 66\|windows" | tail -8
done

R=$GRAFT_REPO_ROOT
cd $R/audiality2_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DVM_SCRATCH -Wl,-soname,liba2amd.so -o /tmp/liba2amd.so a2amd_host.cpp a2amd_sched.cpp a2amd_render.cpp a2amd_dist.cpp a2amd_vm.cpp a2amd_kernels.hip a2amd_fast.hip a2amd_vm.hip a2amd_wavecap.hip a2amd_win.hip 2>&1 | grep -v warning | head -3
cd /tmp
for v in lds scratch; do
  pre="$R/audiality2_amd/liba2amd_walk.so $R/audiality2_amd/liba2amd_units.so"
  [ $v = scratch ] && pre="/tmp/liba2amd.so $pre"
  rm -rf /tmp/prof_e; LD_PRELOAD="$pre" A2REF_BUFFER=4096 rocprofv3 --kernel-trace -d /tmp/prof_e -o p -- $R/oracle/_ref/ref_bench $R/tests/a2s/bench.a2s OscPanScripted 16384 8192 1 2>/dev/null | tail -1 | cut -c1-120
  python $R/tools/rocpd_kernels.py /tmp/prof_e $v | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for k,v in d['kernels'].items():
    if 'vm' in k: print('  ', d['label'], k[:30], v)"
done

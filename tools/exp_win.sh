R=$GRAFT_REPO_ROOT
cd /tmp
pre="$R/audiality2_amd/liba2amd_walk.so $R/audiality2_amd/liba2amd_units.so"
for sl in 1 2 4 8; do
for prog in OscPanScripted OscFilterPanScripted; do
  echo "SLABS=$sl $prog: $(LD_PRELOAD="$pre" A2AMD_WIN_SLABS=$sl A2REF_BUFFER=4096 $R/oracle/_ref/ref_bench $R/tests/a2s/bench.a2s $prog 16384 4096 1 2>&1 | tail -1 | cut -c1-60,180-260)"
done
done
for sl in 1 4; do
for ch in osc-pan osc-filter-pan; do
echo "SLABS=$sl $ch $(A2AMD_WIN_SLABS=$sl python $R/tools/scripted_timing.py --chain $ch --names scripted,scripted2,scripted3,quiet2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(' '.join('%s=%.3f' % (k, d[k]['kernels_ms_per_batch']) for k in ('scripted2','scripted3','quiet2')))")"
done
done
cd $R && A2AMD_WIN_SLABS=4 timeout 600 python -m pytest tests/test_device_vm.py -m gpu -x -q -k "scripted_bench or wake_many or looping or notes_run" 2>&1 | tail -3

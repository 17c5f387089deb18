cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "#### window kernels, 16 384 voices x 64 fragments (tools/scripted_timing.py): kernel ms per batch"
for ch in osc-pan osc2-pan; do python tools/scripted_timing.py --chain $ch --voices 16384 --batch 64 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$ch', {k:(round(v['kernels_ms_per_batch'],4) if isinstance(v,dict) and 'kernels_ms_per_batch' in v else None) for k,v in d.items() if isinstance(v,dict)})"; done
pre="$PWD/audiality2_amd/liba2amd_walk.so $PWD/audiality2_amd/liba2amd_units.so"
echo "#### 2b / 2e engine cells, a2_Run(4096), 12 288 fragments"
for prog in OscPanScripted OscPanEnvScripted; do ( cd tests/a2s; LD_PRELOAD="$pre" A2REF_BUFFER=4096 ../../oracle/_ref/ref_bench bench.a2s $prog 16384 12288 1 2>&1 | grep voice_samples | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$prog', 'vs/s %.4g' % d['voice_samples_per_s'], 'p50 steady us', d['run_us_p50_steady'])" ); done
echo "#### parity"
python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "window or records or scripted or walk" 2>&1 | tail -2

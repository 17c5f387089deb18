cd ${GRAFT_REPO_ROOT:-$(pwd)}
T='import json,sys
d=json.loads(sys.stdin.read())
print({k:(round(v["kernels_ms_per_batch"],4) if isinstance(v,dict) and "kernels_ms_per_batch" in v else None) for k,v in d.items() if isinstance(v,dict) and k!="scripted"})'
echo "#### window kernels with filter12, 16 384 voices x 64 fragments: before (liba2amd_chain8.so: the kernels of commit db86cbb) / blocks of 8 (shipped) / of 4"
for ch in osc-filter-pan osc2-filter-pan; do for lib in build_variants/liba2amd_chain8.so ""; do
  echo -n "$ch ${lib:-shipped}: "; A2AMD_LIB=${lib:+$PWD/$lib} python tools/scripted_timing.py --chain $ch --voices 16384 --batch 64 2>/dev/null | tail -1 | python -c "$T"
done; done
echo "#### parity"
python -m pytest tests/test_gpu_parity.py tests/test_device_vm.py -q -x -p no:cacheprovider -k "window or records or scripted or walk or filter" 2>&1 | tail -2

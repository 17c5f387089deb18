cd ${GRAFT_REPO_ROOT:-$(pwd)}
for m in "A2AMD_VMSPEC_EARLY=1" "A2AMD_VMSPEC_EARLY=0"; do
  fails=0
  for i in 1 2 3 4; do
    env $m timeout 300 python -m pytest tests/test_device_vm.py -q -x -p no:cacheprovider -k "looping_voices or notes_run or wake_many" > /tmp/t.txt 2>&1
    if grep -q failed /tmp/t.txt; then fails=$((fails+1)); echo "---- $m run $i FAILED"; grep -E "^E  " /tmp/t.txt | head -4 | cut -c1-300; fi
  done
  echo "$m: $fails of 4 runs failed; last: $(tail -1 /tmp/t.txt)"
done
echo "#### whole file, default settings"; python -m pytest tests/test_device_vm.py -q -x -p no:cacheprovider 2>&1 | tail -2
pre="$PWD/audiality2_amd/liba2amd_walk.so $PWD/audiality2_amd/liba2amd_units.so"
echo "#### statistics: vmloops (lists change), OscPanScripted 16384 (steady)"
( cd tests/a2s; LD_PRELOAD="$pre" A2AMD_WIN=1 A2AMD_HOSTTIMING=1 ../../oracle/_ref/ref_render vmloops.a2s Main 143360 4096 48000 2 /tmp/x.pcm 0.08 2>&1 | grep "speculative" | cut -c1-600
  LD_PRELOAD="$pre" A2REF_BUFFER=4096 A2AMD_HOSTTIMING=1 ../../oracle/_ref/ref_bench bench.a2s OscPanScripted 16384 12288 1 2>&1 | grep "speculative\|voice_samples" | cut -c1-600 )

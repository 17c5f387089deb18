# how many of the soak's scripts actually launch / take speculative passes (the soak harness discards stderr)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
pre="$PWD/audiality2_amd/liba2amd_walk.so $PWD/audiality2_amd/liba2amd_units.so"
mkdir -p /tmp/fz; launched=0; taken=0; withvm=0; n=0
for seed in $(seq 5000 5059); do
  python -c "
import sys; sys.path.insert(0,'tests')
import fuzz_scripts; open('/tmp/fz/s.a2s','w').write(fuzz_scripts.make_script($seed))"
  buf=$(python -c "print((64, 37, 256, 1024, 17)[$seed % 5])")
  out=$(cd tests/a2s; LD_PRELOAD="$pre" A2AMD_WIN=1 A2AMD_VMSPEC_MIN=1 A2AMD_HOSTTIMING=1 timeout 60 ../../oracle/_ref/ref_render /tmp/fz/s.a2s Main $((96000 / buf * buf)) $buf 48000 2 /tmp/fz/o.pcm 0.15 2>&1)
  n=$((n+1))
  echo "$out" | grep -q "batches with device VM voices" && withvm=$((withvm+1))
  l=$(echo "$out" | sed -n 's/.*: \([0-9]*\) speculative passes.*/\1/p' | head -1); t=$(echo "$out" | sed -n 's/.*, \([0-9]*\) of them taken.*/\1/p' | head -1)
  launched=$((launched + ${l:-0})); taken=$((taken + ${t:-0}))
done
echo "$n scripts (seeds 5000-5059, the soak's first leg's settings): $withvm with device VM voices; speculative passes launched $launched, taken $taken"

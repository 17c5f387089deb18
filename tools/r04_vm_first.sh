#!/bin/bash
# first light of the device VM: bench.a2s scripted variants, CPU engine vs drop-in + walk (+ device VM)
cd tests/a2s
R=../../oracle/_ref/ref_render
W="../../audiality2_amd/liba2amd_walk.so ../../audiality2_amd/liba2amd_units.so"
O=../../gpurun_out/vm1
mkdir -p $O
for prog in OscPanScripted OscFilterPanScripted; do
  for buf in 64 4096; do
    $R bench.a2s $prog $((48000*2)) $buf 48000 2 $O/cpu_$prog$buf.pcm 16 0.01 2>&1 | tail -2
    A2AMD_WALK_STATS=1 A2AMD_VM_TRACE=1 LD_PRELOAD="$W" $R bench.a2s $prog $((48000*2)) $buf 48000 2 $O/vm_$prog$buf.pcm 16 0.01 2>&1 | tail -5
    A2AMD_NO_VM=1 LD_PRELOAD="$W" $R bench.a2s $prog $((48000*2)) $buf 48000 2 $O/novm_$prog$buf.pcm 16 0.01 2>&1 | tail -2
    cmp $O/cpu_$prog$buf.pcm $O/vm_$prog$buf.pcm && echo "SAME vm $prog $buf" || echo "DIFF vm $prog $buf"
    cmp $O/cpu_$prog$buf.pcm $O/novm_$prog$buf.pcm && echo "SAME novm $prog $buf" || echo "DIFF novm $prog $buf"
  done
done
rm -f $O/*.pcm

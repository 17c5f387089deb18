#!/bin/bash
# records of one env voice, host-made (A2AMD_NO_VM=1) and device-made, side by side
cd /root/repo/tests/a2s
python - <<'PY'
src = open("envloops.a2s").read()
head = src[:src.index("export Main")]
open("_envd.a2s", "w").write(head + "export Main(V=.08)\n{\n\t" + "SOLO" + "\n\tfor { d 100000 }\n}\n")
PY
sed -i "s/SOLO/$1/" _envd.a2s
W=/root/repo/audiality2_amd/liba2amd_walk.so; U=/root/repo/audiality2_amd/liba2amd_units.so
A2AMD_VM_DUMP=1 A2AMD_NO_VM=1 LD_PRELOAD="$W $U" ../../oracle/_ref/ref_render _envd.a2s Main ${2:-9600} 64 48000 2 /tmp/d_host.pcm 0.08 2> /root/repo/gpurun_out/envd_host.txt
A2AMD_VM_DUMP=1 A2AMD_VM_TRACE=1 LD_PRELOAD="$W $U" ../../oracle/_ref/ref_render _envd.a2s Main ${2:-9600} 64 48000 2 /tmp/d_vm.pcm 0.08 2> /root/repo/gpurun_out/envd_vm.txt
rm -f _envd.a2s
cmp /tmp/d_host.pcm /tmp/d_vm.pcm | head -2
grep -c REC /root/repo/gpurun_out/envd_host.txt /root/repo/gpurun_out/envd_vm.txt

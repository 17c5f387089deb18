cd /root/repo/tests/a2s
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo/tests")
import fuzz_scripts
open("/tmp/f100.a2s", "w").write(fuzz_scripts.make_script(100))
PY
gcc -O0 -g -shared -fPIC -o /tmp/segv_trace.so /root/repo/tools/segv_trace.c
export A2AMD_DEVICES=2
LD_PRELOAD="/tmp/segv_trace.so /root/repo/audiality2_amd/liba2amd_units.so" ../../oracle/_ref/ref_render /tmp/f100.a2s Main 48000 512 48000 2 /tmp/f.pcm 0.15 2>&1 | tail -25

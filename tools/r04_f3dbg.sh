cd /root/repo/tests/a2s
A2REF_REALTIME=1 A2AMD_WAVE_STATS=1 LD_PRELOAD=/root/repo/audiality2_amd/liba2amd_units.so ../../oracle/_ref/ref_render edge.a2s Main 4800 64 48000 2 /tmp/e.pcm 0.2 2>&1 | tail -12

cd ${GRAFT_REPO_ROOT:-$(pwd)}
for m in "A2AMD_O2F_MIN=512" "A2AMD_O2F_MIN=0" "A2AMD_O2F_MIN=512 A2AMD_VMSPEC=0" "A2AMD_O2F_MIN=512 A2AMD_NO_VM=1"; do
  echo "== $m"; env $m timeout 600 python tests/measure/song_timing.py --seconds 500 2>&1 | tail -2 | python -c "
import sys,json
for ln in sys.stdin:
    try: d=json.loads(ln)
    except Exception: print(ln[:200]); continue
    print('  buffer', d['buffer_frames'], 'cpu %.3f s' % d['cpu_reference_s'], 'drop-in %.3f s' % d['gpu_dropin_s'], 'us/buffer %.1f' % d['dropin_us_per_buffer'], 'bit-identical', d['bit_identical_30s_render'])"
done

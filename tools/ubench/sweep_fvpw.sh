for v in 8 16 24 32 48 64; do echo -n "FVPW $v: "; A2AMD_FVPW=$v python bench.py --config 2 --no-cpu-baseline --no-realtime --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.3g'%d['value'], '%.3f'%d['ms_per_step'], '%.3f'%d['roofline']['avg_launch_ms'], d['parity_vs_golden'])"; done

import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["config"]["voices_per_gpu"], "value=%.4g"%d["value"], "ms/step=%.4f"%d["ms_per_step"], "leaf_ms=%.4f"%d["roofline"]["avg_launch_ms"], "all_ms=%.4f"%d["roofline"]["all_kernels_ms_per_step"], "frac=%.4f"%d["roofline"]["frac"])

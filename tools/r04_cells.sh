#!/bin/bash
cd /root/repo
python tools/engine_cells.py variant > gpurun_out/r04_engine_cells_scripted.jsonl 2> gpurun_out/r04_cells.err
python tools/engine_cells.py --novm "variant 2e" > gpurun_out/r04_engine_cells_env_novm.jsonl 2>> gpurun_out/r04_cells.err
tail -3 gpurun_out/r04_cells.err
python - <<'PY'
import json
for f in ("gpurun_out/r04_engine_cells_scripted.jsonl", "gpurun_out/r04_engine_cells_env_novm.jsonl"):
    for ln in open(f):
        d = json.loads(ln)
        for k, v in d.items():
            if k.startswith("a2_Run"):
                for m in ("units", "units+walk"):
                    if m in v and "error" not in v[m]:
                        x = v[m]
                        print(d["case"], d["device_vm"], k, m, "%.3g v-s/s" % x["voice_samples_per_s"], "p99 %.0f us" % x["us_per_fragment_p99"], "equal", x["hash_equal"], "tail_at", v.get("tail_at"))
                    elif m in v:
                        print(d["case"], k, m, v[m])
                if "error" in v:
                    print(d["case"], k, v["error"])
PY

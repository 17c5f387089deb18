"""The scripted engine-in-loop cells of bench.py (variants 2b / 3b, 16 384 voices) with the device
VM on and off: voice-samples/s, microseconds per fragment, hash against the CPU engine.
usage: python tools/r04_vm_cells.py [voices] > profiles/r04_scripted_engine_in_loop.jsonl"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

voices = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for program in ("OscPanScripted", "OscFilterPanScripted"):
    for buf in (4096, 64):
        hf = 64
        cpu = bench.engine_run(program, voices, max(hf, buf // 64), buf, False, hf)
        for label, extra in (("units+walk+vm", {"A2AMD_WALK_STATS": "1"}), ("units+walk (A2AMD_NO_VM=1)", {"A2AMD_NO_VM": "1"})):
            nfr = 2048 if buf > 64 else 1200
            g = bench.engine_run(program, voices, nfr, buf, True, hf, env_extra=extra, walk=True)
            if "error" in g or "error" in cpu:
                print(json.dumps({"program": program, "buffer": buf, "mode": label, "error": g.get("error", cpu.get("error"))}), flush=True)
                continue
            per = buf // 64
            print(json.dumps({"program": program, "voices": voices, "buffer": buf, "mode": label,
                              "voice_samples_per_s": g["voice_samples_per_s"],
                              "us_per_fragment_p50": g["run_us_p50"] / per, "us_per_fragment_p99": g["run_us_p99"] / per,
                              "us_per_fragment_mean": g["seconds"] / g["fragments"] * 1e6, "fragments": g["fragments"],
                              "hash_equal": g["hashes"][0] == cpu["hashes"][0] and g["active_voices"] == cpu["active_voices"],
                              "cpu_units_voice_samples_per_s": cpu["voice_samples_per_s"],
                              "walk_stats": g.get("walk_stats")}), flush=True)

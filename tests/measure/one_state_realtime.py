#!/usr/bin/env python3
"""How many sustained voices does ONE engine state - one CPU thread for the reference engine's compiler,
VM and voice tree - hold in realtime behind the replaced voice walk (INTEGRATION.md option C)?
FilterTree (tests/a2s/bench.a2s): G top-level groups x 128 sub-groups x 256 wtosc->filter12->panmix voices,
a2_Run(64) = one synchronous GPU round trip per 64-frame fragment, and a2_Run(4096).  The audio of the first
fragments is compared with the same engine + drop-in WITHOUT the walk (the CPU units would take minutes
at these sizes; units vs CPU is what tests/test_dropin.py and bench.py check up to 262 144 voices).

    python tests/measure/one_state_realtime.py [voices ...]      # default 262144 524288 1048576
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [262144, 524288, 1048576]
    for v in sizes:
        for buf in (64, 4096):
            ref = bench.engine_run("FilterTree", v, max(8, buf // 64), buf, True, 8)
            g = bench.engine_run("FilterTree", v, 2048 if buf == 64 else 8192, buf, True, 8, walk=True,
                                 env_extra={"A2AMD_WALK_STATS": "1"})
            row = {"program": "FilterTree", "voices": v, "buffer_frames": buf}
            if "error" in g or "error" in ref:
                row["error"] = g.get("error") or ref.get("error")
            else:
                per = buf // 64
                row.update({"voice_samples_per_s": g["voice_samples_per_s"], "us_per_fragment_p50": g["run_us_p50"] / per,
                            "us_per_fragment_p99": g["run_us_p99"] / per, "us_per_fragment_max": g["run_us_max"] / per,
                            "realtime_at_48k": g["run_us_p99"] / per <= 64e6 / 48000, "active_voices": g["active_voices"],
                            "hash_equal_to_units_without_walk": g["hashes"] == ref["hashes"] and
                            g["active_voices"] == ref["active_voices"],
                            "units_without_walk_us_per_fragment": ref["seconds"] / ref["fragments"] * 1e6,
                            "walk_stats_skipped_made_unread": g.get("walk_stats")})
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

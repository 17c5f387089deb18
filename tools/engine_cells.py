"""Chosen cells of bench.py's engine_in_loop (the reference engine calling a2_Run(), its units replaced by
the drop-in), as bench.py runs them - first and far-end hashes against the CPU engine included - one JSON
line per case.  With --novm the walk hands no voice to the device VM (A2AMD_NO_VM=1: the A/B of DESIGN 7b).
usage: python tools/engine_cells.py [--novm] [substring of a case label ...] > profiles/rNN_engine_cells.jsonl"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = sys.argv[1:]
if "--novm" in args:
    args.remove("--novm")
    os.environ["A2AMD_NO_VM"] = "1"
cases = [c for c in bench.ENGINE_CASES if not args or any(a in c[0] for a in args)]
for case in cases:
    r = bench.engine_in_loop(cases=[case])
    print(json.dumps({"case": case[0], "device_vm": "A2AMD_NO_VM" not in os.environ, **r["cases"][case[0]]}), flush=True)

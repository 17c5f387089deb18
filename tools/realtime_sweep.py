#!/usr/bin/env python3
"""Per-fragment latency in realtime mode (one 64-frame fragment per render,
full round trip: upload -> kernels -> master bus back on the host), and the
largest voice count whose p99 stays inside the 1.333 ms budget of a 48 kHz
fragment (SURVEY.md 8d "max realtime voices").  Prints one JSON line per
(chain, voices).  Host side = the C ABI driven directly (fragment_repeat), i.e.
without the engine's own per-voice walk."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", default="osc-pan,osc-filter-pan")
    ap.add_argument("--voices", default="1024,4096,16384,32768,65536,131072,262144,524288")
    ap.add_argument("--fragments", type=int, default=400)
    args = ap.parse_args()
    import audiality2_amd
    from audiality2_amd import synth
    budget_ms = 64 / 48000 * 1e3
    for chain in args.chains.split(","):
        best = 0
        for n in [int(x) for x in args.voices.split(",")]:
            be = audiality2_amd.open_backend(max_batch=1)
            lib = be.lib
            lib.a2amd_fragment_repeat.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
            sc = synth.Scene(be)
            sc.root()
            sc.add_voices(n, chain=chain)
            sc.run(1, batch=1)
            for _ in range(20):
                lib.a2amd_fragment_repeat(be.ctx, 64, 1)
                be.render(64)
            lat = np.zeros(args.fragments)
            for i in range(args.fragments):
                t0 = time.perf_counter()
                lib.a2amd_fragment_repeat(be.ctx, 64, 1)
                out = be.render(64)
                lat[i] = (time.perf_counter() - t0) * 1e3
            be.close()
            p50, p99, mx = np.percentile(lat, 50), np.percentile(lat, 99), lat.max()
            ok = bool(p99 <= budget_ms)
            if ok:
                best = max(best, n)
            print(json.dumps({"chain": chain, "voices": n, "fragment_ms_p50": round(float(p50), 4),
                              "fragment_ms_p99": round(float(p99), 4), "fragment_ms_max": round(float(mx), 4),
                              "budget_ms": round(budget_ms, 4), "realtime": ok,
                              "nonzero": bool(out.any())}), flush=True)
        print(json.dumps({"chain": chain, "max_realtime_voices_1gpu": best}), flush=True)


if __name__ == "__main__":
    main()

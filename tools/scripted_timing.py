#!/usr/bin/env python3
"""GPU time of a wavetable batch whose voices are driven like a script drives them
(BASELINE variants 2b / 3b; for the FM units see fm_scripted_timing.py).

N voices, B fragments per batch.  Every PERIOD fragments (phases staggered) a voice
gets a sub-fragment window, a control write and the rest of the fragment:
  --what split   no write, only the split window
  --what amp     amplitude set, no ramp
  --what pitch   pitch write with a duration: the pitch ramps from then on (what
                 tests/a2s/bench.a2s OscPanScripted does every 3 ms)
Prints the kernel time per batch (HIP events, a2amd_set_profiling).

    python tools/scripted_timing.py [--voices 16384] [--chain osc-pan] [--batch 64] [--what pitch]
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiality2_amd  # noqa: E402
from audiality2_amd import synth  # noqa: E402


def measure(voices=16384, chain="osc-pan", batch=64, period=2, what="pitch", names=("quiet", "scripted", "scripted2", "scripted3", "quiet2"),
            Stats=None):
    if Stats is None:       # (bench.py passes its own: ctypes ties a2amd_get_stats to one class object)
        from bench import Stats
    be = audiality2_amd.open_backend(48000, None, 2, max_batch=batch)
    lib = be.lib
    sc = synth.Scene(be)
    sc.root()
    sc.add_voices(voices, chain=chain)
    sc.run(1, batch=1)
    rng = np.random.default_rng(3)
    cuts = rng.integers(1, 64, voices)
    nosc = 2 if chain.startswith("osc2") else 1

    def run_batch(scripted):
        for f in range(batch):
            be.fragment(64)
            be.unit_process(sc.rootv[0], 0, 64)
            for k, units in enumerate(sc.leaves):
                if scripted and (f + k) % period == 0:
                    c = int(cuts[k])
                    for u in units:
                        be.unit_process(u, 0, c)
                    if what == "amp":
                        be.unit_write(units[0], 2, synth.fix(0.001 * (1 + (f + k) % 5)), 0, 0)
                    elif what == "pitch":
                        be.unit_write(units[(f + k) % nosc], 1, synth.fix(((k % 61) - 30) / 12.0 + 0.01 * ((f + k) % 3)),
                                      17, 3 * 48 << 8)
                    for u in units:
                        be.unit_process(u, c, 64 - c)
                else:
                    for u in units:
                        be.unit_process(u, 0, 64)
            be.inline_end(sc.rootv[0])
            be.unit_process(sc.rootv[1], 0, 64)
            be.unit_process(sc.rootv[2], 0, 64)
        st0 = Stats()
        lib.a2amd_get_stats(be.ctx, ctypes.byref(st0))
        out = be.render(batch * 64)
        st1 = Stats()
        lib.a2amd_get_stats(be.ctx, ctypes.byref(st1))
        return (st1.timed_all_ms - st0.timed_all_ms), int(np.abs(out).max())

    lib.a2amd_set_profiling(be.ctx, 1)
    res = {}
    for name in names:
        ms, peak = run_batch(name.startswith("scripted"))
        res[name] = {"kernels_ms_per_batch": round(ms, 4),
                     "voice_samples_per_s": float("%.3g" % (voices * batch * 64 / (ms * 1e-3))), "peak": peak}
    be.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voices", type=int, default=16384)
    ap.add_argument("--chain", default="osc-pan")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--period", type=int, default=2)
    ap.add_argument("--what", default="pitch")
    ap.add_argument("--names", default="quiet,scripted,scripted2,scripted3,quiet2",
                    help="the batches to run, in order (quiet*: no records, scripted*: with)")
    args = ap.parse_args()
    res = measure(args.voices, args.chain, args.batch, args.period, args.what, names=tuple(args.names.split(",")))
    print(json.dumps({"voices": args.voices, "chain": args.chain, "fragments_per_batch": args.batch, "what": args.what,
                      "voices_with_records_per_fragment": 1.0 / args.period, **res}))


if __name__ == "__main__":
    main()

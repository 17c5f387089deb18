#!/usr/bin/env python3
"""GPU time of an FM batch whose voices are driven like a script drives them.

N fmK->panmix voices, B fragments per batch.  In the "scripted" batches every
voice gets, every PERIOD fragments (phases staggered), a control write, a
sub-fragment window, two more writes and the rest of the fragment - the
pattern of `*a .9; *fb .9; d 10` envelopes in benchmark/fmtest4.a2s.  Prints
the kernel time per batch (HIP events, a2amd_set_profiling) for a quiet batch
and for scripted ones, and the voice-samples/s they correspond to.

    python tools/fm_scripted_timing.py [--voices 16384] [--chain fm4-pan]
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
from bench import Stats  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voices", type=int, default=16384)
    ap.add_argument("--chain", default="fm4-pan")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--period", type=int, default=7)
    args = ap.parse_args()
    be = audiality2_amd.open_backend(48000, None, 2, max_batch=args.batch)
    lib = be.lib
    sc = synth.Scene(be)
    sc.root()
    sc.add_voices(args.voices, chain=args.chain)
    sc.run(1, batch=1)
    rng = np.random.default_rng(3)
    cuts = rng.integers(1, 64, args.voices)

    def batch(scripted):
        for f in range(args.batch):
            be.fragment(64)
            be.unit_process(sc.rootv[0], 0, 64)
            for k, (fm, pan) in enumerate(sc.leaves):
                if scripted and (f + k) % args.period == 0:
                    c = int(cuts[k])
                    be.unit_write(fm, 2, synth.fix(0.001 * (1 + (f + k) % 5)), 0, 3000 << 8)
                    be.unit_process(fm, 0, c)
                    be.unit_process(pan, 0, c)
                    be.unit_write(fm, 3, synth.fix(0.1 * ((f + k) % 7)), 17, 480 << 8)
                    be.unit_write(pan, 1, synth.fix(((f + k) % 9 - 4) / 4.0), 0, 3000 << 8)
                    be.unit_process(fm, c, 64 - c)
                    be.unit_process(pan, c, 64 - c)
                else:
                    be.unit_process(fm, 0, 64)
                    be.unit_process(pan, 0, 64)
            be.inline_end(sc.rootv[0])
            be.unit_process(sc.rootv[1], 0, 64)
            be.unit_process(sc.rootv[2], 0, 64)
        st0 = Stats()
        lib.a2amd_get_stats(be.ctx, ctypes.byref(st0))
        out = be.render(args.batch * 64)
        st1 = Stats()
        lib.a2amd_get_stats(be.ctx, ctypes.byref(st1))
        return (st1.timed_all_ms - st0.timed_all_ms), int(np.abs(out).max())

    lib.a2amd_set_profiling(be.ctx, 1)
    res = {}
    for name, scripted in (("quiet", False), ("scripted", True), ("scripted2", True), ("quiet2", False)):
        ms, peak = batch(scripted)
        res[name] = {"kernels_ms_per_batch": ms, "voice_samples_per_s": args.voices * args.batch * 64 / (ms * 1e-3),
                     "peak": peak}
    print(json.dumps({"voices": args.voices, "chain": args.chain, "fragments_per_batch": args.batch,
                      "voices_with_records_per_fragment": 1.0 / args.period, **res}))
    be.close()


if __name__ == "__main__":
    main()

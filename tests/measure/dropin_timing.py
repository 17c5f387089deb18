#!/usr/bin/env python3
"""Wall time of the unmodified reference engine rendering a script offline, with
its own units and with the drop-in GPU units (BASELINE configs[0] style:
"plumbing" - a few dozen voices, one a2_Run() per buffer).

Needs oracle/_ref (the compiled reference: it travels with the repo snapshot,
the reference's own .a2s songs do not, so the scripts are tests/a2s/*).

    python tests/measure/dropin_timing.py [--seconds 30] [--buffer 64]
"""
import argparse
import json
import os
import subprocess
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R = os.path.join(ROOT, "oracle", "_ref", "ref_render")
U = os.path.join(ROOT, "audiality2_amd", "liba2amd_units.so")
A2S = os.path.join(ROOT, "tests", "a2s")
CASES = [("scripted", ["0.2"]), ("fm", ["0.15"]), ("fx", ["0.1"]), ("delaybus", ["2", "4", "0.05"]), ("song", ["0.08"])]


def run(script, args, frames, buffer, preload):
    env = dict(os.environ)
    if preload:
        env["LD_PRELOAD"] = U
    out = f"/tmp/dropin_{script}_{int(preload)}.pcm"
    t0 = time.perf_counter()
    subprocess.run([R, f"{A2S}/{script}.a2s", "Main", str(frames), str(buffer), "48000", "2", out] + args,
                   check=True, env=env, cwd=A2S)
    return time.perf_counter() - t0, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--buffer", type=int, default=4096, help="a2_Run() buffer in frames (a2play's default: 4096)")
    a = ap.parse_args()
    frames = int(a.seconds * 48000) // a.buffer * a.buffer
    for script, args in CASES:
        tc, fc = run(script, args, frames, a.buffer, False)
        tg, fg = run(script, args, frames, a.buffer, True)
        same = open(fc, "rb").read() == open(fg, "rb").read()
        print(json.dumps({"script": script, "audio_s": frames / 48000.0, "buffer": a.buffer,
                          "cpu_reference_s": round(tc, 3), "gpu_dropin_s": round(tg, 3),
                          "dropin_over_cpu": round(tg / tc, 2),
                          "us_per_fragment_gpu": round(tg / (frames / 64.0) * 1e6, 1),
                          "us_per_buffer_gpu": round(tg / (frames / float(a.buffer)) * 1e6, 1),
                          "bit_identical": same}))


if __name__ == "__main__":
    main()

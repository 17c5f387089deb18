#!/usr/bin/env python3
"""The whole stack at scale: the unmodified reference engine (compiler, VM,
voice walk: one thread) rendering N sustained voices per 64-frame fragment, with
its own units and with the drop-in GPU units (LD_PRELOAD).  This is the number
an application linking the drop-in gets; the engine's per-voice walk and the
per-unit callbacks stay on that one CPU thread.

    python tests/measure/engine_in_loop.py            # one engine state, 1 024 ... 32 768 voices
    python tests/measure/engine_in_loop.py states     # 1 ... 16 engine states (threads) on one GPU

Uses oracle/_ref/ref_bench (compiled reference + timing harness) and
tests/a2s/bench.a2s.  One JSON line per (program, voices).
"""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
U = os.path.join(ROOT, "audiality2_amd", "liba2amd_units.so")
A2S = os.path.join(ROOT, "tests", "a2s")


BUFFER = int(os.environ.get("A2REF_BUFFER", "64"))     # a2_Run() buffer: 64 = realtime-style, 4096 = a2play's


def run(program, voices, frags, preload, threads=1):
    env = dict(os.environ, A2REF_BUFFER=str(BUFFER))
    if preload:
        env["LD_PRELOAD"] = U
    out = subprocess.run([B, "bench.a2s", program, str(voices), str(frags), str(threads)], env=env, cwd=A2S,
                         capture_output=True, text=True, check=True, timeout=900).stdout
    return json.loads(out.strip().splitlines()[-1])


def main():
    for program, cpu_frags in (("OscPan", 300), ("OscPanScripted", 300), ("OscFilterPan", 200), ("OscFilterPanScripted", 200),
                               ("Osc2Pan", 200),
                               ("Osc2PanGroups", 200), ("Fm1Pan", 300), ("Fm2Pan", 150), ("Fm4Pan", 60)):
        for voices in (1024, 4096, 16384, 32768):
            c = run(program, voices, max(cpu_frags * 1024 // voices, 20), False)
            g = run(program, voices, max(300, 4 * BUFFER // 64), True)
            frag_us = g["seconds"] / g["fragments"] * 1e6
            print(json.dumps({"program": program, "voices": voices, "buffer_frames": BUFFER,
                              "cpu_reference_vs_per_s": c["voice_samples_per_s"],
                              "engine_plus_dropin_vs_per_s": g["voice_samples_per_s"],
                              "speedup": round(g["voice_samples_per_s"] / c["voice_samples_per_s"], 2),
                              "dropin_us_per_fragment": round(frag_us, 1),
                              "realtime_at_48k": frag_us < 1333.3}), flush=True)


def states():
    """T independent engine states (the only thread-safe arrangement,
    audiality2.h.cmake:163-166), one host thread each, 16 384 voices per state,
    all sharing one GPU through the drop-in (one backend context + stream per
    state)."""
    per = 16384
    for program in ("OscPan", "OscFilterPan", "Fm4Pan"):
        for t in (1, 2, 4, 8, 16):
            c = run(program, per * t, 40 if program != "Fm4Pan" else 10, False, t)
            g = run(program, per * t, 200, True, t)
            frag_us = g["seconds"] / g["fragments"] * 1e6
            print(json.dumps({"program": program, "states": t, "voices_total": per * t,
                              "cpu_reference_vs_per_s": c["voice_samples_per_s"],
                              "engine_plus_dropin_vs_per_s": g["voice_samples_per_s"],
                              "speedup": round(g["voice_samples_per_s"] / c["voice_samples_per_s"], 2),
                              "dropin_us_per_fragment": round(frag_us, 1),
                              "realtime_at_48k": frag_us < 1333.3}), flush=True)


if __name__ == "__main__":
    import sys
    if sys.argv[1:] == ["states"]:
        states()
    else:
        main()

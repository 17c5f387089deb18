#!/usr/bin/env python3
"""BASELINE configs[0] on the GPU box: the reference's own player, driven exactly as
its benchmark/benchmark.sh:50 drives it (a2play -dbuffer -r44100 <song> -pSong
-st500), with the engine's own units and with the drop-in preloaded.

The reference's songs cannot travel to the GPU box (they are its source text), so
the song is tests/a2s/song.a2s - four tracks, three delay buses, about 60 short
enveloped voices, 6.0 s of one container core per 500 s of audio against 7.8 s for
the reference's benchmark/k2intro.a2s on the same core (both measured in the build
container; benchmark/RESULTS:24 lists 9.891 s for k2intro on an i9-7940X).  Also
renders 30 s through oracle/_ref/ref_render with a2play's buffer size (4096) both
ways and compares the audio bit for bit.

    python tests/measure/song_timing.py [--seconds 500]
"""
import argparse
import json
import os
import subprocess
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
A2PLAY = os.path.join(ROOT, "oracle", "_ref", "a2play")
R = os.path.join(ROOT, "oracle", "_ref", "ref_render")
U = os.path.join(ROOT, "audiality2_amd", "liba2amd_units.so")
A2S = os.path.join(ROOT, "tests", "a2s")


def play(seconds, preload, buffer=None):
    env = dict(os.environ)
    if preload:
        env["LD_PRELOAD"] = U
    cmd = [A2PLAY, "-dbuffer", "-r44100", "song.a2s", "-pSong", f"-st{seconds}"]
    if buffer:
        cmd.insert(3, f"-b{buffer}")
    t0 = time.perf_counter()
    r = subprocess.run(cmd, env=env, cwd=A2S, capture_output=True, text=True, timeout=3600)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, (r.stdout + r.stderr)[-500:]
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=int, default=500)
    a = ap.parse_args()
    same = None
    if os.path.exists(R):
        pcm = []
        for preload in (False, True):
            env = dict(os.environ)
            if preload:
                env["LD_PRELOAD"] = U
            out = f"/tmp/song_{int(preload)}.pcm"
            subprocess.run([R, f"{A2S}/song.a2s", "Main", str(30 * 44100 // 4096 * 4096), "4096", "44100", "2", out, "0.08"],
                           check=True, env=env, cwd=A2S, timeout=600)
            pcm.append(open(out, "rb").read())
        same = pcm[0] == pcm[1] and any(pcm[0])
    play(2, True)       # (first use of the GPU in this process tree: driver start-up is not the song's)
    for buffer in (None, 16384):
        tc = play(a.seconds, False, buffer)
        tg = play(a.seconds, True, buffer)
        print(json.dumps({"command": f"a2play -dbuffer -r44100 song.a2s -pSong -st{a.seconds}" + (f" -b{buffer}" if buffer else ""),
                          "buffer_frames": buffer or 4096, "audio_s": a.seconds,
                          "cpu_reference_s": round(tc, 3), "gpu_dropin_s": round(tg, 3),
                          "dropin_over_cpu": round(tg / tc, 2),
                          "dropin_us_per_buffer": round(tg / (a.seconds * 44100 / (buffer or 4096)) * 1e6, 1),
                          "bit_identical_30s_render": same}), flush=True)


if __name__ == "__main__":
    main()

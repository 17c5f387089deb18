#!/usr/bin/env python3
"""Regenerate the golden fixtures from the COMPILED REFERENCE.

Runs only where /root/reference is mounted (the build container): builds
oracle/_ref with oracle/Makefile, then for every case below lets the
unmodified reference engine render a script while oracle/_ref/ref_tools logs
the calls it makes through the unit plugin surface.  Committed per case:

  <case>.trace.xz   the call trace (input of every backend)
  <case>.hash.npy   FNV-1a 64 of the reference's output, one per 64-frame
                    fragment, channel 0 then channel 1
  <case>.head.npy   the first 4096 output frames [channels, 4096], for eyeballs

plus reference dumps used to pin the stand-alone pieces of the oracle:
  p2i_probe.npy     (pitch, a2_P2I(pitch)) pairs           src/pitch.c:57
  builtin_waves.npz the engine's 24 built-in waves, all mip levels incl. pads
                    (src/waves.c:629-708)
"""
import lzma
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from audiality2_amd.replay import read_pcm  # noqa: E402

REF = "/root/reference"
TOOLS = os.path.join(ROOT, "oracle", "_ref", "ref_tools")
A2S = os.path.join(ROOT, "tests", "a2s")

# name, script, program, frames, args
CASES = [
    ("sustain", f"{A2S}/sustain.a2s", "Main", 9600, ["4", "0.05"]),
    ("filter", f"{A2S}/filter.a2s", "Main", 48000, ["4", "0.02"]),
    ("delaybus", f"{A2S}/delaybus.a2s", "Main", 48000, ["2", "4", "0.05"]),
    ("scripted", f"{A2S}/scripted.a2s", "Main", 48000, ["0.2"]),
    # corner cases of every unit; renders its own waves at load time, which the
    # reference only manages on an A2_REALTIME master state (see ref_tools.c)
    ("edge", f"{A2S}/edge.a2s", "Main", 4 * 48000, ["0.2"]),
    # the FM oscillator units
    ("fm", f"{A2S}/fm.a2s", "Main", 3 * 48000, ["0.15"]),
    ("fmtest3", f"{REF}/benchmark/fmtest3.a2s", "Song", 5 * 48000, []),
    ("fmtest4", f"{REF}/benchmark/fmtest4.a2s", "Song", 5 * 48000, []),
    # dc, waveshaper, dcblock, limiter
    ("fx", f"{A2S}/fx.a2s", "Main", 3 * 48000, ["0.1"]),
    ("dctest", f"{REF}/benchmark/dctest.a2s", "Song", 5 * 48000, []),
    ("wstest", f"{REF}/benchmark/wstest.a2s", "Song", 5 * 48000, []),
    # the engine's own env unit (stays on the CPU) wired to replaced units
    ("envwire", f"{A2S}/envwire.a2s", "Main", 2 * 48000, ["0.2"]),
    # a wave uploaded by the application and released under running oscillators (A2REF_UPLOAD)
    ("unload", f"{A2S}/unload.a2s", "Main", 48000, ["0.1"]),
    # the reference's own benchmark songs (benchmark/RESULTS): all five use only
    # units on the hot path
    ("k2intro", f"{REF}/benchmark/k2intro.a2s", "Song", 6 * 48000, []),
    # BASELINE configs[0] is quoted at 44.1 kHz (benchmark.sh: a2play -dbuffer -r44100): base pitch,
    # filter coefficients and delay lengths all depend on the rate
    ("k2intro44", f"{REF}/benchmark/k2intro.a2s", "Song", 3 * 44100 // 64 * 64, []),
    ("k2epilogue", f"{REF}/benchmark/k2epilogue.a2s", "Song", 5 * 48000, []),
    ("k2loader", f"{REF}/benchmark/k2loader.a2s", "Song", 5 * 48000, []),
    ("k2trance", f"{REF}/benchmark/k2trance.a2s", "Song", 5 * 48000, []),
    ("pulsetronic", f"{REF}/benchmark/pulsetronic.a2s", "Song", 5 * 48000, []),
    # the reference's test/data scripts (round 4; all 27 that start are soaked for 20 s by
    # tests/measure/soak_long.py testdata): every mode of the env unit, ramps, events and messages,
    # recursion, micro-tuning, noise phase, and one of its demo songs
    ("td_envtest", f"{REF}/test/data/envtest.a2s", "Song", 5 * 48000, []),
    ("td_envtest2", f"{REF}/test/data/envtest2.a2s", "Song", 5 * 48000, []),
    ("td_envtest3", f"{REF}/test/data/envtest3.a2s", "Song", 5 * 48000, []),
    ("td_envtest4", f"{REF}/test/data/envtest4.a2s", "Song", 5 * 48000, []),
    ("td_pitchenvtest", f"{REF}/test/data/pitchenvtest.a2s", "Song", 5 * 48000, []),
    ("td_ramptest", f"{REF}/test/data/ramptest.a2s", "Song", 5 * 48000, []),
    ("td_ramptest2", f"{REF}/test/data/ramptest2.a2s", "Song", 5 * 48000, []),
    ("td_ramptestenv", f"{REF}/test/data/ramptestenv.a2s", "Song", 5 * 48000, []),
    ("td_evtest", f"{REF}/test/data/evtest.a2s", "Song", 5 * 48000, []),
    ("td_recursetest", f"{REF}/test/data/recursetest.a2s", "Song", 5 * 48000, []),
    ("td_microtonal", f"{REF}/test/data/microtonal.a2s", "Song", 5 * 48000, []),
    ("td_noisephase", f"{REF}/test/data/noisephase.a2s", "Song", 5 * 48000, []),
    ("td_evilnoises", f"{REF}/test/data/evilnoises.a2s", "Song", 5 * 48000, []),
]

REALTIME_CASES = {"edge", "unload"}
UPLOAD_CASES = {"unload": "20000"}       # release the uploaded wave after this many frames


def fnv1a_fragments(pcm, frag=64):
    """FNV-1a 64 per fragment over the little-endian bytes of ch0 then ch1..."""
    nfr = pcm.shape[1] // frag
    out = np.zeros(nfr, dtype=np.uint64)
    prime = np.uint64(0x100000001B3)
    for f in range(nfr):
        h = np.uint64(0xCBF29CE484222325)
        blk = np.ascontiguousarray(pcm[:, f * frag:(f + 1) * frag]).view(np.uint8).reshape(-1)
        with np.errstate(over="ignore"):
            for b in blk:
                h = (h ^ np.uint64(b)) * prime
        out[f] = h
    return out


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    tmp = "/tmp/a2amd_goldens"
    os.makedirs(tmp, exist_ok=True)
    only = sys.argv[1:]        # (case names: just those, and not the dumps)
    for name, script, prog, frames, args in CASES:
        if only and name not in only and not ("td" in only and name.startswith("td_")):
            continue
        tr, pcm = f"{tmp}/{name}.trace", f"{tmp}/{name}.pcm"
        env = dict(os.environ)
        if name in REALTIME_CASES:
            env["A2REF_REALTIME"] = "1"
        if name in UPLOAD_CASES:
            env["A2REF_UPLOAD"] = UPLOAD_CASES[name]
        rate = "44100" if name.endswith("44") else "48000"
        subprocess.run([TOOLS, "trace", script, prog, str(frames), "64", rate, "2", tr, pcm] + args,
                       check=True, cwd=os.path.dirname(script), env=env)
        with open(tr, "rb") as f, lzma.open(f"{HERE}/{name}.trace.xz", "wb", preset=9 | lzma.PRESET_EXTREME) as g:
            g.write(f.read())
        audio = read_pcm(pcm, 2, 64)
        np.save(f"{HERE}/{name}.hash.npy", fnv1a_fragments(audio))
        np.save(f"{HERE}/{name}.head.npy", audio[:, :4096])
        print(name, audio.shape, "peak", int(np.abs(audio).max()))
    if only:
        return
    subprocess.run([TOOLS, "dump", tmp], check=True)
    np.save(f"{HERE}/p2i_probe.npy", np.fromfile(f"{tmp}/p2i_probe.bin", dtype="<i4").reshape(-1, 2))
    raw = np.fromfile(f"{tmp}/builtin_waves.bin", dtype="<i2")
    pos, waves = 0, {}
    for k in range(24):
        hdr = raw[pos:pos + 26].view("<i4")
        pos += 26
        sizes = hdr[3:13]
        n = int(sum(1 + int(s) + 131 for s in sizes))
        waves[f"w{k}_hdr"] = hdr.copy()
        waves[f"w{k}_data"] = raw[pos:pos + n].copy()
        pos += n
    np.savez_compressed(f"{HERE}/builtin_waves.npz", **waves)
    print("done")


if __name__ == "__main__":
    main()

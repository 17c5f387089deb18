#!/usr/bin/env python3
"""Long-form parity soak: 20 s of every song of the reference's benchmark
directory (the workload of its benchmark.sh), GPU vs the reference's own render.

Two steps, because the reference exists only in the build container and the GPU
only on the GPU box:

  python tests/measure/soak_long.py capture   # here: oracle/_ref/ref_tools renders every
                                      # song and logs the call traces into
                                      # soak_long/ (git-ignored, travels with gpurun)
  python tests/measure/soak_long.py replay    # GPU box: replay each trace through
                                      # liba2amd.so, compare per-fragment hashes

The replay prints one JSON line per song (fragments compared / differing).
"""
import json
import lzma
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DIR = os.path.join(ROOT, "soak_long")
REF = "/root/reference/benchmark"
SONGS = ["k2intro", "k2epilogue", "k2loader", "k2trance", "pulsetronic", "fmtest3", "fmtest4", "dctest", "wstest"]
SECONDS = 20


def capture():
    from audiality2_amd.replay import read_pcm
    from conftest import fnv1a_fragments
    os.makedirs(DIR, exist_ok=True)
    tools = os.path.join(ROOT, "oracle", "_ref", "ref_tools")
    for s in SONGS:
        tr, pcm = f"/tmp/soak_{s}.trace", f"/tmp/soak_{s}.pcm"
        subprocess.run([tools, "trace", f"{REF}/{s}.a2s", "Song", str(SECONDS * 48000), "64", "48000", "2", tr, pcm],
                       check=True, cwd=REF)
        with open(tr, "rb") as f, lzma.open(f"{DIR}/{s}.trace.xz", "wb", preset=6) as g:
            g.write(f.read())
        np.save(f"{DIR}/{s}.hash.npy", fnv1a_fragments(read_pcm(pcm, 2, 64)))
        print(s, os.path.getsize(f"{DIR}/{s}.trace.xz") >> 10, "KiB")


def replay_all():
    import audiality2_amd
    from audiality2_amd.replay import Trace, replay
    from conftest import fnv1a_fragments
    for s in SONGS:
        tr = Trace(f"{DIR}/{s}.trace.xz")
        cfg = tr.config
        be = audiality2_amd.open_backend(cfg["samplerate"], cfg["basepitch"], cfg["channels"], max_batch=64)
        t0 = time.perf_counter()
        out = replay(tr, be, batch=64, check_noise=True)
        dt = time.perf_counter() - t0
        be.close()
        want = np.load(f"{DIR}/{s}.hash.npy")
        got = fnv1a_fragments(out)
        bad = np.nonzero(got != want)[0]
        print(json.dumps({"song": f"benchmark/{s}.a2s", "seconds": SECONDS, "fragments": int(len(want)),
                          "differing": int(len(bad)), "first_differing": [int(b) for b in bad[:3]],
                          "records": len(tr.records), "replay_wall_s": round(dt, 2)}))


if __name__ == "__main__":
    if sys.argv[1:] == ["capture"]:
        capture()
    elif sys.argv[1:] == ["replay"]:
        replay_all()
    else:
        raise SystemExit(__doc__)

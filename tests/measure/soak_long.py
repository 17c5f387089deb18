#!/usr/bin/env python3
"""Long-form parity soak: 20 s of every song of the reference's benchmark
directory (the workload of its benchmark.sh) and of every script with a `Song` under
its test/data directory, GPU vs the reference's own render.

Two steps, because the reference exists only in the build container and the GPU
only on the GPU box:

  python tests/measure/soak_long.py capture [testdata]  # here: oracle/_ref/ref_tools renders every
                                      # song and logs the call traces into
                                      # soak_long/ (git-ignored, travels with gpurun)
  python tests/measure/soak_long.py replay [testdata]   # GPU box: replay each trace through
                                      # liba2amd.so, compare per-fragment hashes

`testdata` = the scripts of the reference's test/data directory that export a `Song` the reference
itself compiles and starts (four of them stop in its compiler with "Operation or feature not
implemented" and are listed as skipped); traces go to soak_long/testdata/.

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
REF_TESTDATA = "/root/reference/test/data"
SONGS = ["k2intro", "k2epilogue", "k2loader", "k2trance", "pulsetronic", "fmtest3", "fmtest4", "dctest", "wstest"]
SECONDS = 20


def which(kind):
    """(label, reference directory, trace directory, songs)"""
    if kind == "testdata":
        d = os.path.join(DIR, "testdata")
        if os.path.isdir(REF_TESTDATA):
            songs = sorted(f[:-4] for f in os.listdir(REF_TESTDATA)
                           if f.endswith(".a2s") and "export Song(" in open(os.path.join(REF_TESTDATA, f)).read())
        else:       # the GPU box: whatever was captured
            songs = sorted(f[:-9] for f in os.listdir(d) if f.endswith(".trace.xz")) if os.path.isdir(d) else []
        return "test/data", REF_TESTDATA, d, songs
    return "benchmark", REF, DIR, SONGS


def capture(kind):
    from audiality2_amd.replay import read_pcm
    from conftest import fnv1a_fragments
    label, ref, out, songs = which(kind)
    os.makedirs(out, exist_ok=True)
    tools = os.path.join(ROOT, "oracle", "_ref", "ref_tools")
    skipped = {}
    for s in songs:
        tr, pcm = f"/tmp/soak_{s}.trace", f"/tmp/soak_{s}.pcm"
        for f in (tr, pcm):
            if os.path.exists(f):
                os.remove(f)
        r = subprocess.run([tools, "trace", f"{ref}/{s}.a2s", "Song", str(SECONDS * 48000), "64", "48000", "2", tr, pcm],
                           cwd=ref, capture_output=True, text=True)
        if r.returncode or not os.path.exists(pcm) or os.path.getsize(tr) < 1024:
            if kind != "testdata":
                raise SystemExit(f"{s}: {r.stderr[-400:]}")
            why = [l for l in r.stderr.splitlines() if "ERROR" in l or "no program" in l]
            skipped[s] = (why or ["?"])[0].strip()
            print(s, "skipped:", skipped[s])
            continue
        with open(tr, "rb") as f, lzma.open(f"{out}/{s}.trace.xz", "wb", preset=6) as g:
            g.write(f.read())
        np.save(f"{out}/{s}.hash.npy", fnv1a_fragments(read_pcm(pcm, 2, 64)))
        print(s, os.path.getsize(f"{out}/{s}.trace.xz") >> 10, "KiB")
    if kind == "testdata":
        with open(f"{out}/skipped.json", "w") as f:
            json.dump(skipped, f, indent=1)


def replay_all(kind):
    import audiality2_amd
    from audiality2_amd.replay import Trace, replay
    from conftest import fnv1a_fragments
    label, _, DIR, songs = which(kind)
    if os.path.exists(f"{DIR}/skipped.json"):
        for s, why in json.load(open(f"{DIR}/skipped.json")).items():
            print(json.dumps({"song": f"{label}/{s}.a2s", "skipped": "the reference does not start it: " + why}))
    for s in songs:
        if not os.path.exists(f"{DIR}/{s}.trace.xz"):
            continue
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
        print(json.dumps({"song": f"{label}/{s}.a2s", "seconds": SECONDS, "fragments": int(len(want)),
                          "differing": int(len(bad)), "first_differing": [int(b) for b in bad[:3]],
                          "records": len(tr.records), "replay_wall_s": round(dt, 2)}))


if __name__ == "__main__":
    kind = sys.argv[2] if len(sys.argv) > 2 else "benchmark"
    if sys.argv[1:2] == ["capture"]:
        capture(kind)
    elif sys.argv[1:2] == ["replay"]:
        replay_all(kind)
    else:
        raise SystemExit(__doc__)

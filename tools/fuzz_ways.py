#!/usr/bin/env python3
"""One seed of the differential fuzzer (tests/fuzz_scripts.py) behind the walk + drop-in with a list of environments:
which of them differs from the reference's render, and from which sample on.
    python tools/fuzz_ways.py 2635 "A2AMD_WIN=1" "A2AMD_WIN=1 A2AMD_VMWIN=0" ..."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_scripts import make_script  # noqa: E402

R = os.path.join(ROOT, "oracle", "_ref", "ref_render")
PRE = os.path.join(ROOT, "audiality2_amd", "liba2amd_walk.so") + " " + os.path.join(ROOT, "audiality2_amd", "liba2amd_units.so")


def main():
    seed = int(sys.argv[1])
    frames = 96000
    tmp = tempfile.mkdtemp(prefix="a2fz")
    sp = f"{tmp}/f.a2s"
    open(sp, "w").write(make_script(seed))
    rate, buffer, channels = (48000, 44100, 96000, 32000)[seed % 4], (64, 37, 256, 1024, 17)[seed % 5], (2, 2, 1)[seed % 3]
    n = frames * rate // 48000 // buffer * buffer
    base = dict(os.environ)
    if seed % 3 == 0:
        base["A2REF_SINK"] = "1"
        if seed % 6 == 0:
            base["A2REF_SOURCE"] = "1"
    print(f"seed {seed}: rate {rate}, buffer {buffer}, channels {channels}, {n} frames")
    ref = None
    for way in [None] + sys.argv[2:]:
        env = dict(base)
        if way is not None:
            env["LD_PRELOAD"] = PRE
            for kv in way.split():
                k, v = kv.split("=", 1)
                env[k] = v
        out = f"{tmp}/o.pcm"
        r = subprocess.run([R, sp, "Main", str(n), str(buffer), str(rate), str(channels), out, "0.15"], env=env, cwd=tmp,
                           capture_output=True, text=True)
        if r.returncode:
            print(way, "FAILED", r.returncode, r.stderr[-300:])
            continue
        a = np.fromfile(out, dtype="<i4")
        if ref is None:
            ref = a
            continue
        bad = np.nonzero(a != ref)[0]
        print("%-60s %s" % (way, "equal" if not len(bad) else
                            f"{len(bad)} samples differ, first at {bad[0]} = frame {bad[0] // channels} (buffer {bad[0] // channels // buffer}, "
                            f"frame {bad[0] // channels % buffer}): {a[bad[:3]]} for {ref[bad[:3]]}"))


if __name__ == "__main__":
    main()

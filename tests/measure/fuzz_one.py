#!/usr/bin/env python3
"""One seed of the differential fuzzer, several ways (what differs from what, and from where):
the reference twice (is IT deterministic on this script?), the drop-in's units alone, behind the walk, behind the
walk without the device VM, and - when tools/ubench/variants holds them - variant builds of the backend.
    python tests/measure/fuzz_one.py 3779 [frames]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_scripts import make_script  # noqa: E402

R = os.path.join(ROOT, "oracle", "_ref", "ref_render")
U = os.path.join(ROOT, "audiality2_amd", "liba2amd_units.so")
W = os.path.join(ROOT, "audiality2_amd", "liba2amd_walk.so")
V = os.path.join(ROOT, "tools", "ubench", "variants")


def main():
    seed = int(sys.argv[1])
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 96000
    tmp = tempfile.mkdtemp(prefix="a2fuzz1")
    sp = f"{tmp}/f.a2s"
    open(sp, "w").write(make_script(seed))
    rate, buffer, channels = (48000, 44100, 96000, 32000)[seed % 4], (64, 37, 256, 1024, 17)[seed % 5], (2, 2, 1)[seed % 3]
    n = frames * rate // 48000 // buffer * buffer
    modes = [("reference", {}), ("reference again", {}), ("units", {"LD_PRELOAD": U}), ("walk + units", {"LD_PRELOAD": W + " " + U}),
             ("walk + units, no device VM", {"LD_PRELOAD": W + " " + U, "A2AMD_NO_VM": "1"}),
             ("units, A2AMD_BATCH=1", {"LD_PRELOAD": U, "A2AMD_BATCH": "1"})]
    if os.path.isdir(V):
        for fn in sorted(os.listdir(V)):
            if fn.startswith("liba2amd_units_") and fn.endswith(".so"):
                modes.append(("units, variant " + fn[15:-3], {"LD_PRELOAD": os.path.join(V, fn)}))
    ref = None
    for name, extra in modes:
        env = dict(os.environ, **extra)
        if seed % 3 == 0:
            env["A2REF_SINK"] = "1"
            if seed % 6 == 0:
                env["A2REF_SOURCE"] = "1"
        out = f"{tmp}/o.pcm"
        r = subprocess.run([R, sp, "Main", str(n), str(buffer), str(rate), str(channels), out, "0.15"], env=env, cwd=tmp,
                           capture_output=True, text=True)
        if r.returncode:
            print(f"{name}: exit {r.returncode} {r.stderr[-200:]}", flush=True)
            continue
        pcm = np.fromfile(out, dtype="<i4")
        if ref is None:
            ref = pcm
            print(f"{name}: {len(pcm)} samples, peak {int(np.abs(pcm).max())}", flush=True)
            continue
        d = np.nonzero(pcm != ref)[0]
        print(f"{name}: " + ("equal" if not len(d) else
              f"{len(d)} samples differ, first at {int(d[0])} (frame {int(d[0]) // channels}, fragment of 64: {int(d[0]) // channels // 64}): "
              f"{pcm[d[0]:d[0] + 4].tolist()} against {ref[d[0]:d[0] + 4].tolist()}"), flush=True)


if __name__ == "__main__":
    main()

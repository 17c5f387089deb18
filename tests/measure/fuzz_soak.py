#!/usr/bin/env python3
"""One-off differential fuzz soak (see tests/test_fuzz_dropin.py): seeds A..B,
reference engine with its own units vs with the drop-in.  Prints failures and a
summary line.   python tests/measure/fuzz_soak.py 24 400
A2FUZZ_WALK=1: the drop-in behind the replaced voice walk (INTEGRATION.md option C)."""
import json
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
if os.environ.get("A2FUZZ_WALK"):
    U = os.path.join(ROOT, "audiality2_amd", "liba2amd_walk.so") + " " + U


def main():
    a, b = int(sys.argv[1]), int(sys.argv[2])
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 96000
    tmp = tempfile.mkdtemp(prefix="a2fuzz")
    bad, errors, silent, ref_crashed = [], [], 0, []
    for seed in range(a, b):
        sp = f"{tmp}/f.a2s"
        open(sp, "w").write(make_script(seed))
        outs, sinks = [], []
        for pre in (False, True):
            env = dict(os.environ)
            if pre:
                env["LD_PRELOAD"] = U
            if seed % 3 == 0:       # Main is a group with an xinsert (tests/fuzz_scripts.py): give it clients
                env["A2REF_SINK"] = "1"
                if seed % 6 == 0:
                    env["A2REF_SOURCE"] = "1"
            rate, buffer, channels = (48000, 44100, 96000, 32000)[seed % 4], (64, 37, 256, 1024, 17)[seed % 5], (2, 2, 1)[seed % 3]
            n = frames * rate // 48000 // buffer * buffer
            r = subprocess.run([R, sp, "Main", str(n), str(buffer), str(rate), str(channels), f"{tmp}/o{int(pre)}.pcm", "0.15"],
                               env=env, cwd=tmp, capture_output=True, text=True)
            if r.returncode == -11 and not pre:
                # the REFERENCE crashed on its own units (seeds >= 3000: env_ProcessLUT reads its table with whatever
                # index a re-targeted unity ramp left it, src/units/env.c:122-126): nothing to compare with
                ref_crashed.append(seed)
                break
            if r.returncode:
                errors.append((seed, pre, r.stderr[-300:]))
                break
            outs.append(np.fromfile(f"{tmp}/o{int(pre)}.pcm", dtype="<i4"))
            sinks.append([ln for ln in r.stdout.splitlines() if ln.startswith("sink ")])
        if len(outs) == 2 and sinks[0] != sinks[1]:
            bad.append((seed, -1, -1))
            print("SINK MISMATCH", seed, sinks, flush=True)
        if len(outs) == 2:
            if not outs[0].any():
                silent += 1
            d = np.nonzero(outs[0] != outs[1])[0]
            if len(d):
                bad.append((seed, int(len(d)), int(d[0])))
                print("MISMATCH", seed, len(d), d[:3], flush=True)
    for e in errors:
        print("ERROR", e, flush=True)
    print(json.dumps({"seeds": [a, b], "frames": frames, "mismatching_seeds": [x[0] for x in bad],
                      "errors": len(errors), "reference_crashed_on_its_own": ref_crashed, "silent": silent, "walk": bool(os.environ.get("A2FUZZ_WALK"))}))


if __name__ == "__main__":
    main()

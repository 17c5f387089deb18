#!/usr/bin/env python3
"""Which envloops voice differs between the CPU engine and the device VM?  One render per voice type."""
import os, re, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A2S = os.path.join(ROOT, "tests", "a2s")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_render")
UNITS = os.path.join(ROOT, "audiality2_amd", "liba2amd_units.so")
WALK = os.path.join(ROOT, "audiality2_amd", "liba2amd_walk.so")
src = open(os.path.join(A2S, "envloops.a2s")).read()
head = src[:src.index("export Main")]
solos = {
    "pluck": "Pluck -1 V C.EXP3 C.IEXP2", "pluck_spline": "Pluck -.2 V C.SPLINE C.SPLINE", "pluck_inv": "Pluck .9 V C.IEXP1 C.EXP6",
    "swell": "Swell -.5 V C.EXP4", "swell_lin": "Swell .2 V C.LINEAR", "timed": "Timed 0 V", "wander": "Wander -.7 V",
    "siren": "Siren -1.2 V", "wah": "Wah -1.5 V", "sweep": "Sweep -2 V", "plain": "Plain .6 V", "group": "Group .1 V",
    "mortal": "Mortal -.4 V 333",
}
def render(path, tag, preload=None, extra=None, buffer=64, frames=48000):
    out = f"/tmp/envb_{tag}.pcm"
    env = dict(os.environ, A2AMD_WALK_STATS="1")
    env.pop("LD_PRELOAD", None)
    if preload:
        env["LD_PRELOAD"] = preload
    env.update(extra or {})
    r = subprocess.run([REF, path, "Main", str(frames), str(buffer), "48000", "2", out, "0.08"], env=env, cwd=A2S,
                       capture_output=True, text=True, timeout=600)
    if r.returncode:
        print(tag, "FAILED", r.stderr[-300:])
        return None, None
    m = re.search(r"(\d+) voices handed to the device VM, (\d+) taken back", r.stderr)
    return np.fromfile(out, dtype="<i4"), m.groups() if m else None
only = sys.argv[1:]
for name, call in solos.items():
    if only and name not in only:
        continue
    path = os.path.join(A2S, f"_envb_{name}.a2s")
    open(path, "w").write(head + "export Main(V=.08)\n{\n\t" + call + "\n\tfor { d 100000 }\n}\n")
    try:
        cpu, _ = render(path, "cpu")
        for tag, pre, extra in (("units", UNITS, None), ("novm", f"{WALK} {UNITS}", {"A2AMD_NO_VM": "1"}), ("vm", f"{WALK} {UNITS}", None)):
            got, st = render(path, tag, pre, extra)
            if got is None:
                continue
            bad = np.nonzero(cpu != got)[0]
            print(f"{name:14s} {tag:6s} {'same' if not len(bad) else f'{len(bad)} differ, first frame {bad[0] // 2} (buffer {bad[0] // 128})'}  vm {st}")
    finally:
        os.unlink(path)

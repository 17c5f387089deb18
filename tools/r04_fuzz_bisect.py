"""Which line of a fuzz script's Main makes the drop-in (units only) differ from the CPU engine?
usage: python tools/r04_fuzz_bisect.py seed rate buffer channels [ENV=VALUE ...]"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_scripts import make_script
seed, rate, buf, ch = map(int, sys.argv[1:5])
extra = dict(a.split("=", 1) for a in sys.argv[5:])
R = os.path.join(ROOT, "oracle", "_ref", "ref_render")
U = os.path.join(ROOT, "audiality2_amd", "liba2amd_units.so")
W = os.path.join(ROOT, "audiality2_amd", "liba2amd_walk.so")
O = "/tmp/fzb"
os.makedirs(O, exist_ok=True)
text = make_script(seed).split("\n")
m0 = next(i for i, l in enumerate(text) if l.startswith("export Main"))
body = [i for i in range(m0 + 2, len(text)) if text[i].startswith("\t") and not text[i].startswith("\t\t") and
        text[i].strip() and text[i].strip().split()[0] not in ("struct", "vol", "!P", "for", "}", "d")]
inner = [i for i in range(m0 + 2, len(text)) if text[i].startswith("\t\t") and ":" in text[i].split(";")[0] or
         (text[i].startswith("\t\t") and text[i].strip()[0] in "VBHL")]
frames = rate * 3 // 2 // buf * buf


def run(lines, mode):
    open(f"{O}/s.a2s", "w").write("\n".join(lines) + "\n")
    env = dict(os.environ, **extra)
    env.pop("LD_PRELOAD", None)
    if mode == 1:
        env["LD_PRELOAD"] = U
    if mode == 2:
        env["LD_PRELOAD"] = f"{W} {U}"
    r = subprocess.run([R, f"{O}/s.a2s", "Main", str(frames), str(buf), str(rate), str(ch), f"{O}/o{mode}.pcm", "0.15"],
                       env=env, cwd=O, capture_output=True, text=True)
    if r.returncode:
        return None
    return np.fromfile(f"{O}/o{mode}.pcm", dtype="<i4")


def differ(lines, mode=1):
    a, b = run(lines, 0), run(lines, mode)
    if a is None or b is None:
        return "error"
    bad = np.nonzero(a != b)[0]
    return (len(bad), int(bad[0]) // (ch * buf) if len(bad) else -1)


mode = int(os.environ.get("BISECT_MODE", "1"))
print("full:", differ(text, mode), flush=True)
for i in body + inner:
    t = list(text)
    t[i] = ""
    print(f"without line {i} [{text[i].strip()[:60]}]:", differ(t, mode), flush=True)

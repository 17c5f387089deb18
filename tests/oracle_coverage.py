#!/usr/bin/env python3
"""Which lines of the oracle do the committed traces reach?  (test tooling)

Builds oracle/a2o.c with gcov instrumentation in a scratch directory, replays
every tests/golden/*.trace.xz through it and prints the lines of the DSP part
(unit process functions) that were never executed.  Used to decide which corner
cases tests/a2s/edge.a2s still has to provoke; expected output at the time of
writing: only wtosc_check_unloaded (a wave released through the API while an
oscillator plays it - not reachable from a script).

    python tests/oracle_coverage.py
"""
import ctypes
import glob
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from audiality2_amd.replay import Trace, replay  # noqa: E402
from conftest import make_oracle  # noqa: E402


def main():
    tmp = tempfile.mkdtemp(prefix="a2ocov")
    src = open(os.path.join(ROOT, "oracle", "a2o.c")).read()
    src = src.replace('"../include/a2amd.h"', f'"{ROOT}/include/a2amd.h"')
    hdr = open(os.path.join(ROOT, "oracle", "a2o.h")).read()
    hdr = hdr.replace('"../include/a2amd.h"', f'"{ROOT}/include/a2amd.h"')
    open(f"{tmp}/a2o.c", "w").write(src)
    open(f"{tmp}/a2o.h", "w").write(hdr)
    subprocess.run(["gcc", "-O0", "-fwrapv", "-fPIC", "--coverage", "-shared", "-o", "libcov.so", "a2o.c", "-lm"],
                   check=True, cwd=tmp)
    # replay in a child process: the counters are written when it exits
    subprocess.run([sys.executable, os.path.abspath(__file__), "--replay", f"{tmp}/libcov.so"], check=True)
    gcno = glob.glob(f"{tmp}/*.gcno")[0]
    print(subprocess.run(["gcov", "-o", gcno, "a2o.c"], cwd=tmp, capture_output=True, text=True).stdout.strip())
    dsp = False
    for line in open(f"{tmp}/a2o.c.gcov"):
        m = re.match(r"\s*([#\-0-9*]+):\s*(\d+):(.*)", line)
        if not m:
            continue
        if "static void wtosc_run_pitch" in m.group(3):
            dsp = True
        if "static void resolve_out" in m.group(3):
            dsp = False
        if dsp and m.group(1).startswith("#"):
            print("never run:", m.group(2), m.group(3))


def replay_all(libpath):
    lib = ctypes.CDLL(libpath)
    for f in sorted(glob.glob(os.path.join(HERE, "golden", "*.trace.xz"))):
        tr = Trace(f)
        cfg = tr.config
        be = make_oracle(lib, cfg["samplerate"], cfg["basepitch"], cfg["channels"])
        replay(tr, be, batch=64, check_noise=True)
        be.close()
        print("replayed", os.path.basename(f))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--replay":
        replay_all(sys.argv[2])
    else:
        main()

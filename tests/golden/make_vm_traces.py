#!/usr/bin/env python3
"""Fixtures that pin the device VM's interpreter (audiality2_amd/csrc/a2amd_vmcore.h) on the CPU: for each looping
voice program of tests/a2s/vmloops.a2s, what the COMPILED REFERENCE's own VM did with the voice - every unit register
write (a2_VoiceControl, src/core.c:143-149) and every window (a2_VoiceProcess, core.c:1847-1880) over 600 fragments -
together with the program text, VM state and register wiring at the moment the trace starts (oracle/ref_vmtrace.c,
built by oracle/Makefile against the reference's sources where they lie).

    python tests/golden/make_vm_traces.py        ->  tests/golden/vm_traces.json.xz

tests/test_device_vm.py::test_host_interpreter_against_the_reference_vm replays each through a2amd_vm_trace_host()
(the host copy of the interpreter: the same header the kernel is compiled from) and compares record by record."""
import json
import lzma
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_vmtrace")
A2S = os.path.join(ROOT, "tests", "a2s")

# program, arguments (as the script's Main passes them), warm-up fragments
CASES = [("Lfo", [-1, .08, -.5], 20), ("Lfo", [.37, .05, .8], 33), ("Trem", [0, .08], 25), ("Arp", [.5, .08], 31), ("Duo", [-1.5, .08], 27),
         ("Sweep", [-2, .08], 40), ("Pad", [-.5, .08], 22), ("Fm", [0, .08], 29), ("Fx", [.3, .08], 35), ("Dc", [.024], 21),
         ("Echo", [1, .08], 26), ("Tr", [-1, .08], 24)]

if __name__ == "__main__":
    out = []
    for prog, args, warm in CASES:
        r = subprocess.run([EXE, "vmloops.a2s", prog, str(warm), "600"] + [repr(a) for a in args], cwd=A2S, capture_output=True,
                           text=True, check=True)
        d = json.loads(r.stdout)
        print(prog, args, len(d["code"]), "code words,", len(d["events"]), "events")
        out.append(d)
    path = os.path.join(HERE, "vm_traces.json.xz")
    with lzma.open(path, "wt") as f:
        json.dump(out, f, separators=(",", ":"))
    print("written", path, os.path.getsize(path), "bytes")

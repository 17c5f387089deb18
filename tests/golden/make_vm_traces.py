#!/usr/bin/env python3
"""Fixtures that pin the device VM's interpreter (audiality2_amd/csrc/a2amd_vmcore.h) on the CPU: for each looping
voice program of tests/a2s/vmloops.a2s and tests/a2s/envtrace.a2s (voices with env units), what the COMPILED REFERENCE's own VM did with the voice - every unit register
write (a2_VoiceControl, src/core.c:143-149; an env's through its control wire, src/units/env.c:131), every window
(a2_VoiceProcess, core.c:1847-1880) and every window of the root voice (= the backend's fragments) over 600 fragments -
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

# script, program, arguments (as the script's Main passes them), warm-up fragments
CASES = [("vmloops", "Lfo", [-1, .08, -.5], 20), ("vmloops", "Lfo", [.37, .05, .8], 33), ("vmloops", "Trem", [0, .08], 25),
         ("vmloops", "Arp", [.5, .08], 31), ("vmloops", "Duo", [-1.5, .08], 27), ("vmloops", "Sweep", [-2, .08], 40),
         ("vmloops", "Pad", [-.5, .08], 22), ("vmloops", "Fm", [0, .08], 29), ("vmloops", "Fx", [.3, .08], 35),
         ("vmloops", "Dc", [.024], 21), ("vmloops", "Echo", [1, .08], 26), ("vmloops", "Tr", [-1, .08], 24),
         # env units: traced from the program's first delay on, before any env's target was written
         # (mode constants as 16:16 numbers: EXP3 = 4, IEXP2 = -3, SPLINE = -1, EXP7 = 8, IEXP7 = -8, EXP4 = 5, LINEAR = 1)
         ("envtrace", "Front", [-1, .08, 4, -3], 0), ("envtrace", "Front", [-.2, .08, -1, -1], 0),
         ("envtrace", "Front", [.4, .08, 8, -8], 0), ("envtrace", "Front", [.9, .08, -2, 7], 0),
         ("envtrace", "Behind", [-.5, .08, 5], 0), ("envtrace", "Behind", [.2, .08, 1], 0), ("envtrace", "Timed", [0, .08], 0),
         ("envtrace", "Two", [-.7, .08], 0), ("envtrace", "Siren", [-1.2, .08], 0), ("envtrace", "Plain", [.6, .08], 0),
         # a table look-up that runs past the table's end into the next one (round 4's soak, fuzz seed 3779)
         ("envtrace", "Short", [-.5, .08], 0),
         # round 5: programs that run out, sleep, call, divide by registers, draw random numbers (tests/a2s/vmnotes.a2s) -
         # the device VM has them for the stretch in front of the first VM run that needs the engine (a2amd_vm_exit_time);
         # started detached (no handle), the voices that run out are traced until they are gone
         ("vmnotes", "Note", [-1, .08, -.5], 0, "detached"), ("vmnotes", "Note", [.4, .05, .5], 0, "detached"),
         ("vmnotes", "Pluck", [-.5, .08], 0, "detached"), ("vmnotes", "Tail", [.2, .08], 0, "detached"),
         ("vmnotes", "Div", [0, .08], 0, "detached"), ("vmnotes", "Call", [-.3, .08], 0, "detached"),
         ("vmnotes", "Rnd", [.1, .08], 0, "detached"), ("vmnotes", "Sleeper", [-.5, .08], 0, "detached"),
         ("vmnotes", "Hang", [.3, .08], 0)]

if __name__ == "__main__":
    out = []
    for script, prog, args, warm, *how in CASES:
        env = dict(os.environ)
        if "detached" in how:
            env["A2_VMTRACE_DETACHED"] = "1"
        r = subprocess.run([EXE, script + ".a2s", prog, str(warm), "600"] + [repr(a) for a in args], cwd=A2S, capture_output=True,
                           text=True, check=True, env=env)
        d = json.loads(r.stdout)
        print(prog, args, len(d["code"]), "code words,", len(d["events"]), "events", "gone in fragment %d" % d["gone"] if "gone" in d else "")
        out.append(d)
    path = os.path.join(HERE, "vm_traces.json.xz")
    with lzma.open(path, "wt") as f:
        json.dump(out, f, separators=(",", ":"))
    print("written", path, os.path.getsize(path), "bytes")

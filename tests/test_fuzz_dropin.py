"""Differential fuzzing of the whole stack: random A2S scripts (tests/fuzz_scripts.py:
every replaced unit, random structures, ramps, timing, births and deaths, the
engine RNG in play) rendered by the unmodified reference engine with its own
units and with the drop-in GPU units, on the same box.  Any differing sample is
a failure; the seed reproduces the script."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from fuzz_scripts import make_script

REF_RENDER = os.path.join(ROOT, "oracle", "_ref", "ref_render")
UNITS_SO = os.path.join(ROOT, "audiality2_amd", "liba2amd_units.so")
WALK_SO = os.path.join(ROOT, "audiality2_amd", "liba2amd_walk.so")     # INTEGRATION.md option C
SECONDS = 1.5


def engine_config(seed):
    """Sample rate, a2_Run() buffer size and master channels also vary with the seed."""
    return (48000, 44100, 96000, 32000)[seed % 4], (64, 37, 256, 1024, 17)[seed % 5], (2, 2, 1)[seed % 3]


@pytest.mark.gpu
# (>= 1000: with sleeping subtrees, fuzz_scripts.pad_programs; >= 2000: with voices that loop over the device VM's
# part of the instruction set, fuzz_scripts.loop_programs - taken by the device VM, sent messages, killed, detached)
# (3779: found by the end-of-round-4 soak - an env unit's re-targeted unity ramp overshoots 1.0 and the engine's table
# look-up reads on into the next table of its array, env.c:127-133; the drop-in's tables are laid out the same way)
@pytest.mark.parametrize("seed", list(range(24)) + list(range(1000, 1012)) + list(range(2000, 2024)) + list(range(3000, 3016)) +
                         [3779, 2635])    # (3779: round 4's env table finding; 2635: round 5, a cutoff ramp that ends while k_vm_win leaves the voice alone)
def test_random_script_matches_reference(tmp_path, seed):
    if not (os.path.exists(REF_RENDER) and os.path.exists(UNITS_SO)):
        pytest.skip("oracle/_ref (compiled reference) or liba2amd_units.so not built")
    script = tmp_path / f"fuzz{seed}.a2s"
    script.write_text(make_script(seed))
    outs, sinks = [], []
    # the engine's own units; the drop-in units; the drop-in units behind the replaced voice walk
    for preload in (0, 1, 2):
        out = tmp_path / f"o{preload}.pcm"
        env = dict(os.environ)
        if preload:
            env["LD_PRELOAD"] = UNITS_SO if preload == 1 else f"{WALK_SO} {UNITS_SO}"
        if seed % 3 == 0:       # Main is a group with an xinsert: give it clients
            env["A2REF_SINK"] = "1"
            if seed % 6 == 0:
                env["A2REF_SOURCE"] = "1"
        rate, buffer, channels = engine_config(seed)
        frames = int(SECONDS * rate) // buffer * buffer
        r = subprocess.run([REF_RENDER, str(script), "Main", str(frames), str(buffer), str(rate), str(channels),
                            str(out), "0.15"],
                           env=env, cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (seed, preload, r.stderr[-800:])
        outs.append(np.fromfile(out, dtype="<i4"))
        sinks.append([ln for ln in r.stdout.splitlines() if ln.startswith("sink ")])
    assert sinks[0] == sinks[1] == sinks[2] and len(sinks[0]) == (seed % 3 == 0), (seed, sinks)
    assert outs[0].any(), f"seed {seed}: the reference rendered silence"
    for k in (1, 2):
        bad = np.nonzero(outs[0] != outs[k])[0]
        assert len(bad) == 0, (f"seed {seed}{' (voice walk replaced)' if k == 2 else ''}: {len(bad)} samples differ, "
                               f"first at {bad[:3]}; config (rate, buffer, channels) = {engine_config(seed)}; "
                               f"script: python tests/fuzz_scripts.py {seed}")

"""End to end through the reference's own engine: the unmodified compiled
reference (oracle/_ref) loads the drop-in units (liba2amd_units.so) by symbol
interposition, runs its A2S compiler, VM and voice tree on the CPU, and every
unit callback lands on the GPU backend.  The audio a2_Run() hands back must be
the audio the reference renders with its own units."""
import os
import subprocess

import numpy as np
import pytest

from audiality2_amd.replay import read_pcm
from conftest import GOLDEN, ROOT, differing_fragments, fnv1a_fragments

REF_RENDER = os.path.join(ROOT, "oracle", "_ref", "ref_render")
UNITS_SO = os.path.join(ROOT, "audiality2_amd", "liba2amd_units.so")
A2S = os.path.join(ROOT, "tests", "a2s")

# name, program args, frames (as in tests/golden/make_goldens.py)
CASES = [("sustain", ["4", "0.05"], 9600), ("filter", ["4", "0.02"], 48000),
         ("delaybus", ["2", "4", "0.05"], 48000), ("scripted", ["0.2"], 48000),
         # renders three waves in offline substates while the script loads: those
         # states get GPU contexts of their own next to the master state's
         ("edge", ["0.2"], 4 * 48000),
         # the FM oscillator units
         ("fm", ["0.15"], 3 * 48000),
         # dc, waveshaper, dcblock, limiter
         ("fx", ["0.1"], 3 * 48000),
         # the engine's env unit on the CPU driving GPU units through control wires
         ("envwire", ["0.2"], 2 * 48000),
         # the application releases a wave under running oscillators
         ("unload", ["0.1"], 48000)]
# no fixture, compared with the reference run on the same box: 10 s of voice churn
# (20 000 births and deaths, ~300 voices alive)
LIVE_CASES = [("churn", ["0.05"], 10 * 48000)]
REALTIME_CASES = {"edge", "unload"}      # see tests/golden/make_goldens.py
UPLOAD_CASES = {"unload": "20000"}


def need_ref():
    if not (os.path.exists(REF_RENDER) and os.path.exists(UNITS_SO)):
        pytest.skip("oracle/_ref (compiled reference) or liba2amd_units.so not built")


@pytest.mark.gpu
@pytest.mark.parametrize("name,args,frames", CASES)
def test_engine_with_dropin_units_matches_reference(tmp_path, name, args, frames):
    need_ref()
    out = tmp_path / f"{name}.pcm"
    env = dict(os.environ, LD_PRELOAD=UNITS_SO)
    if name in REALTIME_CASES:
        env["A2REF_REALTIME"] = "1"
    if name in UPLOAD_CASES:
        env["A2REF_UPLOAD"] = UPLOAD_CASES[name]
    subprocess.run([REF_RENDER, f"{A2S}/{name}.a2s", "Main", str(frames), "64", "48000", "2", str(out)] + args,
                   check=True, env=env, cwd=A2S, timeout=600)
    audio = read_pcm(out, 2, 64)
    want = np.load(os.path.join(GOLDEN, f"{name}.hash.npy"))
    got = fnv1a_fragments(audio)
    bad = differing_fragments(name, got, want)
    assert len(bad) == 0, f"{len(bad)} fragments differ from the reference render, first {bad[:5]}"


@pytest.mark.gpu
@pytest.mark.parametrize("channels,buffer", [(1, 64), (2, 37), (1, 50)])
def test_dropin_mono_root_and_odd_buffers(tmp_path, channels, buffer):
    """a2_Render-style 1-channel states use the mono root driver (panmix 2->1,
    audiality2.c:282-291); buffers that are not multiples of 64 give partial
    fragments.  Compared with the same binary without the drop-in."""
    need_ref()
    outs = []
    for pre in (None, UNITS_SO):
        out = tmp_path / f"o{int(pre is not None)}.pcm"
        env = dict(os.environ)
        if pre:
            env["LD_PRELOAD"] = pre
        subprocess.run([REF_RENDER, f"{A2S}/scripted.a2s", "Main", str(buffer * 300), str(buffer), "48000",
                        str(channels), str(out), "0.2"], check=True, env=env, cwd=A2S, timeout=600)
        outs.append(np.fromfile(out, dtype="<i4"))
    assert outs[0].any()
    assert np.array_equal(outs[0], outs[1])


def test_reference_render_is_reproducible(tmp_path):
    """The harness itself (no drop-in) reproduces the golden fixture: the
    fixtures are what the reference renders, run to run."""
    need_ref()
    out = tmp_path / "s.pcm"
    subprocess.run([REF_RENDER, f"{A2S}/sustain.a2s", "Main", "9600", "64", "48000", "2", str(out), "4", "0.05"],
                   check=True, cwd=A2S, timeout=120)
    audio = read_pcm(out, 2, 64)
    assert np.array_equal(fnv1a_fragments(audio), np.load(os.path.join(GOLDEN, "sustain.hash.npy")))


def test_dropin_refuses_to_run_without_gpu(tmp_path):
    import torch
    need_ref()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([REF_RENDER, f"{A2S}/sustain.a2s", "Main", "640", "64", "48000", "2",
                        str(tmp_path / "x.pcm"), "1", "0.05"],
                       env=dict(os.environ, LD_PRELOAD=UNITS_SO), cwd=A2S, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_dropin_refuses_mixed_chains(tmp_path):
    """A CPU unit with audio ports inside a GPU-rendered chain would process
    silence: the drop-in aborts with a message instead (no silent fallback); a
    voice that *starts* in a CPU unit is refused at instantiation."""
    need_ref()
    env = dict(os.environ, LD_PRELOAD=UNITS_SO, A2REF_FOREIGN="1")
    for script in ("mixed", "mixedhead"):   # (both fine on the CPU)
        subprocess.run([REF_RENDER, f"{A2S}/{script}.a2s", "Main", "640", "64", "48000", "2", str(tmp_path / "c.pcm"), "0.1"],
                       env=dict(os.environ, A2REF_FOREIGN="1"), cwd=A2S, check=True, timeout=120)
        assert np.fromfile(tmp_path / "c.pcm", dtype="<i4").any()
    r = subprocess.run([REF_RENDER, f"{A2S}/mixed.a2s", "Main", "640", "64", "48000", "2", str(tmp_path / "m.pcm"), "0.1"],
                       env=env, cwd=A2S, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "mixed CPU/GPU chains are not supported" in r.stderr, r.stderr[-500:]
    r = subprocess.run([REF_RENDER, f"{A2S}/mixedhead.a2s", "Main", "640", "64", "48000", "2", str(tmp_path / "h.pcm"), "0.1"],
                       env=env, cwd=A2S, capture_output=True, text=True, timeout=120)
    assert "takes its input from a unit that is not replaced" in r.stderr, r.stderr[-500:]
    # an insert client (reads AND writes) on a voice other than the root would need
    # that voice's audio on the host in the middle of the GPU batch: refused
    cmd = [REF_RENDER, f"{A2S}/sinkgroup.a2s", "Main", "640", "64", "48000", "2", str(tmp_path / "s.pcm"), "0.1"]
    r = subprocess.run(cmd, env=dict(env, A2REF_INSERT="1"), cwd=A2S, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "an insert client" in r.stderr, r.stderr[-500:]


@pytest.mark.gpu
@pytest.mark.parametrize("script,frames", [("sinkgroup", 9600), ("clients", 48000)])
@pytest.mark.parametrize("clients", [(), ("SINK",), ("SOURCE",), ("SOURCE", "SINK"), ("STREAMS",), ("STREAMS", "SINK"),
                                     # ... and the voice is killed 0.7 ms into a buffer: its clients are owed
                                     # the windows of a fragment that is rendered after the voice is gone
                                     ("SINK", "KILL"), ("SOURCE", "STREAMS", "SINK", "KILL")])
def test_dropin_serves_sink_and_source_clients(tmp_path, script, frames, clients):
    """a2_SinkCallback / a2_SourceCallback on a voice in the middle
    of the graph (a group: inline; panmix; xinsert), and their buffered variants
    a2_OpenSink / a2_OpenSource (STREAMS).  The sink is handed what it is handed
    on the CPU (hash over everything it saw, in order), the source's audio is in
    the output, the output is the reference's."""
    need_ref()
    res = []
    for preload in (False, True):
        out = tmp_path / f"c{int(preload)}.pcm"
        env = dict(os.environ, **{f"A2REF_{c}": "1" for c in clients})
        if "KILL" in clients:
            env["A2REF_KILL"] = str(frames // 2)
        if preload:
            env["LD_PRELOAD"] = UNITS_SO
        r = subprocess.run([REF_RENDER, f"{A2S}/{script}.a2s", "Main", str(frames), "64", "48000", "2", str(out), "0.1"],
                           env=env, cwd=A2S, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        res.append(([ln for ln in r.stdout.splitlines() if ln.startswith("sink ")], np.fromfile(out, dtype="<i4")))
    assert res[0][1].any()
    assert len(res[0][0]) == len([c for c in clients if c in ("SINK", "STREAMS")])
    if "KILL" in clients:
        assert all(f" frames {frames // 2 + 33} " in ln for ln in res[0][0]), res[0][0]
    assert not any(" frames 0 " in ln or "sink peak 0 " in ln for ln in res[0][0])
    assert res[0][0] == res[1][0]
    a, b = (r[1].reshape(-1, 2, 64) for r in res)          # [buffer, channel, frame]
    if "STREAMS" in clients:
        # A source stream fills only its own channel (1) of the buffers the engine
        # hands it (a2_sourcestream_process, xinsertapi.c:287-318) and xi_process
        # mixes ALL of them into the output (xinsert.c:113-118) - the others are
        # uninitialised stack arrays (xinsert.c:66).  What the reference adds to
        # channel 0 is whatever its stack held; the drop-in adds silence.
        a, b = a[:, 1], b[:, 1]
    assert np.array_equal(a, b)


@pytest.mark.gpu
def test_several_engine_states_share_the_gpu():
    """Independent engine states on their own host threads (the reference's
    only thread-safe arrangement) each get a backend context and a stream of
    their own from the drop-in; every state's audio equals the CPU render."""
    import json
    bench = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    need_ref()
    res = []
    for preload in (False, True):
        env = dict(os.environ, A2REF_HASH="1")
        if preload:
            env["LD_PRELOAD"] = UNITS_SO
        out = subprocess.run([bench, "bench.a2s", "OscFilterPan", str(6 * 520), "40", "6"], env=env, cwd=A2S,
                             capture_output=True, text=True, check=True, timeout=600).stdout
        res.append(json.loads(out.strip().splitlines()[-1]))
    assert res[0]["active_voices"] == res[1]["active_voices"] > 6 * 500
    assert len(res[0]["hashes"]) == 6 and res[0]["hashes"] == res[1]["hashes"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,args,frames", LIVE_CASES)
def test_engine_with_dropin_units_matches_reference_run(tmp_path, name, args, frames):
    need_ref()
    outs = []
    for preload in (False, True):
        out = tmp_path / f"{name}{int(preload)}.pcm"
        env = dict(os.environ)
        if preload:
            env["LD_PRELOAD"] = UNITS_SO
        subprocess.run([REF_RENDER, f"{A2S}/{name}.a2s", "Main", str(frames), "64", "48000", "2", str(out)] + args,
                       check=True, env=env, cwd=A2S, timeout=900)
        outs.append(np.fromfile(out, dtype="<i4"))
    assert outs[0].any()
    bad = np.nonzero(outs[0] != outs[1])[0]
    assert len(bad) == 0, f"{len(bad)} samples differ, first at frame {bad[0] // 2 if len(bad) else -1}"


@pytest.mark.gpu
@pytest.mark.parametrize("program,env", [
    ("Player", {"A2REF_SOURCE": "1"}), ("Player", {"A2REF_SRCSTREAM": "0"}), ("Player2", {"A2REF_SOURCE": "1"}),
    ("Player", {}), ("Capture", {"A2REF_SINK": "1"}), ("Capture", {"A2REF_SINKSTREAM": "0"}),
    ("Monitor", {"A2REF_SINK": "1"}), ("Monitor", {"A2REF_SINKSTREAM": "1", "A2REF_SINK": "1"}),
    ("Monitor", {"A2REF_SINK": "1", "A2REF_KILL": "4800"}), ("Player2", {"A2REF_SOURCE": "1", "A2REF_KILL": "4800"})])
def test_dropin_xsource_and_xsink_voices(tmp_path, program, env):
    """The voice shapes of the reference's own stream tests (test/streamtest.c,
    test/streamstress.c): 'xsource; panmix' fed by the application through a
    callback or a stream, 'inline; xsink' handing it what plays underneath."""
    need_ref()
    res = []
    for preload in (False, True):
        out = tmp_path / f"x{int(preload)}.pcm"
        e = dict(os.environ, **env)
        if preload:
            e["LD_PRELOAD"] = UNITS_SO
        r = subprocess.run([REF_RENDER, f"{A2S}/streams.a2s", program, "9600", "64", "48000", "2", str(out), "0.3"],
                           env=e, cwd=A2S, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        res.append(([ln for ln in r.stdout.splitlines() if ln.startswith("sink ")], np.fromfile(out, dtype="<i4")))
    sinks = len([k for k in env if "SINK" in k])
    assert len(res[0][0]) == sinks and not any(" frames 0 " in ln for ln in res[0][0])
    assert res[0][0] == res[1][0]
    assert res[0][1].any() == (program != "Capture" and bool(env))
    assert np.array_equal(res[0][1], res[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("name,args,frames", [c for c in CASES if c[0] in ("sustain", "scripted", "fm", "fx")])
def test_in_tree_build_option(tmp_path, name, args, frames):
    """INTEGRATION.md option B: the engine built WITHOUT the replaced units' sources
    and with its four wrapped descriptors renamed <name>_cpu, linked against the
    drop-in (oracle/Makefile: optionb) - no LD_PRELOAD involved."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_render_b")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_render_b not built")
    out = tmp_path / f"{name}.pcm"
    subprocess.run([exe, f"{A2S}/{name}.a2s", "Main", str(frames), "64", "48000", "2", str(out)] + args,
                   check=True, cwd=A2S, timeout=600)
    got = fnv1a_fragments(read_pcm(out, 2, 64))
    want = np.load(os.path.join(GOLDEN, f"{name}.hash.npy"))
    assert len(differing_fragments(name, got, want)) == 0


@pytest.mark.gpu
def test_reference_player_runs_on_the_dropin():
    """The reference's own player, the way its benchmark script drives it
    (benchmark/benchmark.sh: a2play -dbuffer -r44100 <song> -pSong -st500), with the
    drop-in preloaded.  The buffer driver discards the audio, so this only checks that
    the player runs to the end (bit-exactness is what every other test here is about)."""
    a2play = os.path.join(ROOT, "oracle", "_ref", "a2play")
    if not (os.path.exists(a2play) and os.path.exists(UNITS_SO)):
        pytest.skip("oracle/_ref/a2play or liba2amd_units.so not built")
    for script, prog in (("scripted", "Main"), ("fm", "Main")):
        r = subprocess.run([a2play, "-dbuffer", "-r44100", f"{script}.a2s", f"-p{prog}", "-st3"], cwd=A2S,
                           env=dict(os.environ, LD_PRELOAD=UNITS_SO), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "Offline mode" in r.stdout + r.stderr, (r.stdout + r.stderr)[-500:]

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
# INTEGRATION.md option C: the replacement of the engine's voice walk, in front of the units
WALK_SO = os.path.join(ROOT, "audiality2_amd", "liba2amd_walk.so")
A2S = os.path.join(ROOT, "tests", "a2s")


def preload_libs(walk=False):
    return f"{WALK_SO} {UNITS_SO}" if walk else UNITS_SO

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
LIVE_CASES = [("churn", ["0.05"], 10 * 48000),
              # sleeping subtrees (held by the replaced voice walk) next to births, deaths, wake-ups and a
              # group that ends with everything under it asleep
              ("holds", ["0.02"], 3 * 48000)]
REALTIME_CASES = {"edge", "unload"}      # see tests/golden/make_goldens.py
UPLOAD_CASES = {"unload": "20000"}


def need_ref():
    if not (os.path.exists(REF_RENDER) and os.path.exists(UNITS_SO)):
        pytest.skip("oracle/_ref (compiled reference) or liba2amd_units.so not built")


@pytest.mark.gpu
@pytest.mark.parametrize("walk", [False, True, "records-kernels"])
@pytest.mark.parametrize("name,args,frames", CASES)
def test_engine_with_dropin_units_matches_reference(tmp_path, name, args, frames, walk):
    need_ref()
    out = tmp_path / f"{name}.pcm"
    env = dict(os.environ, LD_PRELOAD=preload_libs(walk is True))
    if walk == "records-kernels":       # (the suite forces the window kernels, conftest.py: here what small scenes get by default)
        env["A2AMD_WIN"] = "0"
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
def test_rendered_waves_stay_on_the_device(tmp_path):
    """edge.a2s renders three waves of its own at load time (a2_RenderWave from the compiler,
    src/compiler.c:3359: one-shot, looped, mip-mapped).  The substates render on the GPU; the device copies of
    those waves - samples, pads, mip levels - are built there from what was rendered (a2amd_wave_upload_captured)
    and none of them is copied up from the host; the audio is the reference's, and the same as with the
    waves uploaded from the engine's copies (A2AMD_NO_RESIDENT=1)."""
    need_ref()
    name, args, frames = [c for c in CASES if c[0] == "edge"][0]
    want = np.load(os.path.join(GOLDEN, f"{name}.hash.npy"))
    stats = {}
    for tag, extra in (("resident", {}), ("uploaded", {"A2AMD_NO_RESIDENT": "1"})):
        out = tmp_path / f"{tag}.pcm"
        env = dict(os.environ, LD_PRELOAD=UNITS_SO, A2AMD_WAVE_STATS="1", A2REF_REALTIME="1", **extra)
        r = subprocess.run([REF_RENDER, f"{A2S}/{name}.a2s", "Main", str(frames), "64", "48000", "2", str(out)] + args,
                           env=env, cwd=A2S, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-500:]
        got = fnv1a_fragments(read_pcm(out, 2, 64))
        assert len(differing_fragments(name, got, want)) == 0, tag
        lines = [ln for ln in r.stderr.splitlines() if "waves copied from the host" in ln]
        main = [ln for ln in lines if "state 0:" in ln]
        assert len(main) == 1 and len(lines) == 4, r.stderr[-800:]       # three substates, one master state
        w = main[0].split()
        stats[tag] = (int(w[w.index("waves") - 1]), int(w[w.index("built") - 1]))
    assert stats["resident"][1] == 3 and stats["uploaded"][1] == 0, stats
    assert stats["uploaded"][0] == stats["resident"][0] + 3, stats


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


@pytest.mark.gpu
@pytest.mark.parametrize("ahead", [0, 1, 5, 32])
@pytest.mark.parametrize("script,arg", [("scripted", "0.2"), ("song", "0.08")])
def test_dropin_walk_prefetch_hints_change_nothing(tmp_path, script, arg, ahead):
    """look_ahead() (a2amd_units.c): every unit of ours that is called remembers which
    one the engine called N calls later in the last walk and prefetches its blocks.
    Hints only - but the ring of recent callers points into engine-owned blocks, and
    notes are born and die all the time in these scripts: with the ring off, short,
    odd and at its longest the render must be the reference's."""
    need_ref()
    outs = []
    for pre in (None, UNITS_SO):
        out = tmp_path / f"o{int(pre is not None)}.pcm"
        env = dict(os.environ, A2AMD_WALK_AHEAD=str(ahead))
        if pre:
            env["LD_PRELOAD"] = pre
        subprocess.run([REF_RENDER, f"{A2S}/{script}.a2s", "Main", str(64 * 1500), "64", "48000", "2", str(out), arg],
                       check=True, env=env, cwd=A2S, timeout=600)
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
    # no abort(): the units fail to initialise, the engine cannot start its root voice
    # and a2_Open() fails - the application gets an error, never CPU-rendered audio
    assert r.returncode not in (0, -6) and "no CPU fallback" in r.stderr, (r.returncode, r.stderr[-400:])


@pytest.mark.gpu
def test_dropin_refuses_mixed_chains(tmp_path):
    """A CPU unit with audio ports inside a GPU-rendered chain would process
    silence: the drop-in says so through the engine's error channel (a2r_Error,
    never silently, never by abort()); a voice that *starts* in a CPU unit is
    refused at instantiation."""
    need_ref()
    env = dict(os.environ, LD_PRELOAD=UNITS_SO, A2REF_FOREIGN="1")
    for script in ("mixed", "mixedhead"):   # (both fine on the CPU)
        subprocess.run([REF_RENDER, f"{A2S}/{script}.a2s", "Main", "640", "64", "48000", "2", str(tmp_path / "c.pcm"), "0.1"],
                       env=dict(os.environ, A2REF_FOREIGN="1"), cwd=A2S, check=True, timeout=120)
        assert np.fromfile(tmp_path / "c.pcm", dtype="<i4").any()
    r = subprocess.run([REF_RENDER, f"{A2S}/mixed.a2s", "Main", "640", "64", "48000", "2", str(tmp_path / "m.pcm"), "0.1"],
                       env=env, cwd=A2S, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "mixed CPU/GPU chains are not supported" in r.stderr + r.stdout, r.stderr[-500:]
    r = subprocess.run([REF_RENDER, f"{A2S}/mixedhead.a2s", "Main", "640", "64", "48000", "2", str(tmp_path / "h.pcm"), "0.1"],
                       env=env, cwd=A2S, capture_output=True, text=True, timeout=120)
    assert "takes its input from a unit that is not replaced" in r.stderr, r.stderr[-500:]


@pytest.mark.gpu
@pytest.mark.parametrize("script,frames", [("sinkgroup", 9600), ("clients", 48000)])
@pytest.mark.parametrize("clients", [(), ("SINK",), ("SOURCE",), ("SOURCE", "SINK"), ("STREAMS",), ("STREAMS", "SINK"),
                                     # ... and the voice is killed 0.7 ms into a buffer: its clients are owed
                                     # the windows of a fragment that is rendered after the voice is gone
                                     ("SINK", "KILL"), ("SOURCE", "STREAMS", "SINK", "KILL"),
                                     # an insert client (a2_InsertCallback) on the group: it is handed the
                                     # group's audio and what it returns replaces it, between the two halves
                                     # of the render
                                     ("INSERT",), ("INSERT", "SINK"), ("INSERT", "SOURCE", "SINK"), ("INSERT", "KILL")])
@pytest.mark.parametrize("buffer,walk", [(64, False), (1024, False), (64, True)])
def test_dropin_serves_sink_and_source_clients(tmp_path, script, frames, clients, buffer, walk):
    """a2_SinkCallback / a2_SourceCallback / a2_InsertCallback on a voice
    in the middle of the graph (a group: inline; panmix; xinsert), and the buffered
    variants a2_OpenSink / a2_OpenSource (STREAMS).  The sink is handed what it is
    handed on the CPU (hash over everything it saw, in order), the source's audio is
    in the output, the insert client's output replaces the group's, the output is the
    reference's."""
    need_ref()
    res = []
    for preload in (False, True):
        out = tmp_path / f"c{int(preload)}.pcm"
        env = dict(os.environ, **{f"A2REF_{c}": "1" for c in clients})
        if "KILL" in clients:
            env["A2REF_KILL"] = str(frames // 2)
        if preload:
            env["LD_PRELOAD"] = preload_libs(walk)
        r = subprocess.run([REF_RENDER, f"{A2S}/{script}.a2s", "Main", str(frames), str(buffer), "48000", "2", str(out), "0.1"],
                           env=env, cwd=A2S, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        res.append(([ln for ln in r.stdout.splitlines() if ln.startswith("sink ")], read_pcm(out, 2, buffer).T.reshape(-1)))
    assert res[0][1].any()
    assert len(res[0][0]) == len([c for c in clients if c in ("SINK", "STREAMS")])
    if "KILL" in clients and buffer == 64:
        # (the callback sink sees the 33 frames of the fragment the voice dies in; the
        # harness stops reading the sink STREAM with the buffer before: oracle/ref_render.c)
        assert all(f" frames {frames // 2 + 33} " in ln for ln in res[0][0] if ln.startswith("sink peak")), res[0][0]
    assert not any(" frames 0 " in ln or "sink peak 0 " in ln for ln in res[0][0])
    assert res[0][0] == res[1][0]
    a, b = (r[1].reshape(-1, 2).T for r in res)            # [channel, frame]
    if "STREAMS" in clients:
        # A source stream fills only its own channel (1) of the buffers the engine
        # hands it (a2_sourcestream_process, xinsertapi.c:287-318) and xi_process
        # mixes ALL of them into the output (xinsert.c:113-118) - the others are
        # uninitialised stack arrays (xinsert.c:66).  What the reference adds to
        # channel 0 is whatever its stack held; the drop-in adds silence.
        a, b = a[1], b[1]
    assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [1, 2, 3])
@pytest.mark.parametrize("nest,outer", [(1, False), (2, False), (2, True), (3, True)])
@pytest.mark.parametrize("buffer", [64, 1024])
def test_dropin_serves_insert_clients_at_any_depth(tmp_path, nest, outer, buffer, devices):
    """a2_InsertCallback on group voices nested one to three levels below the root
    (a2_NewGroup under a2_NewGroup ...: oracle/ref_render A2REF_NEST), one insert
    client on the innermost voice and - outer - another on the outermost group: the
    render pauses behind each of those depths, deepest first, and the audio is the
    reference's.  devices > 1 (A2AMD_DEVICES): the voice tree spread over
    that many backend contexts - every context pauses for the clients of its own voices,
    then the root-bus partials are exchanged (a2amd_render_group, A2AMD_RENDER_EXCHANGE)."""
    need_ref()
    outs = []
    for preload in (False, True):
        out = tmp_path / f"n{int(preload)}.pcm"
        env = dict(os.environ, A2REF_NEST=str(nest), A2REF_INSERT="1")
        if outer:
            env["A2REF_INSERT_OUTER"] = "1"
        if preload:
            env["LD_PRELOAD"] = UNITS_SO
            if devices > 1:
                env["A2AMD_DEVICES"] = str(devices)
        r = subprocess.run([REF_RENDER, f"{A2S}/clients.a2s", "Main", "24000", str(buffer), "48000", "2", str(out), "0.1"],
                           env=env, cwd=A2S, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "insert client" not in r.stderr, r.stderr[-500:]
        outs.append(np.fromfile(out, dtype="<i4"))
    assert outs[0].any() and np.array_equal(outs[0], outs[1])


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
    for pre in (0, 1, 2):
        out = tmp_path / f"{name}{pre}.pcm"
        env = dict(os.environ)
        if pre:
            env["LD_PRELOAD"] = preload_libs(pre == 2)
        subprocess.run([REF_RENDER, f"{A2S}/{name}.a2s", "Main", str(frames), "64", "48000", "2", str(out)] + args,
                       check=True, env=env, cwd=A2S, timeout=900)
        outs.append(np.fromfile(out, dtype="<i4"))
    assert outs[0].any()
    for o in outs[1:]:
        bad = np.nonzero(outs[0] != o)[0]
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


# ---------------------------------------------------------------------------
# round 2: one GPU round trip per a2_Run() buffer (the drop-in's Process sits in front
# of the engine's a2_AudioCallback, include/a2amd_plugin.h A2P_audiodriver)
# ---------------------------------------------------------------------------
BATCH_CASES = [c for c in CASES if c[0] in ("sustain", "filter", "delaybus", "scripted", "edge", "fm", "fx", "envwire",
                                            "unload")]


@pytest.mark.gpu
@pytest.mark.parametrize("buffer", [256, 4096, 1000, 20000])
@pytest.mark.parametrize("name,args,frames", BATCH_CASES)
def test_dropin_batches_whole_buffers(tmp_path, name, args, frames, buffer):
    """a2_Run(buffer) with buffers of 4, 64, 15.6 and 312.5 fragments (the last one more
    than the 256 fragments one launch sequence takes: rendered in two), compared
    with the same binary and the same buffer size without the drop-in."""
    need_ref()
    outs = []
    for pre in (None, UNITS_SO):
        out = tmp_path / f"o{int(pre is not None)}.pcm"
        env = dict(os.environ)
        if pre:
            env["LD_PRELOAD"] = pre
        if name in REALTIME_CASES:
            env["A2REF_REALTIME"] = "1"
        if name in UPLOAD_CASES:
            env["A2REF_UPLOAD"] = UPLOAD_CASES[name]
        subprocess.run([REF_RENDER, f"{A2S}/{name}.a2s", "Main", str(frames), str(buffer), "48000", "2", str(out)] + args,
                       check=True, env=env, cwd=A2S, timeout=600)
        outs.append(read_pcm(out, 2, buffer))
    assert outs[0].any()
    bad = np.nonzero((outs[0] != outs[1]).any(axis=0))[0]
    if name in KNOWN_DEVIATIONS_BY_FRAME:
        # exactly the first 64-frame fragment of the buffer that follows the event - no more, and not nothing
        at = KNOWN_DEVIATIONS_BY_FRAME[name]
        lo = -(-at // buffer) * buffer
        assert len(bad) and bad.min() >= lo and bad.max() < lo + 64, f"documented deviation in [{lo}, {lo + 64}), found {bad[:5]} .. {bad[-5:]}"
        bad = bad[(bad < lo) | (bad >= lo + 64)]
    assert len(bad) == 0, f"{len(bad)} frames differ, first {bad[:5]}"


# unload: the one window in which the reference's oscillators notice that their wave is
# gone replays a neighbour's stale scratch; it is the first 64-frame
# fragment of the buffer that follows the release at frame 20000
KNOWN_DEVIATIONS_BY_FRAME = {"unload": 20000}


@pytest.mark.gpu
@pytest.mark.parametrize("buffer", [64, 256, 4096])
@pytest.mark.parametrize("clients", [("SINK",), ("SOURCE",), ("SOURCE", "SINK"), ("STREAMS",), ("INSERT",),
                                     ("INSERT", "SINK")])
def test_dropin_root_voice_clients(tmp_path, clients, buffer):
    """Clients on the ROOT voice's xinsert (where a2play attaches its sink): in a
    batched buffer the master bus comes back after the engine's walk, so sinks are
    served from it afterwards and sources are added on the host; an INSERT client
    (reads and writes the master bus window by window on the CPU) makes the
    drop-in render that buffer window by window, as in round 1."""
    need_ref()
    res = []
    for preload in (False, True):
        out = tmp_path / f"c{int(preload)}.pcm"
        env = dict(os.environ, A2REF_ROOTCLIENTS="1", **{f"A2REF_{c}": "1" for c in clients})
        if preload:
            env["LD_PRELOAD"] = UNITS_SO
        r = subprocess.run([REF_RENDER, f"{A2S}/scripted.a2s", "Main", "48000", str(buffer), "48000", "2", str(out), "0.2"],
                           env=env, cwd=A2S, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        res.append(([ln for ln in r.stdout.splitlines() if ln.startswith("sink ")], read_pcm(out, 2, buffer)))
    assert res[0][1].any()
    assert res[0][0] == res[1][0] and len(res[0][0]) == len([c for c in clients if c in ("SINK", "STREAMS")])
    a, b = res[0][1], res[1][1]
    if "STREAMS" in clients:
        a, b = a[1], b[1]       # (channel 0 gets the reference's uninitialised stack, see above)
    assert np.array_equal(a, b)


@pytest.mark.gpu
def test_wave_released_without_players_then_reloaded(tmp_path):
    """a wave released while nothing plays it must leave the drop-in's
    registry in that engine cycle (the allocator hands its A2_wave address to the next
    upload).  edge.a2s renders, plays and unloads waves of its own; here the harness
    uploads a wave, releases it before any voice played it, uploads another one - very
    likely at the same address - and a voice plays that."""
    need_ref()
    outs = []
    for preload in (False, True):
        out = tmp_path / f"w{int(preload)}.pcm"
        env = dict(os.environ, A2REF_REALTIME="1", A2REF_UPLOAD="999999999", A2REF_REUPLOAD="1")
        if preload:
            env["LD_PRELOAD"] = UNITS_SO
        subprocess.run([REF_RENDER, f"{A2S}/unload.a2s", "Main", "24064", "256", "48000", "2", str(out), "0.1"],
                       check=True, env=env, cwd=A2S, timeout=300)
        outs.append(read_pcm(out, 2, 256))
    assert outs[0].any()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [2, 3])
@pytest.mark.parametrize("name,args,frames", [c for c in CASES if c[0] in ("sustain", "delaybus", "scripted", "edge", "fm",
                                                                           "envwire")] + [("song", ["0.08"], 96000)])
def test_one_engine_state_over_several_contexts(tmp_path, name, args, frames, devices):
    """A2AMD_DEVICES=n: ONE engine state's voice subtrees dealt over n backend contexts
    (one per GPU; here they share the box's GPU), every context with its own copy of the
    root voice, new voices routed when they are first processed, the root-bus partials
    summed per buffer (a2amd_render_group), the root chain on the first context: the
    audio must be the single-context audio = the CPU reference's."""
    need_ref()
    outs = []
    for dev in (None, devices):
        out = tmp_path / f"d{dev or 0}.pcm"
        env = dict(os.environ)
        if dev:
            env["LD_PRELOAD"] = UNITS_SO
            env["A2AMD_DEVICES"] = str(dev)
        if name in REALTIME_CASES:
            env["A2REF_REALTIME"] = "1"
        subprocess.run([REF_RENDER, f"{A2S}/{name}.a2s", "Main", str(frames), "1024", "48000", "2", str(out)] + args,
                       check=True, env=env, cwd=A2S, timeout=600)
        outs.append(read_pcm(out, 2, 1024))
    assert outs[0].any()
    bad = np.nonzero((outs[0] != outs[1]).any(axis=0))[0]
    assert len(bad) == 0, f"{len(bad)} frames differ, first {bad[:5]}"


@pytest.mark.gpu
@pytest.mark.parametrize("clients", [("SINK",), ("SOURCE", "SINK"), ("SINKSTREAM", "SINK"), ("SINK", "KILL")])
@pytest.mark.parametrize("buffer", [64, 1024])
def test_sink_clients_on_a_voice_of_the_second_context(tmp_path, clients, buffer):
    """A2AMD_DEVICES=2 with a sink client (a2_SinkCallback / a2_OpenSink) on a group that
    is routed to the SECOND backend context (A2REF_SIBLING puts another subtree in front
    of it in the engine's walk): a2amd_render_group must bring that context's tapped
    windows back with the audio, or the client is handed zeros (round 2's advisor
    finding).  Sink hash, stream hash and audio against the CPU run."""
    need_ref()
    frames = 9600
    res = []
    for preload in (False, True):
        out = tmp_path / f"c{int(preload)}.pcm"
        env = dict(os.environ, A2REF_SIBLING="1", **{f"A2REF_{c}": "0" if c == "SINKSTREAM" else "1" for c in clients})
        if "KILL" in clients:
            env["A2REF_KILL"] = str(frames // 2)
        if preload:
            env["LD_PRELOAD"] = UNITS_SO
            env["A2AMD_DEVICES"] = "2"
        r = subprocess.run([REF_RENDER, f"{A2S}/sinkgroup.a2s", "Main", str(frames), str(buffer), "48000", "2", str(out), "0.1"],
                           env=env, cwd=A2S, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        res.append((np.fromfile(out, dtype="<i4"), [l for l in r.stdout.splitlines() if "sink" in l or "stream" in l]))
    assert res[0][0].any() and np.array_equal(res[0][0], res[1][0])
    assert res[0][1] and res[0][1] == res[1][1], (res[0][1], res[1][1])


@pytest.mark.gpu
def test_fuzz_scripts_over_two_contexts(tmp_path):
    """A handful of the fuzzer's random scripts (tests/fuzz_scripts.py) with the voice
    tree spread over two contexts."""
    need_ref()
    import fuzz_scripts
    for seed in range(100, 108):
        path = tmp_path / f"f{seed}.a2s"
        path.write_text(fuzz_scripts.make_script(seed))
        outs = []
        for dev in (None, 2):
            out = tmp_path / f"f{seed}_{dev or 0}.pcm"
            env = dict(os.environ)
            if dev:
                env["LD_PRELOAD"] = UNITS_SO
                env["A2AMD_DEVICES"] = "2"
            subprocess.run([REF_RENDER, str(path), "Main", "48000", "512", "48000", "2", str(out), "0.15"],
                           check=True, env=env, cwd=tmp_path, timeout=600)
            outs.append(np.fromfile(out, dtype="<i4"))
        assert np.array_equal(outs[0], outs[1]), f"seed {seed}"


# ---------------------------------------------------------------------------
# round 3: the whole stack at the sizes it is measured at.  bench.py's engine_in_loop
# object times a2_Run() of the reference engine with the drop-in at BASELINE sizes; the
# same runs, hash-compared with the engine's own CPU units, with the walk's prefetch
# hints off and on (A2AMD_WALK_AHEAD), offline buffers and one fragment per a2_Run().
# ---------------------------------------------------------------------------
import functools


@functools.lru_cache(maxsize=None)
def _cpu_engine_hash(program, voices, buffer, hf):
    import bench
    r = bench.engine_run(program, voices, max(hf, buffer // 64), buffer, False, hf)
    assert "error" not in r, r
    return r["hashes"][0], r["active_voices"]


ALL5 = (("ahead12", 64), ("ahead12", 4096), ("ahead0", 64), ("walk", 64), ("walk", 4096))
ENGINE_SIZES = [(p, v, var, b) for p, v, vs in (
    ("OscPan", 65536, ALL5[:1] + ALL5[2:]), ("OscFilterPan", 16384, ALL5),
    # (every CPU leg of the two big grouped scenes costs half a minute and more: one buffer size each)
    ("Osc2PanGroups", 65536, (("ahead12", 4096), ("walk", 4096))),
    ("OscPanScripted", 16384, ALL5), ("OscFilterPanScripted", 16384, ALL5)) for var, b in vs]
# BASELINE configs[4]'s voice tree in ONE engine state: 8 top-level groups x 128 sub-groups x 256 voices -
# behind the replaced voice walk, and with the 8 top-level groups dealt over 8 backend contexts as well
# (A2AMD_DEVICES=8: the 8-GPU layout; on this box they share the GPU and the root-bus sum is the device-local add)
ENGINE_SIZES += [("FilterTree", 262144, "walk", 4096), ("FilterTree", 262144, "walk8", 4096)]


@pytest.mark.gpu
@pytest.mark.parametrize("program,voices,variant,buffer", ENGINE_SIZES)
def test_engine_in_loop_at_measured_sizes_matches_cpu_units(program, voices, buffer, variant):
    """variant: the units alone with their prefetch hints on / off (INTEGRATION option A),
    or behind the replacement of the engine's voice walk (option C: sleeping voices are
    not visited)."""
    import bench
    need_ref()
    hf = 8 if voices >= 262144 else 16 if voices >= 65536 else 48
    want, active = _cpu_engine_hash(program, voices, buffer, hf)
    extra = {"A2AMD_WALK_AHEAD": "0" if variant == "ahead0" else "12", "A2AMD_WALK_STATS": "1"}
    if variant == "walk8":
        extra["A2AMD_DEVICES"] = "8"
    g = bench.engine_run(program, voices, max(2 * hf, 2 * buffer // 64), buffer, True, hf, walk=variant.startswith("walk"),
                         env_extra=extra)
    assert "error" not in g, g
    assert g["active_voices"] == active >= voices
    assert g["hashes"][0] == want, f"{program} x {voices}, a2_Run({buffer}), {variant}: audio differs"
    if variant.startswith("walk"):
        # the short cut was taken: most visits of sleeping voices were skipped
        skipped, made, unread = g["walk_stats"]
        assert skipped > made if "Scripted" not in program and voices < 262144 else skipped > 0, g["walk_stats"]
        # ... and most of those without touching the voice: the lists were taken from memory
        assert unread > skipped // 2 if "Scripted" not in program else unread > 0, g["walk_stats"]


def test_walk_hands_voices_to_the_engine_one_by_one_without_changing_anything(tmp_path):
    """liba2amd_walk.so's slow path - a voice cut out of its sibling list, handed alone to
    the engine's own a2_ProcessVoices, linked back in or dropped when it died - with the
    engine's CPU units (A2AMD_WALK_CUT: the test hook that takes that path in a state the
    drop-in does not serve): notes born and dying all the time, groups, delay buses."""
    need_ref()
    if not os.path.exists(WALK_SO):
        pytest.skip("audiality2_amd/liba2amd_walk.so not built")
    for script, arg, frames in (("scripted", "0.2", 48000), ("song", "0.08", 96000), ("churn", "0.05", 96000),
                                ("delaybus", "2", 24000)):
        outs = []
        for walk in (False, True):
            out = tmp_path / f"{script}{int(walk)}.pcm"
            env = dict(os.environ, A2AMD_WALK_CUT="1", A2AMD_WALK_STATS="1")
            env.pop("LD_PRELOAD", None)
            if walk:
                env["LD_PRELOAD"] = WALK_SO
            r = subprocess.run([REF_RENDER, f"{A2S}/{script}.a2s", "Main", str(frames), "64", "48000", "2", str(out), arg],
                               env=env, cwd=A2S, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-500:]
            if walk:
                assert "voice visits skipped" in r.stderr and " 0 voice visits skipped" in r.stderr, r.stderr[-300:]
            outs.append(np.fromfile(out, dtype="<i4"))
        assert outs[0].any() and np.array_equal(outs[0], outs[1]), script

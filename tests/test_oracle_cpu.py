"""The CPU oracle against fixtures generated from the compiled reference
(tests/golden/make_goldens.py).  Runs without a GPU."""
import ctypes
import os

import numpy as np
import pytest

from audiality2_amd import synth
from audiality2_amd.replay import Trace, replay
from conftest import GOLDEN, differing_fragments, fnv1a_fragments, make_oracle

CASES = ["sustain", "filter", "delaybus", "scripted", "edge", "fm", "fmtest3", "fmtest4", "fx", "dctest", "wstest", "envwire", "unload", "k2intro", "k2intro44", "k2epilogue", "k2loader", "k2trance", "pulsetronic",
         # the reference's own test/data scripts (make_goldens.py: td_*)
         "td_envtest", "td_envtest2", "td_envtest3", "td_envtest4", "td_pitchenvtest", "td_ramptest", "td_ramptest2",
         "td_ramptestenv", "td_evtest", "td_recursetest", "td_microtonal", "td_noisephase", "td_evilnoises"]


@pytest.mark.parametrize("name", CASES)
def test_trace_replay_matches_reference(oracle_lib, name):
    """Feeding the reference's own call trace to the restatement reproduces the
    audio the reference rendered, bit for bit, and the engine-global noise RNG
    stays in step after every oscillator call."""
    tr = Trace(os.path.join(GOLDEN, f"{name}.trace.xz"))
    be = make_oracle(oracle_lib, tr.config["samplerate"], tr.config["basepitch"], tr.config["channels"])
    out = replay(tr, be, batch=64, check_noise=True)
    be.close()
    want = np.load(os.path.join(GOLDEN, f"{name}.hash.npy"))
    head = np.load(os.path.join(GOLDEN, f"{name}.head.npy"))
    assert out.shape[1] == len(want) * 64
    assert np.array_equal(out[:, :head.shape[1]], head)
    got = fnv1a_fragments(out)
    bad = differing_fragments(name, got, want)
    assert len(bad) == 0, f"first differing fragment {bad[:5]}"


def test_replay_batching_is_transparent(oracle_lib):
    tr = Trace(os.path.join(GOLDEN, "scripted.trace.xz"))
    outs = []
    for batch in (1, 7, 64):
        be = make_oracle(oracle_lib, tr.config["samplerate"], tr.config["basepitch"])
        outs.append(replay(tr, be, batch=batch, max_fragments=200))
        be.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_p2i_matches_reference(oracle_lib):
    """a2_P2I over one full octave plus 20000 random pitches in +-128 octaves."""
    probe = np.load(os.path.join(GOLDEN, "p2i_probe.npy"))
    tab = (ctypes.c_uint32 * 128)()
    oracle_lib.a2o_build_pitch_table(tab)
    oracle_lib.a2o_p2i.restype = ctypes.c_uint
    oracle_lib.a2o_p2i.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
    for pitch, want in probe[::7]:
        assert oracle_lib.a2o_p2i(tab, int(pitch)) == (int(want) & 0xFFFFFFFF)


def test_builtin_waves_match_reference(oracle_lib):
    """a2_InitWaves + mip/pad preparation, all 24 waves, every level and pad.
    pulse1[20] is uninitialised stack memory in the reference (waves.c:640-647);
    the dump's value is fed back in."""
    z = np.load(os.path.join(GOLDEN, "builtin_waves.npz"))
    buf = np.zeros(2048, dtype=np.int16)
    dst = np.zeros(8192, dtype=np.int16)
    sizes = (ctypes.c_uint32 * 10)()
    offs = (ctypes.c_uint32 * 10)()
    oracle_lib.a2o_wave_pyramid.restype = ctypes.c_uint
    p16 = ctypes.POINTER(ctypes.c_int16)
    for k in range(24):
        hdr, data = z[f"w{k}_hdr"], z[f"w{k}_data"]
        assert hdr[0] == synth.WMIPWAVE and hdr[1] & synth.LOOPED and hdr[2] == 2048
        hole = int(data[1 + 20]) if k == 0 else 0
        assert oracle_lib.a2o_builtin_wave(k, buf.ctypes.data_as(p16), hole) == 0
        n = oracle_lib.a2o_wave_pyramid(buf.ctypes.data_as(p16), 2048, 1, 10, dst.ctypes.data_as(p16), sizes, offs)
        assert n == len(data)
        assert list(sizes) == [int(s) for s in hdr[3:13]]
        assert np.array_equal(dst[:n], data), f"wave {k}"


def test_host_wave_preparation_matches_oracle(oracle_lib):
    """synth.wave_pyramid (host-side data preparation used by bench/tests) vs
    the restated a2_render_mipmaps, looped and one-shot, odd lengths."""
    rng = np.random.default_rng(3)
    oracle_lib.a2o_wave_pyramid.restype = ctypes.c_uint
    p16 = ctypes.POINTER(ctypes.c_int16)
    for length, looped in [(2048, True), (1000, True), (777, False), (5, True), (1, False)]:
        src = rng.integers(-32768, 32767, length).astype(np.int16)
        sizes, data = synth.wave_pyramid(src, looped)
        dst = np.zeros(length * 2 + 10 * 140, dtype=np.int16)
        sz = (ctypes.c_uint32 * 10)()
        off = (ctypes.c_uint32 * 10)()
        n = oracle_lib.a2o_wave_pyramid(src.ctypes.data_as(p16), length, int(looped), 10,
                                        dst.ctypes.data_as(p16), sz, off)
        assert list(sz) == sizes
        assert np.array_equal(np.concatenate(data), dst[:n])


def test_basepitch_formula():
    assert synth.basepitch_for(48000) == -492789      # value the reference reports (trace CONFIG)


def test_scene_edge_cases(oracle_lib):
    """Empty tree, partial fragments, voices dying: the oracle stays sane."""
    be = make_oracle(oracle_lib)
    sc = synth.Scene(be)
    sc.root()
    out = sc.run(3, batch=2, frames=17)
    assert out.shape == (2, 51) and not out.any()
    sc.add_voices(3)
    out = sc.run(2, batch=8, frames=64)
    assert out.any()
    for units in sc.leaves:
        for u in units:
            be.unit_deinit(u)
    sc.leaves = []
    out = sc.run(2, batch=8, frames=33)
    assert not out.any()
    be.close()


def test_bench_parity_golden_matches_oracle(oracle_lib):
    """bench.py gates its numbers on tests/golden/bench_<chain>_<voices>v_<groups>g.hash.npy
    (data only, 8 steps x 256 fragments at full size, tests/golden/make_bench_golden.py);
    the files must be what the oracle renders for those workloads.  Re-rendered here:
    the first fragments of each (the whole of them takes minutes of oracle time)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mbg", os.path.join(GOLDEN, "make_bench_golden.py"))
    mbg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mbg)
    import bench
    for i, nfr in ((1, 64), (2, 8), (3, 2)):
        cfg = bench.CONFIGS[i]
        want = np.load(bench.golden_path(cfg["voices"], cfg["chain"], cfg["groups"]))
        assert len(want) % mbg.B == 0 and len(want) >= mbg.STEPS * mbg.B    # (round 4: as many steps as bench.py's default run issues)
        got = fnv1a_fragments(mbg.render(cfg["voices"], cfg["chain"], cfg["groups"], nfr))
        assert np.array_equal(got, want[:nfr]), f"configs[{i}]"
    # the private-wave scene (bench.py --config 5) at a sixteenth of its waves' length would be another scene:
    # its first fragment at full size
    cfg = bench.CONFIGS[5]
    want = np.load(bench.golden_path(cfg["voices"], cfg["chain"], cfg["groups"], private=cfg["private"]))
    assert len(want) == 2 * mbg.B
    # configs[4]: the whole job's audio with 1 and 2 top-level groups (32 768 / 65 536 voices)
    cfg = bench.CONFIGS[4]
    for total, nfr in ((32768, 8), (65536, 3)):
        want = np.load(bench.golden_path(total, cfg["chain"], 0, cfg["tree"]))
        assert len(want) == mbg.CFG4_FRAGMENTS
        got = fnv1a_fragments(mbg.render(total, cfg["chain"], 0, nfr, tree=cfg["tree"]))
        assert np.array_equal(got, want[:nfr]), f"configs[4] with {total} voices"


def test_oracle_xinsert_clients(oracle_lib):
    """xi_process (src/units/xinsert.c:60-142) in the oracle: what WRITE-only
    clients produced is added to the unit's output, READ-only clients are
    handed its input, with no clients it is the bypass."""
    from audiality2_amd import synth
    be = make_oracle(oracle_lib)
    sc = synth.Scene(be, nwaves=1)
    il, pm, xi = sc.root()
    sc.add_voices(2, chain="osc-pan", total=64)
    plain = sc.run(2, batch=2)
    assert plain.any()
    be.close()
    be = make_oracle(oracle_lib)
    sc = synth.Scene(be, nwaves=1)
    il, pm, xi = sc.root()
    sc.add_voices(2, chain="osc-pan", total=64)
    be.unit_clients(xi, 3)
    rng = np.random.default_rng(0)
    x = rng.integers(-1 << 20, 1 << 20, (2, 128)).astype(np.int32)

    def walk(frag):
        be.fragment(64)
        be.unit_process(il, 0, 64)
        for units in sc.leaves:
            for u in units:
                be.unit_process(u, 0, 64)
        be.inline_end(il)
        be.unit_process(pm, 0, 64)
        be.unit_inject(xi, 0, x[:, frag * 64:frag * 64 + 30])       # two windows' worth of client output
        be.unit_inject(xi, 30, x[:, frag * 64 + 30:frag * 64 + 64])
        be.unit_process(xi, 0, 30)
        be.unit_process(xi, 30, 34)
    walk(0)
    walk(1)
    out = be.render(128)
    assert np.array_equal(out, plain + x)
    assert np.array_equal(np.concatenate([be.unit_tapped(xi, 0), be.unit_tapped(xi, 1)], axis=1), plain)
    # client callbacks need every window: no "everybody sleeps" fragments while there are clients
    assert be.lib.a2o_fragment_repeat(be.ctx, 64, 2) == -4       # A2AMD_EUNSUPPORTED
    be.unit_clients(xi, 0)
    assert be.lib.a2o_fragment_repeat(be.ctx, 64, 2) == 0
    be.close()


def test_oracle_xsource_and_xsink_units(oracle_lib):
    """xsrc_process / xsink_Process in the oracle (xsource.c:43-137, xsink.c:27-46)
    on the script the GPU test compares (tests/test_gpu_parity.py): what a source
    client produced is what a 'xsource; panmix' voice plays, a sink on
    'wtosc; xsink; panmix' is handed the oscillator."""
    from test_gpu_parity import _xio_voices_script
    be = make_oracle(oracle_lib)
    audio, taps = _xio_voices_script(be)
    be.close()
    assert audio.any() and sorted(taps) == [1, 2, 3, 4, 5] + [100 + f for f in range(2, 9)]
    # the tapped oscillator: a triangle at constant pitch and amplitude, the same peak in every fragment
    peaks = {int(np.abs(t).max()) for f, t in taps.items() if f != 3 and f < 100}
    assert len(peaks) == 1 and peaks.pop() > 1 << 20

"""scripted voices whose VM runs on the device (include/a2amd_vm.h).

GPU tests: the reference engine with the drop-in and the replaced voice walk hands voices whose
programs loop inside the supported subset to the device VM; the audio a2_Run() returns must be
the audio the same engine renders with its own CPU units - with one-fragment buffers (the VM
kernel runs per fragment) and with long ones (256-fragment batches; voices adopted and taken
back in the middle of a batch are carried by the host's copy of the interpreter).

CPU tests (no GPU): the static analysis that decides what the device VM may run, and the host
copy of the interpreter - the same source the kernel is compiled from - on hand-assembled
programs with known answers."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import audiality2_amd
from conftest import ROOT

REF_RENDER = os.path.join(ROOT, "oracle", "_ref", "ref_render")
UNITS_SO = os.path.join(ROOT, "audiality2_amd", "liba2amd_units.so")
WALK_SO = os.path.join(ROOT, "audiality2_amd", "liba2amd_walk.so")
A2S = os.path.join(ROOT, "tests", "a2s")


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def need_ref():
    if not (os.path.exists(REF_RENDER) and os.path.exists(UNITS_SO) and os.path.exists(WALK_SO)):
        pytest.skip("oracle/_ref (compiled reference), liba2amd_units.so or liba2amd_walk.so not built")


def render(tmp_path, tag, script, program, frames, buffer, args, preload=None, env_extra=None, rate=48000, channels=2):
    out = tmp_path / f"{tag}.pcm"
    env = dict(os.environ, A2AMD_WALK_STATS="1")
    env.pop("LD_PRELOAD", None)
    if preload:
        env["LD_PRELOAD"] = preload
    env.update(env_extra or {})
    r = subprocess.run([REF_RENDER, f"{A2S}/{script}.a2s", program, str(frames), str(buffer), str(rate), str(channels),
                        str(out)] + list(args), env=env, cwd=A2S, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-800:]
    m = re.search(r"(\d+) voices handed to the device VM, (\d+) taken back", r.stderr)
    return np.fromfile(out, dtype="<i4"), (int(m.group(1)), int(m.group(2))) if m else None, r.stderr


def first_difference(a, b, channels, buffer):
    bad = np.nonzero(a != b)[0]
    if not len(bad):
        return None
    k = int(bad[0])
    return f"{len(bad)} samples differ, first in buffer {k // (channels * buffer)}, frame {k % buffer}"


@pytest.mark.gpu
@pytest.mark.parametrize("buffer", [64, 4096, 1000])
def test_looping_voices_run_on_the_device_vm_and_sound_like_the_cpu_engine(tmp_path, buffer):
    """tests/a2s/vmloops.a2s: every unit register the device VM writes, under LFOs, arpeggios, sweeps
    (cutoff ramps through the coefficient table), tick delays, explicit ramps; voices that get
    messages, are killed, detached, sit under a group that wakes in mid-fragment or under one that
    ends.  3 s of it, against the engine's own CPU render."""
    need_ref()
    frames = 48000 * 3 // buffer * buffer
    cpu, _, _ = render(tmp_path, "cpu", "vmloops", "Main", frames, buffer, ["0.08"])
    vm, stats, err = render(tmp_path, "vm", "vmloops", "Main", frames, buffer, ["0.08"], preload=f"{WALK_SO} {UNITS_SO}")
    assert cpu.any()
    assert first_difference(cpu, vm, 2, buffer) is None, first_difference(cpu, vm, 2, buffer)
    # the test is about the device VM: it must have been given the looping voices, and the engine
    # must have wanted some of them back (messages, kills, the group's windows)
    assert stats and stats[0] >= 12 and stats[1] >= 4, (stats, err[-400:])


@pytest.mark.gpu
@pytest.mark.parametrize("buffer", [64, 4096, 1000])
def test_env_segments_travel_with_the_voice_to_the_device_vm(tmp_path, buffer):
    """tests/a2s/envloops.a2s: looping voices with env units - every table mode, amplitude / pan /
    volume / pitch / cutoff targets, in front of and behind the target, 'time' overrides, two envelopes on one
    voice - are adopted with their segments in flight; the device steps the tables (a2amd_vmcore.h: env_target,
    env_lut) and the audio is the CPU engine's with ITS env unit (src/units/env.c), sample for sample."""
    need_ref()
    frames = 48000 * 3 // buffer * buffer
    cpu, _, _ = render(tmp_path, "cpu", "envloops", "Main", frames, buffer, ["0.08"])
    vm, stats, err = render(tmp_path, "vm", "envloops", "Main", frames, buffer, ["0.08"], preload=f"{WALK_SO} {UNITS_SO}")
    assert cpu.any()
    assert first_difference(cpu, vm, 2, buffer) is None, first_difference(cpu, vm, 2, buffer)
    assert stats and stats[0] >= 12 and stats[1] >= 4, (stats, err[-400:])


@pytest.mark.gpu
def test_env_voices_without_the_device_vm(tmp_path):
    """The same scene with the drop-in's env unit stepping its segments on the engine thread (A2AMD_NO_VM=1, and
    without the walk at all): the unit is the reference's restated."""
    need_ref()
    frames = 48000 * 2
    cpu, _, _ = render(tmp_path, "cpu", "envloops", "Main", frames, 256, ["0.08"])
    a, sa, _ = render(tmp_path, "novm", "envloops", "Main", frames, 256, ["0.08"], preload=f"{WALK_SO} {UNITS_SO}",
                      env_extra={"A2AMD_NO_VM": "1"})
    b, _, _ = render(tmp_path, "units", "envloops", "Main", frames, 256, ["0.08"], preload=UNITS_SO)
    assert first_difference(cpu, a, 2, 256) is None, first_difference(cpu, a, 2, 256)
    assert first_difference(cpu, b, 2, 256) is None, first_difference(cpu, b, 2, 256)
    assert sa and sa[0] == 0


@pytest.mark.gpu
def test_device_vm_off_is_the_same_audio(tmp_path):
    """A2AMD_NO_VM=1: nothing is handed over - the A/B switch of the measurements."""
    need_ref()
    frames = 48000
    a, sa, _ = render(tmp_path, "on", "vmloops", "Main", frames, 256, ["0.08"], preload=f"{WALK_SO} {UNITS_SO}")
    b, sb, _ = render(tmp_path, "off", "vmloops", "Main", frames, 256, ["0.08"], preload=f"{WALK_SO} {UNITS_SO}",
                      env_extra={"A2AMD_NO_VM": "1"})
    assert np.array_equal(a, b) and a.any()
    assert sa and sa[0] > 0 and sb and sb[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("program,n4", [("OscPanScripted", 256), ("OscFilterPanScripted", 256)])
@pytest.mark.parametrize("buffer", [64, 4096])
def test_scripted_bench_variants_on_the_device_vm(tmp_path, program, n4, buffer):
    """BASELINE variants 2b / 3b (tests/a2s/bench.a2s) at 1 024 voices: every voice's VM wakes every
    3 ms - all of them end up on the device, none comes back, and the audio is the CPU engine's."""
    need_ref()
    frames = 48000
    cpu, _, _ = render(tmp_path, "cpu", "bench", program, frames, buffer, [str(n4), "0.002"])
    vm, stats, _ = render(tmp_path, "vm", "bench", program, frames, buffer, [str(n4), "0.002"], preload=f"{WALK_SO} {UNITS_SO}")
    assert cpu.any()
    assert first_difference(cpu, vm, 2, buffer) is None, first_difference(cpu, vm, 2, buffer)
    assert stats == (4 * n4, 0), stats


@pytest.mark.gpu
@pytest.mark.parametrize("buffer", [64, 4096, 1000])
def test_notes_run_on_the_device_vm_up_to_the_run_that_ends_them(tmp_path, buffer):
    """Round 5, tests/a2s/vmnotes.a2s: voices whose programs run out, sleep, call local functions, divide by registers,
    draw from the engine's RNG, hang around attached - a song's notes.  The device VM takes each for the stretch of its
    future that needs none of that (the look-ahead of a2amd_vm_adopt) and the walk hands it back to the engine in the
    fragment that holds the first VM run that does (ENT.vm_exit, a2amd_walk.c): END, a2_VoiceFree, the CALL, the RAND
    draw, SLEEP and the wake-up message are the engine's own, at the engine's own frame.  2 s against the CPU render."""
    need_ref()
    frames = 48000 * 2 // buffer * buffer
    cpu, _, _ = render(tmp_path, "cpu", "vmnotes", "Main", frames, buffer, ["0.08"])
    vm, stats, err = render(tmp_path, "vm", "vmnotes", "Main", frames, buffer, ["0.08"], preload=f"{WALK_SO} {UNITS_SO}")
    assert cpu.any()
    assert first_difference(cpu, vm, 2, buffer) is None, first_difference(cpu, vm, 2, buffer)
    # 10 rounds of 7 notes + the sleeper's rounds: most of them must have been the device's for a while, and all but
    # the notes still sounding when the render stops came back (none of these programs loops for ever)
    assert stats and stats[0] >= 40 and stats[0] - 6 <= stats[1] <= stats[0], (stats, err[-400:])


@pytest.mark.gpu
@pytest.mark.parametrize("vmwin", ["1", "0"])
@pytest.mark.parametrize("buffer", [64, 4096, 1000])
def test_voices_that_wake_many_times_per_fragment(tmp_path, buffer, vmwin):
    """tests/a2s/vmfast.a2s: VMs that wake every 0.1 to 0.4 ms - up to a dozen windows of the chain per 64-frame
    fragment, bursts of sixty - on all three window classes.  With k_vm_win (A2AMD_VMWIN=1, the default with the window
    kernels) a fragment's further windows beyond the two staged in LDS go to a block of the pool the lane takes for
    that fragment itself (a2amd_vmwin.hip); through records (A2AMD_VMWIN=0) k_win_ctl places them by its up-front
    bound.  Both against the CPU engine's render, sample for sample."""
    need_ref()
    frames = 48000 // buffer * buffer
    cpu, _, _ = render(tmp_path, "cpu", "vmfast", "Main", frames, buffer, ["0.05"])
    vm, stats, err = render(tmp_path, "vm", "vmfast", "Main", frames, buffer, ["0.05"], preload=f"{WALK_SO} {UNITS_SO}",
                            env_extra={"A2AMD_VMWIN": vmwin, "A2AMD_HOSTTIMING": "1"})
    assert cpu.any()
    assert first_difference(cpu, vm, 2, buffer) is None, first_difference(cpu, vm, 2, buffer)
    assert stats and stats[0] >= 40, (stats, err[-400:])
    # (the test is about k_vm_win: most batches must have been its - every batch whose voices and fragments are the
    # ones the batch before predicted, a2amd_vm.cpp vm_predict)
    m = re.search(r"(\d+) batches with device VM voices, (\d+) of them through k_vm_win", err)
    assert m, err[-600:]
    if vmwin == "1":
        assert int(m.group(2)) >= int(m.group(1)) // 4 > 0, m.group(0)
    else:
        assert int(m.group(2)) == 0, m.group(0)


@pytest.mark.gpu
@pytest.mark.parametrize("spec", ["1", "0"])
@pytest.mark.parametrize("program", ["OscPanScripted", "OscFilterPanScripted"])
@pytest.mark.parametrize("buffer", [4096, 1024, 64, 1000])
def test_speculative_vm_pass_is_the_same_audio(tmp_path, program, buffer, spec):
    """Round 6 (vm_speculate, a2amd_vm.cpp): behind a batch of whole 64-frame fragments the class voices' VM + control
    pass for the batch expected NEXT runs on a stream of its own - window entries into a slot set of its own, the
    stepped state into shadow arrays - while the engine thread walks; a batch that is exactly the one predicted takes
    it (k_vm_commit + the render pass in k_vm_win's place), any other ignores it.  BASELINE variants 2b / 3b at 1 024
    voices, 3 s against the CPU engine's render, with the pass (most batches must have TAKEN one: the statistics line of
    A2AMD_HOSTTIMING says so) and without (A2AMD_VMSPEC=0); a 1 000-frame buffer (its batches mostly end in a 40-frame
    fragment: no pass behind those) for the audio alone."""
    need_ref()
    frames = 3 * 48000 // buffer * buffer      # (3 s: the first buffers are the adoption phase - lists change, nothing is predicted)
    cpu, _, _ = render(tmp_path, "cpu", "bench", program, frames, buffer, ["256", "0.002"])
    vm, stats, err = render(tmp_path, "vm", "bench", program, frames, buffer, ["256", "0.002"], preload=f"{WALK_SO} {UNITS_SO}",
                            env_extra={"A2AMD_VMSPEC": spec, "A2AMD_HOSTTIMING": "1", "A2AMD_VMSPEC_MIN": "1"})
    # (A2AMD_VMSPEC_MIN=1: by default batches of fewer than 8 fragments are not speculated on - a loss there, measured -
    # and the one-fragment case would go untested)
    assert cpu.any()
    assert first_difference(cpu, vm, 2, buffer) is None, first_difference(cpu, vm, 2, buffer)
    assert stats == (1024, 0), stats
    m = re.search(r"(\d+) speculative passes .*?, (\d+) of them taken", err)
    if spec == "0":
        assert not m, m.group(0)
    elif buffer % 64 == 0:
        assert m, err[-800:]
        launched, taken = int(m.group(1)), int(m.group(2))
        assert launched > 0 and taken >= launched // 4 > 0, m.group(0)
        # ... and a pass that was taken has counted the voices it left to the quiet kernels - none, in these scenes - so
        # the quiet kernel of the class is not launched to find that out (issue_kernels; by default not for the filter
        # class, whose 80 us of k_leaf_oscfiltpan keep an order of kernels that is worth more: DESIGN 2c)
        # (a 64-frame buffer is one fragment, in which most of these voices - they wake every 3 ms - are left alone:
        # their quiet kernel has work to do, and whether any batch had none is not this test's to say)
        q = re.search(r"(\d+) quiet-kernel launches not made", err)
        assert q, err[-800:]
        if buffer >= 1024:
            assert (int(q.group(1)) > 0) == (program == "OscPanScripted"), (program, q.group(0))
        elif program != "OscPanScripted":
            assert int(q.group(1)) == 0, q.group(0)
    # (a 1 000-frame buffer ends in a 40-frame fragment: such a batch is not followed by a pass - but a buffer the drop-in
    # delivers in two pieces may leave a batch of whole fragments, and a pass behind that is as right as any: no claim)


@pytest.mark.gpu
@pytest.mark.parametrize("buffer", [64, 4096])
def test_device_vm_leaves_chains_it_cannot_name_to_the_engine(tmp_path, buffer):
    """tests/a2s/vmlong.a2s: two looping voices with ten-unit chains that write their ninth and tenth unit, two with
    the plain wtosc-panmix chain.  The backend renders all four (chains of up to 16 units, round 6); the device VM's
    register map names a chain position in three bits (A2D_VM_MAXPOS), so it takes the short ones and refuses the long
    ones at adoption - it used to be handed nothing longer than 8 because the backend took nothing longer.  The audio
    is the CPU engine's either way."""
    need_ref()
    frames = 48000 * 2 // buffer * buffer
    cpu, _, _ = render(tmp_path, "cpu", "vmlong", "Main", frames, buffer, ["0.08"])
    vm, stats, err = render(tmp_path, "vm", "vmlong", "Main", frames, buffer, ["0.08"], preload=f"{WALK_SO} {UNITS_SO}")
    assert cpu.any()
    assert first_difference(cpu, vm, 2, buffer) is None, first_difference(cpu, vm, 2, buffer)
    assert stats and stats[0] == 2, (stats, err[-600:])


@pytest.mark.gpu
@pytest.mark.parametrize("script,args", [("vmloops", ["0.08"]), ("vmnotes", ["0.08"]), ("envloops", ["0.08"])])
def test_device_vm_voices_through_records_are_the_same_audio(tmp_path, script, args):
    """A2AMD_VMWIN=0: the window-class voices of the device VM through k_vm_count / k_vm_emit and k_win_ctl - the path
    the suite's other tests leave to voices of other chains since k_vm_win - against the CPU engine."""
    need_ref()
    buffer = 1000
    frames = 48000 * 2 // buffer * buffer
    cpu, _, _ = render(tmp_path, "cpu", script, "Main", frames, buffer, args)
    vm, stats, err = render(tmp_path, "vm", script, "Main", frames, buffer, args, preload=f"{WALK_SO} {UNITS_SO}",
                            env_extra={"A2AMD_VMWIN": "0"})
    assert first_difference(cpu, vm, 2, buffer) is None, first_difference(cpu, vm, 2, buffer)
    assert stats and stats[0] >= 12, (stats, err[-400:])


@pytest.mark.gpu
@pytest.mark.parametrize("blocks", [True, False])
@pytest.mark.parametrize("buffer", [64, 1000])
def test_births_and_deaths_beside_a_thousand_sleeping_voices(tmp_path, buffer, blocks):
    """tests/a2s/bench.a2s OscFilterPanChurn (the bench line's "churn" cell at 1 024 voices): a pad's voices and ONE
    sequencer voice in the same group - 200 births and deaths per second in the sequencer's own list.  The pad's list
    never sleeps as a whole (the sequencer stands in it); the walk takes it in blocks of 256 voices that slept unseen
    (a2amd_walk.c BLK; A2AMD_WALK_NOBLOCKS=1: voice by voice).  1.5 s against the CPU engine, sample for sample."""
    need_ref()
    frames = 72000 // buffer * buffer
    cpu, _, _ = render(tmp_path, "cpu", "bench", "OscFilterPanChurn", frames, buffer, ["256", "0.002"])
    vm, stats, err = render(tmp_path, "vm", "bench", "OscFilterPanChurn", frames, buffer, ["256", "0.002"],
                            preload=f"{WALK_SO} {UNITS_SO}", env_extra={} if blocks else {"A2AMD_WALK_NOBLOCKS": "1"})
    assert cpu.any()
    assert first_difference(cpu, vm, 2, buffer) is None, first_difference(cpu, vm, 2, buffer)
    m = re.search(r"(\d+) voice visits skipped, (\d+) made", err)
    assert m and int(m.group(1)) > 50 * int(m.group(2)), err[-400:]


@pytest.mark.gpu
def test_song_notes_are_taken_by_the_device_vm(tmp_path):
    """tests/a2s/song.a2s (the plumbing-sized workload): its melodic notes and the kick - programs that run out - are
    the device VM's between their first delay and their last VM run; hats and snares stay the engine's (their noise
    oscillators draw from the engine's one RNG in walk order, a2amd_vm_adopt).  Audio = the CPU engine's."""
    need_ref()
    frames = 48000 * 4
    cpu, _, _ = render(tmp_path, "cpu", "song", "Song", frames, 4096, [])
    vm, stats, err = render(tmp_path, "vm", "song", "Song", frames, 4096, [], preload=f"{WALK_SO} {UNITS_SO}")
    assert cpu.any()
    assert first_difference(cpu, vm, 2, 4096) is None, first_difference(cpu, vm, 2, 4096)
    # 4 s = 8 beats: 16 + 32 arpeggio notes ... per 16 ticks 32 Arp, 6 Bass, 3 Pad, 4 Kick of 69 notes
    assert stats and stats[0] >= 60, (stats, err[-400:])


# ---- CPU: analysis and host interpreter -------------------------------------------------------
def lib():
    L = audiality2_amd.load_library()
    L.a2amd_vm_analyze.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint, ctypes.c_uint, ctypes.c_int32, ctypes.c_uint32,
                                   ctypes.POINTER(VmInfo)]
    L.a2amd_vm_analyze.restype = ctypes.c_int
    return L


class VmInfo(ctypes.Structure):
    _fields_ = [("reason", ctypes.c_int32), ("opcode", ctypes.c_int32), ("at", ctypes.c_int32), ("written", ctypes.c_uint64),
                ("controlled", ctypes.c_uint64), ("reachable", ctypes.c_uint32), ("longest", ctypes.c_uint32)]


class VmState(ctypes.Structure):
    _fields_ = [("waketime", ctypes.c_uint32), ("state", ctypes.c_uint8), ("func", ctypes.c_uint8), ("pc", ctypes.c_uint16),
                ("r", ctypes.c_int32 * 64)]


# opcode numbers (include/a2amd_vm.h; checked against the engine's by a2amd_walk.c and test_plugin_abi.py)
OPS = ("END RETURN CALL JUMP LOOP JZ JNZ JG JL JGE JLE DELAY DELAYR TDELAY TDELAYR SLEEP WAKE FORCE SUBR DIVR P2DR NEGR "
       "LOAD LOADR ADD ADDR MUL MULR MOD MODR QUANT QUANTR RAND RANDR GR LR GER LER EQR NER ANDR ORR XORR NOTR SET SETALL "
       "RAMP RAMPR RAMPALL RAMPALLR PUSH PUSHR SPAWN").split()
OP = {n: i for i, n in enumerate(OPS)}
TWO = {"DELAY", "TDELAY", "LOAD", "ADD", "MUL", "MOD", "QUANT", "RAND", "PUSH", "RAMP", "RAMPALL"}


def asm(*ins):
    """(name, a1, a2[, a3]) -> 32 bit words (A2_instruction, src/internals.h:218-224)."""
    words = []
    for i in ins:
        name, a1, a2 = i[0], i[1], i[2]
        words.append(OP[name] | (a1 << 8) | (a2 << 16))
        if name in TWO:
            words.append(i[3] & 0xffffffff)
    return (ctypes.c_uint32 * len(words))(*words), len(words)


MSDUR = int(np.float32(48000) * np.float32(65.536) + np.float32(0.5))      # src/audiality2.c:499


def analyze(code, n, pc=0, tick=1 << 16):
    info = VmInfo()
    r = lib().a2amd_vm_analyze(code, n, pc, tick, MSDUR, ctypes.byref(info))
    return r, info


def fx(v):
    return int(round(v * 65536))


def test_analysis_takes_a_delay_loop_and_names_what_it_controls():
    # for { +r5 .01; d 3; -r5 .01 (ADD of a negative); d 3 }
    code, n = asm(("ADD", 5, 0, fx(.01)), ("DELAY", 0, 0, fx(3)), ("ADD", 5, 0, fx(-.01)), ("DELAY", 0, 0, fx(3)), ("JUMP", 0, 0))
    r, info = analyze(code, n)
    assert r == 0 and info.reachable == 5
    assert info.written == 1 << 5 and info.controlled == 1 << 5
    assert info.longest == 3        # JUMP, ADD, DELAY: the longest stretch between two certain yields


def test_analysis_refuses_what_the_device_cannot_or_must_not_run():
    # an instruction outside the subset: END, SPAWN, RAND (the engine-global RNG), DIVR
    for bad in (("END", 0, 0), ("SPAWN", 1, 2), ("RAND", 3, 0, 5), ("DIVR", 3, 4), ("SLEEP", 0, 0), ("CALL", 0, 1)):
        code, n = asm(("DELAY", 0, 0, fx(1)), bad, ("JUMP", 0, 0))
        r, info = analyze(code, n)
        assert r == 2 and info.opcode == OP[bad[0]] and info.at == 2, (bad, r, info.opcode, info.at)
    # ... unless it cannot be reached from where the voice stands
    code, n = asm(("SPAWN", 1, 2), ("DELAY", 0, 0, fx(1)), ("JUMP", 0, 1))
    assert analyze(code, n, pc=1)[0] == 0 and analyze(code, n, pc=0)[0] == 2
    # a loop that may never yield: no delay at all, a delay of zero, a delay from a register
    for body in ((("ADD", 3, 0, 1),), (("DELAY", 0, 0, 0),), (("DELAYR", 3, 0),), (("TDELAYR", 3, 0),)):
        code, n = asm(*body, ("JUMP", 0, 0))
        assert analyze(code, n)[0] == 4, body
    # a tick delay is a certain yield while the program leaves R_TICK alone and the tick is not zero
    code, n = asm(("TDELAY", 0, 0, fx(1)), ("JUMP", 0, 0))
    assert analyze(code, n)[0] == 0 and analyze(code, n, tick=0)[0] == 4
    code, n = asm(("TDELAY", 0, 0, fx(1)), ("ADD", 0, 0, 5), ("JUMP", 0, 0))
    assert analyze(code, n)[0] == 4
    # a counted loop without a delay inside: LOOP's back edge is a cycle
    code, n = asm(("LOAD", 4, 0, fx(3)), ("ADD", 5, 0, 1), ("LOOP", 4, 2 + 0), ("DELAY", 0, 0, fx(1)), ("JUMP", 0, 0))
    assert analyze(code, n)[0] == 4
    # 1000 instructions in a row (A2_INSLIMIT) between two delays
    code, n = asm(("DELAY", 0, 0, fx(1)), *[("ADDR", 5, 6)] * 1000, ("JUMP", 0, 0))
    assert analyze(code, n)[0] == 4
    code, n = asm(("DELAY", 0, 0, fx(1)), *[("ADDR", 5, 6)] * 900, ("JUMP", 0, 0))
    assert analyze(code, n)[0] == 0
    # divisors
    for d in (0, -1):
        code, n = asm(("MOD", 3, 0, d), ("DELAY", 0, 0, fx(1)), ("JUMP", 0, 0))
        assert analyze(code, n)[0] == 3
    # a jump out of the function, a truncated two-word instruction
    code, n = asm(("DELAY", 0, 0, fx(1)), ("JUMP", 0, 77))
    assert analyze(code, n)[0] == 1
    code, n = asm(("DELAY", 0, 0, fx(1)), ("JUMP", 0, 0), ("LOAD", 1, 0, 0))
    assert analyze(code, n - 1, pc=3)[0] == 1


def trace_host(code, n, st, wr_unit, wr_reg, kinds, frags, now=0, cap=4096):
    L = audiality2_amd.load_library()
    L.a2amd_vm_trace_host.restype = ctypes.c_int
    wu = (ctypes.c_int32 * 64)(*([-1] * 64))
    wr = (ctypes.c_uint8 * 64)()
    for reg, (u, ur) in zip(wr_unit, wr_reg):
        wu[reg], wr[reg] = u, ur
    kk = (ctypes.c_int32 * len(kinds))(*kinds)
    ff = (ctypes.c_uint8 * len(frags))(*frags)
    recs = (ctypes.c_uint32 * (4 * cap))()
    k = L.a2amd_vm_trace_host(code, ctypes.c_uint(n), ctypes.byref(st), wu, wr, kk, len(kinds), ctypes.c_uint32(now),
                              ctypes.c_uint32(MSDUR), 48000, 0, ff, len(frags), recs, cap)
    assert 0 <= k <= cap, k
    out = []
    for i in range(k):
        head, value, dur, start = recs[4 * i], ctypes.c_int32(recs[4 * i + 1]).value, recs[4 * i + 2], recs[4 * i + 3]
        out.append((head & 0xffff, (head >> 16) & 0xff, (head >> 24) & 15, head >> 28, value, dur, start))
    return out


R_SEG, R_WRITE, R_F1SET, R_F1RAMP = 1, 2, 5, 6
WTOSC, PANMIX, FILTER12 = 0, 1, 2


def test_host_interpreter_known_answers():
    """for { +p .01; d 3; -p .01; d 3 } on a wtosc -> panmix voice, engine time 0: the VM runs in
    the frame its wake time lies in (core.c:1816-1823), the pending register goes out as a ramp over
    the delay (core.c:1719-1726) with the wake time's fraction as start, windows are cut there."""
    code, n = asm(("ADD", 5, 0, fx(.01)), ("DELAY", 0, 0, fx(3)), ("ADD", 5, 0, fx(-.01)), ("DELAY", 0, 0, fx(3)), ("JUMP", 0, 0))
    st = VmState()
    st.state = 1                            # A2_WAITING
    st.waketime = (100 << 8) | 0x40         # frame 100.25 = fragment 1, frame 36
    st.r[5] = fx(.5)
    st.r[1] = fx(1)                         # transpose
    recs = trace_host(code, n, st, [5], [(0, 1)], [WTOSC, PANMIX], [64] * 5)
    dt = ((fx(3) * MSDUR + 0x7fffff) >> 24)
    assert dt == 144 << 8                   # 3 ms at 48 kHz, 24:8
    # fragment 0: nothing (the default window).  fragment 1: window [0, 36), the write, window [36, 64)
    assert recs[0] == (1, R_SEG, 0, 0, 0, 0 | (36 << 16), 0)
    assert recs[1] == (1, R_WRITE, 0, 1, fx(.5) + fx(.01) + fx(1), dt, 0x40)      # value + transpose (+ basepitch 0)
    assert recs[2] == (1, R_SEG, 0, 0, 0, 36 | (28 << 16), 0)
    # next wake 144 frames on: frame 244.25 = fragment 3, frame 52
    assert recs[3] == (3, R_SEG, 0, 0, 0, 0 | (52 << 16), 0)
    assert recs[4][:5] == (3, R_WRITE, 0, 1, fx(.5) + fx(1)) and recs[4][6] == 0x40
    assert recs[5] == (3, R_SEG, 0, 0, 0, 52 | (12 << 16), 0)
    assert len(recs) == 6
    assert st.waketime == (388 << 8) | 0x40 and st.state == 1 and st.pc == 8      # (pc counts 32 bit words)
    # a VM run at the very start of a fragment: the write comes first, the whole fragment is one explicit window
    st2 = VmState()
    st2.state = 1
    st2.waketime = 128 << 8
    recs = trace_host(code, n, st2, [5], [(0, 1)], [WTOSC, PANMIX], [64] * 3)
    assert [r[1] for r in recs] == [R_WRITE, R_SEG] and recs[1] == (2, R_SEG, 0, 0, 0, 64 << 16, 0)


KIND_OF = {n: k for k, n in enumerate("wtosc panmix filter12 fbdelay inline xinsert fm1 fm2 fm3 fm4 fm3p fm4p fm2r fm4r dc waveshaper "
                                       "dcblock limiter xsink xsource".split())}        # a2amd_unitkind, include/a2amd.h


def load_vm_traces():
    import json
    import lzma
    with lzma.open(os.path.join(GOLDEN, "vm_traces.json.xz"), "rt") as f:
        return json.load(f)


class VmEnv(ctypes.Structure):       # a2amd_vm_env, include/a2amd_vm.h
    _fields_ = [("ramper", ctypes.c_int32 * 4), ("lut", ctypes.c_int32), ("scale", ctypes.c_int32), ("offset", ctypes.c_int32),
                ("out", ctypes.c_int32), ("active", ctypes.c_int32), ("regbase", ctypes.c_int32), ("before", ctypes.c_int32),
                ("out_unit", ctypes.c_int32), ("out_reg", ctypes.c_int32)]


def env_tables():
    """env_InitLUTs, src/units/env.c:218-257, with the reference's expressions and types: float constants widened to
    double next to a double, libm's cos / pow (math.* calls the same libm the reference was compiled against)."""
    import math
    f = lambda x: float(np.float32(x))
    t = [[0] * 66 for _ in range(8)]
    for i in range(64):
        t[0][i] = int((1.0 - math.cos(i * math.pi / 63)) * 16384.0 + 0.5)
    for j, deg in enumerate((1, 2, 3, 4, 6, 9, 13)):
        c = math.pow(f(0.1), float(deg))
        rc = f(0.002) + f(0.1) * math.pow(f(0.8), float(deg))
        for i in range(64):
            x = 1.0 - i / 64.0
            r = (1.0 - x) * rc
            t[1 + j][i] = int((math.pow(c, x) * (1.0 - r) + r - c * x) * 32768.0 + 0.5)
    for row in t:
        row[64] = row[65] = 32768
    return (ctypes.c_uint16 * (8 * 66))(*[v for row in t for v in row])


def replay_trace(d, with_exit=False, layout=None):
    """One fixture of tests/golden/vm_traces.json.xz through the host copy of the interpreter: (records it made, the
    reference's events in the same form, the VM state at the end, (has_exit, exit_when, writes of the stay), frames per
    backend fragment)."""
    L = audiality2_amd.load_library()
    L.a2amd_vm_trace_host_env.restype = ctypes.c_int
    L.a2amd_vm_trace_host_exit.restype = ctypes.c_int
    n = len(d["code"])
    code = (ctypes.c_uint32 * n)(*d["code"])
    st = VmState()
    st.waketime, st.state, st.func, st.pc = d["state"]["waketime"], d["state"]["state"], d["state"]["func"], d["state"]["pc"]
    for k, v in enumerate(d["state"]["r"]):
        st.r[k] = v
    # chain positions as the backend counts them: env units are not backend units
    env_units = [e["unit"] for e in d["envs"]]
    bpos = {}
    for k, u in enumerate(d["units"]):
        if k not in env_units:
            bpos[k] = len(bpos)
    names = [u for k, u in enumerate(d["units"]) if k not in env_units]
    kinds = [KIND_OF[u] for u in names]
    wu, wr = [-1] * 64, [0] * 64
    for r, (pos, reg) in enumerate(d["cregs"]):
        if pos in bpos:
            wu[r], wr[r] = bpos[pos], reg
    envs = (VmEnv * 2)()
    assert len(d["envs"]) <= 2
    for k, e in enumerate(d["envs"]):
        wu[e["regbase"]] = -3 - k                       # its 'target' register
        envs[k].regbase = e["regbase"]
        envs[k].before = sum(1 for q in bpos if q < e["unit"])
        envs[k].out_unit, envs[k].out_reg = (-1, 0) if e["out_vmreg"] < 0 else (bpos[d["cregs"][e["out_vmreg"]][0]], d["cregs"][e["out_vmreg"]][1])
    # the backend's fragments = the root voice's windows; every event belongs to the fragment whose "r" follows it
    frames, base, events, pending = [], [], [], []
    for e in d["events"]:
        if e[0] == "r":
            events += [(len(frames), e[2], x) for x in pending]
            pending = []
            frames.append(e[3])
            base.append(e[2])
        else:
            pending.append(e)
    assert not pending and sum(frames) == 64 * d["fragments"]
    if layout is not None:      # (other fragments over the same stretch of time: only the records are of interest then)
        assert sum(layout) == sum(frames)
        frames, base = list(layout), [0] * len(layout)
    nfr = len(frames)
    ff = (ctypes.c_uint8 * nfr)(*frames)
    fb = (ctypes.c_uint8 * nfr)(*base)
    cap = 1 << 16
    recs = (ctypes.c_uint32 * (4 * cap))()
    args = [code, ctypes.c_uint(n), ctypes.byref(st), (ctypes.c_int32 * 64)(*wu), (ctypes.c_uint8 * 64)(*wr),
            (ctypes.c_int32 * len(kinds))(*kinds), len(kinds), ctypes.c_uint32(d["now"]),
            ctypes.c_uint32(d["msdur"]), d["samplerate"], d["basepitch"], ff, fb, nfr, envs, len(d["envs"]),
            env_tables(), recs, cap]
    has_exit, exit_when, stay = ctypes.c_int32(0), ctypes.c_uint32(0), ctypes.c_int32(0)
    if with_exit:
        k = L.a2amd_vm_trace_host_exit(*args, ctypes.byref(has_exit), ctypes.byref(exit_when), ctypes.byref(stay))
    else:
        k = L.a2amd_vm_trace_host_env(*args)
    assert 0 < k < cap, k
    if layout is not None:
        segs = [0] * nfr
        for i in range(k):
            if (recs[4 * i] >> 16) & 0xff == R_SEG:
                segs[recs[4 * i] & 0xffff] += 1
        return segs
    got = []
    for i in range(k):
        head, value, dur, start = recs[4 * i], ctypes.c_int32(recs[4 * i + 1]).value, recs[4 * i + 2], recs[4 * i + 3]
        frag, op, unit, reg = head & 0xffff, (head >> 16) & 0xff, (head >> 24) & 15, head >> 28
        if op == R_SEG:
            if dur != (frames[frag] << 16):     # (a whole-fragment window is the default window, explicit or not)
                got.append((frag, "p", dur & 0xffff, dur >> 16))
        elif op == R_WRITE:
            if names[unit] != "dcblock":        # (its cutoff arrives as a coefficient: the table's business, as below)
                got.append((frag, "w", unit, reg, value, dur, start))
        else:
            assert op in (R_F1SET, R_F1RAMP), op    # (a filter coefficient: compared in the test below and on the GPU)

    def i32(x):
        return ((x + (1 << 31)) & 0xffffffff) - (1 << 31)

    want = []
    for frag, fbase, e in events:
        if e[0] == "p":
            if (e[2] - fbase, e[3]) != (0, frames[frag]):
                want.append((frag, "p", e[2] - fbase, e[3]))
            continue
        _, _, r, value, start, dur, transpose = e
        pos, reg = d["cregs"][r]
        if pos in env_units:
            continue                        # the VM's write to an env's 'target': it starts a segment, no record
        kind = d["units"][pos]
        if (kind == "filter12" and reg == 0) or kind == "dcblock":
            continue                        # cutoff: host ramper + coefficient records / table value
        if (kind == "wtosc" or kind.startswith("fm")) and reg == 1:
            value = i32(value + transpose + d["basepitch"])
        elif kind == "filter12" and reg == 1:
            value = 32768 if value < 512 else (65536 << 8) // value
        elif kind == "limiter":
            v8 = i32((value << 8) & 0xffffffff)
            value = (abs(v8) // d["samplerate"]) * (1 if v8 >= 0 else -1) if reg == 0 else max(256, (value << 8) & 0xffffffff)
        want.append((frag, "w", bpos[pos], reg, value, dur, start & 255))
    return got, want, st, (has_exit.value, exit_when.value, stay.value), frames


@pytest.mark.parametrize("case", [0, 3, 7, 11, 14, 19, 24, 27])
def test_windows_inside_fragments_are_bounded_by_their_count_over_uncut_fragments(case):
    """What k_vm_pool's prediction rests on (a2amd_vm.hip): a window of a voice begins inside a fragment only
    where its VM wakes up, so the number of such windows over a stretch of time cut into 64-frame fragments bounds the
    number over the same stretch in any fragments that are those or pieces of them (the engine cuts a fragment where a
    group's VM wakes up).  The host copy of the interpreter over the trace fixtures' programs, 300 fragments, uncut
    against random cuts: a fragment's further windows = its R_SEG records - 1 (a fragment with one window has none)."""
    d = load_vm_traces()[case]
    nfr = min(d["fragments"], 300)
    d = dict(d, fragments=nfr, events=[e for e in d["events"]])
    # (the fixture's own fragments up to nfr engine fragments: replay_trace checks their sum)
    total, keep = 0, []
    for e in d["events"]:
        keep.append(e)
        if e[0] == "r":
            total += e[3]
            if total == 64 * nfr:
                break
    d["events"] = keep
    uncut = replay_trace(d, with_exit=case >= 23, layout=[64] * nfr)
    bound = sum(max(0, n - 1) for n in uncut)
    assert bound > 0
    rng = np.random.default_rng(case)
    for _ in range(6):
        lay = []
        for _f in range(nfr):
            if rng.random() < 0.4:
                c = int(rng.integers(1, 64))
                lay += [c, 64 - c] if rng.random() < 0.7 else [c // 2 or 1, c - (c // 2 or 1) or 1, 64 - c][:3]
            else:
                lay.append(64)
        lay = [x for x in lay if x > 0]
        # (repair the rare piece arithmetic that does not add up to whole fragments)
        if sum(lay) != 64 * nfr:
            continue
        segs = replay_trace(d, with_exit=case >= 23, layout=lay)
        assert sum(max(0, n - 1) for n in segs) <= bound, (sum(max(0, n - 1) for n in segs), bound)


@pytest.mark.parametrize("case", range(23))
def test_host_interpreter_against_the_reference_vm(case):
    """tests/golden/vm_traces.json.xz (made by tests/golden/make_vm_traces.py from oracle/ref_vmtrace.c: the COMPILED
    REFERENCE's own VM running each looping voice of vmloops.a2s - and, cases 12 on, of envtrace.a2s: voices with env
    units in every table mode, in front of and behind what they drive, with 'time' overrides, two on one voice, and
    (case 22, 'Short') a table look-up that runs past its table's end into the engine's next table - for 600
    fragments, every unit register write, every window and the root voice's windows logged).  The host copy of the
    device VM's interpreter - the header the kernel is compiled from - takes over at the same moment (program text,
    A2_vmstate, register wiring, env placement, engine clock, the engine's fragments as the root voice cut them) and
    must produce the same windows and the same writes: register, value as the unit's callback transforms it
    (a2amd_unit_write: wtosc / fm pitch + transpose + base pitch, filter12's 1/q, limiter's release and threshold; an env
    segment's per-window value through its control wire), duration, start - in the same order, and end in the same VM
    state.  No GPU involved."""
    d = load_vm_traces()[case]
    got, want, st, _, _ = replay_trace(d)
    assert len(want) > 20
    first = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None)
    assert first is None and len(got) == len(want), (first, got[first:first + 3] if first is not None else len(got),
                                                     want[first:first + 3] if first is not None else len(want))
    assert (st.waketime, st.state, st.pc) == (d["end_state"]["waketime"], d["end_state"]["state"], d["end_state"]["pc"])


# what ends each stay of tests/a2s/vmnotes.a2s' voices, and how many frames of the trace it lasts at least / at most
# (ms of the program text in front of the VM run that is the engine's; the trace starts inside the first delay)
STAYS = {"Note": ("END", 25 + 40.5, 25 + 40.5 + 33), "Pluck": ("END", 9 * 11.3, 9 * 11.3 + 27), "Tail": ("END", 5 * 14.7, 6 * 14.7),
         "Div": ("END", 12 * 9.9, 12 * 9.9 + 12), "Call": ("CALL", 20, 40), "Rnd": ("RAND", 4 * 13, 5 * 13),
         "Sleeper": ("SLEEP", 8 * 12.1, 8 * 12.1 + 20), "Hang": ("END", 4 * 21, 4 * 21 + 9)}


@pytest.mark.parametrize("case", range(23, 32))
def test_host_interpreter_stops_where_the_engine_has_to_take_over(case):
    """Round 5 (the rest of SURVEY 8 f4): programs that END, SLEEP, CALL, draw from the engine's RNG, divide by registers
    (tests/a2s/vmnotes.a2s), traced from the compiled reference's VM like the looping ones - the voices that run out
    until they are gone.  a2amd_vm_trace_host_exit() makes the look-ahead a2amd_vm_adopt makes for such a voice and
    stops the interpreter in front of the first VM run that is the engine's: up to there it reproduces the reference's
    windows and writes record for record (register divisors included), the exit lies in the frame the program text
    says, the state handed back is the one the engine's VM had when it started that run - for a voice that runs out,
    the run that ended it: the trace's last window ends exactly there."""
    d = load_vm_traces()[case]
    got, want, st, (has_exit, exit_when, stay), frames = replay_trace(d, with_exit=True)
    assert has_exit == 1
    why, lo_ms, hi_ms = STAYS[d["program"]]
    xf = ((exit_when - d["now"]) & 0xffffffff) >> 8            # frames into the trace
    assert lo_ms * 48 - 64 <= xf <= hi_ms * 48 + 1, (d["program"], xf, lo_ms * 48, hi_ms * 48)
    # where that is on the backend's fragment grid
    frag, acc = 0, 0
    while acc + frames[frag] <= xf:
        acc += frames[frag]
        frag += 1
    off = xf - acc

    def before(ev):
        return ev[0] < frag or (ev[0] == frag and ev[1] == "p" and ev[2] < off)
    # ours: everything in front of the exit; behind it the voice only gets default windows (and the rest of the exit's fragment)
    ours = [e for e in got if before(e) or (e[0] == frag and e[1] == "w" and got.index(e) < next(
        (i for i, g in enumerate(got) if g[0] == frag and g[1] == "p" and g[2] >= off), len(got)))]
    assert not [e for e in got if e[1] == "w" and e not in ours], "a write behind the exit"
    # the reference's: its events up to the window that ends at the exit (what follows belongs to the run that is the engine's)
    cut = len(want)
    for i, e in enumerate(want):
        if e[0] > frag or (e[0] == frag and e[1] == "p" and e[2] >= off):
            cut = i
            break
    if off:     # (the window [0, off) of the exit's fragment is part of the stay; writes behind it are the exit run's)
        k = next((i for i, e in enumerate(want) if e[0] == frag and e[1] == "p" and e[2] + e[3] == off), None)
        assert k is not None, (frag, off, [e for e in want if e[0] == frag])
        cut = k + 1
    else:
        cut = next((i for i, e in enumerate(want) if e[0] >= frag), len(want))
    ref = want[:cut]
    first = next((i for i, (a, b) in enumerate(zip(ours, ref)) if a != b), None)
    assert first is None and len(ours) == len(ref), (first, ours[first:first + 3] if first is not None else len(ours),
                                                     ref[first:first + 3] if first is not None else len(ref))
    assert len([e for e in ref if e[1] == "w"]) == stay and stay >= 1
    assert st.waketime == exit_when and st.state == 1       # (A2_WAITING: asleep in the delay in front of that run)
    if "gone" in d and why == "END":
        # the voice ran out: the engine ended it in the VM run at 'exit_when' - the last window it got ends there
        last = [e for e in d["events"] if e[0] == "p"][-1]
        assert d["fragments"] == d["gone"] + 1
        lastfrag_start = 64 * d["gone"]
        if last[1] == d["gone"]:
            assert lastfrag_start + last[2] + last[3] == xf, (last, xf)
        else:       # (it ended in the first frame of the fragment: no window there)
            assert xf == lastfrag_start, (last, xf)


def test_host_interpreter_cutoff_goes_through_the_coefficient_table():
    """A filter12 cutoff ramp: f12_CutOff keeps the ramper (filter12.c:141-147), every window of the ramp
    gets its coefficient (the head of f12_process, :86-96) - here from the table of f12_pitch2coeff over
    everything a2_P2I can return, which must agree with the formula evaluated directly."""
    import math
    code, n = asm(("LOAD", 6, 0, fx(2)), ("DELAY", 0, 0, fx(4)), ("LOAD", 6, 0, fx(-1)), ("DELAY", 0, 0, fx(4)), ("JUMP", 0, 0))
    st = VmState()
    st.state = 1
    st.waketime = 10 << 8
    recs = trace_host(code, n, st, [6], [(1, 0)], [WTOSC, FILTER12, PANMIX], [64] * 8)
    ramps = [r for r in recs if r[1] == R_F1RAMP]
    assert len(ramps) >= 8 and not [r for r in recs if r[1] == R_F1SET]
    # the first window of the ramp: 54 frames of a 192-frame ramp from 0 to 2.0 (ramper values are 8:24)
    timer = (fx(4) * MSDUR + 0x7fffff) >> 24
    delta = ((fx(2) << 8) * 256) // timer
    value = delta * 54
    pitch = value >> 8

    def p2i(pitch):
        tab = []
        b = 0x80000000
        for i in range(64):
            b2 = int(float(0x80000000) * float(np.float32(2.0) ** np.float32((i + 1) * np.float32(1.0 / 64))) + 0.5)
            tab.append((b, (b2 - b + 128) >> 8))
            b = b2
        nn, octv = pitch & 0xffff, pitch >> 16
        base, coeff = tab[nn >> 10]
        return ((((coeff * (nn & 0x3ff)) & 0xffffffff) >> 2) + base & 0xffffffff) >> ((7 - octv) & 31)
    f = np.float32(p2i(pitch)) * np.float32(261.626 / 16777216.0)
    want = 362 << 16 if f > 12000 else int(np.float32(512.0 * 65536.0) * math.sin(math.pi * float(f) / 48000))
    assert ramps[0][4] == want, (ramps[0], want)

#!/usr/bin/env python3
"""bench.py - voice-samples/sec of the MI355X voice-render path.

Default workload = BASELINE.json configs[3], the largest single-GPU config:
65 536 sustained leaf voices, each `wtosc; wtosc(add); panmix`, 256 per group
voice `inline; fbdelay; fbdelay` with the delay settings of the reference's
benchmark/fmtest4.a2s:83-95 (256 groups, 512 fbdelay instances), 48 kHz,
fragment = 64 frames, stereo.  --config 1 / 2 select configs[1] (1 024 x
wtosc->panmix) and configs[2] (16 384 x wtosc->filter12->panmix); both are
also run briefly and reported as extra keys of the default line.

One "step" = one offline a2_Run() buffer of --batch fragments (default 256 =
16 384 frames = 341 ms of audio) rendered for all voices THROUGH THE PRODUCT
ENTRY POINTS: a2amd_fragment_repeat() records the batch (an engine whose VMs all
sleep), a2amd_render(UPLOAD | SUBTREES | ROOT | READBACK | ASYNC) renders it and
enqueues the master bus for the host, a2amd_collect() delivers the previous
step's audio into host buffers while the GPU renders the next one.  Every step's
audio reaches the host inside the timed region; steps covered by the committed
oracle golden (tests/golden/bench_cfg*.hash.npy, made by make_bench_golden.py)
are compared hash by hash per fragment, the rest by a steady-state check.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns
its own voice subtrees (weak scaling: the config's voices per GPU); per step
each rank renders its subtrees into its partial of the root voice's inline bus,
the partials are summed with ONE RCCL reduce (int32, wrap-around sum => exact
in any order), and rank 0 runs the root chain (panmix must see the sum,
SURVEY.md 8e).

Prints one JSON line (rank 0).  Extra objects: "roofline" (the dominant kernel
against the HBM roofline, timed with HIP events on the launch stream),
"roofline_valu" (the same kernel against the integer-VALU issue rate, from the
PMC instruction counts under profiles/), "realtime" (measured per-fragment round
trips at this voice count) and "cpu_baseline" (the compiled reference timed on
this host's cores on a bounded sample of the same workload).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per voice-fragment (state read + written, bus share)
BYTES_PER_VOICE_FRAGMENT = {
    "osc-pan": 504.0,               # (152 + 96 B state) x 2 + 8 B bus share
    "osc-filter-pan": 792.0,        # + filter12's 144 B x 2
    "osc2-pan": 504.0 + 2 * 152.0,  # + the second oscillator
    "osc2-filter-pan": 792.0 + 2 * 152.0,   # filter12's voice + the second oscillator (round 6: k_leaf_osc2filtpan)
    **{f"{k}-pan": (64.0 * n + 96.0) * 2 + 8.0 for k, n in
       (("fm1", 1), ("fm2", 2), ("fm3", 3), ("fm4", 4), ("fm3p", 3), ("fm4p", 4), ("fm2r", 2), ("fm4r", 4))},
}
FBDELAY_BYTES_PER_FRAGMENT = 24.0 * 64 + 240.0      # 6 x 4 B per sample + state per fragment
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s
# integer VALU issue: 256 CUs x 4 SIMDs x 16 lanes per clock x 2.4 GHz (MI355X_MICROARCH.md)
VALU_PEAK_TLANEOPS = 256 * 4 * 16 * 2.4e9 / 1e12
LEAF_KERNEL = {"osc-pan": "k_leaf_oscpan", "osc-filter-pan": "k_leaf_oscfiltpan", "osc2-pan": "k_leaf_osc2pan",
               "osc2-filter-pan": "k_leaf_osc2filtpan"}

CONFIGS = {   # BASELINE.json configs[i]
    1: dict(voices=1024, chain="osc-pan", groups=0,
            label="1024 voices wtosc->panmix under the root, 48 kHz, fragment=64, stereo (BASELINE configs[1])"),
    2: dict(voices=16384, chain="osc-filter-pan", groups=0,
            label="16384 voices wtosc->filter12->panmix under the root, 48 kHz, fragment=64, stereo "
                  "(BASELINE configs[2])"),
    3: dict(voices=65536, chain="osc2-pan", groups=256,
            label="65536 voices 2xwtosc->panmix, 256 per group voice inline->fbdelay->fbdelay "
                  "(benchmark/fmtest4.a2s delay settings; 256 groups, 512 fbdelays), 48 kHz, fragment=64, "
                  "stereo (BASELINE configs[3])"),
}
CONFIGS[4] = dict(   # BASELINE configs[4] / SURVEY 8(d) config #5: one top-level group per GPU
    voices=32768, chain="osc-filter-pan", groups=0, tree=128,
    label="32768 voices wtosc->filter12->panmix per GPU: one top-level group voice (a2_NewGroup's driver "
          "inline->panmix->xinsert) of 128 sub-groups x 256 voices; 8 GPUs = the 262144 voices of BASELINE "
          "configs[4], the groups' sum into the root bus = ONE ncclReduce per buffer")
TOP_GROUP_VOICES = 32768
# Round 4: the path where north_star says it is bandwidth-bound - SURVEY 8(d)'s "private sample waves" case, which no
# BASELINE config has: configs[1]'s voice (wtosc -> panmix) x 65 536 over 2 048 uploaded sample waves of 65 536 samples
# (0.54 GB of int16 with their mip levels, 3.2 GB as Hermite coefficient entries: past every cache), 32 voices per
# wave at the configs' mixed pitches (+-2.5 octaves: mip levels 0-2, 0.25-2 source samples per frame).
CONFIGS[5] = dict(voices=65536, chain="osc-pan", groups=0, private=(2048, 65536, 256),
                  label="65536 voices wtosc->panmix over 2048 private looped sample waves of 65536 samples (period 256), "
                        "pitches +-2.5 octaves, 48 kHz, fragment=64, stereo (SURVEY 8d: the private-wave case; not a "
                        "BASELINE config)")
# Round 6: the subtractive note - wtosc; wtosc; filter12; panmix, the shape of every lead of the reference's benchmark/k2*.a2s -
# at configs[2]'s size, sustained and settled: what its quiet kernel (k_leaf_osc2filtpan) renders.  Not a BASELINE config.
CONFIGS[6] = dict(voices=16384, chain="osc2-filter-pan", groups=0,
                  label="16384 voices 2xwtosc->filter12->panmix under the root (the subtractive note; configs[2]'s size), "
                        "48 kHz, fragment=64, stereo (not a BASELINE config)")
UP, SUB, ROOTP, RB, KEEP, ASYNC = 4, 1, 2, 8, 16, 32


LINE_LIMIT = 4000       # the driver keeps the last 8 KB of stdout: the contract line stays well inside it
DETAILS_NAME = "bench_details.json"


def _r(x, nd=6):
    """Numbers of the contract line to 6 significant digits (the side file keeps them all)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def contract_line(full):
    """The ONE line the driver parses: the contract's keys, roofline, cpu_baseline and the
    engine-in-the-loop summary - at most LINE_LIMIT characters.  Everything else the run
    measured (engine cells, other configs, sweeps, prose) is in the side file."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                        "scaling", "dtype", "data", "parity_vs_golden", "realtime_factor"))
    line["vs_baseline"] = full.get("vs_baseline")
    cfg = full.get("config", {})
    line["config"] = _pick(cfg, ("workload", "voices_per_gpu", "chain", "groups", "fragments_per_step", "samplerate"))
    line["config"]["sharding"] = (cfg.get("sharding") or "")[:120]
    if isinstance(full.get("parity"), dict):
        line["parity"] = _pick(full["parity"], ("golden_steps_compared", "timed_steps_covered_by_golden",
                                                "golden_fragments_compared"))
    rf = full.get("roofline") or {}
    line["roofline"] = _pick(rf, ("bound", "kernel", "avg_launch_ms", "launches_timed", "algorithmic_bytes_per_launch",
                                  "achieved", "peak", "unit", "frac", "traffic_over_algorithmic", "binding_roofline"))
    line["roofline"]["traffic"] = rf.get("traffic")
    ts = rf.get("traffic_source") or {}
    line["roofline"]["traffic_file"] = ts.get("file")
    line["roofline"]["traffic_stale"] = ts.get("stale")
    rv = full.get("roofline_valu") or {}
    line["roofline_valu"] = _pick(rv, ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_of_nominal",
                                       "valu_insts_per_voice_fragment"))
    if isinstance(full.get("cpu_baseline"), dict):
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "single_thread_value", "host_cores"))
        line["cpu_baseline"]["sample"] = (cb.get("sample") or "")[:200]
    if isinstance(full.get("realtime"), dict):
        line["realtime"] = _pick(full["realtime"], ("voices", "fragment_ms_p50", "fragment_ms_p99", "holds_realtime"))
    if isinstance(full.get("fragment_roundtrip_with_reduce"), dict):
        line["fragment_roundtrip_with_reduce"] = _pick(full["fragment_roundtrip_with_reduce"],
                                                       ("voices", "fragment_ms_p50", "fragment_ms_p99", "holds_realtime"))
    oc = full.get("other_configs")
    if isinstance(oc, dict):
        line["other_configs"] = {}
        for k, v in oc.items():
            e = _pick(v, ("value", "ms_per_step", "parity_vs_golden", "kernel_voice_samples_per_s", "voices_total"))
            if isinstance(v.get("roofline"), dict):
                e["kernel"] = v["roofline"].get("kernel")
                e["frac"] = v["roofline"].get("frac")
            if isinstance(v.get("roofline_valu"), dict):
                e["valu_frac"] = v["roofline_valu"].get("frac")
            line["other_configs"][k.split(" (")[0]] = e
    mrv = full.get("max_realtime_voices")
    if isinstance(mrv, dict):
        line["max_realtime_voices"] = _pick(mrv, ("max_realtime_voices", "largest_size_tried", "n_gpus", "units",
                                                  "units+walk"))
        line["max_realtime_voices"]["scope"] = "GPU side, one synchronous round trip per 64-frame fragment" \
            if "max_realtime_voices" in mrv else "a2_Run(64) of the reference engine, p99 <= 1.333 ms"
    for k in ("value_a2_run", "max_realtime_voices_units", "max_realtime_voices_units_walk"):
        if k in full:
            line[k] = full[k]
    eil = full.get("engine_in_loop")
    if isinstance(eil, dict) and "cases" in eil:
        # one figure per cell: units+walk (else units) voice-samples/s at the large buffer and p99 us at a2_Run(64)
        cells = {}
        for label, entry in eil["cases"].items():
            c = {}
            for bk, e in entry.items():
                if not (isinstance(e, dict) and bk.startswith("a2_Run(")):
                    continue
                m = e.get("units+walk") or e.get("units") or {}
                if "voice_samples_per_s" not in m:
                    c["error"] = True
                elif bk == "a2_Run(64)":
                    c["p99_us_64"] = m.get("us_per_fragment_p99_steady")
                else:
                    c["vsps"] = m.get("voice_samples_per_s")
                    c["cpu_vsps"] = e.get("cpu_units_voice_samples_per_s")
            cells[label.split(" (")[0]] = c
        line["engine_in_loop"] = {"hash_equal": eil.get("hash_equal"), "mode": "units+walk", "cells": cells}
    elif isinstance(eil, dict):
        line["engine_in_loop"] = {"error": str(eil.get("error"))[:120]}
    line["details"] = full.get("details")
    line = _r(line)
    # whatever happens to the optional parts, the line stays inside the limit
    for drop in ("engine_in_loop", "other_configs", "realtime", "fragment_roundtrip_with_reduce", "max_realtime_voices",
                 "parity", "roofline_valu"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(drop, None)
    return line


def emit(full):
    """Side file (and stderr) get everything; stdout gets the contract line."""
    paths = []
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAILS_NAME), "w") as f:
                    json.dump(full, f, indent=1)
                paths.append(os.path.relpath(os.path.join(d, DETAILS_NAME), ROOT))
            except OSError:
                pass
    full["details"] = paths[0] if paths else None
    print("bench.py: details: " + json.dumps(full), file=sys.stderr, flush=True)
    s = json.dumps(contract_line(full))
    assert len(s) <= LINE_LIMIT, len(s)
    print(s, flush=True)


class Stats(ctypes.Structure):
    _fields_ = [("fragments", ctypes.c_uint64), ("voice_fragments", ctypes.c_uint64),
                ("records", ctypes.c_uint64), ("launches", ctypes.c_uint64),
                ("last_kernel_ms", ctypes.c_double), ("last_leaf_ms", ctypes.c_double),
                ("live_units", ctypes.c_uint32), ("live_voices", ctypes.c_uint32),
                ("live_waves", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("timed_leaf_ms", ctypes.c_double), ("timed_all_ms", ctypes.c_double),
                ("timed_batches", ctypes.c_uint64)]


def fnv1a_fragments(pcm, frag=64):
    """FNV-1a 64 of every 64-frame fragment (bytes of ch0, then ch1)."""
    nfr = pcm.shape[1] // frag
    blk = np.ascontiguousarray(pcm[:, :nfr * frag].reshape(pcm.shape[0], nfr, frag).transpose(1, 0, 2)).view(np.uint8)
    blk = blk.reshape(nfr, -1)
    h = np.full(nfr, 0xCBF29CE484222325, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for i in range(blk.shape[1]):
            h = (h ^ blk[:, i].astype(np.uint64)) * np.uint64(0x100000001B3)
    return h


def golden_path(voices, chain, groups, tree=0, private=None):
    if private:
        return os.path.join(ROOT, "tests", "golden", f"bench_{chain}_{voices}v_private{private[0]}x{private[1]}.hash.npy")
    if tree:    # configs[4]: keyed by the voices of the WHOLE job (all ranks)
        return os.path.join(ROOT, "tests", "golden", f"bench_cfg4_{chain}_{voices}v_{tree}sub.hash.npy")
    return os.path.join(ROOT, "tests", "golden", f"bench_{chain}_{voices}v_{groups}g.hash.npy")


def build_scene(be, voices, chain, groups, world=1, rank=0, tree=0, private=None):
    """The synthetic voice tree of a config (every rank plays different voices)."""
    from audiality2_amd import shard, synth
    sc = synth.Scene(be)
    sc.root()
    sc.nvoices = shard.voice_range(rank, voices)[0]
    if private:
        sc.private_waves(*private)
        sc.add_voices(voices, chain=chain, total=voices * world, private=True)
    elif tree:
        # configs[4]: top-level groups of `tree` sub-groups x 256 voices, this rank's share
        assert voices % TOP_GROUP_VOICES == 0 and TOP_GROUP_VOICES % tree == 0
        for _ in range(voices // TOP_GROUP_VOICES):
            top = sc.add_bus_group()
            for _ in range(tree):
                sub = sc.add_bus_group(top)
                sc.add_voices(TOP_GROUP_VOICES // tree, chain=chain, group=sub, total=voices * world)
    elif groups:
        per = voices // groups
        for _ in range(groups):
            grp = sc.add_group(preset="fmtest4")
            sc.add_voices(per, chain=chain, group=grp, total=voices * world)
    else:
        sc.add_voices(voices, chain=chain, total=voices * world)
    return sc


def host_threads():
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cpu_baseline(voices, chain, groups, oracle_fragments=100, private=None):
    """Reference (oracle/_ref/ref_bench) if it travelled, else the C port."""
    program = {"osc-pan": "OscPan", "osc-filter-pan": "OscFilterPan", "osc2-pan": "Osc2Pan", "fm1-pan": "Fm1Pan",
               "fm2-pan": "Fm2Pan", "fm4-pan": "Fm4Pan"}.get(chain)
    if groups and chain == "osc2-pan":
        program = "Osc2PanGroups"
    if private:
        program = None      # (no script uploads sample waves: the C restatement plays the scene bench.py builds)
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    script = os.path.join(ROOT, "tests", "a2s", "bench.a2s")
    ncores = host_threads()
    if program and os.path.exists(exe):
        # about 10 s of one core at ~1e8 (2 oscillators) .. 2e8 voice-samples/s
        per_frag = voices * 64.0
        frags1 = int(max(50, min(7500, 1.5e9 / per_frag)))
        # every thread is an independent engine state and needs whole groups
        tmax = min(ncores, max(1, voices // 256))
        res = {}
        for threads in sorted({1, tmax}):
            frags = frags1 if threads == 1 else int(min(7500, frags1 * min(threads, 16)))
            try:
                out = subprocess.run([exe, script, program, str(voices), str(frags), str(threads)],
                                     capture_output=True, text=True, timeout=300, check=True).stdout
                res[threads] = json.loads(out.strip().splitlines()[-1])
                if res[threads].get("active_voices", 0) < res[threads].get("voices", 0):
                    res[threads] = {"error": f"only {res[threads].get('active_voices')} voices came up"}
            except Exception as e:  # noqa: BLE001
                res[threads] = {"error": str(e)}
        ok = {t: r for t, r in res.items() if "voice_samples_per_s" in r}
        if ok:
            best = max(ok, key=lambda t: ok[t]["voice_samples_per_s"])
            r = ok[best]
            return {"value": r["voice_samples_per_s"], "unit": "voice-samples/s", "cores": best,
                    "kind": "reference",
                    "single_thread_value": ok.get(1, {}).get("voice_samples_per_s"),
                    "host_cores": ncores,
                    "sample": f"{r['voices']} voices {chain}" + (f" in groups of 256 (inline->fbdelay->fbdelay)" if groups else "")
                              + f", {r['fragments']} fragments of 64 frames ({r['fragments'] * 64 / 48000:.2f} s of audio), "
                              f"the compiled reference engine via a2_Run(64), {best} thread(s) = {best} independent "
                              f"engine states (the reference is single-threaded per state); {r['seconds']:.1f} s wall",
                    "errors": {str(t): r["error"] for t, r in res.items() if "error" in r} or None}
    # the C restatement, driven through the same call protocol
    from audiality2_amd import synth
    from audiality2_amd.replay import Backend
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liba2oracle.so"))
    be = Backend(lib, "a2o_", 48000, synth.basepitch_for(48000), 2)
    sc = build_scene(be, voices, chain, groups, private=private)
    sc.walk(64)
    be.render(64)
    lib.a2o_fragment_repeat.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
    t0 = time.perf_counter()
    done = 0
    while done < oracle_fragments:
        n = min(64, oracle_fragments - done)
        assert lib.a2o_fragment_repeat(be.ctx, 64, n) == 0
        be.render(n * 64)
        done += n
    dt = time.perf_counter() - t0
    be.close()
    return {"value": voices * 64.0 * oracle_fragments / dt, "unit": "voice-samples/s", "cores": 1,
            "kind": "port", "host_cores": ncores,
            "sample": f"{voices} voices {chain}, {groups} groups" + (f", {private[0]} private waves of {private[1]} samples" if private else "")
                      + f", {oracle_fragments} fragments of 64 frames, C restatement (oracle/a2o.c), 1 thread"}


# ---------------------------------------------------------------------------------------------
# The PRODUCT as an application gets it: the reference engine itself (A2S compiler, VM, event
# scheduler, voice walk: one CPU thread; oracle/_ref/ref_bench = the compiled reference + a
# timing harness) calling a2_Run(), its units replaced by the drop-in (LD_PRELOAD of
# liba2amd_units.so).  north_star's entry point (src/core.c:2004-2011, bufferdrv.c:28-40).
ENGINE_CASES = [   # label, program of tests/a2s/bench.a2s, voices
    ("configs[1]", "OscPan", 1024), ("configs[2]", "OscFilterPan", 16384),
    ("configs[3]", "Osc2PanGroups", 65536),
    ("configs[4] in one engine state", "FilterTree", 262144),
    ("variant 2b (scripted)", "OscPanScripted", 16384), ("variant 3b (scripted)", "OscFilterPanScripted", 16384),
    ("variant 2e (env unit per voice)", "OscPanEnvScripted", 16384),
    # configs[2]'s voices with 200 note births + 200 deaths per second in one group beside them (the walk's per-list epochs)
    ("churn (configs[2] + 200 notes/s)", "OscFilterPanChurn", 16384),
]
ENGINE_BUFFERS = (4096, 64)     # a2play's offline buffer; one fragment per a2_Run() = a realtime driver's


WALK_SO = os.path.join(ROOT, "audiality2_amd", "liba2amd_walk.so")


def engine_run(program, voices, fragments, buffer, dropin, hash_fragments=0, env_extra=None, wait=True, walk=False):
    """One ref_bench process; returns its JSON (or the Popen when wait=False).
    walk: INTEGRATION.md option C - liba2amd_walk.so in front of the units."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    env = dict(os.environ, A2REF_BUFFER=str(buffer))
    env.pop("LD_PRELOAD", None)
    if hash_fragments:
        env["A2REF_HASH"] = str(hash_fragments)
    if dropin:
        env["LD_PRELOAD"] = (WALK_SO + " " if walk else "") + os.path.join(ROOT, "audiality2_amd", "liba2amd_units.so")
    env.update(env_extra or {})
    p = subprocess.Popen([exe, "bench.a2s", program, str(voices), str(fragments), "1"], env=env,
                         cwd=os.path.join(ROOT, "tests", "a2s"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return engine_result(p) if wait else p


def engine_result(p, timeout=600):
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        return {"error": "timeout"}
    if p.returncode != 0:
        return {"error": f"rc {p.returncode}: {err[-300:]}"}
    try:
        res = json.loads(out.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return {"error": f"no JSON: {out[-200:]} {err[-200:]}"}
    for ln in err.splitlines():     # (A2AMD_WALK_STATS=1)
        if ln.startswith("a2amd walk:") and "voice visits skipped" in ln:
            w = ln.split()
            res["walk_stats"] = (int(w[w.index("voice") - 1]), int(w[w.index("made") - 1]), int(w[w.index("made") + 1].lstrip("(")))
    return res


def engine_in_loop(cases=None, buffers=ENGINE_BUFFERS, hash_fragments=64, gpu_fragments=1024, dropin_env=None):
    """a2_Run() with the reference engine in the loop, drop-in vs the engine's own CPU units:
    voice-samples/s, time per 64-frame fragment (p50 / p99 over the a2_Run() calls), and the
    FNV-1a hash of the first `hash_fragments` fragments after the warm-up from both runs.
    Two ways of loading the drop-in: "units" = LD_PRELOAD=liba2amd_units.so
    (option A: the engine's voice walk untouched), "units+walk" = liba2amd_walk.so in front
    of it (option C: the engine's a2_ProcessVoices replaced, sleeping voices are not visited)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    if not os.path.exists(exe):
        return {"error": "oracle/_ref/ref_bench (the compiled reference) did not travel"}
    cases = list(cases or ENGINE_CASES)
    modes = [("units", False)] + ([("units+walk", True)] if os.path.exists(WALK_SO) else [])
    # the CPU renders: independent processes, one thread each, all at once (the box has the cores)
    cpu, plan = {}, {}
    for label, program, voices in cases:
        for buf in buffers:
            hf = hash_fragments if voices < 65536 else min(hash_fragments, 16 if voices < 262144 else 8)
            # how long each drop-in run is ...
            nfr = max(gpu_fragments if buf > 64 else 600, buf // 64 * 4)
            nruns = {}
            for mode, walk in modes:
                # (without the walk's short cut a fragment costs ~40 ns per voice of engine walk: bound the run)
                # ... and with it a fragment of a quiet scene costs microseconds: four times as many
                # (round 6: the scripted cells as long as the others - the device VM adopts 16 384 voices over the first sixty
                # batches or so, and a run of 1 024 fragments was over before the steady state began)
                nruns[mode] = 4 * nfr if walk else \
                    max(hf, buf // 64, min(nfr, int(2.5e6 / (voices * 0.04))))
            # ... and where the SECOND hashed window lies: as far into the shortest of them as the one-thread CPU
            # render of the same scene gets in about a minute (1.5e9 voice-samples), on a 4 096-frame boundary
            tail = min(int(1.5e9 / (voices * 64.0)), min(nruns.values()) - hf) // 64 * 64
            tail = tail if tail >= hf else None
            plan[(label, buf)] = (hf, nruns, tail)
            ncpu = max(hf, buf // 64, (tail + hf) if tail is not None else 0)
            cpu[(label, buf)] = engine_run(program, voices, ncpu, buf, False, hf, wait=False,
                                           env_extra={"A2REF_HASH_AT": str(tail)} if tail is not None else None)
    cpu = {k: engine_result(p, timeout=1200) for k, p in cpu.items()}
    out = {"entry_point": "a2_Run(frames) of the compiled reference engine (oracle/_ref/ref_bench, one engine state, one "
                          "CPU thread for compiler + VM + voice walk); units: LD_PRELOAD=liba2amd_units.so; units+walk: "
                          "liba2amd_walk.so (the engine's a2_ProcessVoices replaced: sleeping voices are not visited) "
                          "in front of it",
           "hash": "FNV-1a 64 of the first hash_fragments fragments rendered after the warm-up, ch0 then ch1 per 64-frame "
                   "fragment; tail_hash: the same number of fragments from fragment tail_at of the timed run on (as deep "
                   "into the run as the one-thread CPU render gets in about a minute); cpu_hash / cpu_tail_hash = the "
                   "same engine with its own CPU units",
           "cases": {}}
    all_equal = True
    rt = {m: 0 for m, _ in modes}
    for label, program, voices in cases:
        entry = {"program": program, "voices": voices}
        for buf in buffers:
            c = cpu[(label, buf)]
            hf, nruns, tail = plan[(label, buf)]
            e = {}
            if "error" in c:
                e["error"] = "CPU run: " + c["error"]
                all_equal = False
            else:
                e = {"cpu_units_voice_samples_per_s": c["voice_samples_per_s"], "hash_fragments": hf,
                     "cpu_hash": c["hashes"][0], "tail_at": tail,
                     "cpu_tail_hash": c["tail_hashes"][0] if tail is not None else None}
                for mode, walk in modes:
                    g = engine_run(program, voices, nruns[mode], buf, True, hf, walk=walk,
                                   env_extra=dict(dropin_env or {}, **({"A2REF_HASH_AT": str(tail)} if tail is not None else {})))
                    if "error" in g:
                        e[mode] = {"error": g["error"]}
                        all_equal = False
                        continue
                    per = buf // 64
                    m = {"voice_samples_per_s": g["voice_samples_per_s"],
                         "us_per_fragment_p50": g["run_us_p50"] / per, "us_per_fragment_p99": g["run_us_p99"] / per,
                         "us_per_fragment_mean": g["seconds"] / g["fragments"] * 1e6,
                         # (the first quarter of the timed calls - where a scripted scene's voices are handed to the
                         # device VM - and the second half apart)
                         "us_per_fragment_p99_startup": g.get("run_us_p99_startup", g["run_us_p99"]) / per,
                         "us_per_fragment_p99_steady": g.get("run_us_p99_steady", g["run_us_p99"]) / per,
                         "fragments_timed": g["fragments"], "active_voices": g["active_voices"],
                         "realtime_at_48k": bool(g["run_us_p99"] / per <= 64.0 / 48000.0 * 1e6),
                         "hash": g["hashes"][0], "tail_hash": g["tail_hashes"][0] if tail is not None else None,
                         "hash_equal": g["hashes"][0] == c["hashes"][0] and g["active_voices"] == c["active_voices"] and
                                       (tail is None or g["tail_hashes"][0] == c["tail_hashes"][0])}
                    all_equal = all_equal and m["hash_equal"]
                    if buf == 64 and m["realtime_at_48k"] and m["hash_equal"] and "Scripted" not in program:
                        rt[mode] = max(rt[mode], voices)
                    e[mode] = m
            entry[f"a2_Run({buf})"] = e
        out["cases"][label] = entry
    out["hash_equal"] = all_equal
    out["max_realtime_voices_one_engine_state"] = {
        "what": "largest of the sustained-voice cases above whose a2_Run(64) - one synchronous GPU round trip per "
                "64-frame fragment - has p99 <= 1.333 ms, audio equal to the CPU render", **rt}
    return out


def pmc_entry(chain, voices, groups, B):
    """PMC-derived figures for this workload's dominant kernel (profiles/*.json, tools/pmc_summary.py)."""
    key = f"{chain}/{voices}/{groups}/{B}"
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if key in d:
            return dict(d[key], file=f"profiles/{name}")
        old = f"{chain}/{voices}/{B}"
        if not groups and old in d:
            return dict(d[old], file=f"profiles/{name}")
    return {}


_CODE_SHA = None


def kernel_code_sha(kernel):
    """Hash of the kernel's disassembly in the liba2amd.so THIS run loaded (tools/isa_mix.py):
    figures kept under profiles/ carry the hash of the binary they were measured on."""
    global _CODE_SHA
    if _CODE_SHA is None:
        _CODE_SHA = {}
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import hashlib
            import isa_mix
            for name, ins in isa_mix.extract():
                _CODE_SHA[name] = hashlib.sha256("\n".join(f"{m} {ops}" for _, m, ops, _ in ins).encode()).hexdigest()[:16]
        except Exception as e:  # noqa: BLE001  (no llvm-objdump on the box: say so in the line)
            _CODE_SHA = {"error": str(e)}
    for name, sha in _CODE_SHA.items():
        if kernel in name and "commit" not in name:
            return sha
    return None


def valu_mix_entry(kernel):
    """The measured issue rate of the kernel's own instruction mix (tools/valu_mix.py)."""
    try:
        for name in ("r06_valu_mix.json", "r04_valu_mix.json", "r03_valu_mix.json"):
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                with open(path) as f:
                    return json.load(f)["kernels"].get(kernel)
    except (OSError, ValueError, KeyError):
        pass
    return None


class Runner:
    """One config on one GPU through the product entry points."""

    def __init__(self, audiality2_amd, voices, chain, groups, B, device=0, world=1, rank=0, tree=0, private=None):
        self.voices, self.chain, self.groups, self.B = voices, chain, groups, B
        self.rank, self.world, self.tree, self.private = rank, world, tree, private
        self.be = audiality2_amd.open_backend(48000, None, 2, device=device, max_batch=B)
        lib = self.lib = self.be.lib
        lib.a2amd_fragment_repeat.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
        lib.a2amd_get_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(Stats)]
        lib.a2amd_set_profiling.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.a2amd_collect.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)), ctypes.c_uint]
        lib.a2amd_collect.restype = ctypes.c_int
        self.frames = B * 64
        self.bufs = [np.zeros((2, self.frames), dtype=np.int32) for _ in range(2)]
        self.ptrs = []
        for b in self.bufs:
            p = (ctypes.POINTER(ctypes.c_int32) * 2)()
            for c in range(2):
                p[c] = b[c].ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
            self.ptrs.append(p)
        self.sc = build_scene(self.be, voices, chain, groups, world, rank, tree, private)
        self.step_no = 0            # steps issued
        self.got = 0                # steps collected
        self.kept = {}              # step -> audio (the steps the golden covers, and the last one)
        self.keep_upto = 0
        self.last = None

    def err(self):
        return RuntimeError(self.be._err(self.be.ctx).decode(errors="replace"))

    def issue(self):
        if self.step_no == 0:
            self.sc.walk(64)        # the explicit engine walk: voice inits + control writes
            n = self.B - 1
        else:
            n = self.B
        if n and self.lib.a2amd_fragment_repeat(self.be.ctx, 64, n):
            raise self.err()
        if self.be._render(self.be.ctx, UP | SUB | ROOTP | RB | ASYNC, None, 0) < 0:
            raise self.err()
        self.step_no += 1

    def collect(self):
        k = self.got & 1
        if self.rank != 0:          # (multi-GPU: the audio comes out on rank 0)
            self.got += 1
            self.last = self.bufs[k]
            return
        if self.lib.a2amd_collect(self.be.ctx, self.ptrs[k], self.frames) != self.frames:
            raise self.err()
        if self.got < self.keep_upto:
            self.kept[self.got] = self.bufs[k].copy()
        self.last = self.bufs[k]
        self.got += 1

    def run(self, nsteps):
        """nsteps steps, every one delivered to host buffers; at most two in flight."""
        for _ in range(nsteps):
            self.issue()
            if self.step_no - self.got == 2:
                self.collect()
        while self.got < self.step_no:
            self.collect()

    def profile(self, nsteps):
        """Same steps with every launch bracketed by HIP events on the launch stream."""
        self.lib.a2amd_set_profiling(self.be.ctx, 1)
        self.run(nsteps)
        st = Stats()
        self.lib.a2amd_get_stats(self.be.ctx, ctypes.byref(st))
        self.lib.a2amd_set_profiling(self.be.ctx, 0)
        n = max(st.timed_batches, 1)
        return st.timed_leaf_ms / n, st.timed_all_ms / n, int(st.timed_batches)

    def realtime(self, n=300):
        """One 64-frame fragment per render, full synchronous round trip (records up,
        kernels, master bus back): what a realtime driver with buffer=64 would see of
        the GPU side (the engine's own voice walk is not in it)."""
        out = np.zeros((2, 64), dtype=np.int32)
        p = (ctypes.POINTER(ctypes.c_int32) * 2)()
        for c in range(2):
            p[c] = out[c].ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        ts = []
        for i in range(n + 20):
            t0 = time.perf_counter()
            if self.lib.a2amd_fragment_repeat(self.be.ctx, 64, 1):
                raise self.err()
            if self.be._render(self.be.ctx, UP | SUB | ROOTP | RB, p, 64) != 64:
                raise self.err()
            if i >= 20:
                ts.append((time.perf_counter() - t0) * 1e3)
        ts = np.sort(np.array(ts))
        budget = 64.0 / 48000.0 * 1e3
        return {"voices": self.voices, "fragments_timed": n, "fragment_ms_p50": float(ts[len(ts) // 2]),
                "fragment_ms_p99": float(ts[min(len(ts) - 1, int(len(ts) * 0.99))]), "fragment_ms_max": float(ts[-1]),
                "budget_ms": budget, "holds_realtime": bool(ts[min(len(ts) - 1, int(len(ts) * 0.99))] <= budget),
                "scope": "GPU side of one 64-frame fragment through a2amd_fragment_repeat + a2amd_render(ALL), "
                         "synchronous; the engine's own CPU voice walk is not included"}

    def close(self):
        self.be.close()


def check_golden(r, nsteps):
    """Compare every rendered step the committed oracle golden covers; returns
    (steps compared, all equal) or (0, None) without a golden for this workload."""
    job = r.tree or r.world > 1     # (the whole job's audio: every rank's voices and groups in one oracle scene)
    path = golden_path(r.voices * r.world if job else r.voices, r.chain, r.groups * (r.world if job else 1), r.tree, r.private)
    if not os.path.exists(path):
        return 0, None
    gold = np.load(path)
    if job:
        # configs[4], any config at N > 1: the golden holds the first fragments of step 0 (of the whole job's audio)
        if 0 not in r.kept:
            return 0, None
        n = min(len(gold), r.B)
        return 1, bool(np.array_equal(fnv1a_fragments(r.kept[0][:, :n * 64]), gold[:n]))
    n = min(len(gold) // r.B, nsteps)
    ok = True
    for s in range(n):
        if s not in r.kept:
            return s, None
        if not np.array_equal(fnv1a_fragments(r.kept[s]), gold[s * r.B:(s + 1) * r.B]):
            ok = False
    return n, ok


def golden_steps(voices, chain, groups, B, tree=0, private=None):
    path = golden_path(voices, chain, groups, tree, private)
    if tree:
        return 1 if os.path.exists(path) else 0
    return len(np.load(path)) // B if os.path.exists(path) else 0


def measure_single(audiality2_amd, cfg, B, steps, warmup, device=0, with_realtime=True):
    """One config at N=1: returns the dict of measured figures."""
    import gc
    import torch
    r = Runner(audiality2_amd, cfg["voices"], cfg["chain"], cfg["groups"], B, device, tree=cfg.get("tree", 0), private=cfg.get("private"))
    gsteps = golden_steps(cfg["voices"], cfg["chain"], cfg["groups"], B, cfg.get("tree", 0), cfg.get("private"))
    r.keep_upto = gsteps
    r.run(1)                        # step 0: voices are born
    r.run(warmup)
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()                    # a full collection mid-loop costs tens of ms of host time
    t0 = time.perf_counter()
    r.run(steps)                    # (run() returns when the last step's audio is in host memory)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    total_steps = 1 + warmup + steps
    compared, ok = check_golden(r, total_steps)
    if ok is False and os.environ.get("A2AMD_DEBUG"):
        print("bench.py: ablation run (A2AMD_DEBUG set): output differs from the golden, number is NOT a result", file=sys.stderr)
    elif ok is False:
        raise SystemExit("bench.py: GPU render differs from the oracle golden; refusing to report a number")
    last = r.last.copy()
    # steady state beyond the golden: the scene is stationary, so every later step must
    # stay within the level of the verified ones (catches silence, runaway, stuck output)
    peak_ref = max((int(np.abs(a).max()) for a in r.kept.values()), default=None)
    leaf_ms, all_ms, nprof = r.profile(min(steps, 32))
    res = {"voices": cfg["voices"], "chain": cfg["chain"], "groups": cfg["groups"], "private": cfg.get("private"), "seconds": dt,
           "value": float(cfg["voices"]) * B * 64 * steps / dt, "ms_per_step": dt / steps * 1e3,
           "leaf_ms": leaf_ms, "all_ms": all_ms, "launches_timed": nprof,
           "golden_steps_compared": compared, "parity_vs_golden": ok,
           "timed_steps_covered_by_golden": max(0, min(compared, total_steps) - 1 - warmup),
           "last_step_peak": int(np.abs(last).max()), "golden_peak": peak_ref,
           "last_step_nonzero": bool(last.any())}
    if with_realtime:
        res["realtime"] = r.realtime()
    r.close()
    return res


def realtime_sweep(audiality2_amd, device=0, chain="osc-filter-pan", sizes=(262144, 524288, 1048576), n=200):
    """The other half of BASELINE's metric - max realtime voices - on the GPU side: the largest
    N for which ONE 64-frame fragment of N sustained voices (wtosc->filter12->panmix, the
    north-star's voice, in BASELINE configs[4]'s tree: top-level groups of 128 sub-groups x 256
    voices - under ONE bus every voice's mix-down lands on the same 512 bytes) makes the full
    synchronous round trip (record, upload, kernels, master
    bus back on the host) in p99 <= 1.333 ms.  The engine's own CPU voice walk is not in this
    figure; engine_in_loop has the same with the reference engine calling a2_Run(64)."""
    rows, best = [], 0
    for nv in sizes:
        r = Runner(audiality2_amd, nv, chain, 0, 1, device, tree=CONFIGS[4]["tree"])
        r.run(2)                    # voices are born, one quiet fragment
        rt = r.realtime(n)
        r.close()
        rows.append({k: rt[k] for k in ("voices", "fragment_ms_p50", "fragment_ms_p99", "fragment_ms_max", "holds_realtime")})
        if rt["holds_realtime"]:
            best = nv
        else:
            break
    return {"chain": chain, "tree": "top-level groups of 128 sub-groups x 256 voices (a2_NewGroup drivers)",
            "budget_ms": 64.0 / 48000.0 * 1e3, "max_realtime_voices": best,
            "largest_size_tried": rows[-1]["voices"], "sweep": rows,
            "scope": "GPU side: a2amd_fragment_repeat(64, 1) + a2amd_render(UPLOAD|SUBTREES|ROOT|READBACK), synchronous, "
                     "per 64-frame fragment; one MI355X"}


def roofline_objects(res, B):
    chain, voices, groups = res["chain"], res["voices"], res["groups"]
    bpvf = BYTES_PER_VOICE_FRAGMENT[chain]
    private = res.get("private")
    if private:
        # + the window of its wave a voice reads per fragment ("unique wave bytes"): 64 frames x the source
        # samples per frame at the mip level wtosc picks (wtosc.c:250-258: halved until <= 2.0) + 5 samples of
        # interpolation margin, 2 bytes each; averaged over the scene's pitches
        per = private[2]
        wb = 0.0
        for k in range(voices):
            r = 2.0 ** (((k % 61) - 30) / 12.0) * per * 261.626 / 48000.0
            while r > 2.0:
                r *= 0.5
            wb += 2.0 * (64.0 * r + 5.0)
        bpvf = bpvf + wb / voices
    alg = bpvf * voices * B
    # (the private-wave scene is played from the waves' samples unless A2AMD_RAW=0 asks for coefficient entries)
    coef_path = bool(private) and os.environ.get("A2AMD_RAW", "") == "0"
    pmc = pmc_entry(chain + (("-private-coef" if coef_path else "-private") if private else ""), voices, groups, B)
    leaf_s = res["leaf_ms"] * 1e-3
    achieved = alg / leaf_s / 1e9 if leaf_s > 0 else None
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
            "traffic": pmc.get("hbm_bytes_per_launch"),
            "kernel": LEAF_KERNEL.get(chain, "k_leaf_fmpan" if chain.startswith("fm") else "k_voices"),
            "avg_launch_ms": res["leaf_ms"], "launches_timed": res["launches_timed"],
            "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_voice_sample": bpvf / 64.0,
            "timing": "HIP events on the launch stream around every launch, separate pass right after the "
                      "timed region (graph replay off)",
            "all_kernels_ms_per_step": res["all_ms"],
            "other_kernels_ms_per_step": res["all_ms"] - res["leaf_ms"]}
    if roof["traffic"] and leaf_s > 0:
        # what the counters say really moved, against the same peak: for the BASELINE configs (24 built-in waves, state
        # in registers over 256 fragments) a small fraction of the model's bytes; for the private-wave scene the figure
        # that says whether HBM is the bound
        roof["traffic_frac"] = roof["traffic"] / leaf_s / 1e9 / HBM_PEAK_GBPS
        roof["traffic_over_algorithmic"] = roof["traffic"] / alg
    if not private:
        # (say it in the line)  The BASELINE configs play 24 built-in waves and keep their state in registers
        # over the 256 fragments of a launch: the kernel is bound by VALU issue - "roofline_valu" is the bound that
        # applies - and 'achieved' / 'frac' here price SURVEY 8(d)'s MODEL bytes, which the chip does not move
        roof["frac_is"] = "model bytes (SURVEY 8d: unit state + bus share per voice-fragment) over HBM peak - NOT traffic: " \
                          "'traffic' / 'traffic_frac' are what the counters saw move; the binding roofline is roofline_valu"
        roof["binding_roofline"] = "roofline_valu"
    if private:
        roof["binding_roofline"] = "valu-issue (samples path) / hbm (A2AMD_RAW=0, coefficient entries): see traffic_frac"
        roof["wave_data_read_as"] = "Hermite coefficient entries (12 B per sample and tap)" if coef_path else \
            "int16 samples (A2D_WF_RAWTAPS: the footprint policy of a2amd_wave_upload; A2AMD_RAW=0 for the entries)"
        roof["algorithmic_model"] = ("SURVEY 8(d): 504 B of unit state and bus share per voice-fragment + the window of its wave "
                                     "a voice reads per fragment (64 x source samples per frame + 5 samples, int16), averaged "
                                     "over the scene's pitches")
    if groups:
        roof["other_kernels"] = (f"{groups} group chains inline->fbdelay->fbdelay + the root chain: "
                                 f"{2 * groups * FBDELAY_BYTES_PER_FRAGMENT * B / 1e6:.1f} MB algorithmic per step")
    ipvf = pmc.get("valu_insts_per_voice_fragment")
    kern = roof["kernel"]
    sha_now = kernel_code_sha(kern)
    sha_pmc = pmc.get("code_sha")
    # counters are not collected in this run (separate rocprofv3 passes: tools/profile_round3.sh);
    # what is quoted from profiles/ says which binary it was measured on
    roof["traffic_source"] = {"file": pmc.get("file"), "measured_on_code_sha": sha_pmc, "this_run_code_sha": sha_now,
                              "stale": (sha_pmc != sha_now) if (sha_pmc and sha_now) else None}
    mix = valu_mix_entry(kern)
    hot = (mix or {}).get("hot_loops", {}).get("taps") or {}
    meas = hot.get("measured") or {}
    mix_peak = max((v.get("independent_T", 0.0) for v in meas.values() if isinstance(v, dict)), default=0.0) or None
    mix_dep = max((v.get("as_in_kernel_T", 0.0) for v in meas.values() if isinstance(v, dict)), default=0.0) or None
    valu = {"bound": "valu-issue", "peak": mix_peak or VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s",
            "peak_is": ("the measured rate at which the SIMDs issue THIS kernel's hot-loop instruction mix with independent "
                        "operands (tools/isa_mix.py reads the mix off the shipped binary, tools/ubench/valu_mix.hip issues it)")
            if mix_peak else "nominal (no mix measurement for this kernel under profiles/)",
            "nominal_peak": VALU_PEAK_TLANEOPS,
            "nominal_peak_is": "256 CU x 4 SIMD x 16 lanes x 2.4 GHz: every wave64 instruction taken as 4 cycles",
            "mix_rate_with_the_kernels_dependencies": mix_dep,
            "mix": {"fraction_2_cycle": hot.get("fraction_2_cycle"), "by_class": hot.get("by_class"),
                    "additive_model_T": hot.get("additive_mix_rate_T_lane_ops"),
                    "measured_on_code_sha": (mix or {}).get("code_sha"), "this_run_code_sha": sha_now,
                    "stale": ((mix or {}).get("code_sha") != sha_now) if (mix and sha_now) else None},
            "kernel": kern, "valu_insts_per_voice_fragment": ipvf,
            "source": pmc.get("source", "no PMC summary for this workload under profiles/"),
            "insts_measured_on_code_sha": sha_pmc,
            "note": "achieved = wave-level VALU instructions per voice-fragment (PMC SQ_INSTS_VALU / voice-fragments, "
                    "separate rocprofv3 pass) x 64 lanes / this run's kernel time"}
    if ipvf and leaf_s > 0:
        valu["achieved"] = ipvf * 64.0 * voices * B / leaf_s / 1e12
        valu["frac"] = valu["achieved"] / valu["peak"]
        valu["frac_of_nominal"] = valu["achieved"] / VALU_PEAK_TLANEOPS
    else:
        valu["achieved"] = valu["frac"] = valu["frac_of_nominal"] = None
    return roof, valu


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json configs[i]")
    ap.add_argument("--voices", type=int, default=None, help="override: voices per GPU")
    ap.add_argument("--chain", default=None, choices=sorted(BYTES_PER_VOICE_FRAGMENT))
    ap.add_argument("--groups", type=int, default=None,
                    help="override: this many inline->fbdelay->fbdelay group voices over the leaves")
    ap.add_argument("--batch", type=int, default=256, help="fragments per step (256 = 341 ms of audio)")
    ap.add_argument("--reduce-group", type=int, default=8,
                    help="N>1: steps whose root-bus partials are summed by one collective")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[1] / configs[2] extra keys")
    ap.add_argument("--no-realtime", action="store_true")
    ap.add_argument("--no-engine", action="store_true", help="skip the engine-in-the-loop (a2_Run) object")
    args = ap.parse_args()

    cfg = dict(CONFIGS[args.config])
    custom = False
    for k in ("voices", "chain", "groups"):
        if getattr(args, k) is not None and getattr(args, k) != cfg[k]:
            cfg[k] = getattr(args, k)
            custom = True
    if custom and args.config == 4 and cfg["chain"] == CONFIGS[4]["chain"] and cfg["voices"] % TOP_GROUP_VOICES == 0:
        ng = cfg["voices"] // TOP_GROUP_VOICES
        cfg["label"] = (f"{cfg['voices']} voices wtosc->filter12->panmix per GPU: {ng} top-level group voices of 128 "
                        f"sub-groups x 256 voices" + (" = ALL of BASELINE configs[4] on one GPU" if ng == 8 else ""))
    elif custom:
        cfg["label"] = (f"{cfg['voices']} voices/GPU, {cfg['chain']}, {cfg['groups']} groups, 48 kHz, fragment=64, "
                        f"stereo (custom; not a BASELINE config)")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # A2AMD_BENCH_FORCE_DIST=1: take the multi-rank code path (torch stream,
    # phase-split render, RCCL reduce of the root bus) even with one rank
    multi = world > 1 or os.environ.get("A2AMD_BENCH_FORCE_DIST") == "1"

    import audiality2_amd
    B = args.batch

    if not multi:
        res = measure_single(audiality2_amd, cfg, B, args.steps, args.warmup, local_rank,
                             with_realtime=not args.no_realtime)
        roof, valu = roofline_objects(res, B)
        line = {
            "metric": "voice-samples/sec + max realtime voices @48kHz",
            "value": res["value"], "unit": "voice-samples/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": cfg["label"], "voices_per_gpu": cfg["voices"], "chain": cfg["chain"],
                       "groups": cfg["groups"], "fragments_per_step": B, "samplerate": 48000,
                       "sharding": "single GPU",
                       "timed_region": "per step: a2amd_fragment_repeat(64, B) + a2amd_render(UPLOAD|SUBTREES|"
                                       "ROOT|READBACK|ASYNC) + a2amd_collect() of the previous step into host "
                                       "buffers (double-buffered); every step's master bus reaches the host"},
            "realtime_factor": res["value"] / (cfg["voices"] * 48000.0),
            "parity_vs_golden": res["parity_vs_golden"],
            "parity": {k: res[k] for k in ("golden_steps_compared", "timed_steps_covered_by_golden",
                                           "last_step_peak", "golden_peak", "last_step_nonzero")},
            "roofline": roof, "roofline_valu": valu,
        }
        if "realtime" in res:
            line["realtime"] = res["realtime"]
        if not args.no_extra and not custom and args.config == 3:
            # the other two single-GPU BASELINE configs, same code path, shorter runs
            extra = {}
            for i in (1, 2):
                e = measure_single(audiality2_amd, CONFIGS[i], B, max(20, args.steps // 2), args.warmup, local_rank,
                                   with_realtime=not args.no_realtime)
                er, ev = roofline_objects(e, B)
                extra[f"configs[{i}]"] = {
                    "workload": CONFIGS[i]["label"], "value": e["value"], "ms_per_step": e["ms_per_step"],
                    "parity_vs_golden": e["parity_vs_golden"], "golden_steps_compared": e["golden_steps_compared"],
                    "roofline": {k: er[k] for k in ("kernel", "avg_launch_ms", "achieved", "frac", "traffic",
                                                    "all_kernels_ms_per_step")},
                    "roofline_valu": {k: ev[k] for k in ("valu_insts_per_voice_fragment", "achieved", "frac")},
                    "realtime": {k: e["realtime"][k] for k in ("fragment_ms_p50", "fragment_ms_p99", "holds_realtime")}
                    if "realtime" in e else None}
            # variant 2b of SURVEY 8(d): the same voices driven like a script drives them - in
            # every second fragment a voice gets a window that ends inside the fragment, a
            # pitch ramp, and the rest of the fragment (k_leaf_recs).  The driver is Python
            # over the C ABI, so only the kernels are timed (HIP events).
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import scripted_timing
            sres = scripted_timing.measure(voices=4096, chain="osc-pan", batch=64, period=2, what="pitch",
                                           names=("quiet", "scripted", "scripted2"), Stats=Stats)
            extra["variant 2b (scripted)"] = {
                "workload": "4096 voices wtosc->panmix, 64-fragment batches, a split window and a pitch ramp on every "
                            "voice in every second fragment",
                "kernels_ms_per_batch": sres["scripted2"]["kernels_ms_per_batch"],
                "kernel_voice_samples_per_s": sres["scripted2"]["voice_samples_per_s"],
                "quiet_kernels_ms_per_batch": sres["quiet"]["kernels_ms_per_batch"],
                "timing": "HIP events around the batch's kernels; the engine-in-the-loop figure for this "
                          "variant is in profiles/ (tests/measure/engine_in_loop.py)"}
            line["other_configs"] = extra
        if not args.no_realtime and not custom and args.config == 3:
            line["max_realtime_voices"] = realtime_sweep(audiality2_amd, local_rank)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg["voices"], cfg["chain"], cfg["groups"], private=cfg.get("private"),
                                                oracle_fragments=16 if cfg.get("private") else 100)
        if not args.no_engine and not custom and args.config == 3:
            line["engine_in_loop"] = engine_in_loop()
            if "max_realtime_voices" in line and "max_realtime_voices_one_engine_state" in line["engine_in_loop"]:
                line["max_realtime_voices"]["engine_in_loop"] = line["engine_in_loop"]["max_realtime_voices_one_engine_state"]
            # SURVEY 8(d)'s own definition of the metric - "wall time of the render loop (a2_Run calls only)" - at the
            # top level, beside 'value' (which times the C ABI without an engine above it): configs[3] through
            # a2_Run(4096) of the reference engine, with the drop-in behind the unit API alone ("units") and with the
            # replaced voice walk in front of it ("units+walk", INTEGRATION.md option C)
            try:
                c3 = line["engine_in_loop"]["cases"]["configs[3]"]["a2_Run(4096)"]
                line["value_a2_run"] = {"unit": "voice-samples/s", "workload": "configs[3], a2_Run(4096) of the reference engine",
                                        **{m: c3[m]["voice_samples_per_s"] for m in ("units", "units+walk") if m in c3 and
                                           "voice_samples_per_s" in c3[m]},
                                        "hash_equal": all(c3[m].get("hash_equal", False) for m in ("units", "units+walk") if m in c3)}
                rt = line["engine_in_loop"]["max_realtime_voices_one_engine_state"]
                line["max_realtime_voices_units"] = rt.get("units")
                line["max_realtime_voices_units_walk"] = rt.get("units+walk")
            except (KeyError, TypeError):
                pass
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        emit(line)
        return

    # ------------------------------------------------------------------ N > 1
    # One process per GPU.  Every rank builds its own subtrees under its own copy of the
    # root voice and runs the SAME product loop as N = 1; inside a2amd_render() the
    # ranks' root-bus partials are summed by one ncclReduce (RCCL over xGMI, called from
    # liba2amd on the render stream) and rank 0 runs the root chain and gets the audio.
    # torch.distributed only carries the 128 byte RCCL id to the other ranks and the
    # timing barriers the bench contract asks for.
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    # The contract line is to be the ONLY thing on stdout: RCCL writes a version banner to fd 1 through C
    # stdio when its first communicator comes up - until the line is printed, fd 1 is stderr.
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    dist.init_process_group("nccl", rank=rank, world_size=world,
                            device_id=torch.device("cuda", local_rank))
    # barrier = a (pre-warmed) 1-element all-reduce every rank must join, with the
    # device idle on both sides; dist.barrier() itself costs tens of ms on first use
    token = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", local_rank))

    def fence():
        torch.cuda.synchronize()
        dist.all_reduce(token)
        torch.cuda.synchronize()

    def timed_job(cfg, steps, warmup):
        """A Runner of its own RCCL communicator, born, warmed up and timed: (runner, seconds)."""
        r = Runner(audiality2_amd, cfg["voices"], cfg["chain"], cfg["groups"], B, local_rank, world=world, rank=rank,
                   tree=cfg.get("tree", 0))
        r.keep_upto = 1 if os.path.exists(golden_path(cfg["voices"] * world, cfg["chain"], cfg["groups"] * world,
                                                      cfg.get("tree", 0))) else 0
        if world == 1 and not cfg.get("tree"):      # (A2AMD_BENCH_FORCE_DIST=1: the single-GPU golden, every step)
            r.keep_upto = golden_steps(cfg["voices"], cfg["chain"], cfg["groups"], B)
        lib = r.lib
        lib.a2amd_dist_unique_id.argtypes = [ctypes.c_void_p]
        lib.a2amd_dist_init.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        idbuf = (ctypes.c_uint8 * 128)()
        if rank == 0 and lib.a2amd_dist_unique_id(idbuf):
            raise r.err()
        idt = torch.tensor(list(idbuf), dtype=torch.uint8, device=torch.device("cuda", local_rank))
        dist.broadcast(idt, src=0)
        idbuf = (ctypes.c_uint8 * 128)(*idt.cpu().tolist())
        if lib.a2amd_dist_init(r.be.ctx, idbuf, rank, world):
            raise r.err()
        fence()
        r.run(1)                # step 0: voices are born (the first reduce builds RCCL's channels)
        r.run(warmup)
        fence()
        fence()
        import gc
        gc.collect()
        gc.disable()
        t0 = time.perf_counter()
        r.run(steps)
        fence()
        dt = time.perf_counter() - t0
        gc.enable()
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return r, float(t.item())

    r, dt = timed_job(cfg, args.steps, args.warmup)
    leaf_ms, all_ms, nprof = r.profile(min(args.steps, 16))
    last = r.last
    # a realtime driver's view at N > 1: one 64-frame fragment per render - every fragment pays the reduce
    fence()
    frag_rt = None if args.no_realtime else r.realtime(200)
    fence()
    line = None
    if rank == 0:
        compared, ok = check_golden(r, 1 + args.warmup + args.steps)
        if ok is False:
            raise SystemExit("bench.py: the ranks' summed render differs from the oracle golden; refusing to report a number")
        value = float(cfg["voices"]) * world * B * 64 * args.steps / dt
        res = {"chain": cfg["chain"], "voices": cfg["voices"], "groups": cfg["groups"],
               "leaf_ms": leaf_ms, "all_ms": all_ms, "launches_timed": nprof}
        roof, valu = roofline_objects(res, B)
        line = {
            "metric": "voice-samples/sec + max realtime voices @48kHz",
            "value": value, "unit": "voice-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": cfg["label"] + f" per GPU x {world}", "voices_per_gpu": cfg["voices"],
                       "chain": cfg["chain"], "groups": cfg["groups"], "fragments_per_step": B, "samplerate": 48000,
                       "sharding": "voice subtrees per GPU; per step ONE ncclReduce(int32, sum) of the root voice's "
                                   "inline bus over xGMI inside a2amd_render (RCCL called from liba2amd), root "
                                   "chain + audio delivery on rank 0",
                       "timed_region": "per step and rank: a2amd_fragment_repeat(64, B) + a2amd_render(UPLOAD|"
                                       "SUBTREES|ROOT|READBACK|ASYNC); rank 0: a2amd_collect() of the previous "
                                       "step into host buffers"},
            "realtime_factor": value / (cfg["voices"] * world * 48000.0),
            "parity_vs_golden": ok,
            "parity": {"golden_fragments_compared": min(B, 64) if compared else 0,
                       "what": "the first fragments of step 0 of the WHOLE job's audio (all ranks' subtrees summed by "
                               "the reduce, root chain on rank 0) against the CPU oracle's render of the whole scene "
                               "(tests/golden/make_bench_golden.py: every rank's voices and groups in one oracle state)"}
            if compared else None,
            "roofline": roof, "roofline_valu": valu,
            "output_check": {"peak": int(np.abs(last).max()), "nonzero": bool(last.any())},
        }
        if frag_rt:
            frag_rt["scope"] = ("rank 0's side of one 64-frame fragment at N = %d: a2amd_fragment_repeat(64, 1) + a2amd_render(ALL), "
                                "synchronous - records up, this rank's subtrees, the ncclReduce of the root bus (once per "
                                "fragment in this mode), root chain, master bus back; the other ranks run the same loop" % world)
            line["fragment_roundtrip_with_reduce"] = frag_rt
    r.be.close()
    if args.config == 3 and not custom and not args.no_extra:
        # BASELINE configs[4] in the same launch: one top-level group of 128 sub-groups x 256
        # wtosc->filter12->panmix voices per rank - at --gpus 8 that IS configs[4] - the ranks' root-bus
        # partials summed by the same reduce, the whole job's audio gated on the oracle golden of this rank count
        c4 = dict(CONFIGS[4])
        r4, dt4 = timed_job(c4, max(8, args.steps // 2), args.warmup)
        leaf4, all4, _ = r4.profile(8)
        if rank == 0:
            cmp4, ok4 = check_golden(r4, 2)
            if ok4 is False:
                raise SystemExit("bench.py: configs[4]: the ranks' summed render differs from the oracle golden")
            st4 = max(8, args.steps // 2)
            line["other_configs"] = {"configs[4]": {
                "workload": c4["label"] + f" x {world}" + (" = BASELINE configs[4]" if world == 8 else ""),
                "value": float(c4["voices"]) * world * B * 64 * st4 / dt4, "ms_per_step": dt4 / st4 * 1e3,
                "voices_total": c4["voices"] * world, "parity_vs_golden": ok4,
                "golden_fragments_compared": min(B, 64) if cmp4 else 0,
                "kernel": "k_leaf_oscfiltpan", "avg_launch_ms": leaf4, "all_kernels_ms_per_step": all4}}
        r4.be.close()
    dist.destroy_process_group()
    if rank == 0 and not args.no_engine and args.config == 3 and not custom:
        # The north star's claim - 256k voices in realtime on 8 GPUs - as the ENGINE sees it: ONE engine state whose
        # voice tree the drop-in spreads over this run's N GPUs (A2AMD_DEVICES=N: top-level subtrees per GPU, the
        # root-bus partials exchanged per batch, root chain on device 0), a2_Run(4096) throughput and a2_Run(64)
        # p50 / p99 against the 1.333 ms budget, audio hash-compared with the engine's own CPU units.
        # (the other ranks have finished their timed work; their processes end while this runs)
        eil = engine_in_loop(cases=[("configs[4] shape over %d GPU(s), one engine state" % world, "FilterTree", 32768 * world)],
                             dropin_env={"A2AMD_DEVICES": str(world)})
        line["engine_in_loop"] = eil
        rt = eil.get("max_realtime_voices_one_engine_state", {})
        line["max_realtime_voices"] = {"n_gpus": world, "units": rt.get("units"), "units+walk": rt.get("units+walk"),
                                       "what": rt.get("what")}
    if rank == 0 and not args.no_cpu_baseline:
        # (one rank's share of the job on this box's host cores; the other ranks are done)
        line["cpu_baseline"] = cpu_baseline(cfg["voices"], cfg["chain"], cfg["groups"])
        line["cpu_baseline"]["sample"] += "; one rank's share of the job (%d of %d voices)" % (cfg["voices"], cfg["voices"] * world)
    # (what sat in libc's buffer goes where fd 1 points now - stderr - before stdout is put back)
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        emit(line)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py - voice-samples/sec of the MI355X voice-render path.

Workload (BASELINE.json configs[1]): 1 024 sustained voices, wtosc -> panmix,
48 kHz, fragment = 64 frames, stereo, per GPU.  One "step" = one batch of
--batch fragments (default 256 = 16384 frames = 341 ms of audio, an offline
a2_Run() buffer) rendered for all voices: upload happened before the timed
region (the command stream of sustained voices is empty and identical for every
batch, so the same uploaded batch is re-run; each run renders the NEXT 341 ms
of audio).

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns
its own voice subtrees (weak scaling: --voices per GPU); per step each rank
renders its subtrees into its partial of the root voice's inline bus, the
partials are summed with ONE RCCL reduce (int32, wrap-around sum => exact in
any order), and rank 0 runs the root chain (panmix must see the sum,
SURVEY.md 8e).

Prints one JSON line (rank 0).  Extra objects: "roofline" (the dominant kernel
against the HBM roofline, timed with HIP events on the launch stream inside
the timed region) and "cpu_baseline" (the compiled reference, or the C port,
timed on this host's cores on a bounded sample of the same workload).
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

BYTES_PER_VOICE_SAMPLE = {          # algorithmic HBM bytes
    "osc-pan": 504.0 / 64.0,        # 7.9 B: (152+96 B state) x2 + 8 B bus share per fragment
    "osc-filter-pan": 792.0 / 64.0,  # 12.4 B with filter12's 144 B state
    "osc2-pan": (504.0 + 2 * 152.0) / 64.0,
    # fm units: 64 B of operator state per operator + panmix, read + written, + bus share
    **{f"{k}-pan": ((64.0 * n + 96.0) * 2 + 8.0) / 64.0 for k, n in
       (("fm1", 1), ("fm2", 2), ("fm3", 3), ("fm4", 4), ("fm3p", 3), ("fm4p", 4), ("fm2r", 2), ("fm4r", 4))},
}
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s


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


def cpu_baseline(voices, chain, oracle_fragments=600):
    """Reference (oracle/_ref/ref_bench) if it travelled, else the C port."""
    program = {"osc-pan": "OscPan", "osc-filter-pan": "OscFilterPan", "osc2-pan": "Osc2Pan", "fm1-pan": "Fm1Pan", "fm2-pan": "Fm2Pan",
               "fm4-pan": "Fm4Pan"}.get(chain)
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    script = os.path.join(ROOT, "tests", "a2s", "bench.a2s")
    ncores = os.cpu_count() or 1
    if program and os.path.exists(exe):
        frags = {"osc-pan": 7500, "fm4-pan": 1000, "fm2-pan": 2500}.get(chain, 4000)   # a few seconds of one core
        res = {}
        for threads in sorted({1, min(ncores, 16)}):
            try:
                out = subprocess.run([exe, script, program, str(voices), str(frags), str(threads)],
                                     capture_output=True, text=True, timeout=180, check=True).stdout
                res[threads] = json.loads(out.strip().splitlines()[-1])
            except Exception as e:  # noqa: BLE001
                res[threads] = {"error": str(e)}
        ok = {t: r for t, r in res.items() if "voice_samples_per_s" in r}
        if ok:
            best = max(ok, key=lambda t: ok[t]["voice_samples_per_s"])
            return {"value": ok[best]["voice_samples_per_s"], "unit": "voice-samples/s", "cores": best,
                    "kind": "reference",
                    "single_thread_value": ok.get(1, {}).get("voice_samples_per_s"),
                    "host_cores": ncores,
                    "sample": f"{voices} voices {chain}, {frags} fragments of 64 frames "
                              f"({frags * 64 / 48000:.1f} s of audio), reference engine via a2_Run(64), "
                              f"{best} thread(s) = {best} independent engine states"}
    # the C restatement, driven through the same call protocol
    from audiality2_amd import synth
    from audiality2_amd.replay import Backend
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liba2oracle.so"))
    be = Backend(lib, "a2o_", 48000, synth.basepitch_for(48000), 2)
    sc = synth.Scene(be)
    sc.root()
    sc.add_voices(voices, chain=chain)
    sc.run(1, batch=1)
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
            "sample": f"{voices} voices {chain}, {oracle_fragments} fragments of 64 frames, "
                      f"C restatement (oracle/a2o.c), 1 thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--voices", type=int, default=1024, help="voices per GPU")
    ap.add_argument("--chain", default="osc-pan", choices=sorted(BYTES_PER_VOICE_SAMPLE))
    ap.add_argument("--batch", type=int, default=256, help="fragments per step (256 = 341 ms of audio)")
    ap.add_argument("--groups", type=int, default=0,
                    help="put the voices under this many inline->fbdelay->fbdelay group voices (config 4 shape)")
    ap.add_argument("--reduce-group", type=int, default=8,
                    help="N>1: steps whose root-bus partials are summed by one collective")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()

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
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    import audiality2_amd
    from audiality2_amd import shard, synth
    from audiality2_amd.replay import Backend

    B = args.batch
    # N=1: the library launches on its own stream and replays the steady-state
    # step from a hipGraph.  N>1: everything (kernels and the RCCL reduce) is
    # ordered on one torch stream.
    tstream = torch.cuda.Stream(device=local_rank) if multi else None
    if tstream is not None:
        torch.cuda.set_stream(tstream)
    be = audiality2_amd.open_backend(48000, None, 2, device=local_rank, max_batch=B,
                                     stream=tstream.cuda_stream if tstream is not None else None)
    lib = be.lib
    lib.a2amd_fragment_repeat.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
    lib.a2amd_get_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(Stats)]
    lib.a2amd_set_profiling.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.a2amd_rootbus.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
    lib.a2amd_replay.argtypes = [ctypes.c_void_p, ctypes.c_uint]

    # ---- build the voice tree; every rank plays different voices -----------
    sc = synth.Scene(be)
    sc.root()
    sc.nvoices = shard.voice_range(rank, args.voices)[0]
    if args.groups:
        per = args.voices // args.groups
        for gi in range(args.groups):
            grp = sc.add_group()
            sc.add_voices(per, chain=args.chain, group=grp, total=args.voices * world)
    else:
        sc.add_voices(args.voices, chain=args.chain, total=args.voices * world)

    def repeat(n):
        rc = lib.a2amd_fragment_repeat(be.ctx, 64, n)
        if rc:
            raise RuntimeError(be._err(be.ctx))

    # first batch: the explicit engine walk (voice inits + control writes)
    sc.walk(64)
    repeat(B - 1)
    first = be.render(B * 64)

    # parity gate: per-fragment hashes of the first 8 fragments against the
    # committed golden of this exact workload (tests/golden/, rendered by the CPU
    # oracle in the test-suite); other workloads are gated by tests/ only
    parity = None
    gold = os.path.join(ROOT, "tests", "golden", "bench_default_first8.hash.npy")
    if (rank == 0 and world == 1 and not args.no_parity and args.voices == 1024 and args.chain == "osc-pan"
            and not args.groups and B >= 8 and os.path.exists(gold)):
        parity = bool(np.array_equal(fnv1a_fragments(first[:, :8 * 64]), np.load(gold)))
        if not parity:
            raise SystemExit("bench.py: GPU render differs from the golden render; refusing to report a number")

    # ---- steady state: record once, upload once, re-run ----------------------
    UP, SUB, ROOTP, RB, KEEP = 4, 1, 2, 8, 16
    repeat(B)
    be.render(0, phases=UP | KEEP)

    rootbus = None
    if multi:
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_uint64()
        if lib.a2amd_rootbus(be.ctx, ctypes.byref(ptr), ctypes.byref(nbytes)):
            raise RuntimeError(be._err(be.ctx))

        rootbus = shard.wrap_device_bus(ptr.value, nbytes.value, torch.device("cuda", local_rank))

    # N>1: the root-bus sums of `--reduce-group` steps travel in one collective that
    # overlaps the next group's subtree kernels (audiality2_amd/shard.py)
    pipe = None
    if multi:
        lib.a2amd_rootbus_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        render = be._render                 # (the raw entry point: no output arrays in the step loop)

        def phase(ph):
            if render(be.ctx, ph | KEEP, None, 0) < 0:
                raise RuntimeError(be._err(be.ctx))

        def copy_fn(ptr, to_stage):
            if lib.a2amd_rootbus_copy(be.ctx, ptr, to_stage):
                raise RuntimeError(be._err(be.ctx))

        pipe = shard.GroupedRootReduce(rootbus, lambda: phase(SUB), lambda: phase(ROOTP), rank,
                                       group=args.reduce_group, copy_fn=copy_fn)

    def run(nsteps):
        if not multi:
            if lib.a2amd_replay(be.ctx, nsteps):
                raise RuntimeError(be._err(be.ctx))
            return
        pipe.run(nsteps)

    # barrier = a (pre-warmed) 1-element all-reduce every rank must join, with the
    # device idle on both sides; dist.barrier() itself costs tens of ms on first use
    token = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", local_rank)) if multi else None

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.all_reduce(token)
            torch.cuda.synchronize()

    fence()                 # first use builds the RCCL communicator: keep it out of the timing
    run(args.warmup)
    fence()
    fence()
    import gc
    gc.collect()
    gc.disable()            # a full collection mid-loop costs tens of ms of host time
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # kernel durations: the same step, every launch bracketed by HIP events on
    # the launch stream (profiling disables the graph replay; N>1: one step at a
    # time, so that every step's events pair up)
    def run_serial(nsteps):
        for _ in range(nsteps):
            be.render(0, phases=SUB | KEEP)
            shard.reduce_root_bus(rootbus, dst=0)
            if rank == 0:
                be.render(0, phases=ROOTP | KEEP)

    nprof = min(args.steps, 64)
    lib.a2amd_set_profiling(be.ctx, 1)
    (run_serial if multi else run)(nprof)
    st = Stats()
    lib.a2amd_get_stats(be.ctx, ctypes.byref(st))
    lib.a2amd_set_profiling(be.ctx, 0)
    last = be.render(B * 64, phases=RB) if rank == 0 else None
    if rank != 0:
        be.render(0, phases=ROOTP)      # close the batch on the other ranks

    if rank == 0:
        vs_per_step = float(args.voices) * world * B * 64
        value = vs_per_step * args.steps / dt
        leaf_ms = st.timed_leaf_ms / max(st.timed_batches, 1)
        traffic = None      # HBM bytes per launch from PMC counters (profiles/README.md), if collected
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                traffic = json.load(f).get(f"{args.chain}/{args.voices}/{B}", {}).get("hbm_bytes_per_launch")
        except OSError:
            pass
        bpvs = BYTES_PER_VOICE_SAMPLE[args.chain]
        achieved = bpvs * args.voices * B * 64 / (leaf_ms * 1e-3) / 1e9 if leaf_ms > 0 else None
        line = {
            "metric": "voice-samples/sec (max realtime voices @48kHz reported alongside)",
            "value": value, "unit": "voice-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{args.voices} voices/GPU, {args.chain} (wtosc->panmix = BASELINE "
                                   f"configs[1]), 48 kHz, fragment=64, stereo",
                       "voices_per_gpu": args.voices, "chain": args.chain, "groups": args.groups,
                       "fragments_per_step": B,
                       "samplerate": 48000, "sharding": "voice subtrees per GPU + 1 RCCL int32 reduce "
                       "of the root bus per step" if multi else "single GPU"},
            "realtime_factor": value / (args.voices * world * 48000.0),
            "max_realtime_voices_at_this_rate": int(value / 48000.0),
            "parity_vs_golden": parity,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bpvs * args.voices * B * 64,
                         "kernel": {"osc-pan": "k_leaf_oscpan", "osc-filter-pan": "k_leaf_oscfiltpan", "osc2-pan": "k_leaf_osc2pan"}.get(
                             args.chain, "k_leaf_fmpan" if args.chain.startswith("fm") else "k_voices (leaf launch)"),
                         "avg_launch_ms": leaf_ms,
                         "timing": "HIP events on the launch stream around every launch, separate pass of "
                                   f"{nprof} steps right after the timed region (graph replay off)",
                         "launches_timed": int(st.timed_batches),
                         "algorithmic_bytes_per_voice_sample": bpvs,
                         "all_kernels_ms_per_step": st.timed_all_ms / max(st.timed_batches, 1)},
            "output_check": {"peak": int(np.abs(last).max()), "nonzero": bool(last.any())},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.voices, args.chain)
    be.close()
    if multi:
        dist.destroy_process_group()
    # The contract line is the LAST thing on stdout: RCCL writes a version banner
    # through C stdio, which sits in libc's buffer until it is flushed.
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

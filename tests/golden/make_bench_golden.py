#!/usr/bin/env python3
"""Golden check values for bench.py's parity gate and the full-size GPU tests:
per-fragment FNV-1a 64 hashes of the first STEPS x 256 fragments of each
single-GPU BASELINE config at FULL size (configs[1] 1 024 x wtosc->panmix,
configs[2] 16 384 x wtosc->filter12->panmix, configs[3] 65 536 x 2xwtosc->panmix
under 256 inline->fbdelay->fbdelay groups with fmtest4's delay settings),
rendered by the CPU oracle from the scene bench.py builds (bench.build_scene).

  python tests/golden/make_bench_golden.py [1 2 3 4]    (configs[3]: ~5 min of one core)

configs[4] (4): 64 fragments of the whole job's audio for 1, 2, 4 and 8 top-level groups of
128 sub-groups x 256 wtosc->filter12->panmix voices (32 768 ... 262 144 voices).
30: configs[3] as `bench.py --gpus N` plays it, the whole job's audio at N = 2, 4, 8 (64 fragments).

bench.py and tests/test_gpu_parity.py only read the resulting data files;
tests/test_oracle_cpu.py re-renders the head of each with the oracle.  8 steps =
2.73 s of audio: the longest delay tap of configs[3] (1 378.75 ms) has been
feeding back for more than a second by the end."""
import ctypes
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

STEPS, B = 8, 256


CFG4_FRAGMENTS = 64      # configs[4]: 262 144 voices; 64 fragments are ~25 s of oracle time
CFG4_TOTALS = (32768, 65536, 131072, 262144)     # the whole job at 1, 2, 4, 8 GPUs (bench.py --config 4)


def render(voices, chain, groups, fragments, progress=False, tree=0, private=None):
    """int32 [2, fragments * 64] from the oracle, for bench.py's scene."""
    import bench
    from audiality2_amd import synth
    from audiality2_amd.replay import Backend
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liba2oracle.so"))
    lib.a2o_fragment_repeat.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
    be = Backend(lib, "a2o_", 48000, synth.basepitch_for(48000), 2, max_batch=64)
    sc = bench.build_scene(be, voices, chain, groups, tree=tree, private=private)
    outs = []
    sc.walk(64)
    done, pending = 0, 1
    t0 = time.time()
    while done < fragments:
        n = min(64 - pending, fragments - done - pending)
        if n:
            assert lib.a2o_fragment_repeat(be.ctx, 64, n) == 0
        outs.append(be.render((pending + n) * 64).copy())
        done += pending + n
        pending = 0
        if progress:
            print(f"\r{done}/{fragments} fragments, {time.time() - t0:.0f} s", end="", flush=True)
    if progress:
        print()
    be.close()
    return np.concatenate(outs, axis=1)


if __name__ == "__main__":
    import bench
    from conftest import fnv1a_fragments
    # --steps N: that many steps of 256 fragments for configs[1..3] (default 8; bench.py's default run issues 64:
    # round 4 keeps goldens that cover every one of them)
    argv = list(sys.argv[1:])
    if "--steps" in argv:
        k = argv.index("--steps")
        STEPS = int(argv[k + 1])
        del argv[k:k + 2]
    which = [int(a) for a in argv] or [1, 2, 3, 4]
    if 4 in which:
        # BASELINE configs[4]: top-level groups of 128 sub-groups x 256
        # wtosc->filter12->panmix voices, one per GPU; the whole job's audio at 1, 2, 4 and 8 GPUs
        which.remove(4)
        cfg = bench.CONFIGS[4]
        for total in CFG4_TOTALS:
            pcm = render(total, cfg["chain"], 0, CFG4_FRAGMENTS, progress=True, tree=cfg["tree"])
            path = bench.golden_path(total, cfg["chain"], 0, cfg["tree"])
            np.save(path, fnv1a_fragments(pcm))
            np.save(path.replace(".hash.npy", ".head.npy"), pcm[:, :256])
            print("written", path, "peak", int(np.abs(pcm).max()))
    if 30 in which:
        # configs[3] as bench.py --gpus N plays it (weak scaling: every rank its own 65 536 voices under its own 256
        # groups, pitches spread over the whole job): the whole job's audio at 2, 4 and 8 GPUs, 64 fragments each
        which.remove(30)
        cfg = bench.CONFIGS[3]
        for world in (2, 4, 8):
            pcm = render(cfg["voices"] * world, cfg["chain"], cfg["groups"] * world, CFG4_FRAGMENTS, progress=True)
            path = bench.golden_path(cfg["voices"] * world, cfg["chain"], cfg["groups"] * world)
            np.save(path, fnv1a_fragments(pcm))
            np.save(path.replace(".hash.npy", ".head.npy"), pcm[:, :256])
            print("written", path, "peak", int(np.abs(pcm).max()))
    for i in which:
        cfg = bench.CONFIGS[i]
        # (config 5, the private-wave scene: two steps)
        pcm = render(cfg["voices"], cfg["chain"], cfg["groups"], (2 if cfg.get("private") else STEPS) * B, progress=True,
                     private=cfg.get("private"))
        path = bench.golden_path(cfg["voices"], cfg["chain"], cfg["groups"], private=cfg.get("private"))
        np.save(path, fnv1a_fragments(pcm))
        # the head of the audio itself, for a readable diff when a hash differs
        np.save(path.replace(".hash.npy", ".head.npy"), pcm[:, :256])
        print("written", path, "peak", int(np.abs(pcm).max()))

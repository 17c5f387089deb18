#!/usr/bin/env python3
"""Golden check value for bench.py's parity gate: per-fragment FNV-1a 64 of the
first 8 fragments of the default bench workload (1 024 voices wtosc->panmix
under the root voice, synth.Scene parameters), rendered by the CPU oracle.
bench.py only reads the resulting data file; tests/test_oracle_cpu.py checks
that the file still matches the oracle."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def render(voices=1024, chain="osc-pan", fragments=8):
    from audiality2_amd import synth
    from audiality2_amd.replay import Backend
    be = Backend(ctypes.CDLL(os.path.join(ROOT, "oracle", "liba2oracle.so")), "a2o_", 48000,
                 synth.basepitch_for(48000), 2)
    sc = synth.Scene(be)
    sc.root()
    sc.add_voices(voices, chain=chain, total=voices)
    out = sc.run(fragments, batch=fragments)
    be.close()
    return out


if __name__ == "__main__":
    from conftest import fnv1a_fragments
    np.save(os.path.join(HERE, "bench_default_first8.hash.npy"), fnv1a_fragments(render()))
    print("written")

"""The N>1 path on CPU: two processes (gloo), each renders its shard of the
voices with the CPU oracle, the root-bus partials are summed with the same
collective bench.py uses (audiality2_amd.shard.reduce_root_bus), and the sum
equals the single-process render bit for bit."""
import ctypes
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ORACLE_SO, ROOT

VOICES_PER_RANK, FRAGS = 48, 12


def render_shard(rank, world, chain):
    """Leaves of this rank straight into the bus (= the root's inline bus)."""
    import sys
    sys.path.insert(0, ROOT)
    from audiality2_amd import shard, synth
    from audiality2_amd.replay import Backend
    be = Backend(ctypes.CDLL(ORACLE_SO), "a2o_", 48000, synth.basepitch_for(48000), 2)
    sc = synth.Scene(be)
    lo, hi = shard.voice_range(rank, VOICES_PER_RANK)
    sc.nvoices = lo
    sc.add_voices(hi - lo, chain=chain, total=VOICES_PER_RANK * world)
    out = sc.run(FRAGS, batch=4)
    be.close()
    return out


def worker(rank, world, port, chain, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, ROOT)
    from audiality2_amd import shard
    partial = torch.from_numpy(render_shard(rank, world, chain).copy())
    shard.reduce_root_bus(partial, dst=0)
    if rank == 0:
        q.put(partial.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(world, chain):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, chain, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return out


def single_process(world, chain, with_root):
    import sys
    sys.path.insert(0, ROOT)
    from audiality2_amd import synth
    from audiality2_amd.replay import Backend
    be = Backend(ctypes.CDLL(ORACLE_SO), "a2o_", 48000, synth.basepitch_for(48000), 2)
    sc = synth.Scene(be)
    if with_root:
        sc.root()
    sc.add_voices(VOICES_PER_RANK * world, chain=chain)
    out = sc.run(FRAGS, batch=4)
    be.close()
    return out


def test_two_rank_shard_sum_is_exact(oracle_lib):
    for chain in ("osc-pan", "osc-filter-pan"):
        summed = run_world(2, chain)
        whole = single_process(2, chain, with_root=False)
        assert np.array_equal(summed, whole), chain
        # the root chain (vol 1.0, pan 0) is the identity on the summed bus, so the
        # full tree with its root voice renders the same audio
        assert np.array_equal(summed, single_process(2, chain, with_root=True)), chain


def test_int32_sum_wraps_like_the_bus():
    a = torch.tensor([2**31 - 1, -2**31, 5], dtype=torch.int32)
    b = torch.tensor([1, -1, -7], dtype=torch.int32)
    assert (a + b).tolist() == [-2**31, 2**31 - 1, -2]


def test_whole_job_scene_is_the_union_of_the_ranks_scenes(oracle_lib):
    """bench.py's N > 1 line gates the ranks' summed render on an oracle golden of ONE scene holding every rank's voices
    and groups (tests/golden/make_bench_golden.py 30: build_scene(voices x N, groups x N) with world = 1).  That is only
    the same audio if a rank's scene - its voices numbered from rank x voices, pitches spread over the whole job, under
    its own delay groups - is exactly that rank's slice of the big scene: checked here with the oracle at a small size,
    BASELINE configs[3]'s shape (two oscillators -> panmix under inline -> fbdelay -> fbdelay groups), root-bus partials
    summed as the reduce sums them."""
    import sys
    sys.path.insert(0, ROOT)
    from audiality2_amd import synth
    from audiality2_amd.replay import Backend
    voices, groups, world, frags = 64, 4, 2, 12

    def render(nvoices, ngroups, first, total):
        be = Backend(oracle_lib, "a2o_", 48000, synth.basepitch_for(48000), 2)
        sc = synth.Scene(be)
        sc.nvoices = first
        for _ in range(ngroups):
            grp = sc.add_group(preset="fmtest4")
            sc.add_voices(nvoices // ngroups, chain="osc2-pan", group=grp, total=total)
        out = sc.run(frags, batch=4)
        be.close()
        return out.astype(np.int64)

    parts = [render(voices, groups, r * voices, voices * world) for r in range(world)]
    whole = render(voices * world, groups * world, 0, voices * world)
    assert whole.any()
    summed = ((sum(parts) + (1 << 31)) % (1 << 32)) - (1 << 31)        # int32 wrap-around, as ncclSum(int32)
    assert np.array_equal(summed, whole)

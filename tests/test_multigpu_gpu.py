"""The N>1 step on real device memory: two processes share cuda:0 (the GPU box has
one GPU), each renders its shard of the voice subtrees with the HIP kernels
(phase SUBTREES), the partials of the root voice's inline bus - the device
memory a2amd_rootbus() names, viewed in place by torch - are summed with
audiality2_amd.shard.reduce_root_bus (gloo here, RCCL in bench.py: same call),
rank 0 runs the root chain on the sum (phase ROOT) and reads back.  The result
must be the single-process render of all the voices, bit for bit."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

VOICES_PER_RANK, FRAGS, BATCH = 160, 8, 4
UP, SUB, ROOTP, RB = 4, 1, 2, 8


def build_scene(be, lo, n, total, chain, groups):
    from audiality2_amd import synth
    sc = synth.Scene(be)
    sc.root()
    sc.nvoices = lo
    if groups:
        per = n // groups
        for _ in range(groups):
            g = sc.add_group()
            sc.add_voices(per, chain=chain, group=g, total=total)
    else:
        sc.add_voices(n, chain=chain, total=total)
    return sc


def worker(rank, world, port, chain, groups, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import audiality2_amd
    from audiality2_amd import shard
    torch.cuda.set_device(0)
    be = audiality2_amd.open_backend(48000, None, 2, device=0, max_batch=BATCH)
    lib = be.lib
    lib.a2amd_rootbus.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
    lo, hi = shard.voice_range(rank, VOICES_PER_RANK)
    sc = build_scene(be, lo, hi - lo, VOICES_PER_RANK * world, chain, groups)
    chunks = []
    for _ in range(FRAGS // BATCH):
        for _ in range(BATCH):
            sc.walk(64)
        be.render(0, phases=UP | SUB)
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_uint64()
        assert lib.a2amd_rootbus(be.ctx, ctypes.byref(ptr), ctypes.byref(nbytes)) == 0
        bus = shard.wrap_device_bus(ptr.value, nbytes.value, torch.device("cuda", 0))
        torch.cuda.synchronize()
        partial = bus.cpu()                       # gloo reduces host tensors; RCCL would take `bus` itself
        shard.reduce_root_bus(partial, dst=0)
        if rank == 0:
            bus.copy_(partial)
            torch.cuda.synchronize()
            chunks.append(be.render(BATCH * 64, phases=ROOTP | RB))
        else:
            be.render(0, phases=ROOTP)            # closes the batch; this rank's master bus is not used
    if rank == 0:
        q.put(np.concatenate(chunks, axis=1))
    dist.barrier()
    be.close()
    dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("chain,groups", [("osc-pan", 0), ("osc-filter-pan", 0), ("fm2-pan", 0), ("osc2-pan", 4)])
def test_two_rank_step_on_device_memory(chain, groups):
    import audiality2_amd
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, chain, groups, q)) for r in range(world)]
    for p in procs:
        p.start()
    sharded = q.get()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    # one process, all voices.  With groups each rank owns whole groups, so the
    # single process builds rank 0's groups and then rank 1's.
    be = audiality2_amd.open_backend(48000, None, 2, max_batch=BATCH)
    if groups:
        from audiality2_amd import synth
        sc = synth.Scene(be)
        sc.root()
        for r in range(world):
            sc.nvoices = r * VOICES_PER_RANK
            for _ in range(groups):
                g = sc.add_group()
                sc.add_voices(VOICES_PER_RANK // groups, chain=chain, group=g, total=VOICES_PER_RANK * world)
    else:
        sc = build_scene(be, 0, VOICES_PER_RANK * world, VOICES_PER_RANK * world, chain, 0)
    whole = sc.run(FRAGS, batch=BATCH)
    be.close()
    assert whole.any()
    assert np.array_equal(sharded, whole)


def pipeline_worker(rank, world, port, nsteps, group, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import audiality2_amd
    from audiality2_amd import shard
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream(device=0)
    torch.cuda.set_stream(stream)
    be = audiality2_amd.open_backend(48000, None, 2, device=0, max_batch=BATCH, stream=stream.cuda_stream)
    lib = be.lib
    lib.a2amd_rootbus.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
    lib.a2amd_fragment_repeat.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
    lo, hi = shard.voice_range(rank, VOICES_PER_RANK)
    sc = build_scene(be, lo, hi - lo, VOICES_PER_RANK * world, "osc-pan", 0)
    KEEP = 16
    sc.walk(64)                                   # the engine walk: voice set-up
    assert lib.a2amd_fragment_repeat(be.ctx, 64, BATCH - 1) == 0
    be.render(0, phases=UP | KEEP)                # ... uploaded once, then the batch re-runs
    ptr, nbytes = ctypes.c_void_p(), ctypes.c_uint64()
    assert lib.a2amd_rootbus(be.ctx, ctypes.byref(ptr), ctypes.byref(nbytes)) == 0
    bus = shard.wrap_device_bus(ptr.value, nbytes.value, torch.device("cuda", 0))
    chunks = []

    def reduce_fn(t):       # gloo has no reduce() for device tensors: all_reduce gives rank 0 the same sum
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    lib.a2amd_rootbus_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]

    def copy_fn(ptr, to_stage):         # (group 1 takes the torch copies instead)
        assert lib.a2amd_rootbus_copy(be.ctx, ptr, to_stage) == 0

    pipe = shard.GroupedRootReduce(bus, lambda: be.render(0, phases=SUB | KEEP),
                                   lambda: chunks.append(be.render(BATCH * 64, phases=ROOTP | RB | KEEP)),
                                   rank, group=group, reduce_fn=reduce_fn, copy_fn=copy_fn if group > 1 else None)
    pipe.run(nsteps)
    torch.cuda.synchronize()
    if rank == 0:
        q.put(np.concatenate(chunks, axis=1))
    dist.barrier()
    be.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("nsteps,group", [(5, 2), (7, 8), (4, 1)])
def test_grouped_overlapped_root_reduce_is_exact(nsteps, group):
    """bench.py's N>1 step loop (shard.GroupedRootReduce): the root-bus partials of
    `group` steps parked, summed by one asynchronous collective that overlaps the
    next group's subtree kernels, root chains replayed on the sums - against one
    process re-running the same kept batch."""
    import audiality2_amd
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=pipeline_worker, args=(r, world, port, nsteps, group, q)) for r in range(world)]
    for p in procs:
        p.start()
    sharded = q.get()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    be = audiality2_amd.open_backend(48000, None, 2, max_batch=BATCH)
    be.lib.a2amd_fragment_repeat.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
    sc = build_scene(be, 0, VOICES_PER_RANK * world, VOICES_PER_RANK * world, "osc-pan", 0)
    sc.walk(64)
    assert be.lib.a2amd_fragment_repeat(be.ctx, 64, BATCH - 1) == 0
    be.render(0, phases=UP | 16)
    whole = np.concatenate([be.render(BATCH * 64, phases=SUB | ROOTP | RB | 16) for _ in range(nsteps)], axis=1)
    be.close()
    assert whole.any() and sharded.shape == whole.shape
    assert np.array_equal(sharded, whole)

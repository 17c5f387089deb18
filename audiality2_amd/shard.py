"""Sharding voice subtrees across ranks (one process per GPU).

A voice only ever adds into its parent's bus with wrap-around int32 adds, so
sibling subtrees are independent and the only exchange is the sum of the root
voice's inline bus.  Rank r renders voices [r*n, (r+1)*n) of the
global numbering; `reduce_root_bus` sums the partials onto rank 0, which then
runs the root chain (its panmix multiply truncates, so it must see the sum).
"""
import torch
import torch.distributed as dist


def voice_range(rank, voices_per_rank):
    return rank * voices_per_rank, (rank + 1) * voices_per_rank


def reduce_root_bus(partial, dst=0):
    """ONE collective per batch: int32 SUM of the root-bus partials onto `dst`.
    Exact for any reduction order (addition mod 2^32)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(partial, dst=dst, op=dist.ReduceOp.SUM)
    return partial


def wrap_device_bus(ptr, nbytes, device):
    """A torch int32 view of the backend's root-bus device memory (no copy)."""
    class _Wrap:
        __cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<i4",
                                    "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Wrap(), device=device)


class GroupedRootReduce:
    """Steps of a sharded render with the root-bus exchange batched and overlapped.

    A step is: every rank renders its subtrees (`sub()`), the root-bus partials
    are summed onto rank 0, rank 0 runs the root chain on the sum (`root()`).
    Done one step at a time that is one small collective per step, serialised with
    the kernels.  Here `group` steps' partials are parked in a staging tensor and
    summed by ONE larger collective, issued asynchronously so that it overlaps the
    next group's subtree kernels (xGMI collectives are latency bound at this size:
    fewer, larger ones); rank 0 then replays the sums into the root bus and runs
    the root chains.  Every root chain still sees exactly the sum of its own
    step's partials, so the audio is the same, bit for bit.

    `sub` / `root` launch on the current torch stream; `rootbus` is the int32 view
    of the backend's root-bus device memory (shard.wrap_device_bus)."""

    def __init__(self, rootbus, sub, root, rank, group=8, reduce_fn=None, copy_fn=None):
        self.bus, self.sub, self.root, self.rank, self.group = rootbus, sub, root, rank, max(1, group)
        self.stage = [torch.empty((self.group, rootbus.numel()), dtype=torch.int32, device=rootbus.device)
                      for _ in range(2)]
        self.rows = [[st[j] for j in range(self.group)] for st in self.stage]
        self.reduce_fn = reduce_fn or (lambda t: dist.reduce(t, dst=0, op=dist.ReduceOp.SUM, async_op=True))
        # copy_fn(device pointer of a staging row, to_stage): the backend's own stream
        # copy (a2amd_rootbus_copy) costs less host time per step than a torch op
        if copy_fn is not None:
            ptrs = [[r.data_ptr() for r in rows] for rows in self.rows]
            self.park = lambda si, j: copy_fn(ptrs[si][j], 1)
            self.unpark = lambda si, j: copy_fn(ptrs[si][j], 0)
        else:
            self.park = lambda si, j: self.rows[si][j].copy_(self.bus)
            self.unpark = lambda si, j: self.bus.copy_(self.rows[si][j])

    def _finish(self, pending):
        work, si, count = pending
        if work is not None:
            work.wait()                 # the current stream waits for the collective; the host does not
        if self.rank == 0:
            for j in range(count):
                self.unpark(si, j)
                self.root()

    def run(self, nsteps):
        pending, gi, done = None, 0, 0
        multi = dist.is_initialized()       # (also with one rank: the forced-distributed bench run exercises the call)
        while done < nsteps:
            count, si = min(self.group, nsteps - done), gi & 1
            for j in range(count):
                self.sub()
                self.park(si, j)
            work = self.reduce_fn(self.stage[si][:count]) if multi else None
            if pending is not None:
                self._finish(pending)
            pending, gi, done = (work, si, count), gi + 1, done + count
        if pending is not None:
            self._finish(pending)

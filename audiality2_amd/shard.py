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

"""Multi-GPU sharding of a fragment batch: contiguous ranges per rank, replicated patterns and weight table, no
data-path collective; predictions are concatenated on the host in rank order.

Fragments are independent units in the reference (one output line per input record, order-preserving:
util/fasta2skm.py:77-82, opal.py:468-469), so this is the whole distributed design.
"""
import numpy as np

__all__ = ["shard_range", "gather_in_rank_order"]


def shard_range(n, rank, world):
    """Contiguous range [lo, hi) of rank `rank` out of `world`: [rank*n/world, (rank+1)*n/world)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return (rank * n) // world, ((rank + 1) * n) // world


def gather_in_rank_order(local: np.ndarray, dist=None, dst=0):
    """Concatenates every rank's 1-D array in rank order on rank `dst` (None elsewhere).  dist: an initialised
    torch.distributed module (gloo or nccl/RCCL) or None for a single process."""
    local = np.ascontiguousarray(local)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = [None] * world if rank == dst else None
    dist.gather_object(local, parts, dst=dst)     # host-side concat of a few bytes per fragment: not a data-path collective
    if rank != dst:
        return None
    return np.concatenate(parts) if parts else local

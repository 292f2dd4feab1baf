"""
Multi-GPU layout of the ray set (no counterpart in the single-process
reference).  Rays interact only through the 1-D flux profile
(lib/libprop.py:653-658) and the shared column, so the z-major ray array is cut
into contiguous shards, one per rank / GPU; the column is replicated and the
2 x (ngrid-2) flux profile is all-reduced (RCCL over xGMI, inside the C library)
once per RK stage.  This module is the host-side bookkeeping only.
"""
import numpy as np


def shard_bounds(n_total, world, rank):
    """Contiguous [lo, hi) of rank's shard; shards differ by at most 2 rays and
    every boundary is even so that the 16-byte (2-ray) device accesses of a shard
    stay aligned."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    pairs = (n_total + 1) // 2
    lo = 2 * ((pairs * rank) // world)
    hi = 2 * ((pairs * (rank + 1)) // world)
    return min(lo, n_total), min(hi, n_total)


def shard_state(state, world, rank):
    """Slice slots 0..8 (per-ray) of an 11-slot state; slots 9, 10 are replicated."""
    n = len(state[0])
    lo, hi = shard_bounds(n, world, rank)
    return [np.asarray(s)[lo:hi] for s in state[:9]] + [state[9], state[10]]


def shard_statics(statics, n_total, world, rank):
    lo, hi = shard_bounds(n_total, world, rank)
    return {k: (np.asarray(v)[lo:hi] if np.ndim(v) == 1 and len(v) == n_total else v)
            for k, v in statics.items()}


def exchange_unique_id(dist, rank, make_id):
    """Rank 0 creates the 128-byte RCCL unique id (make_id()), everyone receives
    it through the already initialised torch.distributed group (any backend)."""
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]

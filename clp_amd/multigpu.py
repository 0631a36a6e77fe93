"""Multi-GPU attach: one process per GPU (torch.distributed launches and rendezvous), RCCL inside
the engine for the per-pivot exchange.

Protocol: structural columns are split into `world` contiguous,
256-aligned ranges.  Each rank prices, compacts, dual-updates and flip-tests only its own range (the
rows are replicated); per pivot the ranks all-gather (RCCL over xGMI) their candidate lists
{sequence, alpha, dj, range} + {count, min ratio} before the ratio test and their bound-flip records
after it -- a few tens of KB -- and everything else (CHUZR, solves, ratio test on the merged list,
basis update) runs replicated and deterministic, so the ranks stay in lock step.  Rank-major
concatenation of the lists is the single-GPU order, so the pivot sequence is the single-GPU one.  The
ncclUniqueId is created by rank 0 inside the engine and broadcast here through torch.distributed (any
backend; gloo on CPU for the tests).
"""
from __future__ import annotations

import ctypes as C

from .sharding import column_ranges


def broadcast_unique_id(make_id, rank: int, world: int) -> bytes:
    """rank 0 calls make_id() -> 128 bytes; every rank returns the same bytes."""
    import torch.distributed as dist

    payload = [make_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(payload, src=0)
    uid = payload[0]
    if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
        raise RuntimeError("bad ncclUniqueId broadcast")
    return bytes(uid)


def attach_communicator(engine, rank: int, world: int):
    """Give `engine` (a ClpGpuSimplex with a loaded problem) its shard and its RCCL communicator."""
    from . import engine as _engine

    import os

    if world <= 1 and not os.environ.get("CLPGPU_FORCE_COMM"):
        return column_ranges(engine.n, 1)[0]
    lib = _engine.lib()

    def make_id():
        buf = C.create_string_buffer(128)
        if lib.clpgpu_comm_unique_id(buf) != 0:
            raise RuntimeError("clpgpu_comm_unique_id failed (RCCL not loadable)")
        return buf.raw

    uid = broadcast_unique_id(make_id, rank, world)
    rc = lib.clpgpu_comm_init(engine._h, int(rank), int(world), uid)
    if rc != 0:
        raise RuntimeError(f"clpgpu_comm_init failed ({rc}): {lib.clpgpu_last_error(engine._h).decode()}")
    return column_ranges(engine.n, world)[rank]

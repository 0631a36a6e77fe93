"""CPU test of the N > 1 protocol with world_size 2 over gloo: the unique-id broadcast, the shard
ranges, and the merge rule -- per-rank fused pricing of a column range (oracle standing in for the
kernel), gathered rank-major, must equal the single-rank result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    import torch.distributed as dist

    from clp_amd import problems as P
    from clp_amd.multigpu import broadcast_unique_id
    from clp_amd.sharding import column_ranges, merge_candidates
    from oracle.oracle import OracleSimplex

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        uid = broadcast_unique_id(lambda: bytes(range(128)), rank, world)
        assert uid == bytes(range(128))
        lp = P.sparse_lp(300, 1500, 6, seed=9)
        o = OracleSimplex(lp)
        rng = np.random.default_rng(4)  # same inputs on every rank
        m, n = lp.m, lp.n
        idx = np.sort(rng.choice(m, 40, replace=False)).astype(np.int32)
        val = rng.standard_normal(40)
        status = rng.choice([1, 2, 3], size=n + m, p=[0.2, 0.3, 0.5]).astype(np.uint8)
        dj = np.where((status & 3) == 2, -1.0, 1.0) * rng.uniform(0, 2, n + m)
        a, b = column_ranges(n, world)[rank]
        st = status.copy()
        st[:a] = 1
        st[b:n] = 1
        if rank:
            st[n:] = 1  # the slack part is priced once
        mine = o.price_row_fused(idx, val, st, dj)
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        merged = merge_candidates(parts)
        full = o.price_row_fused(idx, val, status, dj)
        ok = all(np.array_equal(x, y) for x, y in zip(merged[:4], full[:4])) and merged[4] == full[4]
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_pricing_protocol(built):
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}

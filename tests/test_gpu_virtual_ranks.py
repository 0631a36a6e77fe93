"""Column-sharded engine with MORE THAN ONE RANK on a one-GPU box: N loopback ranks -- N contexts
of this process, one host thread and stream each -- run the sharded chain with real rank offsets
(k_shard_pack_cands / k_shard_merge_cands / k_shard_pack_flips / k_shard_merge_flips, reduced costs owned by the
rank that owns the column); what RCCL's all-gathers carry between GPUs travels by device-to-device copies
(include/clpgpu.h, clpgpu_virtual_*).  The sharded runs must reproduce the unsharded run: same pivots on every
rank, same solution.  Reference shape: AbcSimplexDual.cpp:1450-1528 (dualColumn2First per chunk), :1623-1634
(combine), ClpPackedMatrix.cpp:1823-1854 (ABOCA_LITE column chunks)."""
import numpy as np
import pytest

from clp_amd import problems as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(built):
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clp_amd import engine

    return engine


def configure(e, **opts):
    e.set_option("pivot_rule", 1)
    e.set_option("max_pivots", 0)
    e.set_option("factor_mode", 0)
    for k, v in opts.items():
        e.set_option(k, v)


def same_pivots(a, b, n):
    return np.array_equal(a["sequenceIn"][:n], b["sequenceIn"][:n]) and np.array_equal(a["sequenceOut"][:n], b["sequenceOut"][:n]) \
        and np.array_equal(a["pivotRow"][:n], b["pivotRow"][:n])


@pytest.mark.parametrize("nranks,mode", [(2, 2), (4, 2), (8, 2), (4, 1)])
def test_virtual_ranks_match_unsharded(gpu, nranks, mode):
    lp = P.sparse_lp(2000, 9000, 12, seed=13)
    base = gpu.ClpGpuSimplex(0).loadProblem(lp)
    configure(base)
    assert base.dual() == 0
    # (exchange buffers large enough for this LP's candidate lists -- 4000 and more per pivot: an overflow would send
    # the run to the dense row exchange after an extra resync, a different refactorization schedule; that path has
    # its own test below)
    vr = gpu.VirtualRanks(lp, nranks, configure=lambda e: configure(e, comm_mode=mode, shard_cand_cap=16384, shard_flip_cap=4096))
    assert vr.dual_steps(-1) == [0] * nranks
    ref = base.pivotLog()
    for r, e in enumerate(vr.engines):
        log = e.pivotLog()
        assert len(log) == len(ref), f"rank {r}"
        assert same_pivots(log, ref, len(ref)), f"rank {r}"
        assert e.objectiveValue() == base.objectiveValue()
        assert np.array_equal(e.solution(), base.solution())


@pytest.mark.parametrize("nranks", [2, 4])
def test_virtual_ranks_in_lu_mode_match_unsharded(gpu, nranks):
    """The sharded chain with the LU factorization (sparse front + dense tail + eta file, from a nucleus of 256 on) instead of the
    explicit inverse: factorization, solves and eta file are replicated row-space work, so every rank must make the pivots of the
    unsharded LU-mode engine and end on the same solution bits."""
    lp = P.sparse_lp(2000, 9000, 12, seed=13)

    def lu(e, **opts):
        configure(e, **opts)
        e.set_option("factor_mode", -1)
        e.set_option("lu_min_k", 256)

    base = gpu.ClpGpuSimplex(0).loadProblem(lp)
    lu(base)
    assert base.dual() == 0 and base.stats()["lu_factorizations"] > 0
    vr = gpu.VirtualRanks(lp, nranks, configure=lambda e: lu(e, comm_mode=2, shard_cand_cap=16384, shard_flip_cap=4096))
    assert vr.dual_steps(-1) == [0] * nranks
    ref = base.pivotLog()
    for r, e in enumerate(vr.engines):
        assert e.stats()["lu_factorizations"] > 0, f"rank {r} never factorized in LU mode"
        log = e.pivotLog()
        assert len(log) == len(ref), f"rank {r}"
        assert same_pivots(log, ref, len(ref)), f"rank {r}"
        assert e.objectiveValue() == base.objectiveValue()
        assert np.array_equal(e.solution(), base.solution())


@pytest.mark.parametrize("grow,lu", [(1, 0), (0, 0), (0, 1)])
def test_virtual_ranks_overflow_agrees(gpu, grow, lu):
    """exchange buffers far too small: every rank sees the overflow at the same pivot.  Default (shard_grow 1): the buffers grow
    to the shard's size and the list exchange carries on; shard_grow 0: all ranks fall back to the dense row-slice exchange
    together -- also in LU mode, where the windowed SELL copy moves its window base with the fall-back.
    Either way every rank makes the same pivots and reaches the unsharded optimum."""
    lp = P.sparse_lp(2000, 9000, 12, seed=13)

    def conf(e, **opts):
        configure(e, **opts)
        if lu:
            e.set_option("factor_mode", -1)
            e.set_option("lu_min_k", 256)

    base = gpu.ClpGpuSimplex(0).loadProblem(lp)
    conf(base)
    assert base.dual() == 0
    vr = gpu.VirtualRanks(lp, 4, configure=lambda e: conf(e, shard_cand_cap=4, shard_flip_cap=2, shard_grow=grow))
    assert vr.dual_steps(-1) == [0, 0, 0, 0]
    its = {e.numberIterations() for e in vr.engines}
    assert len(its) == 1
    for e in vr.engines:
        st = e.stats()
        assert st["comm_mode"] == (2 if grow else 1), st["comm_mode"]
        assert (st["shard_cand_cap"] >= 2000) == bool(grow)
        assert abs(e.objectiveValue() - base.objectiveValue()) <= 1e-9 * abs(base.objectiveValue())


@pytest.mark.parametrize("form", ["sell", "lds"])
def test_virtual_ranks_full_size_from_the_mature_basis(gpu, form):
    """config 5 in miniature, in the regime a solve of this LP lives in: the 50 000 x 200 000 LP of config 4 warm-started from the
    committed mature basis (nucleus 10 514, LU mode, ~10^5 candidates per pivot = 12 500 per rank), columns sharded over 8
    loopback ranks, 200 pivots against the unsharded engine: identical pivots on every rank, identical solution bits, the
    list exchange still on (buffers sized for the shard: no overflow, no extra resync)."""
    import os

    lp = P.sparse_lp()
    status = (np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "basis_sparse_30000.npy")) & 7).astype(np.uint8)

    def conf(e, **opts):
        e.set_option("pivot_rule", 1)
        e.set_option("max_pivots", 0)
        for k, v in opts.items():
            e.set_option(k, v)
        e.setStatusArray(status)

    base = gpu.ClpGpuSimplex(0).loadProblem(lp)
    conf(base)
    assert base.dual_steps(200) == -1
    assert base.stats()["lu_active"] == 1
    # the ranks' shards are 98 windows wide: below the width the LDS pricing form is laid out for by default ("sell": the L2-gather form,
    # what a sharded run takes at this width); "lds": the shards laid out for the LDS form all the same (price_lds_min_windows 1) and
    # priced with it on every pivot (price_lds 2) -- 25 workgroups of k_price_lds per rank, the same bits
    def pre(e):
        if form == "lds":
            e.set_option("price_lds_min_windows", 1)
            e.set_option("price_lds", 2)
        else:
            e.set_option("price_lds", 0)

    vr = gpu.VirtualRanks(lp, 8, configure=lambda e: conf(e, shard_cand_cap=32768, shard_flip_cap=8192), preconfigure=pre)
    assert vr.dual_steps(200) == [-1] * 8
    ref = base.pivotLog()
    for r, e in enumerate(vr.engines):
        st = e.stats()
        assert st["lu_active"] == 1 and st["comm_mode"] == 2, f"rank {r}"
        assert st["price_form"] == (1 if form == "lds" else 0), f"rank {r}"
        assert same_pivots(e.pivotLog(), ref, 200), f"rank {r}"
        assert np.array_equal(e.solution(), base.solution())

"""GPU parity test of the "objective going backwards" restore (src/ClpSimplexDual.cpp:5326-5488) and of the ClpSimplexProgress
bookkeeping around it, through the C ABI against the CPU oracle.  Fault injection on both sides (option debug_backwards_at, see
tests/test_oracle_progress.py): the engine goes back to the basis of the last good status check -- statuses and nonbasic values
from the host snapshot, a re-factorization, original bounds + fake bounds again, weights restored by sequence (saveWeights 4),
forceFactorization 1 -- and from there has to make the oracle's pivots."""
import numpy as np
import pytest

from clp_amd import problems as P

pytestmark = pytest.mark.gpu
RTOL = 1e-8


@pytest.fixture(scope="module")
def gpu_cls(built):
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clp_amd.engine import ClpGpuSimplex

    return ClpGpuSimplex


def both(gpu_cls, lp, rule, **opts):
    from oracle.oracle import OracleSimplex

    g = gpu_cls().loadProblem(lp)
    o = OracleSimplex(lp)
    for s in (g, o):
        s.set_option("pivot_rule", rule)
        for k, v in opts.items():
            s.set_option(k, v)
    return g, g.dual(), o, o.dual()


@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("maker,args,at", [("sparse_lp", (300, 1200, 8, 11), 300), ("netlib_shaped_lp", (400, 1500, 9000, 3), 250)])
def test_backwards_restore_identical_pivot_sequence(gpu_cls, maker, args, at, rule):
    lp = getattr(P, maker)(*args)
    g, sg, o, so = both(gpu_cls, lp, rule, debug_backwards_at=at)
    assert sg == so == 0
    assert g.stats()["backwards_restores"] == o.backwards == 1
    lg, lo = g.pivotLog(), o.pivot_log()
    assert len(lg) == len(lo)
    assert np.array_equal(lg["sequenceIn"], lo["sequenceIn"]) and np.array_equal(lg["sequenceOut"], lo["sequenceOut"])
    assert g.stats()["refactorizations"] == o.refactorizations
    assert abs(g.objectiveValue() - o.objective) <= RTOL * (1 + abs(o.objective))
    assert float(np.max(np.abs(g.solution() - o.solution()) / (1.0 + np.abs(o.solution())))) < RTOL


@pytest.mark.parametrize("rule", [0, 1])
def test_backwards_restore_on_a_degenerate_instance(gpu_cls, rule):
    g, sg, o, so = both(gpu_cls, P.nqueens(20), rule, debug_backwards_at=200)
    assert sg == so == 0
    assert g.stats()["backwards_restores"] == o.backwards == 1
    assert abs(g.objectiveValue() + 20.0) < 1e-7


def test_no_restore_and_no_loop_flag_on_a_healthy_solve(gpu_cls):
    g, sg, o, so = both(gpu_cls, P.sparse_lp(300, 1200, 8, 11), 1)
    assert sg == so == 0
    st = g.stats()
    assert st["backwards_restores"] == 0 and st["loop_flags"] == 0 and o.backwards == 0 and o.loop_flags == 0
    assert np.array_equal(g.pivotLog()["sequenceIn"], o.pivot_log()["sequenceIn"])


@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("maker,args,at", [("sparse_lp", (300, 1200, 8, 11), 300), ("netlib_shaped_lp", (400, 1500, 9000, 3), 250)])
def test_bad_accuracy_restore_identical_pivot_sequence(gpu_cls, maker, args, at, rule):
    """Errors beyond 1e15 after a refactorization are treated as a singular basis (src/ClpSimplexDual.cpp:5237-5318): previous
    basis back, the leaving variable flagged, a refactorization after every pivot.  Fault injection on both sides (option
    debug_bad_accuracy_at); from there on the engine has to make the oracle's pivots."""
    lp = getattr(P, maker)(*args)
    g, sg, o, so = both(gpu_cls, lp, rule, debug_bad_accuracy_at=at)
    assert sg == so == 0
    assert g.stats()["accuracy_restores"] == o.accuracy_restores == 1
    lg, lo = g.pivotLog(), o.pivot_log()
    assert len(lg) == len(lo)
    assert np.array_equal(lg["sequenceIn"], lo["sequenceIn"]) and np.array_equal(lg["sequenceOut"], lo["sequenceOut"])
    assert g.stats()["refactorizations"] == o.refactorizations
    assert abs(g.objectiveValue() - o.objective) <= RTOL * (1 + abs(o.objective))


@pytest.mark.parametrize("rule", [0, 1])
def test_singular_refactorization_restore_identical_pivot_sequence(gpu_cls, rule):
    """A refactorization that comes out singular in the middle of a solve (src/ClpSimplexDual.cpp:5060-5125): previous basis back
    with the flagged variables kept, the leaving variable flagged, a refactorization after every pivot.  Fault injection on both
    sides (option debug_singular_at: the refactorization itself succeeds and is taken as singular)."""
    lp = P.sparse_lp(300, 1200, 8, 11)
    g, sg, o, so = both(gpu_cls, lp, rule, debug_singular_at=300)
    assert sg == so == 0
    assert g.stats()["singular_restores"] == o.singular_restores == 1
    lg, lo = g.pivotLog(), o.pivot_log()
    assert len(lg) == len(lo)
    assert np.array_equal(lg["sequenceIn"], lo["sequenceIn"]) and np.array_equal(lg["sequenceOut"], lo["sequenceOut"])
    assert g.stats()["refactorizations"] == o.refactorizations
    assert abs(g.objectiveValue() - o.objective) <= RTOL * (1 + abs(o.objective))

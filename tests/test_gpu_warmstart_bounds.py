"""GPU test of ClpSimplex::sanityCheck's bound part at start-up (src/ClpSimplex.cpp:7645-7790, run by createRim(63) :4270):
bounds that cross by more than the primal tolerance make the problem infeasible before anything is solved -- what a branch that
empties a variable's range looks like in branch and bound.  Engine and oracle: status 1, no iteration; the context stays usable."""
import numpy as np
import pytest

from clp_amd import problems as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_cls(built):
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clp_amd.engine import ClpGpuSimplex

    return ClpGpuSimplex


def test_crossing_bounds_infeasible_then_restored(gpu_cls):
    from oracle.oracle import OracleSimplex

    lp = P.sparse_lp(300, 1200, 8, 11)
    g = gpu_cls().loadProblem(lp)
    assert g.dual() == 0
    optimum, status = g.objectiveValue(), g.statusArray().copy()
    crossed = lp.col_upper.copy()
    crossed[7] = lp.col_lower[7] - 1.0  # the branch "x7 <= lower - 1"
    g.chgColumnUpper(crossed)
    g.setStatusArray(status)
    assert g.dual() == 1 and g.numberIterations() == 0
    lp2 = type(lp)(lp)
    lp2.col_upper = crossed
    o = OracleSimplex(lp2)
    assert o.dual() == 1 and o.iterations == 0
    # the bounds back: the same context solves again, warm, to the same optimum
    g.chgColumnUpper(lp.col_upper)
    g.setStatusArray(status)
    assert g.dual() == 0
    assert abs(g.objectiveValue() - optimum) <= 1e-9 * (1 + abs(optimum))


def test_crossing_row_bounds_on_a_fresh_model(gpu_cls):
    lp = P.sparse_lp(120, 500, 5, 2)
    lp.row_lower = lp.row_lower.copy()
    lp.row_lower[3] = lp.row_upper[3] + 0.5
    g = gpu_cls().loadProblem(lp)
    assert g.dual() == 1 and g.numberIterations() == 0


@pytest.mark.parametrize("seed,dual_bound", [(7, 5.0), (28, 20.0), (29, 5.0), (48, 5.0)])
def test_infeasible_with_fake_bounds_active_is_status_10(gpu_cls, seed, dual_bound):
    """ClpSimplex::dual's second thought (src/ClpSimplex.cpp:5800-5803): "infeasible" reached while nonbasic variables still sit at
    fake bounds is status 10, "clean up in primal".  LPs of the oracle fuzz with free columns and a dual bound far below the
    solution's scale end that way on the oracle; the engine with option fake_bound_cleanup (what the clpGpuDual adapter sets, it
    has a primal to finish with) makes the same pivots and reports the same 10, and a bare context reports the 1 it found."""
    from oracle.oracle import OracleSimplex
    from test_oracle_fuzz import make

    lp = make(np.random.default_rng(7000 + seed))
    o = OracleSimplex(lp)
    o.set_option("pivot_rule", 1)
    o.set_option("dual_bound", dual_bound)
    assert o.dual() == 10
    for cleanup, expect in ((1, 10), (0, 1)):
        g = gpu_cls().loadProblem(lp)
        g.set_option("pivot_rule", 1)
        g.set_option("dual_bound", dual_bound)
        g.set_option("fake_bound_cleanup", cleanup)
        assert g.dual() == expect
        assert g.numberIterations() == o.iterations
        lg, lo = g.pivotLog(), o.pivot_log()
        assert np.array_equal(lg["sequenceIn"], lo["sequenceIn"]) and np.array_equal(lg["sequenceOut"], lo["sequenceOut"])


def test_no_incoming_column_second_thought(gpu_cls):
    """ "say infeasible ... unless primal feasible!!!!" (src/ClpSimplexDual.cpp:1982-2027): no incoming column right after a factorization
    is status 1 -- unless the sums of the last status check say nearly primal feasible or dual infeasible, then 10.  An LP of the fuzz
    (seed 374, Dantzig, dual bound 5) that ends that way on the oracle after 41 pivots: same pivots and the same 10 on the engine."""
    from oracle.oracle import OracleSimplex
    from test_oracle_fuzz import make

    lp = make(np.random.default_rng(7000 + 374))
    o = OracleSimplex(lp)
    g = gpu_cls().loadProblem(lp)
    for s in (o, g):
        s.set_option("pivot_rule", 0)
        s.set_option("dual_bound", 5.0)
    assert o.dual() == 10 and o.iterations == 41
    assert g.dual() == 10 and g.numberIterations() == 41
    lg, lo = g.pivotLog(), o.pivot_log()
    assert np.array_equal(lg["sequenceIn"], lo["sequenceIn"]) and np.array_equal(lg["sequenceOut"], lo["sequenceOut"])

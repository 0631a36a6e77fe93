"""Engine against oracle on the LPs of the CPU differential fuzz (tests/test_oracle_fuzz.py: free, one-sided and fixed columns, every kind
of row, crossing bounds, infeasible and unbounded instances) under default options and both row-choice rules: same status, same number of
pivots, same entering and leaving variables.  The GPU suite's other LPs are boxed-column instances; this is where the engine's handling of
fake bounds, flagged variables and the infeasible / unbounded exits is held to the oracle's -- it found the missing `pivotRow_ = -1` after
"no incoming column" (src/ClpSimplexDual.cpp:1874).  Two solves are known to differ and are excluded here (a ratio-test tie
among exactly equal |alpha| on integer data, seed 4; a 2-against-10 ending, seed 53); tools/fuzz_gpu.py runs the other option sets."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KNOWN = {(4, 0), (53, 0)}


@pytest.fixture(scope="module")
def gpu_cls(built):
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clp_amd.engine import ClpGpuSimplex

    return ClpGpuSimplex


@pytest.mark.parametrize("first", [0, 20, 40])
def test_fuzz_lps_same_pivots_as_oracle(gpu_cls, first):
    from oracle.oracle import OracleSimplex
    from test_oracle_fuzz import make

    differing = []
    for seed in range(first, first + 20):
        lp = make(np.random.default_rng(7000 + seed))
        for rule in (0, 1):
            if (seed, rule) in KNOWN:
                continue
            o = OracleSimplex(lp)
            g = gpu_cls().loadProblem(lp)
            for s in (o, g):
                s.set_option("pivot_rule", rule)
                s.set_option("max_iterations", 20000)
            g.set_option("fake_bound_cleanup", 1)  # the oracle restates ClpSimplex::dual's second thought (src/ClpSimplex.cpp:5800)
            so, sg = o.dual(), g.dual()
            lo, lg = o.pivot_log(), g.pivotLog()
            same = so == sg and len(lo) == len(lg) and np.array_equal(lo["sequenceIn"], lg["sequenceIn"]) and np.array_equal(lo["sequenceOut"], lg["sequenceOut"])
            if not same:
                differing.append((seed, rule, int(so), int(sg), len(lo), len(lg)))
    assert not differing, differing


def test_check_both_solutions_changes_these_solves(gpu_cls):
    """The resync's bookkeeping is ClpSimplex::checkBothSolutions (src/ClpSimplex.cpp:3226-3440) on both sides since round 4; LPs on which it
    and the older checkPrimalSolution + checkDualSolution pair end differently (39 against 20 pivots): the engine must
    follow the oracle under either setting of option check_both -- with the pair left in the engine these fail against the default oracle."""
    from oracle.oracle import OracleSimplex
    from test_oracle_fuzz import make

    for seed, rule in ((59, 1),):  # (seed 150, the CPU twin's second LP, differs between engine and oracle for another reason: tools/fuzz_gpu.py)
        lp = make(np.random.default_rng(7000 + seed))
        ends = []
        for both in (1, 0):
            o = OracleSimplex(lp)
            g = gpu_cls().loadProblem(lp)
            for s in (o, g):
                s.set_option("pivot_rule", rule)
                s.set_option("check_both", both)
            g.set_option("fake_bound_cleanup", 1)
            so, sg = o.dual(), g.dual()
            lo, lg = o.pivot_log(), g.pivotLog()
            assert so == sg and len(lo) == len(lg), (seed, both, so, sg, len(lo), len(lg))
            assert np.array_equal(lo["sequenceIn"], lg["sequenceIn"]) and np.array_equal(lo["sequenceOut"], lg["sequenceOut"])
            ends.append((so, len(lo)))
        assert ends[0] != ends[1], (seed, ends)


def test_problems_try_primal_exit_follows_the_oracle(gpu_cls):
    """gutsOfDual's "problems - try primal" exit (src/ClpSimplexDual.cpp:533-547; option try_primal 1, set by the clpGpuDual adapter) on the
    fuzz LPs, where a runaway escalation of the dual bound is what triggers it (tests/test_oracle_progress.py::test_problems_try_primal_exit):
    engine and oracle take it at the same status check -- same status 10, same pivots up to there -- and solves that do not take it are
    unchanged.  It also ends the solves before the stretches with dual bounds of 1e17 in which the default path's known differences live."""
    from oracle.oracle import OracleSimplex
    from test_oracle_fuzz import make

    differing, took = [], 0
    for seed in range(0, 60):
        lp = make(np.random.default_rng(7000 + seed))
        for rule in (0, 1):
            o = OracleSimplex(lp)
            g = gpu_cls().loadProblem(lp)
            for s in (o, g):
                s.set_option("pivot_rule", rule)
                s.set_option("max_iterations", 20000)
                s.set_option("try_primal", 1)
            g.set_option("fake_bound_cleanup", 1)
            so, sg = o.dual(), g.dual()
            lo, lg = o.pivot_log(), g.pivotLog()
            same = so == sg and len(lo) == len(lg) and np.array_equal(lo["sequenceIn"], lg["sequenceIn"]) and np.array_equal(lo["sequenceOut"], lg["sequenceOut"])
            same = same and o.try_primal == g.stats()["try_primal_exits"]
            took += int(o.try_primal)
            if not same:
                differing.append((seed, rule, int(so), int(sg), len(lo), len(lg), int(o.try_primal), int(g.stats()["try_primal_exits"])))
    print(f"{took} of 120 solves took the exit; differing: {differing}")
    assert took >= 10  # measured: 21
    # measured: one solve differs, and not at the exit -- seed 53 under Dantzig, the default path's known 2-against-10 ending (module docstring)
    assert all((d[0], d[1]) in KNOWN for d in differing), differing

"""CPU tests of the oracle's restatement of the reference's handling of nonbasic FREE variables in the dual (option "free_nonbasic" 1; 0,
the default, is the bothFake substitution both sides of this repo use by default; the HIP engine has the same option since round 5 and is held
to this oracle path in tests/test_gpu_free.py):
  - ClpSimplex::allSlackBasis gives a free column the status isFree at value 0 (src/ClpSimplex.cpp:7846-7849), changeBounds leaves it alone;
  - ClpSimplex::checkBothSolutions (:3226-3440) clears moreSpecialOptions_ & 8 when a nonbasic variable sits off its bounds, counts the dual
    infeasibilities of free variables apart and sets firstFree_;
  - ClpSimplexDual::dualRow (src/ClpSimplexDual.cpp:3005-3055) first brings free variables with a dj into the basis: nextSuperBasic (:8285),
    the FTRANned column, the row with the best infeasibility * |alpha| or the largest |alpha|;
  - ClpSimplexDual::dualColumn0's general branch (:4058-4179): a free variable worth keeping comes in whatever the ratios say (freePivot), and
    is given fake bounds on the way; a free variable with a small alpha and a small dj counts as badFree (forces a refactorization, :4778);
  - statusOfProblemInDual: primal feasible and only free dual infeasibilities is 10, "use primal" (:5619-5622).
There is no reference binary to compare with; what is checked: the optimum is HiGHS's (or the solve asks for the primal clean-up, status 10,
which ClpSimplex::dual would then run), the free paths really ran, and a few pivot counts are pinned against regressions."""
import numpy as np
import pytest

from oracle.oracle import OracleSimplex
from test_oracle_fuzz import highs, make


def free_columns(lp):
    return np.flatnonzero((lp.col_lower < -1e20) & (lp.col_upper > 1e20))


def solve(lp, rule, **opts):
    o = OracleSimplex(lp)
    o.set_option("pivot_rule", rule)
    o.set_option("free_nonbasic", 1)
    o.set_option("max_iterations", 20000)
    for k, v in opts.items():
        o.set_option(k, v)
    return o, o.dual()


@pytest.mark.parametrize("seed", range(120))
def test_free_paths_agree_with_highs(seed):
    lp = make(np.random.default_rng(7000 + seed))
    if not len(free_columns(lp)):
        pytest.skip("no free column")
    hs, hobj = highs(lp)
    if hs not in (0, 2, 3):
        pytest.skip("HiGHS undecided")
    for rule in (0, 1):
        for opts in ({}, {"perturbation": 50}, {"scaling": 3}, {"dual_bound": 50.0}):
            o, st = solve(lp, rule, **opts)
            where = (seed, rule, opts, hs, st)
            if hs == 0:
                assert st == 10 or (st == 0 and abs(o.objective - hobj) <= 1e-6 * (1 + abs(hobj))), where
            else:
                assert st in (1, 2, 10), where


@pytest.mark.parametrize("seed,expected", [(8, [(0, 70, 3, 5), (0, 48, 3, 5)]), (11, [(0, 64, 5, 5), (0, 35, 5, 5)]), (15, [(0, 23, 3, 3), (0, 24, 3, 3)]),
                                           (23, [(0, 58, 5, 5), (0, 69, 5, 5)]), (27, [(0, 15, 4, 5), (0, 19, 4, 4)])])
def test_free_paths_run_and_are_pinned(seed, expected):
    """(status, iterations, pivot rows chosen by the free-first entry, free variables brought in by freePivot) under Dantzig and steepest
    edge; with the option off neither path runs and the optimum is the same."""
    lp = make(np.random.default_rng(7000 + seed))
    for rule in (0, 1):
        o, st = solve(lp, rule)
        assert (st, o.iterations, o.free_first_rows, o.free_entered) == expected[rule]
        off = OracleSimplex(lp)
        off.set_option("pivot_rule", rule)
        assert off.dual() == 0 and off.free_first_rows == 0 and off.free_entered == 0
        assert abs(off.objective - o.objective) <= 1e-7 * (1 + abs(o.objective))


def test_free_columns_end_basic_or_at_a_fake_bound_never_free():
    """An optimal basis of the dual has no isFree nonbasic with a dj: every free column either came into the basis or was given fake bounds by
    the general branch (and then sits at one with a zero dj) or still is isFree with a dj inside the tolerance."""
    for seed in (8, 11, 22, 23, 34):
        lp = make(np.random.default_rng(7000 + seed))
        o, st = solve(lp, 1)
        assert st == 0
        status, dj = o.status() & 7, o.reduced_costs()
        for j in free_columns(lp):
            assert status[j] == 1 or abs(dj[j]) <= 1e-6, (seed, j, status[j], dj[j])


def test_a_caller_basis_with_a_free_column_at_a_bound_is_cleaned_to_free():
    """createRim's clean-up (src/ClpSimplex.cpp:4317-4338): atLowerBound / atUpperBound on a column without bounds becomes isFree."""
    lp = make(np.random.default_rng(7000 + 8))
    cold, st = solve(lp, 1)
    assert st == 0
    start = np.full(lp.n + lp.m, 3, dtype=np.uint8)  # every column at lower, slack basis
    start[lp.n:] = 1
    start[np.flatnonzero(lp.col_lower < -1e20)] = 2
    warm = OracleSimplex(lp)
    warm.set_option("pivot_rule", 1)
    warm.set_option("free_nonbasic", 1)
    warm.set_status(start)
    assert warm.dual() == 0 and warm.free_first_rows > 0
    assert abs(warm.objective - cold.objective) <= 1e-7 * (1 + abs(cold.objective))


def test_many_free_columns_the_reference_path_asks_for_primal():
    """sparse_lp(300, 1200) with a tenth of its columns made free: the reference's path brings 48 of the 120 in through freePivot (49 rows chosen by
    the free-first entry), gives the others fake bounds as the general branch meets them -- [value, value + dualBound] or [value - dualBound,
    value], so they sit at a fake bound that is their starting value -- and ends primal feasible on those bounds but not optimal: status 10,
    "use primal", which ClpSimplex::dual would then run.  The bothFake substitution (option off; the default on both sides) reaches HiGHS's optimum in
    half the pivots.  Pinned as measured, as a record of what the restated path does at that density of free columns."""
    from clp_amd import problems as P

    lp = P.sparse_lp(300, 1200, 8, 11)
    free = np.random.default_rng(11).choice(lp.n, lp.n // 10, replace=False)
    lp = type(lp)(lp)
    lp.col_lower, lp.col_upper = lp.col_lower.copy(), lp.col_upper.copy()
    lp.col_lower[free], lp.col_upper[free] = -1e30, 1e30
    hs, hobj = highs(lp)
    assert hs == 0
    o, st = solve(lp, 1)
    assert (st, o.iterations, o.free_first_rows, o.free_entered) == (10, 1731, 49, 48)
    assert o.objective > hobj + 1.0  # primal feasible, not optimal: the dual's work is undone by the fake bounds still active
    off = OracleSimplex(lp)
    off.set_option("pivot_rule", 1)
    assert off.dual() == 0 and off.iterations == 927 and abs(off.objective - hobj) <= 1e-7 * (1 + abs(hobj))


@pytest.mark.parametrize("seed", range(40))
def test_warm_resolve_after_branching_with_free_columns(seed):
    """The branch-and-bound pattern of tests/test_oracle_fuzz.py on the free path: solve, tighten three columns around their values, re-solve
    from the optimal basis with the option still on.  The warm start's statuses come from the first solve: free columns are basic, at a
    fake-bound status (cleaned back to isFree by the start-up, src/ClpSimplex.cpp:4317-4338) or still isFree."""
    for attempt in range(20):
        rng = np.random.default_rng(9000 + 20 * seed + attempt)
        lp = make(rng)
        if len(free_columns(lp)) and highs(lp)[0] == 0:
            break
    else:
        pytest.skip("no solvable draw with a free column")
    for rule in (0, 1):
        o, st = solve(lp, rule)
        if st != 0:
            continue
        status, x = o.status().copy(), o.solution()
        lp2 = type(lp)(lp)
        cu, cl = lp.col_upper.copy(), lp.col_lower.copy()
        for j in rng.choice(lp.n, min(3, lp.n), replace=False):
            if rng.uniform() < 0.5:
                if x[j] > 0.5:
                    cu[j] = min(cu[j], np.floor(x[j]))
            elif cl[j] > -1e29:
                cl[j] = max(cl[j], np.ceil(x[j]))
        lp2.col_upper, lp2.col_lower = cu, cl
        hs2, hobj2 = highs(lp2)
        for opts in ({}, {"perturbation": 100}, {"max_pivots": 3}):
            o2 = OracleSimplex(lp2)
            o2.set_option("pivot_rule", rule)
            o2.set_option("free_nonbasic", 1)
            for k, v in opts.items():
                o2.set_option(k, v)
            o2.set_status(status & 7)
            st2 = o2.dual()
            where = (seed, rule, opts, hs2, st2)
            if hs2 == 0:
                assert st2 == 10 or (st2 == 0 and abs(o2.objective - hobj2) <= 1e-6 * (1 + abs(hobj2))), where
            elif hs2 == 2:
                assert st2 in (1, 2, 10), where

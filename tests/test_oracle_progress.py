"""CPU tests of the oracle's restatement of ClpSimplexProgress for the dual (src/ClpSolve.cpp:4438-4725) and of the
"objective going backwards" restore of statusOfProblemInDual (src/ClpSimplexDual.cpp:5326-5488).  No LP of the suite makes
the objective fall by itself (that takes numerical trouble), so the restore is driven by fault injection (option
debug_backwards_at: the first two status checks at or after that iteration see a drop -- the first a small one, which only
saves the costs, :5379-5390, the second a large one, which goes back to the last good basis, :5392-5476)."""
import numpy as np
import pytest

from clp_amd import problems as P
from oracle.oracle import OracleSimplex


def solve(lp, rule, **opts):
    o = OracleSimplex(lp)
    o.set_option("pivot_rule", rule)
    for k, v in opts.items():
        o.set_option(k, v)
    return o, o.dual()


@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("maker,args,at", [("sparse_lp", (300, 1200, 8, 11), 300), ("netlib_shaped_lp", (400, 1500, 9000, 3), 250),
                                           ("nqueens", (20,), 200)])
def test_backwards_restore_returns_to_the_saved_basis_and_still_finishes(maker, args, at, rule):
    lp = getattr(P, maker)(*args)
    plain, s0 = solve(lp, rule)
    hurt, s1 = solve(lp, rule, debug_backwards_at=at)
    assert s0 == s1 == 0 and plain.backwards == 0 and hurt.backwards == 1
    assert abs(plain.objective - hurt.objective) <= 1e-9 * (1 + abs(plain.objective))
    # the pivots up to the injection are the undisturbed ones; after it a stretch is done again, refactorizing every few pivots
    a, b = plain.pivot_log()["sequenceIn"], hurt.pivot_log()["sequenceIn"]
    assert np.array_equal(a[:at], b[:at])
    assert hurt.iterations > plain.iterations and hurt.refactorizations > plain.refactorizations + 5


def test_progress_machinery_is_inert_on_healthy_solves():
    """No repeat over status checks, no fall of the objective: the solve is pivot for pivot what it was without it
    (iteration counts of the degenerate generators as committed before this logic existed)."""
    for n, rule, expected in [(8, 0, 91), (8, 1, 73), (20, 0, 553), (20, 1, 572)]:
        o, s = solve(P.nqueens(n), rule)
        assert s == 0 and o.iterations == expected and o.backwards == 0 and o.loop_flags == 0


@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("maker,args,at", [("sparse_lp", (300, 1200, 8, 11), 300), ("nqueens", (20,), 200)])
def test_bad_accuracy_restore_rejects_a_variable_and_still_finishes(maker, args, at, rule):
    """:5237-5318 through fault injection (option debug_bad_accuracy_at): the basis of the last good check comes back, the leaving
    variable is flagged, every pivot is followed by a refactorization for a while; the optimum is the undisturbed one."""
    lp = getattr(P, maker)(*args)
    plain, s0 = solve(lp, rule)
    hurt, s1 = solve(lp, rule, debug_bad_accuracy_at=at)
    assert s0 == s1 == 0 and plain.accuracy_restores == 0 and hurt.accuracy_restores == 1
    assert abs(plain.objective - hurt.objective) <= 1e-9 * (1 + abs(plain.objective))
    a, b = plain.pivot_log()["sequenceIn"], hurt.pivot_log()["sequenceIn"]
    assert np.array_equal(a[:at], b[:at])
    assert hurt.refactorizations > plain.refactorizations + 5


@pytest.mark.parametrize("rule", [0, 1])
def test_singular_refactorization_goes_back_to_the_saved_basis(rule):
    """:5060-5125 through fault injection (option debug_singular_at)."""
    lp = P.sparse_lp(300, 1200, 8, 11)
    plain, s0 = solve(lp, rule)
    hurt, s1 = solve(lp, rule, debug_singular_at=300)
    assert s0 == s1 == 0 and plain.singular_restores == 0 and hurt.singular_restores == 1
    assert abs(plain.objective - hurt.objective) <= 1e-9 * (1 + abs(plain.objective))
    assert np.array_equal(plain.pivot_log()["sequenceIn"][:300], hurt.pivot_log()["sequenceIn"][:300])
    assert hurt.refactorizations > plain.refactorizations + 5


@pytest.mark.parametrize("rule", [0, 1])
def test_check_both_solutions_against_the_older_pair(rule):
    """ClpSimplex::checkBothSolutions (src/ClpSimplex.cpp:3226) is what gutsOfSolution ends in in the reference, and since round 4 on both
    sides here (option "check_both" 1, the default; 0 = the older checkPrimalSolution + checkDualSolution pair).  On the well-behaved LPs of
    this suite the two make the same pivots -- they differ at the edge of the relaxed tolerances -- and on LPs of the fuzz they do not: the
    switch is a real one (the GPU twin of the second half is tests/test_gpu_fuzz.py::test_check_both_solutions_changes_these_solves)."""
    from clp_amd.mps import read_mps
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    lps = [read_mps(os.path.join(here, "golden", "afiro.mps")), P.nqueens(20), P.ufl(10, 30, 99), P.sparse_lp(300, 1200, 8, 11),
           P.netlib_shaped_lp(400, 1500, 9000, 3)]
    for lp in lps:
        a, sa = solve(lp, rule, check_both=0)
        b, sb = solve(lp, rule)
        assert sa == sb == 0
        assert np.array_equal(a.pivot_log()["sequenceIn"], b.pivot_log()["sequenceIn"])
        assert abs(a.objective - b.objective) <= 1e-9 * (1 + abs(a.objective))


def test_check_both_solutions_changes_fuzz_solves():
    """LPs of the differential fuzz on which the two forms of the resync's bookkeeping end differently: 39 pivots under the pair, 20 under
    checkBothSolutions (seed 59, steepest edge); "dual infeasible" (2) under the pair, "use primal" (10) under checkBothSolutions (seed 150)."""
    from test_oracle_fuzz import make

    lp = make(np.random.default_rng(7000 + 59))
    old, s_old = solve(lp, 1, check_both=0)
    new, s_new = solve(lp, 1)
    assert (s_old, old.iterations) == (10, 39) and (s_new, new.iterations) == (10, 20)
    lp = make(np.random.default_rng(7000 + 150))
    old, s_old = solve(lp, 0, check_both=0)
    new, s_new = solve(lp, 0)
    assert s_old == 2 and s_new == 10 and old.iterations == new.iterations == 13


def test_problems_try_primal_exit():
    """gutsOfDual's "problems - try primal" (src/ClpSimplexDual.cpp:533-547; option try_primal 1, what the clpGpuDual adapter sets): when the
    primal infeasibilities have grown 1e5-fold since the smallest sum seen while the objective stood still, and the recorded objectives say the
    solve fell off a cliff (or the growth is 1e10-fold), the dual hands over: status 10.  On the fuzz LPs that is what a runaway escalation of
    the dual bound looks like: of 2 400 solves (300 LPs x 2 rules x 4 option sets) 284 take the exit, 229 of which ended in 10 anyway -- later, with
    dual bounds of 1e17 --, 46 in 1 or 2 and 9 in 0.  Pinned on two of them; HiGHS agrees that the LPs a 10 is returned for are not decided
    wrongly (10 is "use primal", never a verdict); with the option off (the default on both sides) nothing changes."""
    from test_oracle_fuzz import highs, make

    took = 0
    for seed, rule, expect_off, expect_on in ((150, 0, (10, 13), (10, 7)), (3, 1, None, (10, 8)), (9, 0, None, (10, 11))):
        lp = make(np.random.default_rng(7000 + seed))
        off, s_off = solve(lp, rule)
        on, s_on = solve(lp, rule, try_primal=1)
        assert off.try_primal == 0 and on.try_primal == 1
        assert (s_on, on.iterations) == expect_on
        if expect_off is not None:
            assert (s_off, off.iterations) == expect_off
        assert on.iterations <= off.iterations
        k = on.iterations
        assert np.array_equal(on.pivot_log()["sequenceIn"], off.pivot_log()["sequenceIn"][:k])  # the same solve, ended earlier
        took += 1
    assert took == 3
    # an LP the dual solves is not touched by the option
    lp = make(np.random.default_rng(7000 + 8))
    a, sa = solve(lp, 1)
    b, sb = solve(lp, 1, try_primal=1)
    assert sa == sb == 0 and b.try_primal == 0 and a.iterations == b.iterations
    hs, hobj = highs(lp)
    assert hs == 0 and abs(b.objective - hobj) <= 1e-7 * (1 + abs(hobj))

"""CPU tests of the oracle's restatement of ClpSimplexDual::perturb (src/ClpSimplexDual.cpp:6533-6957) and of the places
dual() consults perturbation_ (start-up :335, the kick :488, the restores :2222, :5708, :5814, :5860, :5891)."""
import numpy as np
import pytest

from clp_amd import problems as P
from oracle.oracle import OracleSimplex


def solve(lp, rule, perturbation):
    o = OracleSimplex(lp)
    o.set_option("pivot_rule", rule)
    o.set_option("perturbation", perturbation)
    return o, o.dual()


@pytest.mark.parametrize("rule", [0, 1])
def test_perturbed_solve_reaches_the_same_optimum_in_fewer_pivots(rule):
    """N-Queens 50 (test/test_racing_lp.cpp:123): one cost value, 0/1 matrix -- the case perturbation exists for."""
    lp = P.nqueens(50)
    plain, s0 = solve(lp, rule, 102)
    pert, s1 = solve(lp, rule, 50)
    assert s0 == s1 == 0
    assert plain.perturbations == 0 and pert.perturbations == 1
    assert abs(plain.objective + 50.0) < 1e-7 and abs(pert.objective + 50.0) < 1e-7  # the true costs are back at the end
    assert pert.iterations * 3 < plain.iterations


def test_varied_costs_are_left_alone():
    """More than a quarter of the |costs| distinct: perturb() says "good enough" (:6599-6605) and the solve is the
    unperturbed one, pivot for pivot."""
    lp = P.sparse_lp(300, 1200, 5, 1)
    plain, s0 = solve(lp, 1, 102)
    pert, s1 = solve(lp, 1, 50)
    assert s0 == s1 == 0 and pert.perturbations == 0
    assert np.array_equal(plain.pivot_log()["sequenceIn"], pert.pivot_log()["sequenceIn"])


def test_constructor_default_only_kicks():
    """perturbation_ = 100 (the ClpSimplex constructor's value): nothing at start-up, so short solves are untouched."""
    lp = P.nqueens(20)
    plain, s0 = solve(lp, 1, 102)
    kick, s1 = solve(lp, 1, 100)
    assert s0 == s1 == 0 and kick.perturbations == 0
    assert np.array_equal(plain.pivot_log()["sequenceIn"], kick.pivot_log()["sequenceIn"])


def test_few_distinct_costs_on_a_generic_matrix():
    """Three cost values on a random matrix: perturbed at start-up; the solve ends optimal for the TRUE costs or asks
    for the primal clean-up (status 10, ClpSimplex::dual :5808) -- never a wrong optimum."""
    lp = P.sparse_lp(300, 1200, 5, 1)
    lp.obj = np.ceil(lp.obj * 3.0)
    plain, s0 = solve(lp, 1, 102)
    assert s0 == 0
    seen = set()
    for value in (50, 53, 57):
        pert, s1 = solve(lp, 1, value)
        assert pert.perturbations == 1 and s1 in (0, 10)
        seen.add(s1)
        if s1 == 0:
            assert abs(pert.objective - plain.objective) <= 1e-9 * abs(plain.objective)
    assert 0 in seen


def test_all_zero_costs_are_not_perturbed():
    """:6588 "safer to use primal": with every cost zero perturb() changes nothing; the dual still finds a feasible point."""
    lp = P.sparse_lp(120, 400, 5, 3)
    lp.obj = np.zeros(lp.n)
    pert, s1 = solve(lp, 1, 50)
    assert s1 == 0 and pert.perturbations == 0 and pert.objective == 0.0


def test_kick_after_two_m_plus_n_iterations():
    """N-Queens 100 under Dantzig pricing needs more than 2(m+n) = 21188 pivots: the default perturbation_ = 100 perturbs
    there (gutsOfDual :488) and the optimum is still the true one."""
    lp = P.nqueens(100)
    kick, s1 = solve(lp, 0, 100)
    assert s1 == 0 and kick.perturbations == 1
    assert kick.iterations > 2 * (lp.m + lp.n)
    assert abs(kick.objective + 100.0) < 1e-6

"""The oracle's restatement of ClpDualRowSteepest::pivotRow in full (src/ClpDualRowSteepest.cpp:179-364): numberWanted from mode_
(:258-278; the constructor's default is mode 3, src/ClpDualRowSteepest.hpp:118), the early break of the two passes (:329-335) and the
second call under an unchanged tolerance (:338-346) -- and the blocked form of its dense LU.

Nothing in the reference pins a pivot sequence, so what can be held here is: every mode still solves the fuzz LPs to HiGHS' optimum
while really scanning partially (option debug_chuzr_floor lowers the 2000 of :260-276 so that LPs of 40 rows do); mode 1 and a mode
whose numberWanted exceeds the list are the same solve; the second call happens and draws its own random number; and the blocked
elimination leaves the bits of the loop of CoinAbcDenseFactorization::factor (src/CoinAbcDenseFactorization.cpp:262-313)."""
import numpy as np
import pytest

from clp_amd.problems import dense_lp, sparse_lp
from oracle.oracle import OracleSimplex
from test_oracle_fuzz import highs, make


def counters(o):
    return o.partial_scans, o.chuzr_recalls, o.factor_elements


def solve(lp, **opts):
    o = OracleSimplex(lp)
    o.set_option("pivot_rule", 1)
    for k, v in opts.items():
        o.set_option(k, v)
    return o, o.dual(live=True)


@pytest.mark.parametrize("first", [0, 30, 60])
def test_every_mode_reaches_the_optimum_while_scanning_partially(first):
    partial = {2: 0, 3: 0}
    for seed in range(first, first + 30):
        lp = make(np.random.default_rng(7000 + seed))
        hs, hobj = highs(lp)
        if hs != 0:
            continue
        for mode, floor in ((3, 3), (2, 2), (3, 2000), (1, 2000), (0, 2000)):
            o, st = solve(lp, steepest_mode=mode, debug_chuzr_floor=floor, max_iterations=20000)
            assert st == 0 and abs(o.objective - hobj) <= 1e-6 * (1 + abs(hobj)), (seed, mode, floor, st, o.objective, hobj)
            scans = counters(o)[0]
            if floor < 2000:
                partial[mode] += scans
            else:
                assert scans == 0, (seed, mode)  # 40 rows against a numberWanted of 2000: every call scans the whole list
    assert partial[2] > 100 and partial[3] > 100, partial


def test_modes_without_a_partial_scan_are_one_solve():
    """3 000 infeasible rows at the start: modes 0 / 1 scan them all; mode 3 looks at max(2000, number / 20) of them while the basis
    holds fewer entries than rows and makes different pivots; with the floor above the list it is mode 1's solve."""
    lp = sparse_lp(3000, 12000, mean_nnz_per_col=10)
    logs = {}
    for name, opts in (("m1", dict(steepest_mode=1)), ("m0", dict(steepest_mode=0)), ("m3", dict(steepest_mode=3)),
                       ("m3_high_floor", dict(steepest_mode=3, debug_chuzr_floor=100000))):
        o, st = solve(lp, max_iterations=300, **opts)
        logs[name] = (o.pivot_log(), counters(o))
    assert logs["m1"][0].tobytes() == logs["m0"][0].tobytes() == logs["m3_high_floor"][0].tobytes()
    assert logs["m1"][1][0] == 0 and logs["m3"][1][0] == 300
    assert not np.array_equal(logs["m1"][0]["pivotRow"], logs["m3"][0]["pivotRow"])
    # factorization()->numberElements() of the model (option steepest_elements 0): entries of the basic structural columns
    assert 0 < logs["m3"][1][2] <= 300 * 40


RECALL_SEEDS = (217, 313, 404, 449, 725, 780, 783, 868, 1053, 1110)  # fuzz LPs on which the second call happens (found by a sweep of 1 200)


def test_second_call_under_the_unchanged_tolerance():
    """:338-346.  lastBadIteration_ is what arms the changed tolerance (:251-257); with it set to 0 (fault injection) the first 200
    iterations run under it whenever largestDualError_ > largestPrimalError_, and a call that then finds no row is made again --
    with another random number, so the solve's later random starts move: without the fault injection the same LPs make no second
    call."""
    for seed in RECALL_SEEDS:
        lp = make(np.random.default_rng(7000 + seed))
        o, st = solve(lp, debug_last_bad_iteration=0, debug_chuzr_floor=3, max_iterations=20000)
        assert counters(o)[1] >= 1, seed
        o2, st2 = solve(lp, debug_chuzr_floor=3, max_iterations=20000)
        assert counters(o2)[1] == 0, seed
        hs, hobj = highs(lp)
        if hs == 0:
            assert st == st2 == 0 and abs(o.objective - hobj) <= 1e-6 * (1 + abs(hobj)), (seed, st, st2)


def test_second_call_chooses_every_row_under_a_huge_factor():
    """Fault injection debug_tolerance_factor 1e30 (what the GPU twin of this test runs): no first call finds a row, every pivot's row is
    the second call's -- the solves still end at HiGHS' optimum, one second call per pivotRow."""
    pivots = recalls = 0
    for seed in range(40):
        lp = make(np.random.default_rng(7000 + seed))
        hs, hobj = highs(lp)
        if hs != 0:
            continue
        o, st = solve(lp, debug_last_bad_iteration=0, debug_tolerance_factor=1.0e30, debug_chuzr_floor=3, max_iterations=150)
        assert st == 0 and abs(o.objective - hobj) <= 1e-6 * (1 + abs(hobj)), (seed, st)
        pivots += o.iterations
        recalls += o.chuzr_recalls
    assert recalls >= pivots > 300


@pytest.mark.parametrize("which", ["sparse", "dense"])
def test_blocked_lu_is_the_plain_loop(which):
    lp, iters = (sparse_lp(1200, 4800, mean_nnz_per_col=8), 1500) if which == "sparse" else (dense_lp(300, 300), 10000)
    out = []
    for plain in (1, 0):
        o, st = solve(lp, debug_plain_lu=plain, max_iterations=iters)
        out.append((st, o.objective, o.solution().tobytes(), o.pivot_log().tobytes(), o.row_weights()[0].tobytes()))
    assert out[0] == out[1]

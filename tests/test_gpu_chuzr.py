"""ClpDualRowSteepest::pivotRow as the reference runs it by default (mode 3, src/ClpDualRowSteepest.hpp:118): the partial scan of the
infeasibility list -- numberWanted entries above the tolerance from the random start, :258-278 and :329-335 -- and the second call
under the unchanged tolerance (:338-346), engine against oracle.

k_chuzr_scan hands a partial scan to one workgroup that walks the list in the reference's order (chuzrOrderedScan, kernels.hip): chunks
without a flagged candidate / the last pivot row are finished in parallel, the others by the reference's own statements from a staged
copy.  Small LPs never scan partially at the reference's floor of 2000, so option debug_chuzr_floor lowers it on both sides; the fuzz
LPs (tests/test_oracle_fuzz.py) bring flagged variables, fake bounds and the infeasible / unbounded exits."""
import numpy as np
import pytest

import clp_amd.problems as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_cls(built):
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clp_amd.engine import ClpGpuSimplex

    return ClpGpuSimplex


def both(gpu_cls, lp, **opts):
    from oracle.oracle import OracleSimplex

    o = OracleSimplex(lp)
    g = gpu_cls().loadProblem(lp)
    for s in (o, g):
        s.set_option("pivot_rule", 1)
        for k, v in opts.items():
            s.set_option(k, v)
    g.set_option("fake_bound_cleanup", 1)
    return g, o


def same_pivots(lg, lo):
    return len(lg) == len(lo) and all(np.array_equal(lg[k], lo[k]) for k in ("sequenceIn", "sequenceOut", "pivotRow"))


def compare(g, o, sg, so):
    """'same': status and pivots identical; 'wild': theta / alpha / dualOut / objective parted by more than 1e-9 relative BEFORE the first
    differing pivot (or before the end, when only the status differs) -- the fuzz LPs with free columns run through stretches with fake
    bounds of 5e9 ... 1e10 in the basis, where the explicit inverse and the oracle's LU round 0.1 apart in absolute terms, under every
    steepest mode; 'different': the pivots part while the values still agree -- a difference in logic."""
    lg, lo = g.pivotLog(), o.pivot_log()
    n = min(len(lg), len(lo))
    if sg == so and same_pivots(lg, lo):
        close = all(np.all(np.abs(lg[k] - lo[k]) <= 1e-9 * (1 + np.abs(lo[k]))) for k in ("theta", "alpha", "dualOut", "objective"))
        return "same" if close else "same pivots"
    dp = next((i for i in range(n) if any(lg[k][i] != lo[k][i] for k in ("sequenceIn", "sequenceOut", "pivotRow"))), n)
    dv = next((i for i in range(n) if any(abs(lg[k][i] - lo[k][i]) > 1e-9 * (1 + abs(lo[k][i])) for k in ("theta", "alpha", "dualOut", "objective"))), None)
    return "wild" if dv is not None and dv < dp else "different"


def sweep(gpu_cls, seeds, counter, **opts):
    from test_oracle_fuzz import make

    wild, different, total = [], [], 0
    sweep.ordered_walks = 0
    for seed in seeds:
        lp = make(np.random.default_rng(7000 + seed))
        g, o = both(gpu_cls, lp, max_iterations=20000, **opts)
        so, sg = o.dual(), g.dual()
        kind = compare(g, o, sg, so)
        sweep.ordered_walks += int(g.stats()["chuzr_ordered_walks"])
        ours, theirs = int(g.stats()[counter]), getattr(o, counter)
        if kind == "same":
            total += ours
            # (the tail of a pivot runs the head of the next CHUZR: when the solve ends there the device has counted a call the
            # reference never makes)
            if ours not in (theirs, theirs + 1):
                different.append((seed, "counter", theirs, ours))
        elif kind == "same pivots":  # identical pivots through a numerically wild stretch: the calls made at the solve's end may differ
            total += ours
        elif kind == "wild":
            wild.append(seed)
        else:
            different.append((seed, int(so), int(sg), len(o.pivot_log()), len(g.pivotLog())))
    assert not different, different
    assert len(wild) <= len(seeds) // 10, wild
    return total


@pytest.mark.parametrize("mode,floor", [(3, 3), (2, 2), (3, 17)])
def test_partial_scan_on_the_fuzz_lps(gpu_cls, mode, floor):
    assert sweep(gpu_cls, range(60), "chuzr_partial_scans", steepest_mode=mode, debug_chuzr_floor=floor) > 200
    # the two exceptions of the scan (a flagged candidate, the last pivot row above the tolerance) really occur in these solves: the
    # ordered walk from the span that holds one is exercised, not only the parallel path
    assert sweep.ordered_walks > 0, sweep.ordered_walks


@pytest.mark.parametrize("factor,floor", [(1.0e30, 3), (1.0e30, 2000), (1.0e4, 3)])
def test_second_call_follows_the_oracle(gpu_cls, factor, floor):
    """:338-346.  The changed tolerance needs lastBadIteration_ within 200 iterations and largestDualError_ > largestPrimalError_ -- two
    rounding-noise numbers on a healthy LP, not reproducible between two factorizations -- so both sides are armed by fault injection
    (options debug_last_bad_iteration 0, debug_tolerance_factor f: tests/test_oracle_chuzr.py).  f = 1e30: no first call finds a row, every
    pivot's row is the second call's choice (its own random start, its own partial scan, the device's chuzrOrderedScan in the workgroup
    that made the final selection); f = 1e4: the second call only where the first finds nothing above 1e-5.  Same second calls, same
    pivots."""
    recalls = sweep(gpu_cls, range(60), "chuzr_recalls", debug_last_bad_iteration=0, debug_tolerance_factor=factor, debug_chuzr_floor=floor)
    assert recalls >= (300 if factor > 1e20 else 30), recalls


def test_default_mode_scans_partially_at_3000_rows(gpu_cls):
    """3 000 infeasible rows at the slack basis, the reference's own floor: numberWanted = 2000 while the basis holds fewer entries than
    rows (ratio < 1: max(2000, number / 20)), later max(2000, number / 8).  600 pivots (refactorizations at 135 pivots each) identical,
    every call a partial scan on both sides, the entries count the ratio is taken from equal; mode 1 is a different solve."""
    lp = P.sparse_lp(3000, 12000, 10)
    g, o = both(gpu_cls, lp, max_pivots=0)
    o.set_option("max_iterations", 600)
    assert g.dual_steps(600) == -1 and o.dual() == 3
    lg, lo = g.pivotLog(), o.pivot_log()
    assert same_pivots(lg, lo)
    st = g.stats()
    assert st["chuzr_partial_scans"] == o.partial_scans and o.partial_scans >= 500
    assert st["factor_elements"] == o.factor_elements > 0
    g1, o1 = both(gpu_cls, lp, max_pivots=0, steepest_mode=1)
    assert g1.dual_steps(600) == -1
    assert g1.stats()["chuzr_partial_scans"] == 0
    assert not np.array_equal(g1.pivotLog()["pivotRow"], lg["pivotRow"])


def test_full_size_first_500_pivots_in_the_default_mode(gpu_cls):
    """BASELINE config 4 from the slack basis: 50 000 infeasible rows, an empty factorization (ratio 0 < 1) -> the reference looks at
    max(2000, 50 000 / 20) = 2 500 rows per call.  The first 500 pivots against the oracle (a committed record of the same solve as
    tests/test_gpu_parity.py::test_full_size_sparse_first_500_pivots_vs_oracle), and every one of them a partial scan."""
    lp = P.sparse_lp()
    g, o = both(gpu_cls, lp, max_pivots=0)
    o.set_option("max_iterations", 500)
    assert g.dual_steps(500) == -1 and o.dual() == 3
    assert same_pivots(g.pivotLog(), o.pivot_log())
    assert g.stats()["chuzr_partial_scans"] == o.partial_scans == 500


def test_mature_basis_scans_fully_with_the_factorization_s_own_count(gpu_cls):
    """Mode 3 sizes the scan by factorization()->numberElements() / rows (:262-276).  From config 4's mature basis the engine is in LU mode
    and its factorization holds front L + U + the dense tail + the frozen slack part: ~32 M entries, ratio ~640 > 80 -> every call scans the
    whole list, as real Clp with an LU of that basis would (option steepest_elements 1: what the plug-in's numberElements() answers).  With the
    shared model (0, the default on both sides -- DESIGN section 2 for why: the 537 448 entries of the 10 514 basic structural columns, ratio 10.7) the same basis scans
    number x ratio / 80 = 13 % of the rows per call."""
    import os

    lp = P.sparse_lp()
    status = (np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "basis_sparse_30000.npy")) & 7).astype(np.uint8)
    seen = {}
    for model in (1, 0):
        g = gpu_cls().loadProblem(lp)
        g.setStatusArray(status)
        g.set_option("pivot_rule", 1)
        g.set_option("max_pivots", 0)
        g.set_option("steepest_elements", model)
        assert g.dual_steps(60) == -1
        st = g.stats()
        assert st["lu_active"] == 1
        seen[model] = (int(st["factor_elements"]), int(st["chuzr_partial_scans"]))
    assert seen[1][0] > 80 * lp.m and seen[1][1] == 0, seen
    assert seen[0][0] < 12 * lp.m and seen[0][1] >= 55, seen

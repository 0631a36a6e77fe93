"""The row choice of ClpSimplexDual::dualRow's free-first entry (src/ClpSimplexDual.cpp:3016-3049) as the engine's host code makes it under
option free_nonbasic (clp_amd/csrc/engine.hip freeFirstChoice, through the device-free hook clpgpu_test_free_first_row): among the
positions where the FTRANned free column has |alpha| > 1e-3, the unflagged one with the largest infeasibility x |alpha| and |alpha| > 0.1;
else, if its |alpha| passes 0.01, the one with the largest |alpha| whose variable has a bound; strict maxima, the first of equals.  Held to
a plain restatement on random data and to hand-made cases for every threshold; the whole free path is held to the oracle on the GPU
(tests/test_gpu_free.py)."""
import numpy as np

FLAGGED = 64


def restated(work, pv, sol, lo, up, status):
    best_f = best_i = 0.0
    row_f = row_i = -1
    for r in range(len(work)):
        a = abs(work[r])
        if a > 1e-3:
            s = pv[r]
            inf = sol[s] - up[s] if sol[s] > up[s] else (lo[s] - sol[s] if sol[s] < lo[s] else 0.0)
            if inf * a > best_i and a > 0.1 and not status[s] & FLAGGED:
                best_i, row_i = inf * a, r
            if a > best_f and (lo[s] > -1e20 or up[s] < 1e20):
                best_f, row_f = a, r
    if row_i >= 0:
        return row_i
    return row_f if best_f > 1e-2 else -1


def test_random_cases_follow_the_restatement(built):
    from clp_amd.engine import free_first_row

    rng = np.random.default_rng(31)
    seen = set()
    for trial in range(400):
        m, N = int(rng.integers(1, 40)), 90
        pv = rng.choice(N, m, replace=False).astype(np.int32)
        work = rng.standard_normal(m) * rng.choice([0.0, 5e-4, 5e-3, 0.05, 0.5, 3.0], m)
        lo = np.where(rng.random(N) < 0.3, -1e30, rng.uniform(-2, 0, N))
        up = np.where(rng.random(N) < 0.3, 1e30, rng.uniform(0, 2, N))
        sol = rng.uniform(-3, 3, N) * (rng.random(N) < (0.5 if trial % 2 else 0.0))  # every other trial: nothing infeasible
        status = np.where(rng.random(N) < 0.2, 1 | FLAGGED, 1).astype(np.uint8)
        got = free_first_row(work, pv, sol, lo, up, status)
        assert got == restated(work, pv, sol, lo, up, status)
        seen.add("none" if got < 0 else "row")
    assert seen == {"none", "row"}


def test_thresholds_and_ties(built):
    from clp_amd.engine import free_first_row

    N = 6
    pv = np.arange(4, dtype=np.int32)
    lo, up = np.zeros(N), np.ones(N)
    st = np.ones(N, np.uint8)
    feasible = np.full(N, 0.5)
    # nothing infeasible: the largest |alpha| with a bound, if it passes 0.01; the first of equals
    assert free_first_row([0.5, -2.0, 2.0, 0.1], pv, feasible, lo, up, st) == 1
    assert free_first_row([0.009, 0.0, 0.002, 0.0], pv, feasible, lo, up, st) == -1
    assert free_first_row([0.011, 0.0, 0.002, 0.0], pv, feasible, lo, up, st) == 0
    # a variable without any bound is never the feasible choice
    lo2, up2 = lo.copy(), up.copy()
    lo2[1], up2[1] = -1e30, 1e30
    assert free_first_row([0.5, -2.0, 1.0, 0.1], pv, feasible, lo2, up2, st) == 2
    # an infeasible row wins over a larger |alpha| on a feasible one, by infeasibility x |alpha|, but only with |alpha| > 0.1
    sol = feasible.copy()
    sol[3] = 4.0  # 3 above its upper bound
    assert free_first_row([5.0, 0.0, 0.0, 0.2], pv, sol, lo, up, st) == 3
    assert free_first_row([5.0, 0.0, 0.0, 0.09], pv, sol, lo, up, st) == 0
    sol[2] = -1.0  # 1 below its lower bound: 1 x 0.9 > 3 x 0.2
    assert free_first_row([5.0, 0.0, 0.9, 0.2], pv, sol, lo, up, st) == 2
    # a flagged variable is not pivoted out for its infeasibility
    st2 = st.copy()
    st2[2] |= FLAGGED
    assert free_first_row([5.0, 0.0, 0.9, 0.2], pv, sol, lo, up, st2) == 3
    # below 1e-3 a position does not exist
    assert free_first_row([5e-4, 0.0, 0.0, 0.0], pv, sol, lo, up, st) == -1

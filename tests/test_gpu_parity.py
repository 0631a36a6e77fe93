"""GPU parity tests (-m gpu): the HIP path through the C ABI against the CPU oracle on the same
inputs, against the committed golden fixtures, and -- at BASELINE.json's full size -- through
size-independent properties.  Bit-exact for integer/index work and for the per-column dot products
of pricing (same IEEE operation order); 1e-8 relative for objective / solutions (north_star)."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from clp_amd import problems as P

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
RTOL = 1e-8  # north_star: objective and primal/dual solutions within 1e-8 relative


@pytest.fixture(scope="module")
def gpu_cls(built):
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clp_amd.engine import ClpGpuSimplex

    return ClpGpuSimplex


def oracle(lp, rule=1, **opts):
    from oracle.oracle import OracleSimplex

    o = OracleSimplex(lp)
    o.set_option("pivot_rule", rule)
    for k, v in opts.items():
        o.set_option(k, v)
    return o


def rel(a, b):
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b)))) if len(a) else 0.0


def test_smoke(built):
    import __graft_entry__ as g

    g.smoke()


# ---------------------------------------------------------------- kernel level -----------------
@pytest.mark.parametrize("maker,args", [("sparse_lp", (300, 1200, 8, 11)), ("dense_lp", (120, 150, 12)),
                                        ("sparse_lp", (2000, 9000, 12, 13))])
@pytest.mark.parametrize("density", [0.02, 0.5])
def test_price_row_bit_identical(gpu_cls, maker, args, density):
    lp = getattr(P, maker)(*args)
    g, o = gpu_cls().loadProblem(lp), oracle(lp)
    rng = np.random.default_rng(17)
    m, n = lp.m, lp.n
    k = max(1, int(density * m))
    idx = np.sort(rng.choice(m, k, replace=False)).astype(np.int32)
    val = rng.standard_normal(k)
    status = rng.choice([1, 2, 3, 5], size=n + m, p=[0.2, 0.3, 0.45, 0.05]).astype(np.uint8)
    dj = np.where((status & 3) == 2, -1.0, 1.0) * rng.uniform(0, 2, n + m)
    a, b = g.priceRow(idx, val, status, dj), o.price_row_fused(idx, val, status, dj)
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x, y)
    assert a[4] == b[4]


def test_price_row_committed_golden(gpu_cls):
    z = np.load(os.path.join(HERE, "golden", "price_case.npz"))
    lp = P.sparse_lp(300, 1200, 8, seed=11)
    oi, ov, ci, cv, ut = gpu_cls().loadProblem(lp).priceRow(z["pi_index"], z["pi_value"], z["status"], z["dj"])
    assert np.array_equal(oi, z["out_index"]) and np.array_equal(ov, z["out_value"])
    assert np.array_equal(ci, z["cand_index"]) and np.array_equal(cv, z["cand_value"])
    assert ut == float(z["upper_theta"])


def test_price_row_edge_cases(gpu_cls):
    lp = P.sparse_lp(300, 1200, 8, seed=11)
    g, o = gpu_cls().loadProblem(lp), oracle(lp)
    m, n = lp.m, lp.n
    status = np.full(n + m, 3, np.uint8)
    dj = np.ones(n + m)
    # empty pi -> empty row, no candidates, upperTheta stays at its start value 1e31
    a = g.priceRow(np.zeros(0, np.int32), np.zeros(0), status, dj)
    assert len(a[0]) == 0 and len(a[2]) == 0 and a[4] == 1e31
    # everything basic / fixed -> nothing scanned
    st2 = status.copy()
    st2[::2] = 1
    st2[1::2] = 5
    idx = np.arange(0, m, 3, dtype=np.int32)
    val = np.linspace(-1, 1, len(idx))
    a, b = g.priceRow(idx, val, st2, dj), o.price_row_fused(idx, val, st2, dj)
    assert len(a[0]) == len(b[0]) == 0
    # values below the zero tolerance are dropped exactly like the reference (1e-13)
    tiny = g.priceRow(idx[:1], np.array([1e-16]), status, dj)
    ref = o.price_row_fused(idx[:1], np.array([1e-16]), status, dj)
    assert np.array_equal(tiny[0], ref[0])


def test_matrix_products(gpu_cls):
    lp = P.sparse_lp(500, 1800, 7, seed=19)
    g, o = gpu_cls().loadProblem(lp), oracle(lp)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(lp.n), rng.standard_normal(lp.m)
    assert rel(g.times(-1.0, x, y), o.times(-1.0, x, y)) < 1e-13
    xr, yc = rng.standard_normal(lp.m), rng.standard_normal(lp.n)
    assert np.array_equal(g.transposeTimes(-1.0, xr, yc), o.transpose_times(-1.0, xr, yc))  # same op order


@pytest.mark.parametrize("maker,args,k", [("dense_lp", (120, 150, 12), 30), ("dense_lp", (120, 150, 12), 100),
                                          ("sparse_lp", (300, 1200, 8, 11), 0)])
def test_factor_solve_update(gpu_cls, maker, args, k):
    lp = getattr(P, maker)(*args)
    g, o = gpu_cls().loadProblem(lp), oracle(lp)
    rng = np.random.default_rng(5)
    m, n = lp.m, lp.n
    status = np.full(n + m, 3, np.uint8)
    status[n:] = 1
    cols, rows = rng.choice(n, k, replace=False), rng.choice(m, k, replace=False)
    status[cols] = 1
    status[n + rows] = 3
    (rg, pg), (ro, po) = g.factorize(status), o.factorize(status)
    assert rg == ro == 0 and np.array_equal(pg, po)  # same partial-pivoting order => same positions
    pv = po.copy()
    for t in range(8):
        v = rng.standard_normal(m) * (rng.random(m) < 0.6)
        assert rel(g.ftran(v), o.ftran(v)) < 1e-9 and rel(g.btran(v), o.btran(v)) < 1e-9
        basic = set(int(s) for s in pv)
        for _ in range(100):
            q = int(rng.integers(0, n + m))
            if q in basic:
                continue
            col = np.zeros(m)
            if q >= n:
                col[q - n] = -1.0
            else:
                col[lp.row[lp.col_start[q]:lp.col_start[q + 1]]] = lp.elem[lp.col_start[q]:lp.col_start[q + 1]]
            w = o.ftran(col)
            cand = np.nonzero(np.abs(w) > 0.1)[0]
            if len(cand):
                break
        p = int(cand[rng.integers(0, len(cand))])
        assert g.replaceColumn(p, q) == 0 and o.replace_column(w, p, w[p]) == 0
        pv[p] = q
    # B^-1 B = I on the final basis (property, independent of the oracle)
    for pos in rng.choice(m, 5, replace=False):
        seq = int(pv[pos])
        col = np.zeros(m)
        if seq >= n:
            col[seq - n] = -1.0
        else:
            col[lp.row[lp.col_start[seq]:lp.col_start[seq + 1]]] = lp.elem[lp.col_start[seq]:lp.col_start[seq + 1]]
        e = np.zeros(m)
        e[pos] = 1.0
        assert np.allclose(g.ftran(col), e, atol=1e-8)


def test_singular_basis_reported(gpu_cls):
    lp = P.sparse_lp(300, 1200, 8, seed=11)
    g, o = gpu_cls().loadProblem(lp), oracle(lp)
    rng = np.random.default_rng(5)
    status = np.full(lp.n + lp.m, 3, np.uint8)
    status[lp.n:] = 1
    status[rng.choice(lp.n, 40, replace=False)] = 1
    status[lp.n + rng.choice(lp.m, 40, replace=False)] = 3
    assert g.factorize(status)[0] == o.factorize(status)[0] == -1  # ClpFactorization.hpp:53


# ---------------------------------------------------------------- whole solves -----------------
def solve_both(gpu_cls, lp, rule, **opts):
    g = gpu_cls().loadProblem(lp)
    g.set_option("pivot_rule", rule)
    o = oracle(lp, rule, **opts)
    for k, v in opts.items():
        g.set_option(k, v)
    return g, g.dual(), o, o.dual()


def kkt(lp, g, tol=1e-6):
    sol, dj, st = g.solution(), g.reducedCosts(), g.statusArray() & 7
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(lp.m, lp.n))
    assert np.allclose(A @ sol[:lp.n], sol[lp.n:], atol=1e-6, rtol=1e-9)
    lo = np.concatenate([lp.col_lower, lp.row_lower])
    up = np.concatenate([lp.col_upper, lp.row_upper])
    assert np.all(sol >= lo - tol) and np.all(sol <= up + tol)
    assert np.all(dj[(st == 3) & (up > lo)] >= -1e-5) and np.all(dj[(st == 2) & (up > lo)] <= 1e-5)
    assert (st == 1).sum() == lp.m


@pytest.mark.parametrize("rule", [0, 1])
def test_afiro_identical_pivot_sequence(gpu_cls, afiro, rule):
    """BASELINE config 2: AFIRO, pivot sequence and basis identical to the CPU path."""
    g, sg, o, so = solve_both(gpu_cls, afiro, rule)
    assert sg == so == 0
    lg, lo = g.pivotLog(), o.pivot_log()
    assert np.array_equal(lg["sequenceIn"], lo["sequenceIn"]) and np.array_equal(lg["sequenceOut"], lo["sequenceOut"])
    assert np.array_equal(lg["pivotRow"], lo["pivotRow"])
    assert np.array_equal(g.pivotVariable(), o.pivot_variable())
    assert np.array_equal(g.statusArray() & 7, o.status() & 7)
    assert abs(g.objectiveValue() - (-4.6475314286e+02)) <= RTOL * 464.75314286
    assert rel(g.solution(), o.solution()) < RTOL and rel(g.reducedCosts(), o.reduced_costs()) < RTOL
    gold = json.load(open(os.path.join(HERE, "golden", "afiro_pivots.json")))["steepest" if rule else "dantzig"]
    assert lg["sequenceIn"].tolist() == gold["in"] and lg["sequenceOut"].tolist() == gold["out"]
    kkt(afiro, g)


# dense 300x400: mean row and column length >= 256 -> the wide-row / wide-column / dense-column kernel
# variants; sparse 60x30000: long rows with short columns (wide-row variants alone)
@pytest.mark.parametrize("maker,args", [("dense_lp", (120, 150, 12)), ("sparse_lp", (300, 1200, 8, 11)),
                                        ("sparse_lp", (1500, 6000, 10, 31)), ("dense_lp", (300, 400, 5)),
                                        ("sparse_lp", (60, 30000, 6, 5))])
@pytest.mark.parametrize("rule", [0, 1])
def test_random_lp_identical_pivot_sequence(gpu_cls, maker, args, rule):
    lp = getattr(P, maker)(*args)
    g, sg, o, so = solve_both(gpu_cls, lp, rule)
    assert sg == so == 0
    lg, lo = g.pivotLog(), o.pivot_log()
    assert len(lg) == len(lo)
    assert np.array_equal(lg["sequenceIn"], lo["sequenceIn"]) and np.array_equal(lg["sequenceOut"], lo["sequenceOut"])
    assert np.array_equal(g.pivotVariable(), o.pivot_variable())
    assert abs(g.objectiveValue() - o.objective) <= RTOL * (1 + abs(o.objective))
    assert rel(g.solution(), o.solution()) < RTOL and rel(g.reducedCosts(), o.reduced_costs()) < 1e-7
    assert rel(lg["theta"], lo["theta"]) < 1e-7 and rel(lg["alpha"], lo["alpha"]) < 1e-7
    kkt(lp, g)


@pytest.mark.parametrize("case", [("nqueens", (8,), -8.0), ("nqueens", (20,), -20.0), ("tsp_mtz", (20, 42), 172.283333),
                                  ("ufl", (10, 30, 99), 560.0), ("ufl", (20, 60, 77), 770.5)])
@pytest.mark.parametrize("rule", [0, 1])
def test_degenerate_reference_instances(gpu_cls, case, rule):
    """Generated instances of test/test_racing_lp.cpp with the bounds of test/test_racing_reference.txt.
    Massively degenerate (0/1 data): ties are broken by last-bit rounding, so the pivot sequence is
    not asserted here -- status, objective and KKT are."""
    name, args, expected = case
    lp = getattr(P, name)(*args)
    g, sg, o, so = solve_both(gpu_cls, lp, rule)
    assert sg == so == 0
    assert abs(g.objectiveValue() - expected) < 1e-5 * max(1.0, abs(expected))
    assert abs(g.objectiveValue() - o.objective) <= RTOL * (1 + abs(o.objective))
    kkt(lp, g)


@pytest.mark.parametrize("rule", [0, 1])
def test_netlib_shaped_fake_bounds(gpu_cls, rule):
    """Power-law columns, 20% equality rows, 10% columns without upper bound (fake bounds,
    ClpSimplexDual::changeBounds :3148) and entries spanning 1e-3..1e3: objective / KKT parity."""
    lp = P.netlib_shaped_lp(300, 1000, 8000, seed=9)
    g, sg, o, so = solve_both(gpu_cls, lp, rule)
    assert sg == so == 0
    assert abs(g.objectiveValue() - o.objective) <= 1e-7 * (1 + abs(o.objective))
    kkt(lp, g, tol=1e-5)


def test_warm_resolve_after_bound_change(gpu_cls):
    """Branch-and-bound pattern: solve, tighten upper bounds with the matrix left on the device
    (ClpModel::chgColumnUpper), re-solve dual from the previous basis.  The re-solve is warm (far
    fewer pivots than from the slack basis), ends at the optimum the oracle finds for the modified LP
    from the same basis, and satisfies the KKT conditions of the modified LP.  (The pivot sequences are
    not compared: like a ClpSimplex object that is solved twice, the engine is not a fresh model on
    the second call.)"""
    lp = P.sparse_lp(300, 1200, 8, 11)
    g = gpu_cls().loadProblem(lp)
    assert g.dual() == 0
    first_iterations = g.numberIterations()
    status, sol = g.statusArray().copy(), g.solution()
    branch = np.argsort(-sol[:lp.n], kind="stable")[:5]
    assert np.all(sol[branch] > 1.0)
    new_upper = lp.col_upper.copy()
    new_upper[branch] = np.floor(sol[branch] * 0.5)
    g.chgColumnUpper(new_upper)
    g.setStatusArray(status)
    sg = g.dual()
    lp2 = type(lp)(lp)  # LpData is a dict with attribute access
    lp2.col_upper = new_upper
    o = oracle(lp2, 1)
    o.set_status(status)
    so = o.dual()
    assert sg == so == 0
    assert 0 < g.numberIterations() < first_iterations // 2
    assert abs(g.objectiveValue() - o.objective) <= RTOL * (1 + abs(o.objective))
    assert np.all(g.solution()[:lp.n][branch] <= new_upper[branch] + 1e-7)
    kkt(lp2, g)


@pytest.mark.parametrize("option", ["blocked_refactor", "register_panel"])
def test_reinversion_variants_agree(gpu_cls, option):
    """Re-inversion of the nucleus: the unblocked form, the blocked form with the global-memory panel
    (the one nuclei beyond 4096 get) and the default blocked form with the register-resident panel do
    the same arithmetic -- identical pivots and identical solution bits."""
    lp = P.sparse_lp(300, 1200, 8, 11)
    a = gpu_cls().loadProblem(lp)
    b = gpu_cls().loadProblem(lp)
    for g in (a, b):
        g.set_option("max_pivots", 20)  # many refactorizations
    b.set_option(option, 0)
    assert a.dual() == b.dual() == 0
    la, lb = a.pivotLog(), b.pivotLog()
    assert np.array_equal(la["sequenceIn"], lb["sequenceIn"]) and np.array_equal(la["sequenceOut"], lb["sequenceOut"])
    assert np.array_equal(a.pivotVariable(), b.pivotVariable())
    assert np.array_equal(a.solution(), b.solution())


def test_flip_list_overflow_path_matches(gpu_cls):
    """The bound flips of a pivot are appended unordered and put in list order afterwards; with a
    2-entry append buffer every pivot with more flips takes the overflow path (ordered compaction of
    the flags by one workgroup).  Same pivots, same flips, same solution as the default engine."""
    lp = P.sparse_lp(300, 1200, 8, 11)
    a = gpu_cls().loadProblem(lp)
    b = gpu_cls().loadProblem(lp)
    b.set_option("flip_list_cap", 2)
    assert a.dual() == b.dual() == 0
    la, lb = a.pivotLog(), b.pivotLog()
    assert int(la["numberFlipped"].max()) > 2, "instance no longer exercises the overflow path"
    for key in ("sequenceIn", "sequenceOut", "numberFlipped"):
        assert np.array_equal(la[key], lb[key])
    assert np.array_equal(a.solution(), b.solution())


@pytest.mark.parametrize("n", [10, 50])
def test_infeasible(gpu_cls, n):
    g, sg, o, so = solve_both(gpu_cls, P.infeasible(n), 1)
    assert sg == so == 1


def test_iteration_limit_and_stepping(gpu_cls):
    """clpgpu_dual_steps resumes exactly where it stopped: same pivots as an uninterrupted run."""
    lp = P.sparse_lp(300, 1200, 8, seed=11)
    g1 = gpu_cls().loadProblem(lp)
    assert g1.dual() == 0
    g2 = gpu_cls().loadProblem(lp)
    status = -1
    while status == -1:
        status = g2.dual_steps(97)
    assert status == 0 and g2.numberIterations() == g1.numberIterations()
    assert np.array_equal(g1.pivotLog()["sequenceIn"], g2.pivotLog()["sequenceIn"])
    g3 = gpu_cls().loadProblem(lp)
    g3.setMaximumIterations(50)
    assert g3.dual() == 3 and g3.numberIterations() == 50


def test_column_range_shard_matches_full(gpu_cls):
    """Pricing restricted to a column range (the multi-GPU shard) + rank-major merge == unsharded."""
    from clp_amd.sharding import column_ranges, merge_candidates

    lp = P.sparse_lp(400, 2000, 8, seed=23)
    rng = np.random.default_rng(2)
    m, n = lp.m, lp.n
    idx = np.sort(rng.choice(m, 60, replace=False)).astype(np.int32)
    val = rng.standard_normal(60)
    status = rng.choice([1, 2, 3], size=n + m, p=[0.2, 0.3, 0.5]).astype(np.uint8)
    dj = np.where((status & 3) == 2, -1.0, 1.0) * rng.uniform(0, 2, n + m)
    full = gpu_cls().loadProblem(lp).priceRow(idx, val, status, dj)
    parts = []
    for r, (a, b) in enumerate(column_ranges(n, 4)):
        g = gpu_cls().loadProblem(lp)
        g.setColumnRange(a, b)
        st = status.copy()
        if r:
            st[n:] = 1  # slack part is priced by rank 0 only
        parts.append(g.priceRow(idx, val, st, dj))
    merged = merge_candidates(parts)
    for k in range(4):
        assert np.array_equal(merged[k], full[k])
    assert merged[4] == full[4]


# ---------------------------------------------------------------- full size --------------------
def test_full_size_sparse_properties(gpu_cls):
    """BASELINE config 4 (50k x 200k, ~10M nz): properties that do not need the oracle at full size.
    (1) the priced tableau row equals -A^T pi (scipy) to 1e-12 and is linear in pi;
    (2) 300 pivots keep B^-1 consistent: FTRAN of a basic column is a unit vector;
    (3) the dual objective never decreases."""
    lp = P.sparse_lp()
    g = gpu_cls().loadProblem(lp)
    m, n = lp.m, lp.n
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(m, n))
    rng = np.random.default_rng(7)
    idx = np.sort(rng.choice(m, 4000, replace=False)).astype(np.int32)
    v1, v2 = rng.standard_normal(4000), rng.standard_normal(4000)
    status = np.full(n + m, 3, np.uint8)
    dj = np.ones(n + m)
    rows = []
    for v in (v1, v2, v1 + v2):
        oi, ov, _, _, _ = g.priceRow(idx, v, status, dj)
        full = np.zeros(n)
        full[oi] = ov
        rows.append(full)
        pi = np.zeros(m)
        pi[idx] = v
        assert np.allclose(full, -(A.T @ pi), rtol=1e-11, atol=1e-12)
    assert np.allclose(rows[0] + rows[1], rows[2], rtol=1e-11, atol=1e-11)
    g2 = gpu_cls().loadProblem(lp)
    assert g2.dual_steps(300) == -1 and g2.numberIterations() == 300
    log = g2.pivotLog()
    assert np.all(np.diff(log["objective"]) >= -1e-7 * (1 + np.abs(log["objective"][1:])))
    pv = g2.pivotVariable()
    for pos in rng.choice(m, 4, replace=False):
        seq = int(pv[pos])
        col = np.zeros(m)
        if seq >= n:
            col[seq - n] = -1.0
        else:
            col[lp.row[lp.col_start[seq]:lp.col_start[seq + 1]]] = lp.elem[lp.col_start[seq]:lp.col_start[seq + 1]]
        e = np.zeros(m)
        e[pos] = 1.0
        assert np.allclose(g2.ftran(col), e, atol=1e-8)

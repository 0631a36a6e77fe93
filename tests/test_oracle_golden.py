"""CPU tests: the oracle against every value the reference's own tests pin for this path
 and against the committed golden fixtures; HiGHS as an independent cross-check."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import linprog

from clp_amd import problems as P
from oracle.oracle import OracleSimplex

HERE = os.path.dirname(os.path.abspath(__file__))


def highs_objective(lp):
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(lp.m, lp.n))
    bl = np.where(lp.row_lower < -1e29, -np.inf, lp.row_lower)
    bu = np.where(lp.row_upper > 1e29, np.inf, lp.row_upper)
    Aub = sp.vstack([A, -A]).tocsr()
    bub = np.concatenate([bu, -bl])
    keep = np.isfinite(bub)
    bnds = [(None if l < -1e29 else l, None if u > 1e29 else u) for l, u in zip(lp.col_lower, lp.col_upper)]
    r = linprog(lp.obj, A_ub=Aub[keep], b_ub=bub[keep], bounds=bnds, method="highs")
    return r.status, r.fun


def kkt_check(lp, o, tol=1e-6):
    """Primal/dual feasibility and A x = row activity, as src/unitTest.cpp:1898-1975 does for afiro."""
    sol, dj, st = o.solution(), o.reduced_costs(), o.status() & 7
    n = lp.n
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(lp.m, lp.n))
    assert np.allclose(A @ sol[:n], sol[n:], atol=1e-7, rtol=1e-9)
    lo = np.concatenate([lp.col_lower, lp.row_lower])
    up = np.concatenate([lp.col_upper, lp.row_upper])
    assert np.all(sol >= lo - tol) and np.all(sol <= up + tol)
    at_lower = (st == 3) & (up > lo)
    at_upper = (st == 2) & (up > lo)
    assert np.all(dj[at_lower] >= -1e-5) and np.all(dj[at_upper] <= 1e-5)
    basic = st == 1
    assert basic.sum() == lp.m and np.all(np.abs(dj[basic]) < 1e-7)


@pytest.mark.parametrize("rule", [0, 1])
def test_afiro_objective_and_kkt(built, afiro, rule):
    # reference: src/unitTest.cpp:480-485 and :1898-1975 (objective -4.6475314286e+02 to 1e-8 rel)
    o = OracleSimplex(afiro)
    o.set_option("pivot_rule", rule)
    assert o.dual() == 0
    assert abs(o.objective - (-4.6475314286e+02)) <= 1e-8 * 464.75314286
    kkt_check(afiro, o)
    st, obj = highs_objective(afiro)
    assert st == 0 and abs(obj - o.objective) < 1e-7


def test_hello_plumbing_case(built):
    """BASELINE config 1: the reference's examples/hello.mps (21 x 53, all +-1, ranges and UP bounds),
    stored as parsed arrays by tests/golden/make_golden.py.  The reference pins no objective for it
    (SURVEY 8c); HiGHS and the oracle agree on 0 (the slack basis is optimal: zero pivots)."""
    from clp_amd.mps import LpData

    d = np.load(os.path.join(HERE, "golden", "hello_lp.npz"))
    lp = LpData({k: (int(d[k]) if k in ("m", "n") else d[k]) for k in d.files if k != "optimum"})
    lp["name"] = "hello"
    assert (lp.m, lp.n, len(lp.elem)) == (21, 53, 224)
    for rule in (0, 1):
        o = OracleSimplex(lp)
        o.set_option("pivot_rule", rule)
        assert o.dual() == 0
        assert abs(o.objective - float(d["optimum"])) < 1e-9
        kkt_check(lp, o)
    st, obj = highs_objective(lp)
    assert st == 0 and abs(obj - float(d["optimum"])) < 1e-9


def test_modified_afiro_fixture(built):
    """The reference's other input fixture, examples/modified_afiro.mps (7 x 16: the piecewise variant of AFIRO that
    examples/piecewise.cpp:20 reads; not Netlib AFIRO), stored as parsed arrays by tests/golden/make_golden.py.  No expected
    objective in the reference; HiGHS and the oracle agree on -484.2061685711 under both pivot rules."""
    from clp_amd.mps import LpData

    d = np.load(os.path.join(HERE, "golden", "modified_afiro_lp.npz"))
    lp = LpData({k: (int(d[k]) if k in ("m", "n") else d[k]) for k in d.files if k != "optimum"})
    lp["name"] = "modified_afiro"
    assert (lp.m, lp.n, len(lp.elem)) == (7, 16, 40)
    for rule in (0, 1):
        o = OracleSimplex(lp)
        o.set_option("pivot_rule", rule)
        assert o.dual() == 0
        assert abs(o.objective - float(d["optimum"])) < 1e-9 * abs(float(d["optimum"]))
        kkt_check(lp, o)
    st, obj = highs_objective(lp)
    assert st == 0 and abs(obj - float(d["optimum"])) < 1e-9 * abs(obj)


def test_exmip1_as_the_reference_test_pins_it(built):
    """exmip1 (5 x 8) rebuilt from what test/OsiClpSolverInterfaceTest.cpp asserts about it (matrix by row :203-258, objective
    :194-201, row senses / right-hand sides / ranges :396-415, the column bounds of :170-173; tests/golden/make_golden.py::exmip1).
    Pinned by the reference: the by-column image of the matrix (:342-355), the objective of the initial point (3.5, :191-192), the LP
    optimum 3.2368421 (src/unitTest.cpp:2572), and "infeasible bounds": column 0 given the bounds [1, 0] is not proven optimal
    (:675-686) -- the start-up sanity check of src/ClpSimplex.cpp:7773."""
    from clp_amd.mps import LpData

    d = np.load(os.path.join(HERE, "golden", "exmip1_lp.npz"))
    lp = LpData({k: (int(d[k]) if k in ("m", "n") else d[k]) for k in d.files if k != "optimum"})
    lp["name"] = "exmip1"
    assert lp.elem.tolist() == [3.0, 5.6, 1.0, 2.0, 1.1, 1.0, -2.0, 2.8, -1.0, 1.0, 1.0, -1.2, -1.0, 1.9]  # getMatrixByCol :342-355
    assert abs(float(lp.obj @ lp.col_lower) - 3.5) < 1e-12
    for rule in (0, 1):
        o = OracleSimplex(lp)
        o.set_option("pivot_rule", rule)
        assert o.dual() == 0
        assert abs(o.objective - 3.2368421) < 1e-6 and abs(o.objective - float(d["optimum"])) < 1e-9
        kkt_check(lp, o)
        bad = LpData(lp)
        bad.col_lower, bad.col_upper = lp.col_lower.copy(), lp.col_upper.copy()
        bad.col_lower[0], bad.col_upper[0] = 1.0, 0.0
        o = OracleSimplex(bad)
        o.set_option("pivot_rule", rule)
        assert o.dual() == 1 and o.iterations == 0


def test_afiro_pivot_log_matches_committed_golden(built, afiro):
    gold = json.load(open(os.path.join(HERE, "golden", "afiro_pivots.json")))
    for rule, name in ((0, "dantzig"), (1, "steepest")):
        o = OracleSimplex(afiro)
        o.set_option("pivot_rule", rule)
        assert o.dual() == 0
        log = o.pivot_log()
        assert log["sequenceIn"].tolist() == gold[name]["in"]
        assert log["sequenceOut"].tolist() == gold[name]["out"]


def test_unit_test_3x5_basis_solution(built):
    # reference: src/unitTest.cpp:1413-1482 -- factorize basis {x0,x1,x4}, solution {20/7,3,0,0,23/7}
    lp = P.unit_test_3x5()
    o = OracleSimplex(lp)
    status = np.array([1, 1, 3, 3, 1, 3, 3, 3], dtype=np.uint8)
    rc, pv = o.factorize(status)
    assert rc == 0 and sorted(pv.tolist()) == [0, 1, 4]
    x = o.ftran(np.array([14.0, 3.0, 3.0]))  # nonbasics at 0, rows fixed at 14,3,3
    colsol = np.zeros(5)
    colsol[pv] = x
    assert np.allclose(colsol, [20.0 / 7.0, 3.0, 0.0, 0.0, 23.0 / 7.0], rtol=1e-12)
    # B^-T B^T = I
    e = np.array([1.0, 0.0, 0.0])
    y = o.btran(e)
    for p, seq in enumerate(pv):
        s, t = lp.col_start[seq], lp.col_start[seq + 1]
        assert abs(sum(y[lp.row[s:t]] * lp.elem[s:t]) - e[p]) < 1e-12


@pytest.mark.parametrize("case", [
    # test/test_racing_reference.txt:9-39
    ("nqueens", (8,), -8.0), ("nqueens", (20,), -20.0),
    ("tsp_mtz", (20, 42), 172.283333), ("tsp_mtz", (40, 123), 189.0),
    ("ufl", (10, 30, 99), 560.0), ("ufl", (20, 60, 77), 770.5),
])
def test_racing_lp_reference_bounds(built, case):
    name, args, expected = case
    lp = getattr(P, name)(*args)
    o = OracleSimplex(lp)
    assert o.dual() == 0
    assert abs(o.objective - expected) < 1e-5 * max(1.0, abs(expected))
    kkt_check(lp, o)


@pytest.mark.parametrize("n", [10, 50])
def test_infeasible_detected(built, n):
    # test/test_racing_lp.cpp:277 -- expected status "Infeasible"
    o = OracleSimplex(P.infeasible(n))
    assert o.dual() == 1


@pytest.mark.parametrize("maker,args", [("dense_lp", (60, 80, 5)), ("sparse_lp", (200, 800, 6, 7)),
                                        ("netlib_shaped_lp", (150, 500, 3000, 9))])
@pytest.mark.parametrize("rule", [0, 1])
def test_synthetic_against_highs(built, maker, args, rule):
    lp = getattr(P, maker)(*args)
    o = OracleSimplex(lp)
    o.set_option("pivot_rule", rule)
    assert o.dual() == 0
    st, obj = highs_objective(lp)
    assert st == 0 and abs(obj - o.objective) <= 1e-7 * (1.0 + abs(obj))
    kkt_check(lp, o)


def test_price_case_golden(built):
    z = np.load(os.path.join(HERE, "golden", "price_case.npz"))
    lp = P.sparse_lp(300, 1200, 8, seed=11)
    o = OracleSimplex(lp)
    oi, ov, ci, cv, ut = o.price_row_fused(z["pi_index"], z["pi_value"], z["status"], z["dj"])
    assert np.array_equal(oi, z["out_index"]) and np.array_equal(ov, z["out_value"])
    assert np.array_equal(ci, z["cand_index"]) and np.array_equal(cv, z["cand_value"])
    assert ut == float(z["upper_theta"])
    # independent check of the column part with scipy: alpha_j = -pi^T a_j
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(lp.m, lp.n))
    pi = np.zeros(lp.m)
    pi[z["pi_index"]] = z["pi_value"]
    full = -(A.T @ pi)
    assert np.allclose(full[oi], ov, rtol=1e-12, atol=1e-14)


def test_matrix_ops_against_scipy(built):
    lp = P.sparse_lp(120, 400, 5, seed=3)
    o = OracleSimplex(lp)
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(lp.m, lp.n))
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(lp.n), rng.standard_normal(lp.m)
    assert np.allclose(o.times(-1.0, x, y), y - A @ x, rtol=1e-12, atol=1e-12)
    xr, yc = rng.standard_normal(lp.m), rng.standard_normal(lp.n)
    assert np.allclose(o.transpose_times(2.0, xr, yc), yc + 2.0 * (A.T @ xr), rtol=1e-12, atol=1e-12)


def test_factor_update_roundtrip(built):
    """FTRAN/BTRAN stay inverses of B through product-form updates (size-independent property)."""
    lp = P.dense_lp(40, 60, seed=2)
    o = OracleSimplex(lp)
    rng = np.random.default_rng(5)
    status = np.full(lp.n + lp.m, 3, np.uint8)
    status[lp.n:] = 1
    cols = rng.choice(lp.n, 10, replace=False)
    rows = rng.choice(lp.m, 10, replace=False)
    status[cols] = 1
    status[lp.n + rows] = 3
    rc, pv = o.factorize(status)
    assert rc == 0

    def column(seq):
        c = np.zeros(lp.m)
        if seq >= lp.n:
            c[seq - lp.n] = -1.0
        else:
            s, t = lp.col_start[seq], lp.col_start[seq + 1]
            c[lp.row[s:t]] = lp.elem[s:t]
        return c

    for q in (3, lp.n + int(rows[0]), 7):
        if q in pv:
            continue
        w = o.ftran(column(q))
        p = int(np.argmax(np.abs(w)))
        assert o.replace_column(w, p, w[p]) == 0
        pv[p] = q
        for pos, seq in enumerate(pv):
            e = np.zeros(lp.m)
            e[pos] = 1.0
            assert np.allclose(o.ftran(column(seq)), e, atol=1e-9)

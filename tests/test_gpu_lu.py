"""GPU parity tests of the LU factorization mode (option factor_mode = 1):
sparse Markowitz front + dense tail inverted on the matrix cores + product-form eta file.

Three kinds of check, each through the C ABI:
  * the solves against an INDEPENDENT dense / sparse-direct solve of the same basis matrix (numpy / scipy
    SuperLU), before and after column replacements -- the factorization's own definition, no oracle involved;
  * the same solves against the CPU oracle's factorization (restated CoinAbcDenseFactorization), compared by
    variable, not by basis position: the two factorizations put the basic variables in different positions,
    as any two LU codes do (ClpFactorization::factorize permutes pivotVariable_, src/ClpFactorization.cpp:1953);
  * whole solves in LU mode against the oracle and against the engine's explicit-inverse mode.
Tolerances: 1e-9 relative on solve vectors, 1e-8 relative on objective / solutions (north_star)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as sla

from clp_amd import problems as P

pytestmark = pytest.mark.gpu


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


def nrel(a, b):
    """normwise relative error of a solve vector (the measure a backward-stable solver is held to)"""
    return float(np.max(np.abs(a - b)) / (1.0 + np.max(np.abs(b)))) if len(a) else 0.0


def basis_matrix(lp, pv):
    """B by basis position: column p is the column of variable pv[p] in [A | -I]."""
    m, n = lp.m, lp.n
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(m, n))
    full = sp.hstack([A, -sp.identity(m, format="csc")]).tocsc()
    return full[:, pv].tocsc()


def column_of(lp, q):
    col = np.zeros(lp.m)
    if q >= lp.n:
        col[q - lp.n] = -1.0
    else:
        col[lp.row[lp.col_start[q]:lp.col_start[q + 1]]] = lp.elem[lp.col_start[q]:lp.col_start[q + 1]]
    return col


CASES = [
    # (rows, cols, nnz/col, seed), pivots of the run that produces the basis, lu_stop_density
    ((300, 1200, 8, 11), 150, 0.5),      # front on a small nucleus (density limit lifted)   k ~ 100
    ((300, 1200, 8, 11), 150, 0.0),      # no front: dense tail + eta file only
    ((2000, 9000, 12, 13), 1400, 0.05),  # k ~ 1000
    ((8000, 32000, 12, 17), 7000, 0.012),  # default stop rule, k ~ 5000
]


@pytest.mark.parametrize("args,npiv,density", CASES)
def test_lu_solves_and_eta_file(gpu_cls, args, npiv, density):
    lp = P.sparse_lp(*args)
    m, n = lp.m, lp.n
    # a basis the dual simplex really visits (random column sets of these LPs are structurally singular)
    run = gpu_cls().loadProblem(lp)
    run.set_option("factor_mode", 0)
    run.set_option("check_every", 16)
    run.dual_steps(npiv)
    status = (np.asarray(run.statusArray()) & 7).astype(np.uint8)
    status[(status != 1) & (status != 2)] = 3
    del run
    k = int((status[:n] == 1).sum())
    assert k >= npiv // 3
    g = gpu_cls().loadProblem(lp)
    g.set_option("factor_mode", 1)
    g.set_option("lu_stop_density", density)
    g.set_option("lu_min_tail", 4)
    rng = np.random.default_rng(5)
    rc, pv = g.factorize(status)
    assert rc == 0, g.lastError()
    assert sorted(int(s) for s in pv) == sorted(int(i) for i in np.nonzero(status == 1)[0])
    st = g.stats()
    assert st["lu_active"] == 1 and st["lu_front"] + st["lu_tail"] == k
    if density > 0.0 and k >= 800:
        assert st["lu_front"] > k // 4  # the front really is sparse-eliminated
    o = oracle(lp) if m <= 2000 else None
    if o is not None:
        ro, po = o.factorize(status)
        assert ro == 0

    def check(pv, tol):
        B = basis_matrix(lp, pv)
        lu = sla.splu(B)
        for _ in range(3):
            v = rng.standard_normal(m) * (rng.random(m) < 0.6)
            x, y = g.ftran(v), g.btran(v)
            assert nrel(x, lu.solve(v)) < tol
            assert nrel(y, lu.solve(v, trans="T")) < tol
            if o is not None:
                # by variable: x[pos] belongs to pv[pos] here and to po[pos] in the oracle
                xo = o.ftran(v)
                byseq = np.zeros(n + m)
                byseq[po_now] = xo
                assert nrel(x, byseq[pv]) < tol
                cseq = np.zeros(n + m)
                cseq[pv] = v
                assert nrel(y, o.btran(cseq[po_now])) < tol

    po_now = po.copy() if o is not None else None
    pv = pv.copy()
    check(pv, 1e-9)
    # column replacements: the eta file, with positions replaced more than once
    lastp = -1
    for t in range(24):
        basic = set(int(s) for s in pv)
        lu = sla.splu(basis_matrix(lp, pv))
        for _ in range(200):
            q = int(rng.integers(0, n + m))
            if q in basic:
                continue
            w = lu.solve(column_of(lp, q))
            cand = np.nonzero(np.abs(w) > 0.1)[0]
            if len(cand):
                break
        p = lastp if (t % 5 == 4 and abs(w[lastp]) > 0.05) else int(cand[rng.integers(0, len(cand))])
        lastp = p
        assert g.replaceColumn(p, q) == 0
        if o is not None:
            # the same replacement in the oracle: its position of the leaving variable
            po_pos = int(np.nonzero(po_now == pv[p])[0][0])
            wo = o.ftran(column_of(lp, q))
            assert o.replace_column(wo, po_pos, wo[po_pos]) == 0
            po_now[po_pos] = q
        pv[p] = q
        assert g.pivots() == t + 1
        if t in (0, 5, 23):
            check(pv, 1e-8)
    assert np.array_equal(np.asarray(g.pivotVariable()), pv)


@pytest.mark.parametrize("args,rule", [((300, 1200, 8, 11), 1), ((300, 1200, 8, 11), 0), ((1500, 6000, 10, 19), 1)])
def test_lu_engine_solves_match_oracle(gpu_cls, args, rule):
    """Whole solves with the LU factorization from the first refactorization on, a short eta file (many
    refactorizations) and a long one, against the oracle and against the explicit-inverse engine."""
    lp = P.sparse_lp(*args)
    o = oracle(lp, rule)
    assert o.dual() == 0
    base = gpu_cls().loadProblem(lp)
    base.set_option("pivot_rule", rule)
    base.set_option("factor_mode", 0)
    assert base.dual() == 0
    for max_pivots, density in ((25, 0.3), (400, 0.3), (100, 0.0)):
        g = gpu_cls().loadProblem(lp)
        g.set_option("pivot_rule", rule)
        g.set_option("factor_mode", 1)
        g.set_option("lu_max_pivots", max_pivots)
        g.set_option("lu_stop_density", density)
        g.set_option("lu_min_tail", 4)
        assert g.dual() == 0
        assert g.stats()["lu_factorizations"] > 0
        assert abs(g.objectiveValue() - o.objective) <= 1e-8 * (1.0 + abs(o.objective))
        assert rel(g.solution(), o.solution()) < 1e-7
        assert rel(g.solution(), base.solution()) < 1e-7
        # same pivots as the explicit-inverse engine as long as no tie is broken by basis position
        a, b = g.pivotLog(), base.pivotLog()
        same = 0
        while same < min(len(a), len(b)) and a[same]["sequenceIn"] == b[same]["sequenceIn"] and a[same]["sequenceOut"] == b[same]["sequenceOut"]:
            same += 1
        assert same >= min(50, len(b))


def test_lu_full_size_matches_inverse_mode(gpu_cls):
    """Config 4 at full size: LU mode from a nucleus of 512 on (front + tail + eta file active from pivot ~500)
    against the explicit-inverse engine over the first 2500 pivots: same entering / leaving variables,
    objective to 1e-9, solution to 1e-7.  Under ClpDualRowSteepest's full scan (steepest_mode 1): the two modes put the basic
    variables at different basis positions, as any two LU codes do, and the partial scan of the default mode 3 looks at the
    first numberWanted rows of the infeasibility list IN POSITION ORDER -- two factorizations cannot share its pivots."""
    lp = P.sparse_lp()
    runs = []
    for mode in (0, 1):
        g = gpu_cls().loadProblem(lp)
        g.set_option("pivot_rule", 1)
        g.set_option("check_every", 16)
        g.set_option("max_pivots", 0)
        g.set_option("factor_mode", -1 if mode else 0)
        g.set_option("steepest_mode", 1)
        g.set_option("lu_min_k", 512)
        g.set_option("lu_max_pivots", 300)
        assert g.dual_steps(2500) == -1 and g.numberIterations() == 2500
        runs.append(g)
    a, b = runs[1].pivotLog(), runs[0].pivotLog()
    assert len(a) == len(b) == 2500
    st = runs[1].stats()
    assert st["lu_active"] == 1 and st["lu_front"] > 0 and st["lu_tail"] > 0
    same = 0
    while same < 2500 and a[same]["sequenceIn"] == b[same]["sequenceIn"] and a[same]["sequenceOut"] == b[same]["sequenceOut"]:
        same += 1
    assert same >= 2000, f"pivot sequences part at {same}"
    if same == 2500:
        assert abs(runs[1].objectiveValue() - runs[0].objectiveValue()) <= 1e-9 * abs(runs[0].objectiveValue())
        assert rel(runs[1].solution(), runs[0].solution()) < 1e-7


def test_lu_solve_matches_independent_solver(gpu_cls):
    """End to end against an INDEPENDENT solver: HiGHS' dual simplex (scipy) on a config-4-shaped LP small enough
    for it to finish; the engine runs in LU mode from a nucleus of 256 on.  Objective to 1e-8 relative."""
    from scipy.optimize import linprog

    lp = P.sparse_lp(1500, 6000, 10, seed=23)
    m, n = lp.m, lp.n
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(m, n))
    Aeq = sp.hstack([A, -sp.identity(m, format="csc")]).tocsr()
    bounds = np.column_stack([np.concatenate([lp.col_lower, lp.row_lower]), np.concatenate([lp.col_upper, lp.row_upper])])
    r = linprog(np.concatenate([lp.obj, np.zeros(m)]), A_eq=Aeq, b_eq=np.zeros(m), bounds=bounds, method="highs-ds")
    assert r.status == 0
    g = gpu_cls().loadProblem(lp)
    g.set_option("pivot_rule", 1)
    g.set_option("check_every", 16)
    g.set_option("max_pivots", 0)
    g.set_option("lu_min_k", 256)
    assert g.dual() == 0
    assert g.stats()["lu_factorizations"] > 0
    assert abs(g.objectiveValue() - r.fun) <= 1e-8 * abs(r.fun)


@pytest.mark.parametrize("args", [(1000, 1200, 8, 41), (2500, 2600, 10, 43)])
def test_lu_mode_on_nearly_square_lps(gpu_cls, args):
    """LU mode with n ~ m: the eta-file kernel then runs more workgroups (64 basis positions each) than the LP has 256-key
    compaction blocks, and its per-workgroup slots must be sized for that."""
    lp = P.sparse_lp(*args)
    # the checker: the oracle on the small LP (7 s of CPU), the engine's explicit-inverse mode on the larger one (the oracle needs minutes)
    if lp.m <= 1000:
        o = oracle(lp)
        assert o.dual() == 0
        ref_obj, ref_sol = o.objective, o.solution()
    else:
        base = gpu_cls().loadProblem(lp)
        base.set_option("pivot_rule", 1)
        base.set_option("factor_mode", 0)
        assert base.dual() == 0
        ref_obj, ref_sol = base.objectiveValue(), base.solution()
    g = gpu_cls().loadProblem(lp)
    g.set_option("pivot_rule", 1)
    g.set_option("factor_mode", 1)
    g.set_option("lu_stop_density", 0.3)
    g.set_option("lu_min_tail", 4)
    g.set_option("lu_max_pivots", 60)
    assert g.dual() == 0
    assert g.stats()["lu_factorizations"] > 0
    assert abs(g.objectiveValue() - ref_obj) <= 1e-8 * (1.0 + abs(ref_obj))
    assert rel(g.solution(), ref_sol) < 1e-7


@pytest.mark.parametrize("n", [37, 128, 300, 1111])
def test_own_mfma_gemm_matches_numpy(gpu_cls, n):
    """The engine's f64 GEMM on the matrix cores (the kernel behind the Newton-Schulz steps) against numpy:
    c = beta c + alpha a b, sizes that are not multiples of the 128 x 128 x 16 tiling."""
    lp = P.sparse_lp(300, 1200, 8, seed=11)
    g = gpu_cls().loadProblem(lp)
    rng = np.random.default_rng(n)
    a, b, c = rng.standard_normal((n, n)), rng.standard_normal((n, n)), rng.standard_normal((n, n))
    for alpha, beta in ((1.0, 0.0), (-1.0, 1.0), (0.5, -2.0)):
        out = g.dgemm(alpha, a, b, beta, c)
        ref = beta * c + alpha * (a @ b)
        assert np.max(np.abs(out - ref)) <= 1e-12 * n * (1.0 + np.max(np.abs(ref)))


# ---------------------------------------------------------------- row pricing with pi in LDS ----------------
@pytest.mark.parametrize("args,grid", [((20000, 40000, 8, 29), 256), ((20000, 40000, 8, 29), 8), ((4200, 70000, 6, 31), 256),
                                       ((50000, 200000, 50, 20260926), 256)])
def test_lds_pricing_bit_identical_to_oracle(gpu_cls, args, grid):
    """Dense tableau rows are priced by k_price_lds (row tiles of pi in LDS, jagged tile-by-tile streams of the SELL windows): the
    tableau row, the candidate list and upperTheta must be the oracle's bit for bit -- on a two-tile LP, with eight workgroups
    only (every workgroup walks five rounds of windows), on a one-tile LP, and on config 4 (three tiles)."""
    lp = P.sparse_lp(*args)
    g = gpu_cls()
    g.set_option("price_lds_min_windows", 1)
    g.set_option("price_lds_grid", grid)
    g.loadProblem(lp)
    o = oracle(lp)
    rng = np.random.default_rng(17)
    m, n = lp.m, lp.n
    for density in (1.0, 0.5, 0.12):
        k = int(density * m)
        idx = np.sort(rng.choice(m, k, replace=False)).astype(np.int32)
        val = rng.standard_normal(k)
        status = rng.choice([1, 2, 3, 5], size=n + m, p=[0.2, 0.3, 0.45, 0.05]).astype(np.uint8)
        dj = np.where((status & 3) == 2, -1.0, 1.0) * rng.uniform(0, 2, n + m)
        a, b = g.priceRow(idx, val, status, dj), o.price_row_fused(idx, val, status, dj)
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y)
        assert a[4] == b[4]


def test_lds_pricing_engine_runs_match_plain(gpu_cls):
    """The same solve stretch with the LDS form on every pivot (price_lds 2), chosen per batch by the host (1, the default) and
    never (0): identical pivots and solution bits, far enough into the solve for pi to be dense on most pivots; the default
    run must really have switched forms."""
    lp = P.sparse_lp(20000, 40000, 8, seed=29)
    runs = []
    for mode in (2, 1, 0):
        g = gpu_cls()
        g.set_option("price_lds_min_windows", 1)
        g.set_option("price_lds", mode)
        g.loadProblem(lp)
        g.set_option("pivot_rule", 1)
        g.set_option("check_every", 16)
        g.set_option("max_pivots", 0)
        g.set_option("factor_mode", 0)
        g.dual_steps(6000)
        runs.append(g)
    a = runs[2].pivotLog()
    for r in runs[:2]:
        b = r.pivotLog()
        assert len(a) == len(b) and np.array_equal(a["sequenceIn"], b["sequenceIn"]) and np.array_equal(a["sequenceOut"], b["sequenceOut"])
        assert np.array_equal(r.solution(), runs[2].solution())
    st = [r.stats() for r in runs]
    assert st[0]["price_form"] == 1 and st[2]["price_form"] == 0 and st[2]["price_form_switches"] == 0
    print({k: st[1][k] for k in ("price_form", "price_form_switches", "dense_pi_launches", "iterations")})
    assert st[1]["price_form_switches"] >= 1 and st[1]["dense_pi_launches"] > 100, st[1]  # (pi is dense on ~400 of these 6000 pivots)


def test_lds_pricing_with_long_columns(gpu_cls):
    """Power-law column counts: the columns longer than SELL_LONG = 128 entries stay outside the SELL windows and keep their own
    workgroups (priceLongBody), launched beside k_price_lds when the chain prices with pi in LDS.  The same 2 500 pivots with the
    LDS form on every pivot and never: identical pivots and solution bits."""
    lp = P.netlib_shaped_lp(4500, 9000, 150000, seed=23)
    assert (np.diff(lp.col_start) > 128).sum() >= 10, "instance no longer has long columns"
    runs = []
    for mode in (2, 0):
        g = gpu_cls()
        g.set_option("price_lds_min_windows", 1)
        g.set_option("price_lds", mode)
        g.loadProblem(lp)
        g.set_option("pivot_rule", 1)
        g.set_option("factor_mode", 0)
        g.dual_steps(2500)
        runs.append(g)
    assert runs[0].stats()["price_form"] == 1 and runs[1].stats()["price_form"] == 0
    a, b = runs[0].pivotLog(), runs[1].pivotLog()
    assert len(a) == len(b) and np.array_equal(a["sequenceIn"], b["sequenceIn"]) and np.array_equal(a["sequenceOut"], b["sequenceOut"])
    assert np.array_equal(runs[0].solution(), runs[1].solution())


def test_compact_eta_file_against_the_full_file(gpu_cls):
    """The chain's FTRAN through the compact copy of the eta file (option lu_compact_eta, default: the etas over the positions that hold or
    held a structural, the other positions from their own rows of B x = v) against the full-file form on config 4
    from the committed mature basis, 900 pivots each (the eta file grows to 900, ~60 positions convert): the two differ in rounding only
    -- a long shared prefix of pivots, alpha / theta / objective to rounding over it, and no btran / ftran alpha check raised."""
    lp = P.sparse_lp()
    status = (np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "basis_sparse_30000.npy")) & 7).astype(np.uint8)
    logs, stats = [], []
    for compact in (1, 0):
        g = gpu_cls().loadProblem(lp)
        g.setStatusArray(status)
        g.set_option("pivot_rule", 1)
        g.set_option("max_pivots", 0)
        g.set_option("steepest_mode", 1)
        g.set_option("lu_compact_eta", compact)
        assert g.dual_steps(900) == -1
        logs.append(g.pivotLog())
        stats.append(g.stats())
    a, b = logs
    assert stats[0]["eta_compact_slots"] > 10514 and stats[1]["eta_compact_slots"] == 0
    assert stats[0]["exits_alpha_check"] == stats[1]["exits_alpha_check"] == 0
    same = 0
    while same < 900 and a[same]["sequenceIn"] == b[same]["sequenceIn"] and a[same]["sequenceOut"] == b[same]["sequenceOut"]:
        same += 1
    print(f"compact against full eta file from the mature basis: {same} of 900 pivots identical, slots at the end {stats[0]['eta_compact_slots']}")
    assert same >= 200, same
    pre = slice(0, same)
    # (the bar of tests/test_gpu_mature_parity.py for two factorizations of these bases, condition 1e10: alpha to 1e-4; measured 4e-6)
    worst = float(np.max(np.abs(a["alpha"][pre] - b["alpha"][pre]) / np.abs(b["alpha"][pre])))
    print(f"  alpha: worst relative difference over the shared prefix {worst:.2e}")
    assert worst < 1e-4
    assert np.max(np.abs(a["objective"][pre] - b["objective"][pre]) / np.abs(b["objective"][pre])) < 1e-8

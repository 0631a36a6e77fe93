"""The regime the bench times -- config 4 warm-started from the committed mature basis (tests/golden/basis_sparse_30000.npy:
nucleus 10 514, pi dense, pricing by column, LU mode) -- held against checkers that do not share the engine's factorization:

  * the CPU oracle (dense nucleus LU, restated ClpSimplexDual) warm-started from the same statuses: same entering / leaving
    VARIABLES, alpha to 1e-4 relative (measured 1e-5), for as long as no tie is broken by basis position (the oracle's solve is a committed
    record, tests/golden/oracle_cache/, written by tests/golden/make_oracle_cache.py: ~25 minutes of one CPU core);
  * the basis matrix itself (scipy sparse, no factorization at all): residuals of the engine's FTRAN / BTRAN,
    ||B x - v|| and ||B^T y - v||, right after the factorization of that basis and again behind an eta file of 800
    column replacements.

Reference semantics: src/ClpSimplexDual.cpp:1447-1501 (the btran / ftran alpha check), src/ClpFactorization.cpp:2584-3106.
Tolerances are written at each assertion."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from clp_amd import problems as P

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MATURE = os.path.join(ROOT, "tests", "golden", "basis_sparse_30000.npy")
ORACLE_PIVOTS = 400  # below the refactorization interval of this LP (475): one dense LU of order 10 514 on the CPU


@pytest.fixture(scope="module")
def gpu_cls(built):
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clp_amd.engine import ClpGpuSimplex

    return ClpGpuSimplex


def mature_status():
    return (np.load(MATURE) & 7).astype(np.uint8)


@pytest.mark.parametrize("stop_density,need", [(None, 100), (0.012, 400)])
def test_lu_mode_from_the_mature_basis_follows_the_oracle(gpu_cls, stop_density, need):
    """Engine in LU mode (sparse front + dense tail + eta file) against the oracle's dense LU from the same warm start.  The two
    factorizations put the basic variables at different positions and round differently, so the comparison is by variable and
    ends where a near-tie is decided differently.  Two front / tail splits: the engine's DEFAULT (lu_stop_density 0.03: front
    4 869 + tail 5 645) -- measured: the first 147 of 400 pivots identical -- and the round-4 split (0.012: 4 267 + 6 247) --
    measured: all 400 identical.  Asserts the shared prefix and prints it."""
    from oracle.oracle import OracleSimplex

    lp = P.sparse_lp()
    status = mature_status()
    g = gpu_cls().loadProblem(lp)
    g.setStatusArray(status)
    g.set_option("pivot_rule", 1)
    g.set_option("max_pivots", 0)
    g.set_option("steepest_mode", 1)
    if stop_density is not None:
        g.set_option("lu_stop_density", stop_density)
    assert g.dual_steps(ORACLE_PIVOTS) == -1
    st = g.stats()
    assert st["lu_active"] == 1 and st["lu_front"] > 0 and st["lu_tail"] > 0, "the bench's regime is LU mode"
    o = OracleSimplex(lp)
    o.set_option("pivot_rule", 1)
    o.set_option("max_pivots", 0)
    o.set_option("steepest_mode", 1)
    o.set_status(status)
    o.set_option("max_iterations", ORACLE_PIVOTS)
    assert o.dual() == 3
    a, b = g.pivotLog(), o.pivot_log()
    assert len(a) == len(b) == ORACLE_PIVOTS
    if os.environ.get("CLPGPU_DUMP_LOGS"):
        np.save(os.path.join(os.environ["CLPGPU_DUMP_LOGS"], "mature_engine_log.npy"), a)
        np.save(os.path.join(os.environ["CLPGPU_DUMP_LOGS"], "mature_oracle_log.npy"), b)
    same = 0
    while same < ORACLE_PIVOTS and a[same]["sequenceIn"] == b[same]["sequenceIn"] and a[same]["sequenceOut"] == b[same]["sequenceOut"]:
        same += 1
    print(f"mature basis, LU mode (stop density {stop_density or 'default'}) vs oracle: {same} of {ORACLE_PIVOTS} pivots identical (entering and leaving variables)")
    assert same >= need, f"pivot sequences part at pivot {same}"
    if same < ORACLE_PIVOTS:
        _assert_near_tie(gpu_cls, lp, status, stop_density, same, a[same], b[same])
    pre = slice(0, same)
    # alpha, theta, the leaving variable's infeasibility and the objective over the shared prefix.  The two sides solve with
    # different factorizations of bases whose condition numbers pass 1e10 in this stretch (the oracle's plain dense LU carries
    # its own rounding).  Measured on the MI355X over all 400 pivots of the 0.012 split (profiles/r05_mature_parity.txt): alpha
    # 1.2e-5 relative at worst (median 4e-8); theta 1.8e-7 ABSOLUTE at worst -- thetas of this stretch go down to 4e-8, a ratio of
    # two numbers of the size of the dual tolerance, where a relative figure says nothing --; the leaving variable's infeasibility
    # 4e-4 of (1 + value) at worst (basic values reach 1e4 here; median 3e-7); the objective 1.2e-8 relative.
    def report(f, err):
        print(f"  {f}: worst difference {float(err.max()):.2e} at pivot {int(err.argmax())}, median {float(np.median(err)):.2e}")
        return float(err.max())

    assert report("alpha (relative)", np.abs(a["alpha"][pre] - b["alpha"][pre]) / np.abs(b["alpha"][pre])) < 1e-4
    assert report("theta (absolute)", np.abs(a["theta"][pre] - b["theta"][pre]) / (1.0 + 1e3 * np.abs(b["theta"][pre]))) < 1e-6
    assert report("dualOut (of 1 + value)", np.abs(a["dualOut"][pre] - b["dualOut"][pre]) / (1.0 + np.abs(b["dualOut"][pre]))) < 2e-3
    assert report("objective (relative)", np.abs(a["objective"][pre] - b["objective"][pre]) / np.abs(b["objective"][pre])) < 1e-7
    assert np.array_equal(a["numberFlipped"][pre], b["numberFlipped"][pre])
    if same == ORACLE_PIVOTS:
        assert abs(g.objectiveValue() - o.objective) <= 1e-7 * abs(o.objective)


def _assert_near_tie(gpu_cls, lp, status, stop_density, same, rec_engine, rec_oracle):
    """Where the two sides part, the difference has to be a near-tie -- shown with both sides' own numbers at that pivot, not asserted
    from the outside.  Both are brought to the state after `same` identical pivots (the oracle live: its dense LU of
    the nucleus + `same` pivots, about two CPU-minutes); each side's tableau row of the leaving variable is formed from its own BTRAN
    (rho = B^-T e_p through the plug-in call, alpha = rho^T [A | -I]) and its own reduced costs.
      * leaving variables differ: infeasibility^2 / weight of the two rows agree to 1e-6 relative on BOTH sides (a tie of CHUZR);
      * entering variables differ: the breakpoints dj / alpha of the long-step ratio test come in the same order on both sides up to the
        first place where they do not, that place is not behind the earlier of the two choices, and the two breakpoints that swap there are
        closer on both sides than the dual tolerance allows telling apart: |ratio_x - ratio_y| <= 1e-7 (1 / |alpha_x| + 1 / |alpha_y|) (the
        sides' reduced costs agree to ~2e-9 absolute after 147 pivots through bases of condition 1e10; ClpSimplexDual::dualColumn's choice
        inside a lot follows the order of the breakpoints)."""
    from oracle.oracle import OracleSimplex

    g2 = gpu_cls().loadProblem(lp)
    g2.setStatusArray(status)
    o2 = OracleSimplex(lp)
    for s_ in (g2, o2):
        s_.set_option("pivot_rule", 1)
        s_.set_option("max_pivots", 0)
        s_.set_option("steepest_mode", 1)
    if stop_density is not None:
        g2.set_option("lu_stop_density", stop_density)
    o2.set_status(status)
    o2.set_option("max_iterations", same)
    assert g2.dual_steps(same) == -1 and o2.dual() == 3
    pg, po = np.asarray(g2.pivotVariable()), np.asarray(o2.pivot_variable())
    assert set(pg.tolist()) == set(po.tolist()), "the two bases differ although the pivots were identical"
    print(f"first differing pivot {same + 1}: engine {rec_engine['sequenceOut']} -> {rec_engine['sequenceIn']} (alpha {rec_engine['alpha']:.9g}, "
          f"{rec_engine['numberFlipped']} flips), oracle {rec_oracle['sequenceOut']} -> {rec_oracle['sequenceIn']} (alpha {rec_oracle['alpha']:.9g}, "
          f"{rec_oracle['numberFlipped']} flips)")
    if rec_engine["sequenceOut"] != rec_oracle["sequenceOut"]:
        for name, pv, (w, inf) in (("engine", pg, g2.rowWeights()), ("oracle", po, o2.row_weights())):
            r = []
            for var in (rec_engine["sequenceOut"], rec_oracle["sequenceOut"]):
                pos = int(np.nonzero(pv == var)[0][0])
                r.append(inf[pos] / w[pos])
                print(f"  {name}: leaving candidate {var}: infeasibility^2 / weight = {r[-1]:.17g}")
            assert abs(r[0] - r[1]) <= 1e-6 * max(r), f"{name}: not a tie of CHUZR"
        return
    out = int(rec_engine["sequenceOut"])
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(lp.m, lp.n))
    lower = np.concatenate([lp.col_lower, lp.row_lower])
    upper = np.concatenate([lp.col_upper, lp.row_upper])
    sides = {}
    for name, pv, s_, dj in (("engine", pg, g2, np.asarray(g2.reducedCosts())), ("oracle", po, o2, np.asarray(o2.reduced_costs()))):
        unit = np.zeros(lp.m)
        unit[int(np.nonzero(pv == out)[0][0])] = 1.0
        rho = np.asarray(s_.btran(unit))
        alpha = np.concatenate([A.T @ rho, -rho])
        nonbasic = np.ones(lp.m + lp.n, bool)
        nonbasic[pv] = False
        idx = np.nonzero(nonbasic & (np.abs(alpha) > 1e-9) & (dj / np.where(alpha == 0, 1.0, alpha) > 0) & (upper > lower))[0]
        ratio = dj[idx] / alpha[idx]
        order = np.argsort(ratio, kind="stable")
        sides[name] = (idx[order], ratio[order], {int(v): float(r) for v, r in zip(idx, ratio)}, alpha)
    (eo, er, emap, ealpha), (oo, orr, omap, oalpha) = sides["engine"], sides["oracle"]
    ce, co = int(rec_engine["sequenceIn"]), int(rec_oracle["sequenceIn"])
    for var in (ce, co):
        print(f"  candidate {var}: engine alpha {ealpha[var]:.12g} ratio {emap[var]:.12g} (breakpoint no. {int(np.nonzero(eo == var)[0][0])}); "
              f"oracle alpha {oalpha[var]:.12g} ratio {omap[var]:.12g} (breakpoint no. {int(np.nonzero(oo == var)[0][0])})")
        assert abs(abs(ealpha[var]) - abs(oalpha[var])) <= 1e-5 * abs(oalpha[var]), "the two sides do not see the same tableau row"
    n = min(len(eo), len(oo))
    first = next((i for i in range(n) if eo[i] != oo[i]), None)
    assert first is not None, "identical breakpoint orders, different choices: not a tie -- a difference in the ratio test's logic"
    earlier = min(int(np.nonzero(eo == ce)[0][0]), int(np.nonzero(eo == co)[0][0]), int(np.nonzero(oo == ce)[0][0]), int(np.nonzero(oo == co)[0][0]))
    x, y = int(eo[first]), int(oo[first])
    print(f"  the breakpoint orders part at no. {first} (the earlier choice is no. {earlier}): engine has {x} there, the oracle {y}; "
          f"engine ratios {emap.get(x)} / {emap.get(y)}, oracle ratios {omap.get(x)} / {omap.get(y)}")
    assert first <= earlier, "the orders part only behind the earlier choice: the choices differ for another reason"
    # a breakpoint dj / alpha is known to dj's accuracy over |alpha|; reduced costs closer than the dual tolerance (1e-7) are the same
    # number to ClpSimplexDual (and the two sides' differ by ~2e-9 here): two breakpoints whose distance is below
    # dualTolerance (1 / |alpha_x| + 1 / |alpha_y|) have no order that rounding does not decide
    for name_, m_, al_ in (("engine", emap, ealpha), ("oracle", omap, oalpha)):
        assert x in m_ and y in m_
        gap, room = abs(m_[x] - m_[y]), 1.0e-7 * (1.0 / abs(al_[x]) + 1.0 / abs(al_[y]))
        print(f"  {name_}: breakpoints {x} / {y}: |alpha| {abs(al_[x]):.6g} / {abs(al_[y]):.6g}, distance {gap:.3g} in ratio units against {room:.3g} "
              "= the dual tolerance over the two |alpha|")
        assert gap <= room, "the swapped breakpoints are not a near-tie"


def _basis(lp, pv):
    m, n = lp.m, lp.n
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(m, n))
    full = sp.hstack([A, -sp.identity(m, format="csc")]).tocsc()
    return full[:, pv].tocsc()


def _column(lp, q):
    col = np.zeros(lp.m)
    if q >= lp.n:
        col[q - lp.n] = -1.0
    else:
        col[lp.row[lp.col_start[q]:lp.col_start[q + 1]]] = lp.elem[lp.col_start[q]:lp.col_start[q + 1]]
    return col


def test_solves_at_the_mature_basis_have_small_residuals(gpu_cls):
    """No second factorization involved: with B assembled by scipy from the engine's pivotVariable, the FTRAN / BTRAN
    of the default LU mode must satisfy B x = v and B^T y = v to 1e-9 of the solution's scale -- right after the
    factorization (front + MFMA tail inversion + polish) and behind 800 product-form etas."""
    lp = P.sparse_lp()
    m, n = lp.m, lp.n
    status = mature_status()
    g = gpu_cls().loadProblem(lp)
    g.set_option("lu_max_pivots", 2000)
    rc, pv = g.factorize(status)
    assert rc == 0, g.lastError()
    st = g.stats()
    assert st["lu_active"] == 1 and st["lu_tail"] > 2000
    pv = np.asarray(pv).copy()
    rng = np.random.default_rng(20260926)

    def residuals(tag):
        B = _basis(lp, pv)
        worst = 0.0
        for kind in range(3):
            if kind == 0:
                v = rng.standard_normal(m)
            elif kind == 1:
                v = rng.standard_normal(m) * (rng.random(m) < 0.01)  # a sparse right-hand side (an entering column's shape)
            else:
                v = np.zeros(m)
                v[int(rng.integers(0, m))] = 1.0  # the pivot's unit vector
            x, y = g.ftran(v), g.btran(v)
            rf = float(np.max(np.abs(B @ x - v)) / (np.max(np.abs(v)) + np.max(np.abs(B)) * np.max(np.abs(x))))
            rb = float(np.max(np.abs(B.T @ y - v)) / (np.max(np.abs(v)) + np.max(np.abs(B)) * np.max(np.abs(y))))
            worst = max(worst, rf, rb)
        print(f"{tag}: worst normwise backward error of FTRAN / BTRAN {worst:.2e}")
        return worst

    assert residuals("after the factorization") < 1e-9
    basic = set(int(s) for s in pv)
    lastp = -1
    done = 0
    while done < 800:
        q = int(rng.integers(0, n + m))
        if q in basic:
            continue
        w = g.ftran(_column(lp, q))
        cand = np.nonzero(np.abs(w) > 0.1 * np.max(np.abs(w)))[0]
        if not len(cand):
            continue
        p = lastp if (done % 7 == 6 and lastp >= 0 and abs(w[lastp]) > 0.05 * np.max(np.abs(w))) else int(cand[rng.integers(0, len(cand))])
        assert g.replaceColumn(p, q) == 0
        basic.discard(int(pv[p]))
        basic.add(q)
        pv[p] = q
        lastp = p
        done += 1
        if done in (100, 400):
            assert residuals(f"behind {done} etas") < 1e-8
    assert g.pivots() == 800
    assert residuals("behind 800 etas") < 1e-8
    assert np.array_equal(np.asarray(g.pivotVariable()), pv)

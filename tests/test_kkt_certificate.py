"""The KKT certificate the time-to-optimal ladder accepts where no independent solver finishes (tools/kkt_certificate.py):
it must accept the optimum of LPs whose optimum is known two ways (the CPU oracle and HiGHS) and refuse points that are
feasible but not optimal, optimal-looking but infeasible, or carry the wrong duals."""
import numpy as np
import pytest
import scipy.sparse as sp

from clp_amd import problems as P
from tools.kkt_certificate import certify, row_duals_from_engine


def solved(lp, rule=1):
    from oracle.oracle import OracleSimplex

    o = OracleSimplex(lp)
    o.set_option("pivot_rule", rule)
    assert o.dual() == 0
    return o


@pytest.mark.parametrize("args", [(60, 200, 6, 3), (300, 1200, 8, 11), (200, 260, 7, 5)])
def test_certificate_accepts_the_optimum_and_agrees_with_highs(built, args):
    from scipy.optimize import linprog

    lp = P.sparse_lp(*args)
    o = solved(lp)
    cert = certify(lp, o.solution(), row_duals_from_engine(lp, o))
    assert cert["optimal"], cert
    assert abs(cert["primal_objective"] - o.objective) <= 1e-9 * (1 + abs(o.objective))
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(lp.m, lp.n))
    r = linprog(lp.obj, A_ub=sp.vstack([A, -A]).tocsr(), b_ub=np.concatenate([lp.row_upper, -lp.row_lower]),
                bounds=np.column_stack([lp.col_lower, lp.col_upper]), method="highs-ds")
    assert r.status == 0 and abs(r.fun - cert["primal_objective"]) <= 1e-8 * (1 + abs(r.fun))


def test_certificate_refuses_what_is_not_optimal(built):
    lp = P.sparse_lp(300, 1200, 8, 11)
    o = solved(lp)
    x, y = o.solution()[: lp.n].copy(), row_duals_from_engine(lp, o)
    # a feasible interior move along a column that is nonbasic at its lower bound with a positive reduced cost: objective rises
    dj = np.asarray(o.reduced_costs())[: lp.n]
    j = int(np.argmax(dj))
    assert dj[j] > 1e-3
    worse = x.copy()
    worse[j] += 1e-3
    c1 = certify(lp, worse, y)
    assert not c1["optimal"] and (c1["duality_gap_relative"] > 1e-8 or c1["primal_infeasibility"] > 1e-7 or c1["dual_infeasibility"] > 1e-7)
    # the right point with the duals of another basis
    c2 = certify(lp, x, np.zeros(lp.m))
    assert not c2["optimal"]
    # an infeasible point that looks cheaper
    cheaper = x.copy()
    cheaper[np.argmax(x)] -= 1.0
    c3 = certify(lp, cheaper, y)
    assert not c3["optimal"] and c3["primal_infeasibility"] > 1e-7
    # an early (dual feasible, primal infeasible) iterate of the dual simplex
    from oracle.oracle import OracleSimplex

    early = OracleSimplex(lp)
    early.set_option("pivot_rule", 1)
    early.set_option("max_iterations", 20)
    assert early.dual() == 3
    c4 = certify(lp, early.solution(), row_duals_from_engine(lp, early))
    assert not c4["optimal"] and c4["primal_infeasibility"] > 1e-7

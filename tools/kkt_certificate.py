#!/usr/bin/env python3
"""Optimality certificate computed OUTSIDE the engine (numpy / scipy, f64): given the LP and a returned point -- the
structural values x and the row duals y -- decide optimality from first principles, with no factorization and none of the
engine's own bookkeeping:

    primal feasibility    l <= x <= u,  rl <= A x <= ru                                   (to `feas_tol`, relative to the bound)
    dual feasibility      d = c - A^T y recomputed here;  d_j >= 0 where x_j sits at its lower bound only, <= 0 at its upper
                          bound only, = 0 strictly between;  the same for y_i against the row activity    (to `feas_tol`)
    zero duality gap      c^T x  ==  sum_i y_i * (rl_i if y_i > 0 else ru_i) + sum_j d_j * (l_j if d_j > 0 else u_j)
                                                                                           (to `gap_tol`, relative)

A point passing all three is optimal (weak duality): this is what the time-to-optimal ladder accepts on the rungs no
independent solver finishes (HiGHS hit its ten-hour limit on rungs 7 000 and 10 000: profiles/r04_highs_rungs_7000_10000_time_limit.jsonl).
Convention: minimisation, slack form A x - s = 0 with rl <= s <= ru, so the reduced cost of slack i is y_i
(src/ClpSimplex.cpp:3442-3474: slack column -e_i).  Test infrastructure: used by tests/, tools/ladder.py and bench.py's ladder leg."""
import numpy as np
import scipy.sparse as sp


def certify(lp, x, y, feas_tol=1e-7, gap_tol=1e-8):
    m, n = int(lp.m), int(lp.n)
    A = sp.csc_matrix((np.asarray(lp.elem, dtype=np.float64), np.asarray(lp.row), np.asarray(lp.col_start)), shape=(m, n))
    x = np.asarray(x, dtype=np.float64)[:n]
    y = np.asarray(y, dtype=np.float64)[:m]
    c = np.asarray(lp.obj, dtype=np.float64)
    lo, up = np.asarray(lp.col_lower, dtype=np.float64), np.asarray(lp.col_upper, dtype=np.float64)
    rl, ru = np.asarray(lp.row_lower, dtype=np.float64), np.asarray(lp.row_upper, dtype=np.float64)
    act = A @ x
    d = c - A.T @ y

    def below(v, b):  # violation of v >= b, relative to the bound's size; infinite bounds never bind
        return np.where(np.isfinite(b), np.maximum(b - v, 0.0) / (1.0 + np.abs(np.where(np.isfinite(b), b, 0.0))), 0.0)

    primal = max(float(below(x, lo).max(initial=0.0)), float(below(-x, -up).max(initial=0.0)),
                 float(below(act, rl).max(initial=0.0)), float(below(-act, -ru).max(initial=0.0)))

    def dual_violation(v, lower, upper, dj):
        # where the variable may still move up (not at its upper bound) a negative dj is an improving direction, and vice versa
        span = 1.0 + np.abs(v)
        at_lower = np.isfinite(lower) & (v - lower <= feas_tol * span)
        at_upper = np.isfinite(upper) & (upper - v <= feas_tol * span)
        scale = 1.0 + np.abs(dj)
        viol = np.zeros_like(dj)
        viol = np.where(~at_upper, np.maximum(viol, np.maximum(-dj, 0.0)), viol)  # can increase: dj must be >= 0
        viol = np.where(~at_lower, np.maximum(viol, np.maximum(dj, 0.0)), viol)   # can decrease: dj must be <= 0
        return float((viol / scale).max(initial=0.0))

    dual = max(dual_violation(x, lo, up, d), dual_violation(act, rl, ru, y))
    primal_obj = float(c @ x) + float(getattr(lp, "obj_offset", 0.0) or 0.0)

    def bound_term(mult, lower, upper):
        # a multiplier pushing against an infinite bound makes the dual objective -infinity: reported as a gap
        b = np.where(mult > 0.0, lower, upper)
        live = np.abs(mult) > 0.0
        if np.any(live & ~np.isfinite(b)):
            tiny = np.abs(mult) <= feas_tol * (1.0 + np.abs(mult))
            if np.any(live & ~np.isfinite(b) & ~tiny):
                return -np.inf
            b = np.where(np.isfinite(b), b, 0.0)
        return float(np.sum(np.where(live, mult * np.where(np.isfinite(b), b, 0.0), 0.0)))

    dual_obj = bound_term(y, rl, ru) + bound_term(d, lo, up) + float(getattr(lp, "obj_offset", 0.0) or 0.0)
    gap = abs(primal_obj - dual_obj) / (1.0 + abs(primal_obj)) if np.isfinite(dual_obj) else np.inf
    return {"certificate": "kkt", "primal_infeasibility": primal, "dual_infeasibility": dual, "duality_gap_relative": float(gap),
            "primal_objective": primal_obj, "dual_objective": float(dual_obj), "feas_tol": feas_tol, "gap_tol": gap_tol,
            "optimal": bool(primal <= feas_tol and dual <= feas_tol and gap <= gap_tol)}


def row_duals_from_engine(lp, engine):
    """y of `certify` from an engine / oracle object: the reduced costs of the row slacks (sequences n .. n+m-1)."""
    dj = engine.reducedCosts() if hasattr(engine, "reducedCosts") else engine.reduced_costs()
    return np.asarray(dj, dtype=np.float64)[int(lp.n):]

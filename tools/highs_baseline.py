#!/usr/bin/env python3
"""Independent optimum + CPU context for the bench LPs: HiGHS' serial dual simplex (the one bundled
with scipy), presolve off, explicit slack columns.  Not the reference and not what bench.py times --
an independent solver that says (1) what the optimal objective of the bench LP is (committed under
tests/golden/bench_optima.json, the number the engine's time-to-optimal leg is checked against) and
(2) where a production CPU dual simplex with a sparse LU sits on the same LP.

usage: highs_baseline.py [sparse|netlib|dense] [time limit s] [--presolve] [--out file.json]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from scipy.optimize import linprog

from clp_amd import problems as P

args = [a for a in sys.argv[1:] if not a.startswith("--")]
which = args[0] if args else "sparse"
limit = float(args[1]) if len(args) > 1 else 60.0
presolve = "--presolve" in sys.argv
out = None
if "--out" in sys.argv:
    out = sys.argv[sys.argv.index("--out") + 1]
lp = {"sparse": P.sparse_lp, "netlib": P.netlib_shaped_lp, "dense": P.dense_lp}[which]()
m, n = lp.m, lp.n
A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(m, n))
Aeq = sp.hstack([A, -sp.identity(m, format="csc")]).tocsr()  # A x - s = 0, row bounds on s


def inf(v):
    out = np.array(v, dtype=float)
    out[out > 1e29] = np.inf
    out[out < -1e29] = -np.inf
    return out


bounds = np.column_stack([np.concatenate([inf(lp.col_lower), inf(lp.row_lower)]),
                          np.concatenate([inf(lp.col_upper), inf(lp.row_upper)])])
c = np.concatenate([lp.obj, np.zeros(m)])
t0 = time.time()
r = linprog(c, A_eq=Aeq, b_eq=np.zeros(m), bounds=bounds, method="highs-ds",
            options={"presolve": presolve, "time_limit": limit, "disp": True})
dt = time.time() - t0
rec = {"lp": lp.name, "m": m, "n": n, "nnz": int(lp.col_start[-1]), "solver": f"HiGHS dual simplex (scipy {__import__('scipy').__version__})",
       "presolve": presolve, "status": int(r.status), "message": str(r.message)[:80], "iterations": int(r.nit),
       "seconds": round(dt, 2), "iterations_per_s": round(r.nit / dt, 1),
       "objective": (float(r.fun) if r.fun is not None else None), "cores": 1}
print(json.dumps(rec))
if out:
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")

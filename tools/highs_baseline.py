#!/usr/bin/env python3
"""Context for the CPU baseline: HiGHS' serial dual simplex (the one bundled with scipy) on the bench LP.
Not the reference and not what bench.py reports -- only an independent data point showing where a
production CPU dual simplex sits on the same LP (presolve off, explicit slack columns, time limit)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from scipy.optimize import linprog

from clp_amd import problems as P

limit = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
lp = P.sparse_lp()
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
            options={"presolve": False, "time_limit": limit, "disp": False})
dt = time.time() - t0
print(f"HiGHS dual simplex (scipy {__import__('scipy').__version__}), presolve off: {r.nit} iterations in {dt:.1f} s "
      f"= {r.nit / dt:.0f} iterations/s (status {r.status}: {r.message[:40]})")

"""Prints what the 2(m+n) kick did on N-Queens 100 under Dantzig pricing (engine, option perturbation = 100 vs 102)."""
import json
import sys
import time

sys.path.insert(0, ".")
from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex

lp = P.nqueens(100)
for value in (102, 100, 50):
    g = ClpGpuSimplex().loadProblem(lp)
    g.set_option("pivot_rule", 0)
    g.set_option("perturbation", value)
    t = time.time()
    status = g.dual()
    print(json.dumps({"lp": "nqueens(100)", "m": lp.m, "n": lp.n, "two_m_plus_n": 2 * (lp.m + lp.n), "pivot_rule": "dantzig", "perturbation": value,
                      "status": status, "iterations": g.numberIterations(), "objective": g.objectiveValue(),
                      "perturbations": g.stats()["perturbations"], "seconds": round(time.time() - t, 2)}))

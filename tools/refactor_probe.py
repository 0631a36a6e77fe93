#!/usr/bin/env python3
"""Where the wall clock of a mature config-4 stretch goes: pivots against status checks (refactorization + resync), from the committed
mature basis, 6000 pivots (argv[2]: another count), option log_level 2 (per status check: its wall time; per LU factorization: host front / tail inversion / build)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex

lp = P.sparse_lp()
g = ClpGpuSimplex(0).loadProblem(lp)
g.set_option("pivot_rule", 1)
g.set_option("max_pivots", 0)
g.set_option("log_level", 2)
for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else "").split(",")):
    k, v = kv.split("=")
    g.set_option(k, float(v))
g.setStatusArray(np.load(os.path.join(ROOT, "tests", "golden", "basis_sparse_30000.npy")))
g.dual_steps(100)
torch.cuda.synchronize()
s0 = g.stats()
t0 = time.perf_counter()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
g.dual_steps(N)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
s1 = g.stats()
print("PROBE %d pivots in %.3f s = %.1f it/s; LU factorizations %d: host front %.0f ms, tail inversion %.0f ms, build %.0f ms; refactorizations %d; nucleus %d tail %d; objective %.6f" % (
    N, dt, N / dt, s1["lu_factorizations"] - s0["lu_factorizations"], s1["lu_front_ms"] - s0["lu_front_ms"], s1["lu_invert_ms"] - s0["lu_invert_ms"],
    s1["lu_build_ms"] - s0["lu_build_ms"], s1["refactorizations"] - s0["refactorizations"], s1["nucleus"], s1["lu_tail"], g.objectiveValue()))
print("PROBE what sent the loop to its status checks: scheduled (eta file / forced) %d, alpha check %d, objective backwards %d, bad update %d; pricing form LDS %d (%d switches)" % (
    s1["exits_scheduled"] - s0["exits_scheduled"], s1["exits_alpha_check"] - s0["exits_alpha_check"], s1["exits_backwards"] - s0["exits_backwards"],
    s1["exits_bad_update"] - s0["exits_bad_update"], s1["price_form"], s1["price_form_switches"]))

#!/usr/bin/env python3
"""Option free_nonbasic on LPs with many free columns (sparse_lp with a tenth of its columns freed: tests/test_oracle_free.py), engine against
oracle under both pivot rules: status, pivots, first difference, free counters.  On the GPU box: python tools/free_many.py [engine option=value ...]
(e.g. factor_mode=1: the LU form of the factorization under the free path)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex
from oracle.oracle import OracleSimplex
EXTRA = dict((a.split("=")[0], float(a.split("=")[1])) for a in sys.argv[1:])
for (m, n) in ((300, 1200), (600, 2400)):
    lp = P.sparse_lp(m, n, 8, 11)
    free = np.random.default_rng(11).choice(lp.n, lp.n // 10, replace=False)
    lp = type(lp)(lp)
    lp.col_lower, lp.col_upper = lp.col_lower.copy(), lp.col_upper.copy()
    lp.col_lower[free], lp.col_upper[free] = -1e30, 1e30
    for rule in (1, 0):
        o = OracleSimplex(lp); g = ClpGpuSimplex(0).loadProblem(lp)
        for s in (o, g):
            s.set_option("pivot_rule", rule); s.set_option("free_nonbasic", 1); s.set_option("max_iterations", 20000)
        g.set_option("fake_bound_cleanup", 1)
        for k, v in EXTRA.items():
            g.set_option(k, v)
        so, sg = o.dual(), g.dual()
        lo, lg = o.pivot_log(), g.pivotLog()
        k = min(len(lo), len(lg))
        d = np.nonzero((lo["sequenceIn"][:k] != lg["sequenceIn"][:k]) | (lo["sequenceOut"][:k] != lg["sequenceOut"][:k]))[0]
        st = g.stats()
        print(m, n, "rule", rule, "status", so, sg, "iterations", len(lo), len(lg), "first difference", int(d[0]) + 1 if len(d) else None,
              "oracle rows/entered", o.free_first_rows, o.free_entered, "engine", st["free_first_rows"], st["free_entered"], "objective", o.objective, g.objectiveValue(), flush=True)

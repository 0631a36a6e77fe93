#!/usr/bin/env python3
"""engine with an option set vs the oracle: pivot sequences on a few LPs (development aid for kernel variants)
    python tools/variant_check.py dc_variant=1"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex
from oracle.oracle import OracleSimplex

opts = dict(kv.split("=") for kv in sys.argv[1:])
cases = [("sparse1500", P.sparse_lp(1500, 6000, 10, 31), 0), ("dense300", P.dense_lp(300, 400, 5), 0), ("sparse300", P.sparse_lp(300, 1200, 8, 11), 0),
         ("config4_600", P.sparse_lp(), 600)]
for name, lp, limit in cases:
    g = ClpGpuSimplex(0).loadProblem(lp)
    o = OracleSimplex(lp)
    for x in (g, o):
        x.set_option("pivot_rule", 1)
        if limit:
            x.set_option("max_pivots", 0)
    for k, v in opts.items():
        g.set_option(k, float(v))
    if limit:
        o.set_option("max_iterations", limit)
        sg, so = g.dual_steps(limit), o.dual()
    else:
        sg, so = g.dual(), o.dual()
    lg, lo = g.pivotLog(), o.pivot_log()
    n = min(len(lg), len(lo))
    same = len(lg) == len(lo) and bool((lg["sequenceIn"] == lo["sequenceIn"]).all() and (lg["sequenceOut"] == lo["sequenceOut"]).all())
    first = -1 if same else int(np.nonzero((lg["sequenceIn"][:n] != lo["sequenceIn"][:n]) | (lg["sequenceOut"][:n] != lo["sequenceOut"][:n]))[0][:1].sum())
    print(name, opts, "status", sg, so, "pivots", len(lg), len(lo), "identical", same, "first diff", first, "obj", g.objectiveValue(), o.objective, flush=True)

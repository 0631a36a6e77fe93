#!/usr/bin/env python3
"""Ratio test in the mature regime of config 4: how often the working-set path answers and how often the full list is walked
(CLPGPU_DEBUG_STATS counters of k_dual_column), over 2000 pivots from the committed mature basis; candidate-list lengths and flips from
the pivot log.    CLPGPU_DEBUG_STATS=1 python tools/dc_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CLPGPU_DEBUG_STATS"] = "1"
import numpy as np
import torch  # noqa: F401

from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex

lp = P.sparse_lp()
g = ClpGpuSimplex(0).loadProblem(lp)
g.set_option("pivot_rule", 1)
g.set_option("max_pivots", 0)
g.setStatusArray(np.load(os.path.join(ROOT, "tests", "golden", "basis_sparse_30000.npy")))
g.dual_steps(200)
g.stats()
print("---- after warm-up; the next line covers warm-up + 2000 pivots (cumulative counters)", flush=True)
g.dual_steps(2000)
g.stats()
log = g.pivotLog()[200:2200]
nc = log["reserved"] & ((1 << 30) - 1)
print("candidates per pivot: median %d p90 %d max %d; flips per pivot: median %d p90 %d max %d" % (
    np.median(nc), np.percentile(nc, 90), nc.max(), np.median(log["numberFlipped"]), np.percentile(log["numberFlipped"], 90), log["numberFlipped"].max()))

#!/usr/bin/env python3
"""Where the by-column pricing kernel's time goes with a dense pi (config 4, mature basis): clpgpu_debug_price_bench launches the kernel
with parts switched off (1 candidate-count atomics, 2 by-column scatter of the tableau row -> SELL order, 4 status / dj gathers,
8 the matrix sweep, 16 the gather of pi) and times each form with HIP events.   python tools/price_probe.py [reps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401

from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lp = P.sparse_lp()
g = ClpGpuSimplex(0).loadProblem(lp)
g.set_option("pivot_rule", 1)
g.set_option("max_pivots", 0)
g.setStatusArray(np.load(os.path.join(ROOT, "tests", "golden", "basis_sparse_30000.npy")))
assert g.dual_steps(32) == -1
masks = [0, 1, 2, 3, 4, 7, 8, 15, 16, 23, 24, 31]
us = g.debugPriceBench(masks, reps)
nnz = len(lp.elem)
for mk, t in zip(masks, us):
    off = [name for bit, name in ((1, "atomics"), (2, "column scatter"), (4, "status/dj gathers"), (8, "matrix sweep"), (16, "pi gather")) if mk & bit]
    print(json.dumps({"mask": mk, "off": off, "us_per_launch": round(float(t), 2),
                      "GBps_if_full_stream": round(12.0 * nnz / (t * 1e-6) / 1e9, 1)}), flush=True)
# the context still solves after the probe
assert g.dual_steps(64) == -1
print(json.dumps({"after_probe_iterations": g.numberIterations(), "objective": g.objectiveValue()}))

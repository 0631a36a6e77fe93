#!/usr/bin/env python3
"""Engine and oracle state after the first N pivots of a fuzz LP (tools/fuzz_gpu.py): DSE weights, infeasibility array, basic values.
    fuzz_state.py seed:rule:pivots ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401

from clp_amd.engine import ClpGpuSimplex
from oracle.oracle import OracleSimplex
from test_oracle_fuzz import make

np.set_printoptions(linewidth=250, precision=8)
for spec in sys.argv[1:]:
    seed, rule, piv = (int(x) for x in spec.split(":"))
    lp = make(np.random.default_rng(7000 + seed))
    o = OracleSimplex(lp)
    o.set_option("pivot_rule", rule)
    o.set_option("max_iterations", piv)
    so = o.dual()
    g = ClpGpuSimplex(0).loadProblem(lp)
    g.set_option("pivot_rule", rule)
    sg = g.dual_steps(piv)
    wo, io = o.row_weights()
    wg, ig = g.rowWeights()
    print(f"== seed {seed} rule {rule} after {piv} pivots: oracle status {so} its {o.iterations} | engine status {sg} its {g.numberIterations()}")
    print("pivotVariable O", o.pivot_variable().tolist())
    print("pivotVariable E", g.pivotVariable().tolist())
    print("weights O", wo)
    print("weights E", wg)
    print("infeas  O", io)
    print("infeas  E", ig)
    print("sol O", o.solution())
    print("sol E", g.solution())
    print("status O", o.status().tolist())
    print("status E", g.statusArray().tolist())
    print("dj O", o.reduced_costs())
    print("dj E", g.reducedCosts())

#!/usr/bin/env python3
"""Run the engine on a bench LP and save the status array at given pivot counts (for offline study of
the nucleus: LU fill under different orderings).  usage: dump_basis.py workload pivots[,pivots...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401

from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex

which = sys.argv[1]
marks = [int(x) for x in sys.argv[2].split(",")]
lp = {"sparse": P.sparse_lp, "netlib": P.netlib_shaped_lp}[which]()
g = ClpGpuSimplex(0).loadProblem(lp)
g.set_option("pivot_rule", 1)
g.set_option("check_every", 16)
g.set_option("max_pivots", 0)
os.makedirs("gpurun_out", exist_ok=True)
done = 0
for mk in marks:
    st = g.dual_steps(mk - done)
    done = g.numberIterations()
    np.save(f"gpurun_out/basis_{which}_{done}.npy", np.asarray(g.statusArray(), dtype=np.uint8))
    print(which, done, st, g.stats()["nucleus"], g.objectiveValue(), flush=True)
    if st != -1:
        break

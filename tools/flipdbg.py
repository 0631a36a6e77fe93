#!/usr/bin/env python3
"""development aid: flip statistics and per-kernel times of the first pivots of config 4
    CLPGPU_DEBUG_STATS=1 python tools/flipdbg.py flip_scatter=1"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex

opts = dict(kv.split("=") for kv in sys.argv[1:])
pivots = int(opts.pop("pivots", 400))
lp = P.sparse_lp()
g = ClpGpuSimplex(0).loadProblem(lp)
g.set_option("pivot_rule", 1)
g.set_option("max_pivots", 0)
g.set_option("timing", 2)
for k, v in opts.items():
    g.set_option(k, float(v))
g.dual_steps(pivots)
st = g.stats()
fl = g.pivotLog()["numberFlipped"]
print(opts, "pivots", len(fl), "flips mean", fl.mean(), "max", fl.max(), "hist", np.histogram(fl, [0, 1, 2, 8, 32, 128, 512, 1024, 4096, 1 << 20])[0].tolist())
kt = g.kernelTimes()
print({k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in kt.items() if "flip" in k or "dj_flags" in k}, flush=True)

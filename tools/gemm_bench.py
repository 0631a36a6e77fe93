#!/usr/bin/env python3
"""The engine's own MFMA f64 GEMM on a square problem (run under rocprofv3 --kernel-trace --stats for the kernel time):
    python tools/gemm_bench.py [n] [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = ClpGpuSimplex(0).loadProblem(P.sparse_lp(300, 1200, 8, seed=11))
rng = np.random.default_rng(0)
a, b, c = rng.standard_normal((n, n)), rng.standard_normal((n, n)), np.zeros((n, n))
for _ in range(rep):
    out = g.dgemm(1.0, a, b, 0.0, c)
print("n", n, "flop", 2.0 * n ** 3, "max |err| vs numpy", float(np.max(np.abs(out - a @ b))))

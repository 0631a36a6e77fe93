#!/usr/bin/env python3
"""PCIe-inclusive rate of the plug-in level pricing call (host buffers in, host buffers out) at config 4."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex

lp = P.sparse_lp()
g = ClpGpuSimplex().loadProblem(lp)
m, n = lp.m, lp.n
rng = np.random.default_rng(3)
idx = np.sort(rng.choice(m, 2000, replace=False)).astype(np.int32)
val = rng.standard_normal(2000)
status = np.full(n + m, 3, np.uint8)
dj = np.ones(n + m)
g.priceRow(idx, val, status, dj)
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    g.priceRow(idx, val, status, dj)
dt = (time.perf_counter() - t0) / reps
alg = 12.0 * len(lp.elem) + 4.0 * (n + 1) + n + 8.0 * m
print(f"clpgpu_price_row, host buffers in/out: {dt * 1e3:.2f} ms per call = {alg / dt / 1e9:.1f} GB/s of algorithmic bytes")

#!/usr/bin/env python3
"""Pivot logs of engine and oracle side by side for chosen LPs of the fuzz (tools/fuzz_gpu.py):  fuzz_show.py seed:rule[:opt=value,...] ..."""
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

for spec in sys.argv[1:]:
    parts = spec.split(":")
    seed, rule = int(parts[0]), int(parts[1])
    opts = dict(kv.split("=") for kv in parts[2].split(",")) if len(parts) > 2 and parts[2] else {}
    lp = make(np.random.default_rng(7000 + seed))
    o = OracleSimplex(lp)
    g = ClpGpuSimplex(0).loadProblem(lp)
    for s in (o, g):
        s.set_option("pivot_rule", rule)
        s.set_option("log_level", 3)
        for k, v in opts.items():
            s.set_option(k, float(v))
    g.set_option("fake_bound_cleanup", 1)
    so, sg = o.dual(), g.dual()
    lo, lg = o.pivot_log(), g.pivotLog()
    print(f"== seed {seed} rule {rule} {opts}: m {lp.m} n {lp.n}  oracle status {so} its {len(lo)} refac {o.refactorizations} | engine status {sg} its {len(lg)} refac {g.stats()['refactorizations']}")
    print("   col kinds: lower", lp.col_lower.tolist(), "upper", lp.col_upper.tolist())
    for i in range(max(len(lo), len(lg))):
        a = lo[i] if i < len(lo) else None
        b = lg[i] if i < len(lg) else None
        fmt = lambda r: "-" if r is None else f"in {r['sequenceIn']:3d} out {r['sequenceOut']:3d} row {r['pivotRow']:3d} flips {r['numberFlipped']:2d} theta {r['theta']:.6g} alpha {r['alpha']:.6g} dualOut {r['dualOut']:.6g} obj {r['objective']:.8g}"
        print(f"  {i + 1:3d}  O {fmt(a)}\n       E {fmt(b)}")
    print("   oracle final status", o.status().tolist())
    print("   engine final status", g.statusArray().tolist())

#!/usr/bin/env python3
"""free_nonbasic, one LP of the fuzz, engine beside oracle pivot by pivot:  python tools/free_dbg.py seed rule [log level] [key=value ...]"""
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

seed, rule = int(sys.argv[1]), int(sys.argv[2])
level = int(sys.argv[3]) if len(sys.argv) > 3 else 0
opts = dict((a.split("=")[0], float(a.split("=")[1])) for a in sys.argv[4:])
lp = make(np.random.default_rng(7000 + seed))
free = np.flatnonzero((lp.col_lower < -1e20) & (lp.col_upper > 1e20))
print("seed", seed, "rule", rule, "m", lp.m, "n", lp.n, "free columns", free.tolist(), flush=True)
o = OracleSimplex(lp)
g = ClpGpuSimplex(0).loadProblem(lp)
for s in (o, g):
    s.set_option("pivot_rule", rule)
    s.set_option("max_iterations", 20000)
    s.set_option("free_nonbasic", 1)
    s.set_option("log_level", level)
    for k, v in opts.items():
        s.set_option(k, v)
g.set_option("fake_bound_cleanup", 1)
sys.stderr.write("---- oracle\n")
sys.stderr.flush()
so = o.dual()
sys.stderr.write("---- engine\n")
sys.stderr.flush()
sg = g.dual()
lo, lg = o.pivot_log(), g.pivotLog()
st = g.stats()
print("status", so, sg, "iterations", len(lo), len(lg), "free rows / entered", (o.free_first_rows, o.free_entered), (st["free_first_rows"], st["free_entered"]),
      "refactorizations", o.refactorizations if hasattr(o, "refactorizations") else None, st.get("refactorizations"))
for i in range(max(len(lo), len(lg))):
    a = lo[i] if i < len(lo) else None
    b = lg[i] if i < len(lg) else None
    f = lambda r: "-" if r is None else f"in {int(r['sequenceIn']):4d} out {int(r['sequenceOut']):4d} row {int(r['pivotRow']):3d} flips {int(r['numberFlipped']):2d} alpha {float(r['alpha']): .6e} theta {float(r['theta']): .6e} dualOut {float(r['dualOut']): .4e} obj {float(r['objective']): .8e}"
    mark = " " if (a is not None and b is not None and a["sequenceIn"] == b["sequenceIn"] and a["sequenceOut"] == b["sequenceOut"]) else "*"
    print(f"{i + 1:4d}{mark} O {f(a)}\n      E {f(b)}")

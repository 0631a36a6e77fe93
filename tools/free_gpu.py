#!/usr/bin/env python3
"""Option free_nonbasic (isFree nonbasics as in the reference: src/ClpSimplexDual.cpp:3005-3055, :4058-4179) -- engine against oracle
on the LPs of the CPU fuzz that have free columns: status, pivots, the two free counters, first differing pivot.  On the GPU box:
    python tools/free_gpu.py [first seed] [count] [detail 0|1] [option sets 1..5]"""
import json
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

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
detail = int(sys.argv[3]) if len(sys.argv) > 3 else 1
OPTS = [{}, {"dual_bound": 50.0}, {"perturbation": 50}, {"scaling": 3}, {"dual_bound": 5.0}][: int(sys.argv[4]) if len(sys.argv) > 4 else 5]
bad = total = ran_rows = ran_entered = 0
for seed in range(first, first + count):
    lp = make(np.random.default_rng(7000 + seed))
    if not np.any((lp.col_lower < -1e20) & (lp.col_upper > 1e20)):
        continue
    for rule in (0, 1):
        for opts in OPTS:
            o = OracleSimplex(lp)
            g = ClpGpuSimplex(0)
            if "scaling" in opts:
                g.set_option("scaling", opts["scaling"])  # (a layout option: before the load)
            g.loadProblem(lp)
            for s in (o, g):
                s.set_option("pivot_rule", rule)
                s.set_option("max_iterations", 20000)
                s.set_option("free_nonbasic", 1)
                for k, v in opts.items():
                    if not (k == "scaling" and s is g):
                        s.set_option(k, v)
            g.set_option("fake_bound_cleanup", 1)
            so, sg = o.dual(), g.dual()
            lo, lg = o.pivot_log(), g.pivotLog()
            st = g.stats()
            nmin = min(len(lo), len(lg))
            diff = np.nonzero((lo["sequenceIn"][:nmin] != lg["sequenceIn"][:nmin]) | (lo["sequenceOut"][:nmin] != lg["sequenceOut"][:nmin]))[0]
            total += 1
            ran_rows += int(st["free_first_rows"])
            ran_entered += int(st["free_entered"])
            counters = (int(o.free_first_rows), int(o.free_entered), int(st["free_first_rows"]), int(st["free_entered"]))
            if so != sg or len(lo) != len(lg) or len(diff) or counters[:2] != counters[2:]:
                bad += 1
                rec = {"seed": seed, "rule": rule, "opts": opts, "m": int(lp.m), "n": int(lp.n), "oracle_status": int(so), "engine_status": int(sg),
                       "oracle_iterations": int(len(lo)), "engine_iterations": int(len(lg)), "oracle_free_rows_entered": counters[:2],
                       "engine_free_rows_entered": counters[2:], "first_different_pivot": int(diff[0]) + 1 if len(diff) else None}
                if detail and len(diff):
                    i = int(diff[0])
                    rec["oracle_pivot"] = {k: (float(lo[k][i]) if lo[k].dtype.kind == "f" else int(lo[k][i])) for k in ("sequenceIn", "sequenceOut", "pivotRow", "alpha", "theta")}
                    rec["engine_pivot"] = {k: (float(lg[k][i]) if lg[k].dtype.kind == "f" else int(lg[k][i])) for k in ("sequenceIn", "sequenceOut", "pivotRow", "alpha", "theta")}
                print(json.dumps(rec), flush=True)
print(json.dumps({"summary": True, "solves": total, "mismatches": bad, "engine_free_first_rows": ran_rows, "engine_free_entered": ran_entered}))

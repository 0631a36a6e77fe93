#!/usr/bin/env python3
"""config 4 up to N pivots under several option sets: status, iterations, last error (development aid)"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex, lib

lp = P.sparse_lp()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
for name, opts in [("default", {}), ("refactor_mode=1", {"refactor_mode": 1}), ("row_price_frac=0", {"row_price_frac": 0.0}),
                   ("both", {"refactor_mode": 1, "row_price_frac": 0.0}), ("refactor_mode=2", {"refactor_mode": 2})]:
    g = ClpGpuSimplex(0).loadProblem(lp)
    g.set_option("pivot_rule", 1)
    g.set_option("check_every", 16)
    g.set_option("max_pivots", 0)
    g.set_option("log_level", 1)
    for k, v in opts.items():
        g.set_option(k, v)
    t = time.perf_counter()
    st = -1
    while st == -1 and g.numberIterations() < N:
        st = g.dual_steps(500)
        s = g.stats()
        print(name, "it", g.numberIterations(), "k", s["nucleus"], "cap", s["nucleus_capacity"], "refac", s["refactorizations"], "obj", g.objectiveValue(), "st", st, flush=True)
    print(json.dumps({"case": name, "status": st, "iterations": g.numberIterations(), "seconds": round(time.perf_counter() - t, 2),
                      "error": lib().clpgpu_last_error(g._h).decode(), "objective": g.objectiveValue()}), flush=True)
    del g

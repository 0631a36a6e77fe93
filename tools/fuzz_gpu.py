#!/usr/bin/env python3
"""Engine against oracle on the LPs of the CPU differential fuzz (tests/test_oracle_fuzz.py: every kind of row and column, crossing
bounds, infeasible and unbounded instances): status, iteration count and first differing pivot per LP and option set.  On the GPU box:
    python tools/fuzz_gpu.py [first seed] [count]"""
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
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
OPTS = [{}, {"dual_bound": 5.0}, {"dual_bound": 20.0}, {"perturbation": 50}, {"scaling": 3}]
bad = 0
total = 0
for seed in range(first, first + count):
    lp = make(np.random.default_rng(7000 + seed))
    for rule in (0, 1):
        for opts in OPTS:
            o = OracleSimplex(lp)
            o.set_option("pivot_rule", rule)
            o.set_option("max_iterations", 20000)
            g = ClpGpuSimplex(0)
            if "scaling" in opts:
                g.set_option("scaling", opts["scaling"])
            g.loadProblem(lp)
            g.set_option("pivot_rule", rule)
            g.set_option("max_iterations", 20000)
            g.set_option("fake_bound_cleanup", 1)
            for k, v in opts.items():
                o.set_option(k, v)
                if k != "scaling":
                    g.set_option(k, v)
            so, sg = o.dual(), g.dual()
            lo, lg = o.pivot_log(), g.pivotLog()
            nmin = min(len(lo), len(lg))
            diff = np.nonzero((lo["sequenceIn"][:nmin] != lg["sequenceIn"][:nmin]) | (lo["sequenceOut"][:nmin] != lg["sequenceOut"][:nmin]))[0]
            total += 1
            if so != sg or len(lo) != len(lg) or len(diff):
                bad += 1
                print(json.dumps({"seed": seed, "rule": rule, "opts": opts, "m": int(lp.m), "n": int(lp.n), "oracle_status": int(so), "engine_status": int(sg),
                                  "oracle_iterations": int(len(lo)), "engine_iterations": int(len(lg)),
                                  "first_different_pivot": int(diff[0]) + 1 if len(diff) else None}), flush=True)
print(json.dumps({"summary": True, "solves": total, "mismatches": bad}))

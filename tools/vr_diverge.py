#!/usr/bin/env python3
"""first pivot at which N loopback ranks part from the unsharded engine (diagnostic)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from clp_amd import problems as P, engine as E

nr, mode = int(sys.argv[1]), int(sys.argv[2])
opts = dict(kv.split("=") for kv in sys.argv[3].split(",")) if len(sys.argv) > 3 and sys.argv[3] else {}
lp = P.sparse_lp(2000, 9000, 12, seed=13)
def conf(e, **kw):
    e.set_option("pivot_rule", 1); e.set_option("max_pivots", 0); e.set_option("factor_mode", 0)
    for k, v in opts.items(): e.set_option(k, float(v))
    for k, v in kw.items(): e.set_option(k, v)
base = E.ClpGpuSimplex(0).loadProblem(lp); conf(base); print("base", base.dual(), base.numberIterations(), base.objectiveValue())
vr = E.VirtualRanks(lp, nr, configure=lambda e: conf(e, comm_mode=mode))
print("ranks", vr.dual_steps(-1), [e.numberIterations() for e in vr.engines], [e.objectiveValue() for e in vr.engines])
a, b = vr.engines[0].pivotLog(), base.pivotLog()
n = min(len(a), len(b))
bad = np.nonzero((a["sequenceIn"][:n] != b["sequenceIn"][:n]) | (a["sequenceOut"][:n] != b["sequenceOut"][:n]))[0]
print("first divergence", bad[:1], "of", n)
if len(bad):
    i = int(bad[0])
    for j in range(max(0, i - 2), i + 2):
        print(j, "sharded", a[j]); print(j, "base   ", b[j])
for r in range(1, nr):
    c = vr.engines[r].pivotLog()
    print("rank", r, "== rank 0:", np.array_equal(c["sequenceIn"], a["sequenceIn"]))

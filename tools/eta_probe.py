"""per-kernel times of 256 mature pivots behind an eta file of ~800, compact eta file on / off (lab tool)"""
import os
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex

lp = P.sparse_lp()
basis = np.load("tests/golden/basis_sparse_30000.npy")
for opts in sys.argv[1:] or ["lu_compact_eta=1", "lu_compact_eta=0"]:
    g = ClpGpuSimplex(0).loadProblem(lp)
    g.set_option("pivot_rule", 1)
    g.set_option("max_pivots", 0)
    for kv in opts.split(","):
        k, v = kv.split("=")
        g.set_option(k, float(v))
    g.setStatusArray(basis)
    g.dual_steps(800)
    g.set_option("timing", 2)
    k0, s0 = g.kernelTimes(), g.stats()
    g.dual_steps(256)
    torch.cuda.synchronize()
    k1, s1 = g.kernelTimes(), g.stats()
    rows = []
    for name, (ms, cnt) in k1.items():
        ms0, cnt0 = k0.get(name, (0.0, 0))
        if cnt > cnt0:
            rows.append((1e3 * (ms - ms0) / 256, name, (cnt - cnt0) / 256))
    rows.sort(reverse=True)
    print(f"== {opts}: eta count {s0['eta_count']} -> {s1['eta_count']}, nucleus {s1['nucleus']}, tail {s1['lu_tail']}, refactorizations in the window {s1['refactorizations'] - s0['refactorizations']}; "
          f"sum {sum(r[0] for r in rows):.1f} us per pivot (eager launches, event after each)")
    print("   partial scans", s1["chuzr_partial_scans"] - s0["chuzr_partial_scans"], "ordered walks", s1["chuzr_ordered_walks"] - s0["chuzr_ordered_walks"])
    for us, name, per in rows[:30]:
        print(f"   {name:28s} {us:8.2f} us per pivot ({per:.2f} launches)")

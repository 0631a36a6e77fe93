"""per-kernel times of the column-sharded chain over a one-rank RCCL communicator, 256 mature pivots (lab tool)"""
import os
import sys

sys.path.insert(0, ".")
os.environ["CLPGPU_FORCE_COMM"] = "1"
import numpy as np
import torch

from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex
from clp_amd.multigpu import attach_communicator

lp = P.sparse_lp()
basis = np.load("tests/golden/basis_sparse_30000.npy")
g = ClpGpuSimplex(0).loadProblem(lp)
g.set_option("pivot_rule", 1)
g.set_option("max_pivots", 0)
g.set_option("shard_cand_cap", 262144)
g.set_option("shard_flip_cap", 65536)
g.setStatusArray(basis)
attach_communicator(g, 0, 1)
g.dual_steps(800)
g.set_option("timing", 2)
k0 = g.kernelTimes()
g.dual_steps(256)
torch.cuda.synchronize()
k1 = g.kernelTimes()
rows = sorted(((1e3 * (ms - k0.get(n, (0.0, 0))[0]) / 256, n) for n, (ms, cnt) in k1.items() if cnt > k0.get(n, (0.0, 0))[1]), reverse=True)
print(f"sum {sum(r[0] for r in rows):.1f} us per pivot (eager, event after each launch)")
for us, n in rows[:34]:
    print(f"   {n:28s} {us:8.2f}")

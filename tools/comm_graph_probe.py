"""the column-sharded chain over a one-rank RCCL communicator, eager against hipGraph batches (option comm_graph): it/s of 400 mature pivots (lab tool)"""
import os
import sys
import time

sys.path.insert(0, ".")
os.environ["CLPGPU_FORCE_COMM"] = "1"
import numpy as np
import torch

from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex
from clp_amd.multigpu import attach_communicator

lp = P.sparse_lp()
basis = np.load("tests/golden/basis_sparse_30000.npy")
logs = []
for graph in (0, 1):
    g = ClpGpuSimplex(0).loadProblem(lp)
    g.set_option("pivot_rule", 1)
    g.set_option("max_pivots", 0)
    g.set_option("log_level", 1)
    g.set_option("comm_graph", graph)
    g.set_option("shard_cand_cap", 262144)
    g.set_option("shard_flip_cap", 65536)
    g.setStatusArray(basis)
    attach_communicator(g, 0, 1)
    g.dual_steps(100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.dual_steps(400)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    logs.append(g.pivotLog()[:500].copy())
    print(f"== comm_graph {graph}: {400 / dt:.1f} it/s ({1e3 * dt / 400:.3f} ms per pivot), comm_mode {g.stats()['comm_mode']}", flush=True)
print("identical pivots:", bool((logs[0]["sequenceIn"] == logs[1]["sequenceIn"]).all() and (logs[0]["sequenceOut"] == logs[1]["sequenceOut"]).all()))

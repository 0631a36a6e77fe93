"""wall time of LU factorizations of the committed mature basis (tail 5 645) under option sets (lab tool): python tools/refactor_time.py "a=1" "a=0" """
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex

lp = P.sparse_lp()
status = (np.load("tests/golden/basis_sparse_30000.npy") & 7).astype(np.uint8)
for opts in sys.argv[1:] or [""]:
    g = ClpGpuSimplex(0).loadProblem(lp)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("=")
        g.set_option(k, float(v))
    g.factorize(status)
    torch.cuda.synchronize()
    s0 = g.stats()
    t0 = time.perf_counter()
    for _ in range(3):
        rc, pv = g.factorize(status)
        assert rc == 0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    s1 = g.stats()
    v = np.random.default_rng(1).standard_normal(lp.m)
    x = g.ftran(v)
    print(f"== {opts or 'default'}: {1e3 * dt:.1f} ms per factorization (front {(s1['lu_front_ms'] - s0['lu_front_ms']) / 3:.1f}, tail inversion {(s1['lu_invert_ms'] - s0['lu_invert_ms']) / 3:.1f}, "
          f"build {(s1['lu_build_ms'] - s0['lu_build_ms']) / 3:.1f} ms), tail {s1['lu_tail']}, checksum of an FTRAN {float(np.sum(x)):.12g}")

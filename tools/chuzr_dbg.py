"""engine vs oracle on fuzz LPs under a lowered numberWanted floor: first differing pivot and the CHUZR counters (lab tool)"""
import sys

import numpy as np

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from test_oracle_fuzz import make  # noqa: E402

from clp_amd.engine import ClpGpuSimplex  # noqa: E402
from oracle.oracle import OracleSimplex  # noqa: E402

for arg in sys.argv[1:]:
    parts = arg.split(":")
    seed, mode, floor, bad = (int(x) for x in parts[:4])
    factor = float(parts[4]) if len(parts) > 4 else 0.0
    lp = make(np.random.default_rng(7000 + seed))
    o, g = OracleSimplex(lp), ClpGpuSimplex().loadProblem(lp)
    for s in (o, g):
        s.set_option("pivot_rule", 1)
        s.set_option("steepest_mode", mode)
        s.set_option("debug_chuzr_floor", floor)
        s.set_option("max_iterations", 20000)
        if bad >= 0:
            s.set_option("debug_last_bad_iteration", bad)
        if factor:
            s.set_option("debug_tolerance_factor", factor)
    g.set_option("fake_bound_cleanup", 1)
    so, sg = o.dual(), g.dual()
    lo, lg = o.pivot_log(), g.pivotLog()
    st = g.stats()
    print(f"seed {seed} mode {mode} floor {floor}: m {lp.m} n {lp.n} status {so}/{sg} pivots {len(lo)}/{len(lg)} partial {o.partial_scans}/{st['chuzr_partial_scans']} "
          f"recalls {o.chuzr_recalls}/{st['chuzr_recalls']} refactorizations {o.refactorizations}/{st['refactorizations']}")
    n = min(len(lo), len(lg))
    d = [i for i in range(n) if any(lo[k][i] != lg[k][i] for k in ("sequenceIn", "sequenceOut", "pivotRow"))]
    first = d[0] if d else n
    v = [i for i in range(n) if any(abs(lo[k][i] - lg[k][i]) > 1e-9 * (1 + abs(lo[k][i])) for k in ("theta", "alpha", "dualOut", "objective"))]
    print("    first pivot with different values:", v[0] if v else None)
    first = min(first, v[0]) if v else first
    for i in range(max(0, first - 2), min(max(len(lo), len(lg)), first + 3)):
        print("   ", i, lo[i] if i < len(lo) else None, "|", lg[i] if i < len(lg) else None)

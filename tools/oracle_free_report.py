#!/usr/bin/env python3
"""CPU only.  The oracle's two treatments of nonbasic free columns -- the bothFake substitution (option free_nonbasic 0: the default of oracle and engine)
and the reference's isFree path (1) -- against HiGHS on the LPs of tests/test_oracle_fuzz.py that have free columns, and on sparse_lp with a tenth
of the columns made free.  Output: the table kept as profiles/r04_oracle_free_nonbasic.txt.

    python tools/oracle_free_report.py [number of fuzz seeds, default 300]"""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from clp_amd import problems as P  # noqa: E402
from oracle.oracle import OracleSimplex  # noqa: E402
from test_oracle_fuzz import highs, make  # noqa: E402


def run(lp, rule, free, **opts):
    o = OracleSimplex(lp)
    o.set_option("pivot_rule", rule)
    o.set_option("free_nonbasic", free)
    o.set_option("max_iterations", 50000)
    for k, v in opts.items():
        o.set_option(k, v)
    return o, o.dual()


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    lps = []
    for seed in range(seeds):
        lp = make(np.random.default_rng(7000 + seed))
        if np.any((lp.col_lower < -1e20) & (lp.col_upper > 1e20)):
            lps.append((seed, lp, highs(lp)))
    solvable = [x for x in lps if x[2][0] == 0]
    print(f"# fuzz LPs with free columns among seeds 0..{seeds - 1}: {len(lps)}; HiGHS optimal on {len(solvable)}, infeasible / unbounded on {len(lps) - len(solvable)}")
    print("# outcome on the solvable ones, both pivot rules: status 0 with HiGHS's objective (1e-6) / status 10 / status 1 / anything else")
    print("dual_bound  free_nonbasic  optimal  use_primal_10  infeasible_1  other  pivots(total)  free_first_rows  free_entered")
    for db in (5.0, 20.0, 1.0e10):
        for free in (0, 1):
            c, pivots, ffr, fe = collections.Counter(), 0, 0, 0
            for seed, lp, (hs, hobj) in solvable:
                for rule in (0, 1):
                    o, st = run(lp, rule, free, dual_bound=db)
                    pivots += o.iterations
                    ffr += o.free_first_rows
                    fe += o.free_entered
                    if st == 0:
                        st = 0 if abs(o.objective - hobj) <= 1e-6 * (1 + abs(hobj)) else -99
                    c[st] += 1
            other = sum(v for k, v in c.items() if k not in (0, 10, 1))
            print(f"{db:10g}  {free:13d}  {c[0]:7d}  {c[10]:13d}  {c[1]:12d}  {other:5d}  {pivots:13d}  {ffr:15d}  {fe:12d}")
    wrong = 0
    for seed, lp, (hs, hobj) in lps:
        if hs == 0:
            continue
        for rule in (0, 1):
            for free in (0, 1):
                o, st = run(lp, rule, free)
                wrong += st not in (1, 2, 10)
    print(f"# infeasible / unbounded ones, default dual bound, both settings and rules: {wrong} answers outside (1, 2, 10)")
    print("#\n# sparse_lp(m, 4m, ., seed) with a tenth of the columns made free, steepest edge")
    print("rows  cols  free  free_nonbasic  status  pivots  objective  highs_objective  free_first_rows  free_entered")
    for m, n, k, seed in ((300, 1200, 8, 11), (400, 1600, 6, 5), (600, 2400, 8, 3)):
        lp = P.sparse_lp(m, n, k, seed)
        free_cols = np.random.default_rng(seed).choice(n, n // 10, replace=False)
        lp = type(lp)(lp)
        lp.col_lower, lp.col_upper = lp.col_lower.copy(), lp.col_upper.copy()
        lp.col_lower[free_cols], lp.col_upper[free_cols] = -1e30, 1e30
        hs, hobj = highs(lp)
        for free in (0, 1):
            o, st = run(lp, 1, free)
            print(f"{m:4d}  {n:4d}  {len(free_cols):4d}  {free:13d}  {st:6d}  {o.iterations:6d}  {o.objective:.6f}  {hobj:.6f}  {o.free_first_rows:15d}  {o.free_entered:12d}")


if __name__ == "__main__":
    main()

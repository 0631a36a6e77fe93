#!/usr/bin/env python3
"""Objective against time on one LP for several option sets, on the GPU box.  The LP is generated
once; every variant starts a fresh context from the slack basis and runs for --budget seconds in chunks of --chunk pivots.
Per chunk, from the pivot log: share of pivots with a dual step below 1e-9, bound flips per pivot, median |alpha|, share of
|alpha| below 1e-5, and the pivot mix (slack leaves + structural enters = the nucleus grows; structural for structural = it does
not).  One JSON line per chunk on stdout, tagged with the variant; a table at the end.

    python tools/race.py --budget 45 --variants "lu_adaptive:;lu_475:lu_max_pivots=475;explicit:factor_mode=0"
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="sparse")
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--rung", default="")
    ap.add_argument("--budget", type=float, default=45.0)
    ap.add_argument("--chunk", type=int, default=4000)
    ap.add_argument("--variants", default="default:")
    ap.add_argument("--common", default="")
    args = ap.parse_args()
    import numpy as np
    import torch

    from clp_amd import problems as P
    from clp_amd.engine import ClpGpuSimplex

    if args.rung:
        from tools.ladder import ladder_lp
        lp = ladder_lp(args.rung)
    elif args.workload == "dense":
        lp = P.dense_lp(args.rows or 5000, args.cols or 5000)
    elif args.workload == "netlib":
        m, n = args.rows or 50000, args.cols or 200000
        lp = P.netlib_shaped_lp(m, n, m * n // 1000)
    else:
        lp = P.sparse_lp(args.rows or 50000, args.cols or 200000)
    n = lp.n
    table = []
    for spec in filter(None, args.variants.split(";")):
        tag, _, opts = spec.partition(":")
        g = ClpGpuSimplex(0).loadProblem(lp)
        g.set_option("pivot_rule", 1)
        g.set_option("check_every", 16)
        g.set_option("max_pivots", 0)
        for kv in filter(None, (args.common + "," + opts).split(",")):
            key, val = kv.split("=")
            g.set_option(key, float(val))
        t0 = time.perf_counter()
        status, last_it = -1, 0
        rec = {}
        while status == -1 and time.perf_counter() - t0 < args.budget:
            status = g.dual_steps(args.chunk)
            torch.cuda.synchronize()
            now = time.perf_counter()
            st = g.stats()
            it = g.numberIterations()
            extra = {}
            if it > last_it:
                part = g.pivotLog()[last_it:it]
                if len(part):
                    a = np.abs(part["alpha"])
                    sin, sout = part["sequenceIn"] < n, part["sequenceOut"] < n
                    extra = {"tiny_step_share": round(float((np.abs(part["theta"]) < 1.0e-9).mean()), 4),
                             "flips_per_pivot": round(float(part["numberFlipped"].mean()), 2),
                             "alpha_median": float(np.median(a)), "alpha_small_share": round(float((a < 1.0e-5).mean()), 4),
                             "grow": round(float((sin & ~sout).mean()), 3), "swap": round(float((sin & sout).mean()), 3),
                             "shrink": round(float((~sin & sout).mean()), 3), "flip_only": round(float((part["sequenceIn"] == part["sequenceOut"]).mean()), 3)}
            rec = {"variant": tag, **extra, "iterations": it, "elapsed_s": round(now - t0, 2), "nucleus": st["nucleus"],
                   "refactorizations": st["refactorizations"], "lu": [st["lu_active"], st["lu_front"], st["lu_tail"]],
                   "objective": g.objectiveValue(), "status": status}
            print(json.dumps(rec), flush=True)
            last_it = it
        table.append((tag, opts, rec.get("iterations"), rec.get("elapsed_s"), rec.get("nucleus"), rec.get("objective"), status))
        del g
    print("| variant | options | pivots | seconds | nucleus | dual objective | status |")
    print("|---|---|---|---|---|---|---|")
    for row in table:
        print("| " + " | ".join(str(x) for x in row) + " |")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Whole-solve profile on the GPU box: pivots/s, nucleus size k and refactorizations as the solve
proceeds, until optimal or a time budget.  Answers "does config 4 finish, and how does k grow".

    python tools/solve_profile.py [--workload sparse|dense|netlib] [--budget 120] [--chunk 1000]
writes one JSON line per chunk to stdout (and a summary line at the end)."""
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
    ap.add_argument("--budget", type=float, default=120.0)
    ap.add_argument("--chunk", type=int, default=1000)
    ap.add_argument("--opts", default="")
    ap.add_argument("--theta-stats", action="store_true",
                    help="per chunk, from the pivot log: share of pivots with a dual step below 1e-9, mean bound flips per pivot (DESIGN 8, item 0)")
    args = ap.parse_args()
    import torch  # noqa: F401

    from clp_amd import problems as P
    from clp_amd.engine import ClpGpuSimplex

    if args.workload == "dense":
        lp = P.dense_lp(args.rows or 5000, args.cols or 5000)
    elif args.workload == "netlib":
        m, n = args.rows or 50000, args.cols or 200000
        lp = P.netlib_shaped_lp(m, n, m * n // 1000)
    else:
        lp = P.sparse_lp(args.rows or 50000, args.cols or 200000)
    g = ClpGpuSimplex(0).loadProblem(lp)
    g.set_option("pivot_rule", 1)
    g.set_option("check_every", 16)
    g.set_option("max_pivots", 0)
    for kv in filter(None, args.opts.split(",")):
        key, val = kv.split("=")
        g.set_option(key, float(val))
    t0 = time.perf_counter()
    status, last_t, last_it = -1, t0, 0
    while status == -1 and time.perf_counter() - t0 < args.budget:
        status = g.dual_steps(args.chunk)
        torch.cuda.synchronize()
        now = time.perf_counter()
        st = g.stats()
        it = g.numberIterations()
        extra = {}
        if args.theta_stats and it > last_it:
            part = g.pivotLog()[last_it:it]  # the log holds 2^20 pivots
            if len(part):
                extra = {"tiny_step_share": round(float((abs(part["theta"]) < 1.0e-9).mean()), 4),
                         "flips_per_pivot": round(float(part["numberFlipped"].mean()), 2)}
        print(json.dumps({**extra, "iterations": it, "elapsed_s": round(now - t0, 3), "chunk_it_per_s": round((it - last_it) / max(now - last_t, 1e-9), 1),
                          "nucleus": st["nucleus"], "capacity": st["nucleus_capacity"], "refactorizations": st["refactorizations"],
                          "refreshes": st["refreshes"], "refreshes_rejected": st["refreshes_rejected"], "lu": [st["lu_active"], st["lu_front"], st["lu_tail"], st["lu_factorizations"], round(st["lu_front_ms"]), round(st["lu_invert_ms"]), round(st["lu_build_ms"])],
                          "objective": g.objectiveValue(), "status": status}), flush=True)
        last_t, last_it = now, it
    total = time.perf_counter() - t0
    print(json.dumps({"summary": True, "workload": args.workload, "rows": int(lp.m), "cols": int(lp.n), "status": status,
                      "iterations": g.numberIterations(), "seconds": round(total, 3),
                      "time_to_optimal_s": round(total, 3) if status == 0 else None, "objective": g.objectiveValue(),
                      "nucleus": g.stats()["nucleus"], "refactorizations": g.stats()["refactorizations"],
                      "refreshes": g.stats()["refreshes"], "refreshes_rejected": g.stats()["refreshes_rejected"]}), flush=True)


if __name__ == "__main__":
    main()

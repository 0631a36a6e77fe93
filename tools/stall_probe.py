#!/usr/bin/env python3
"""Why the config-4 solve stalls: at a few pivot counts of one run, who is primal infeasible and what dual steepest edge
sees -- infeasible rows split by slack / structural basic, their infeasibilities and weights, the top rows by infeasibility^2 / weight --
and the pivot log of a stretch saved for offline reading (re-entry distances, step sizes).
    python tools/stall_probe.py --opts lu_max_pivots=475 --marks 6000,12000,16000 --save 12000 16000"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--opts", default="")
    ap.add_argument("--marks", default="6000,12000,16000")
    ap.add_argument("--save", nargs=2, type=int, default=None)
    ap.add_argument("--tag", default="probe")
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401

    from clp_amd import problems as P
    from clp_amd.engine import ClpGpuSimplex

    lp = P.sparse_lp()
    n, m = lp.n, lp.m
    g = ClpGpuSimplex(0).loadProblem(lp)
    g.set_option("pivot_rule", 1)
    g.set_option("check_every", 16)
    g.set_option("max_pivots", 0)
    for kv in filter(None, args.opts.split(",")):
        k, v = kv.split("=")
        g.set_option(k, float(v))
    lo = np.concatenate([lp.col_lower, lp.row_lower])
    up = np.concatenate([lp.col_upper, lp.row_upper])
    done = 0
    for mark in [int(x) for x in args.marks.split(",")]:
        st = g.dual_steps(mark - done)
        done = g.numberIterations()
        w, inf = g.rowWeights()
        pv = g.pivotVariable()
        x = g.solution()
        xb = x[pv]
        viol = np.maximum(lo[pv] - xb, 0) + np.maximum(xb - up[pv], 0)
        struct = pv < n
        bad = viol > 1e-7
        score = np.where(bad, viol * viol / np.maximum(w, 1e-30), 0.0)
        top = np.argsort(-score)[:12]
        rec = {"iterations": int(done), "status": int(st), "nucleus": int(struct.sum()), "objective": g.objectiveValue(),
               "infeasible_slack_rows": int((bad & ~struct).sum()), "infeasible_structural_rows": int((bad & struct).sum()),
               "slack_violation_median": float(np.median(viol[bad & ~struct])) if (bad & ~struct).any() else 0.0,
               "struct_violation_median": float(np.median(viol[bad & struct])) if (bad & struct).any() else 0.0,
               "struct_violation_max": float(viol[struct].max()) if struct.any() else 0.0,
               "weight_median_slack": float(np.median(w[~struct])), "weight_median_struct": float(np.median(w[struct])) if struct.any() else 0.0,
               "weight_max_struct": float(w[struct].max()) if struct.any() else 0.0,
               "score_best_slack": float(score[~struct].max()) if (~struct).any() else 0.0,
               "score_best_struct": float(score[struct].max()) if struct.any() else 0.0,
               "share_struct_in_top_1000_scores": float(struct[np.argsort(-score)[:1000]].mean()),
               "top": [[int(pv[i] < n), float(viol[i]), float(w[i]), float(score[i])] for i in top]}
        print(json.dumps(rec), flush=True)
        if st != -1:
            break
    if args.save:
        a, b = args.save
        log = g.pivotLog()[a:b]
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.save(os.path.join(ROOT, "gpurun_out", f"pivotlog_{args.tag}_{a}_{b}.npy"), log)


if __name__ == "__main__":
    main()

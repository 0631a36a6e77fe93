"""config 4 (or a ladder rung) from the slack basis for a long budget: one JSON line per chunk, and -- if the solve ends optimal -- the
KKT certificate computed outside the engine (tools/kkt_certificate.py).  python tools/long_solve.py <budget s> [rung] [opts]"""
import json
import sys
import time

sys.path.insert(0, ".")
import torch

from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
rung = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
if rung:
    from tools.ladder import ladder_lp

    lp = ladder_lp(rung)
else:
    lp = P.sparse_lp()
g = ClpGpuSimplex(0).loadProblem(lp)
g.set_option("pivot_rule", 1)
g.set_option("max_pivots", 0)
for kv in filter(None, (sys.argv[3] if len(sys.argv) > 3 else "").split(",")):
    k, v = kv.split("=")
    g.set_option(k, float(v))
t0 = time.perf_counter()
status, last_t, last_it = -1, t0, 0
while status == -1 and time.perf_counter() - t0 < budget:
    status = g.dual_steps(20000)
    torch.cuda.synchronize()
    now, st, it = time.perf_counter(), g.stats(), g.numberIterations()
    print(json.dumps({"iterations": it, "elapsed_s": round(now - t0, 1), "chunk_it_per_s": round((it - last_it) / max(now - last_t, 1e-9), 1), "nucleus": st["nucleus"],
                      "lu_tail": st["lu_tail"], "refactorizations": st["refactorizations"], "objective": g.objectiveValue(), "status": status,
                      "partial_scans": st["chuzr_partial_scans"]}), flush=True)
    last_t, last_it = now, it
out = {"summary": True, "rows": int(lp.m), "cols": int(lp.n), "status": int(status), "iterations": int(g.numberIterations()),
       "seconds": round(time.perf_counter() - t0, 1), "objective": g.objectiveValue()}
if status == 0:
    from tools.kkt_certificate import certify, row_duals_from_engine

    out["certificate"] = certify(lp, g.solution(), row_duals_from_engine(lp, g))
print(json.dumps(out), flush=True)

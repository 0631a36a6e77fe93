#!/usr/bin/env python3
"""Time-to-optimal ladder: config-4-shaped LPs (clp_amd.problems.sparse_lp: ranged rows of width <= 2 around A x*,
columns in [0, 100], costs in [0.1, 1], four columns per row) at sizes an independent solver still finishes.

    python tools/ladder.py highs <rung> [time limit s]   # HiGHS serial dual simplex (scipy), presolve off -> one JSON line
    python tools/ladder.py oracle <rung> [max iterations] # the CPU oracle port (steepest edge), same LP -> one JSON line
    python tools/ladder.py engine <rung> [time limit s]   # the HIP engine (needs the MI355X) + a KKT certificate computed outside it
    python tools/ladder.py merge <files...>               # collects the lines into tests/golden/ladder_optima.json

The rungs are fixed here (rows, columns, entries per column, seed); `ladder_lp(name)` is what the GPU test and bench.py build.
The density follows BASELINE config 4 (0.1 %) with a floor of 10 entries per column, so that the small rungs are not trivially
sparse; rung "1500" is the LP tests/test_gpu_lu.py already solves against HiGHS.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RUNGS = {
    "1500": (1500, 6000, 10, 31),
    "2000": (2000, 8000, 10, 38),
    "3000": (3000, 12000, 10, 32),
    "4000": (4000, 16000, 10, 39),
    "5000": (5000, 20000, 10, 33),
    "7000": (7000, 28000, 10, 34),
    "10000": (10000, 40000, 10, 35),
    "14000": (14000, 56000, 14, 36),
    "20000": (20000, 80000, 20, 37),
}


def ladder_lp(name):
    from clp_amd import problems as P
    m, n, k, seed = RUNGS[str(name)]
    return P.sparse_lp(m, n, k, seed)


def run_highs(name, limit):
    import numpy as np
    import scipy
    import scipy.sparse as sp
    from scipy.optimize import linprog
    lp = ladder_lp(name)
    m, n = lp.m, lp.n
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(m, n))
    t0 = time.time()
    r = linprog(lp.obj, A_ub=sp.vstack([A, -A]).tocsr(), b_ub=np.concatenate([lp.row_upper, -lp.row_lower]),
                bounds=np.column_stack([lp.col_lower, lp.col_upper]), method="highs-ds",
                options={"presolve": False, "time_limit": float(limit)})
    dt = time.time() - t0
    return {"rung": str(name), "m": m, "n": n, "nnz": int(lp.col_start[-1]), "solver": f"HiGHS serial dual simplex (scipy {scipy.__version__}), presolve off",
            "status": int(r.status), "iterations": int(r.nit), "seconds": round(dt, 2), "cores": 1,
            "objective": (float(r.fun) if (r.status == 0 and r.fun is not None) else None)}


def run_oracle(name, max_iterations):
    from oracle.oracle import OracleSimplex
    lp = ladder_lp(name)
    o = OracleSimplex(lp)
    o.set_option("pivot_rule", 1)
    if max_iterations:
        o.set_option("max_iterations", max_iterations)
    t0 = time.time()
    st = o.dual()
    dt = time.time() - t0
    return {"rung": str(name), "solver": "oracle port (dense LU, one core)", "status": int(st), "iterations": int(o.iterations),
            "seconds": round(dt, 2), "cores": 1, "objective": float(o.objective)}


def run_engine(name, limit):
    """The engine in its default mode from the slack basis to status 0, and the optimality certificate of the point it returns --
    computed by tools/kkt_certificate.py from the LP and (x, y) alone: what the ladder accepts where HiGHS does not finish."""
    import torch  # noqa: F401

    from clp_amd.engine import ClpGpuSimplex
    from tools.kkt_certificate import certify, row_duals_from_engine
    lp = ladder_lp(name)
    g = ClpGpuSimplex(0).loadProblem(lp)
    g.set_option("pivot_rule", 1)
    g.set_option("max_pivots", 0)
    t0 = time.perf_counter()
    st = -1
    while st == -1 and time.perf_counter() - t0 < float(limit):
        st = g.dual_steps(20000)
    dt = time.perf_counter() - t0
    cert = certify(lp, g.solution(), row_duals_from_engine(lp, g)) if st == 0 else None
    info = g.stats()
    return {"rung": str(name), "m": int(lp.m), "n": int(lp.n), "nnz": int(lp.col_start[-1]), "solver": "kkt: HIP engine (default options) + KKT certificate computed outside it",
            "status": int(st), "iterations": int(g.numberIterations()), "seconds": round(dt, 2), "objective": float(g.objectiveValue()) if st == 0 else None,
            "refactorizations": int(info["refactorizations"]), "nucleus": int(info["nucleus"]), "certificate": cert}


def main():
    what = sys.argv[1]
    if what == "engine":
        print(json.dumps(run_engine(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else 900)), flush=True)
        return
    if what == "highs":
        print(json.dumps(run_highs(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else 3600)), flush=True)
    elif what == "oracle":
        print(json.dumps(run_oracle(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)), flush=True)
    elif what == "merge":
        path = os.path.join(ROOT, "tests", "golden", "ladder_optima.json")
        out = json.load(open(path)) if os.path.exists(path) else {}
        for f in sys.argv[2:]:
            for line in open(f):
                line = line.strip()
                if not line.startswith("{"):
                    continue
                rec = json.loads(line)
                slot = out.setdefault(rec["rung"], {"rows": RUNGS[rec["rung"]][0], "columns": RUNGS[rec["rung"]][1],
                                                    "entries_per_column": RUNGS[rec["rung"]][2], "seed": RUNGS[rec["rung"]][3]})
                slot["oracle" if rec["solver"].startswith("oracle") else ("kkt" if rec["solver"].startswith("kkt") else "highs")] = rec
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
        print(path)


if __name__ == "__main__":
    main()

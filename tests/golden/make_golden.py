#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory.

The reference (coin-or/Clp) cannot be built or imported in the authoring container (CoinUtils is
absent), so vectors that the reference's own tests do not pin are produced by the
oracle (oracle/clp_dual_oracle.c) and are labelled "oracle" inside the files.  What the reference
DOES pin is asserted directly in tests/test_oracle_golden.py (objective values, basis solution).

  afiro_pivots.json : per-iteration (sequenceIn, sequenceOut) of the dual simplex on AFIRO, Dantzig
                      and steepest-edge row choice
  price_case.npz    : one fused row-pricing call on a 300x1200 sparse LP (inputs and outputs)
  hello_lp.npz      : BASELINE config 1 (examples/hello.mps of the reference, 21 x 53, the plumbing
                      case) as parsed arrays -- written only where /root/reference exists; the
                      reference pins no objective for it, the stored optimum is HiGHS' and the oracle's
  modified_afiro_lp.npz : the reference's other input fixture, examples/modified_afiro.mps (7 x 16 piecewise
                      variant of AFIRO read by examples/piecewise.cpp:20), the same way
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from clp_amd import problems as P  # noqa: E402
from clp_amd.mps import read_mps  # noqa: E402
from oracle.oracle import OracleSimplex  # noqa: E402


def main():
    lp = read_mps(os.path.join(HERE, "afiro.mps"))
    out = {"source": "oracle/clp_dual_oracle.c (reference unbuildable here)", "objective": -4.6475314286e+02}
    for rule, name in ((0, "dantzig"), (1, "steepest")):
        o = OracleSimplex(lp)
        o.set_option("pivot_rule", rule)
        assert o.dual() == 0
        log = o.pivot_log()
        out[name] = {"in": log["sequenceIn"].tolist(), "out": log["sequenceOut"].tolist(),
                     "objective": o.objective}
    with open(os.path.join(HERE, "afiro_pivots.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    lp = P.sparse_lp(300, 1200, 8, seed=11)
    o = OracleSimplex(lp)
    rng = np.random.default_rng(3)
    m, n = lp.m, lp.n
    idx = np.sort(rng.choice(m, 90, replace=False)).astype(np.int32)
    val = rng.standard_normal(90)
    status = rng.choice([1, 2, 3, 5], size=n + m, p=[0.2, 0.3, 0.45, 0.05]).astype(np.uint8)
    dj = np.where((status & 3) == 2, -1.0, 1.0) * rng.uniform(0, 2, n + m)
    oi, ov, ci, cv, ut = o.price_row_fused(idx, val, status, dj)
    np.savez_compressed(os.path.join(HERE, "price_case.npz"), pi_index=idx, pi_value=val, status=status, dj=dj,
                        out_index=oi, out_value=ov, cand_index=ci, cand_value=cv, upper_theta=ut)
    hello()
    print("golden fixtures written")


def hello():
    # the two input fixtures the reference carries under examples/: hello.mps (read by hello.cpp:24) and
    # modified_afiro.mps (a 7 x 16 piecewise variant of AFIRO, read by piecewise.cpp:20)
    for name, out in (("hello.mps", "hello_lp.npz"), ("modified_afiro.mps", "modified_afiro_lp.npz")):
        src = os.path.join("/root/reference/examples", name)
        if not os.path.exists(src):
            continue
        from scipy.optimize import linprog
        import scipy.sparse as sp

        lp = read_mps(src)
        o = OracleSimplex(lp)
        assert o.dual() == 0
        A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(lp.m, lp.n))
        lo, up = np.where(lp.row_lower < -1e29, -np.inf, lp.row_lower), np.where(lp.row_upper > 1e29, np.inf, lp.row_upper)
        keep_up, keep_lo = np.isfinite(up), np.isfinite(lo)
        r = linprog(lp.obj, A_ub=sp.vstack([A[keep_up], -A[keep_lo]]), b_ub=np.concatenate([up[keep_up], -lo[keep_lo]]),
                    bounds=[(None if a < -1e29 else a, None if b > 1e29 else b) for a, b in zip(lp.col_lower, lp.col_upper)], method="highs")
        assert r.status == 0 and abs(r.fun - o.objective) < 1e-9 * (1 + abs(r.fun)), (name, r.fun, o.objective)
        np.savez_compressed(os.path.join(HERE, out), m=lp.m, n=lp.n, col_start=lp.col_start, row=lp.row, elem=lp.elem,
                            col_lower=lp.col_lower, col_upper=lp.col_upper, obj=lp.obj, row_lower=lp.row_lower,
                            row_upper=lp.row_upper, optimum=r.fun)


if __name__ == "__main__":
    main()

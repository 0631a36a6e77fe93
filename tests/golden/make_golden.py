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
  exmip1_lp.npz     : exmip1 rebuilt from the values the reference's OSI test asserts (see exmip1() below)
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


def exmip1():
    """exmip1 (5 x 8) as the reference's own OSI test pins it -- test/OsiClpSolverInterfaceTest.cpp: matrix by row :203-258
    (14 elements, starts, indices), objective :194-201, row senses / right-hand sides / ranges :396-415 (G 2.5, L 2.1, E 4.0,
    R [1.8, 5.0], R [3.0, 15.0]), column bounds :170-173 (cl0 2.5, cl1 0, cu1 4.1, cu2 1.0), the objective of the initial point 3.5
    (:191-192, which fixes cl4 = 0.5).  The bounds that test does not assert are those of the public Data/Sample exmip1.mps
    (cu3 = 1, cu4 = 4.0, cu7 = 4.3).  LP optimum 3.2368421 (src/unitTest.cpp:2572)."""
    from scipy.optimize import linprog
    import scipy.sparse as sp

    elements = [3.0, 1.0, -2.0, -1.0, -1.0, 2.0, 1.1, 1.0, 1.0, 2.8, -1.2, 5.6, 1.0, 1.9]
    starts = [0, 5, 7, 9, 11, 14]
    indices = [0, 1, 3, 4, 7, 1, 2, 2, 5, 3, 6, 0, 4, 7]
    A = sp.csr_matrix((elements, indices, starts), shape=(5, 8)).tocsc()
    A.sort_indices()
    big = 1.0e30
    lp = dict(m=5, n=8, col_start=A.indptr.astype(np.int32), row=A.indices.astype(np.int32), elem=A.data.astype(np.float64),
              col_lower=np.array([2.5, 0, 0, 0, 0.5, 0, 0, 0.0]), col_upper=np.array([big, 4.1, 1.0, 1.0, 4.0, big, big, 4.3]),
              obj=np.array([1.0, 0, 0, 0, 2.0, 0, 0, -1.0]), row_lower=np.array([2.5, -big, 4.0, 1.8, 3.0]),
              row_upper=np.array([big, 2.1, 4.0, 5.0, 15.0]))
    assert abs(float(lp["obj"] @ lp["col_lower"]) - 3.5) < 1e-12
    r = linprog(lp["obj"], A_ub=sp.vstack([A[[1, 3, 4]], -A[[0, 3, 4]]]), b_ub=np.array([2.1, 5.0, 15.0, -2.5, -1.8, -3.0]), A_eq=A[[2]], b_eq=[4.0],
                bounds=[(a, None if b > 1e29 else b) for a, b in zip(lp["col_lower"], lp["col_upper"])], method="highs")
    assert r.status == 0 and abs(r.fun - 3.2368421) < 1e-6, r.fun
    np.savez_compressed(os.path.join(HERE, "exmip1_lp.npz"), optimum=r.fun, **lp)


if __name__ == "__main__":
    main()
    exmip1()

#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory.

The reference (coin-or/Clp) cannot be built or imported in the authoring container (CoinUtils is
absent), so vectors that the reference's own tests do not pin are produced by the
oracle (oracle/clp_dual_oracle.c) and are labelled "oracle" inside the files.  What the reference
DOES pin is asserted directly in tests/test_oracle_golden.py (objective values, basis solution).

  afiro_pivots.json : per-iteration (sequenceIn, sequenceOut) of the dual simplex on AFIRO, Dantzig
                      and steepest-edge row choice
  price_case.npz    : one fused row-pricing call on a 300x1200 sparse LP (inputs and outputs)
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
    print("golden fixtures written")


if __name__ == "__main__":
    main()

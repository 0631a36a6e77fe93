"""The engine's host restatement of ClpSimplexDual::perturb (clp_amd/csrc/perturb_host.h, src/ClpSimplexDual.cpp:6533-6957)
against the oracle's, on the CPU and bit for bit: both are written from the reference separately, and on a generic matrix the
two solves only stay together pivot for pivot if every perturbed cost agrees to the last bit (tests/test_gpu_perturbation.py
asserts that on the device).  Covers the start-up decision (:6562-6606), the fixed-fraction values 51-69, the "user is in
charge" branch, the kick (numberIterations > 0), statuses from the middle of a solve, and one-sided / free columns."""
import os
import struct
import subprocess

import numpy as np
import pytest

from clp_amd import problems as P
from oracle.oracle import OracleSimplex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("perturb") / "harness")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "perturb_harness.cpp")])
    return exe


def engine_side(harness, tmp_path, lp, status, perturbation, iterations):
    N = lp.m + lp.n
    src, dst = str(tmp_path / "rim.bin"), str(tmp_path / "out.bin")
    with open(src, "wb") as f:
        f.write(struct.pack("6q", lp.m, lp.n, len(lp.elem), iterations, perturbation, 1234567))
        f.write(struct.pack("2d", 1.0e-7, 1.0e15))
        f.write(np.asarray(lp.col_start, dtype=np.int32).tobytes())
        f.write(np.asarray(lp.elem, dtype=np.float64).tobytes())
        f.write(np.concatenate([lp.col_lower, lp.row_lower]).astype(np.float64).tobytes())
        f.write(np.concatenate([lp.col_upper, lp.row_upper]).astype(np.float64).tobytes())
        f.write(np.asarray(status, dtype=np.uint8).tobytes())
        f.write(np.asarray(lp.obj, dtype=np.float64).tobytes())
        f.write(np.concatenate([lp.obj, np.zeros(lp.m)]).astype(np.float64).tobytes())
    subprocess.check_call([harness, src, dst])
    buf = open(dst, "rb").read()
    rc, after, _seed = struct.unpack_from("3q", buf, 0)
    return rc, after, np.frombuffer(buf, dtype=np.float64, count=N, offset=24).copy()


def statuses_after(lp, pivots):
    o = OracleSimplex(lp)
    o.set_option("max_iterations", pivots)
    o.dual()
    return o.status() & 7


def few_costs():
    lp = P.sparse_lp(300, 1200, 5, 1)
    lp.obj = np.ceil(lp.obj * 3.0)
    return lp


def one_sided():
    """Columns with only an upper bound, free columns, fixed columns and zero costs among the rest."""
    lp = P.sparse_lp(200, 900, 6, 9)
    lp.obj = np.ceil(lp.obj * 2.0)
    lp.col_lower[0:200] = -1.0e30
    lp.col_upper[100:300] = 1.0e30
    lp.col_upper[300:340] = lp.col_lower[300:340]
    lp.obj[400:500] = 0.0
    return lp


CASES = {"nqueens20": lambda: P.nqueens(20), "ufl": lambda: P.ufl(10, 30, 99), "tsp": lambda: P.tsp_mtz(12, 5), "few_costs": few_costs,
         "one_sided": one_sided, "varied": lambda: P.sparse_lp(120, 500, 5, 2)}


@pytest.mark.parametrize("name", sorted(CASES))
def test_perturbed_costs_bit_identical(harness, tmp_path, name):
    lp = CASES[name]()
    changed = 0
    for pivots in (0, 40):
        status = statuses_after(lp, pivots)
        for perturbation, iterations in [(50, 0), (51, 0), (53, 0), (57, 0), (60, 0), (65, 0), (75, 0), (100, 0), (-6, 0), (50, 5000), (100, 5000)]:
            rc_o, after_o, cost_o = OracleSimplex(lp).test_perturb(perturbation, iterations, status)
            rc_e, after_e, cost_e = engine_side(harness, tmp_path, lp, status, perturbation, iterations)
            assert (rc_o, after_o) == (rc_e, after_e), (name, pivots, perturbation, iterations)
            assert np.array_equal(cost_o, cost_e), (name, pivots, perturbation, iterations, int(np.sum(cost_o != cost_e)))
            base = np.concatenate([lp.obj, np.zeros(lp.m)])
            moved = cost_o != base
            assert not np.any(moved[lp.n:])  # row costs are never touched (:6745)
            assert not np.any(moved[:lp.n] & ((status[:lp.n] == 1) | (lp.col_lower >= lp.col_upper)))  # nor basic / fixed columns
            assert after_o in ((101,) if moved.any() else (100, 101, perturbation))
            changed += int(moved.any())
    if name in ("varied", "tsp"):
        # more than a quarter of the costs distinct: only the kick (iterations > 0) and the user value perturb
        assert changed > 0
    else:
        assert changed >= 12

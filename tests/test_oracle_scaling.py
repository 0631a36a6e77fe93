"""Oracle-side restatement of ClpPackedMatrix::scale (src/ClpPackedMatrix.cpp:4120-4760) and of its
application in createRim / the unscaling of the results -- groundwork for SURVEY 8(f)3.  CPU only; the
HIP engine does not scale yet, so nothing here is a parity claim about it.  The reference pins no
scale factors; what is checked is what the algorithm guarantees and that the scaled solves land on
the optimum of the unscaled problem (itself pinned on the reference's values / HiGHS elsewhere)."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from clp_amd import problems as P  # noqa: E402
from clp_amd.mps import read_mps  # noqa: E402
from oracle.oracle import OracleSimplex  # noqa: E402
from test_oracle_golden import kkt_check  # noqa: E402


def make(name):
    if name == "afiro":
        return read_mps(os.path.join(HERE, "golden", "afiro.mps"))
    if name == "sparse300":
        return P.sparse_lp(300, 1200, 8, 11)
    return P.netlib_shaped_lp(400, 1600, 6000)


@pytest.mark.parametrize("name", ["afiro", "sparse300", "netlib400"])
@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_scaled_solve_reaches_the_unscaled_optimum(built, name, mode):
    lp = make(name)
    ref = OracleSimplex(lp)
    assert ref.dual() == 0
    o = OracleSimplex(lp)
    o.set_option("scaling", mode)
    assert o.dual() == 0
    applied, rs, cs = o.scale_factors()
    assert applied and np.all(rs > 0) and np.all(cs > 0)
    assert abs(o.objective - ref.objective) <= 1e-9 * (1 + abs(ref.objective))
    kkt_check(lp, o)  # primal/dual feasibility and complementarity in ORIGINAL units


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_scale_factor_properties(built, mode):
    # a well-conditioned LP whose rows and columns were multiplied by factors spanning six orders of
    # magnitude: exactly the defect row/column scaling can undo
    from clp_amd.mps import LpData

    base = P.sparse_lp(300, 1200, 8, 11)
    rng = np.random.default_rng(5)
    r, cfac = 10.0 ** rng.uniform(-3, 3, base.m), 10.0 ** rng.uniform(-3, 3, base.n)
    lp = LpData(base)
    col_of = np.repeat(np.arange(base.n), np.diff(base.col_start))
    lp["elem"] = base.elem * r[base.row] * cfac[col_of]
    lp["row_lower"], lp["row_upper"] = base.row_lower * r, base.row_upper * r
    lp["col_lower"], lp["col_upper"] = base.col_lower / cfac, base.col_upper / cfac
    lp["obj"] = base.obj * cfac
    o = OracleSimplex(lp)
    o.set_option("scaling", mode)
    assert o.dual() == 0
    applied, rs, cs = o.scale_factors()
    assert applied
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(lp.m, lp.n))
    S = sp.diags(rs) @ abs(A) @ sp.diags(cs)
    S = S.tocsc()
    cols = [j for j in range(lp.n) if S.indptr[j + 1] > S.indptr[j] and lp.col_upper[j] > lp.col_lower[j] + 1e-12]
    colmax = np.array([S.data[S.indptr[j]:S.indptr[j + 1]].max() for j in cols])
    # final pass (:4528-4590): every scaled column's largest entry equals overallLargest in [1, 100],
    # unless the "make gap larger" rule (:4573) capped its scale at (upper - lower) / 1e-5
    top = colmax.max()
    assert 1.0 - 1e-12 <= top <= 100.0 * (1 + 1e-12)
    gap_limited = np.isclose(cs[cols], (lp.col_upper[cols] - lp.col_lower[cols]) / 1.0e-5, rtol=1e-12)
    assert np.allclose(colmax[~gap_limited], top, rtol=1e-12)
    assert np.all(colmax[gap_limited] <= top * (1 + 1e-12)) and gap_limited.sum() < len(cols) // 4
    before = abs(A).data.max() / abs(A).data.min()
    after = S.data.max() / S.data.min()
    assert before > 1e9 and after < 1e4  # the spread of the entries collapses


def test_well_scaled_matrix_is_left_alone(built):
    d = np.load(os.path.join(HERE, "golden", "hello_lp.npz"))  # all entries +-1 (:4273 "don't bother")
    from clp_amd.mps import LpData

    lp = LpData({k: (int(d[k]) if k in ("m", "n") else d[k]) for k in d.files if k != "optimum"})
    lp["name"] = "hello"
    o = OracleSimplex(lp)
    o.set_option("scaling", 3)
    assert o.dual() == 0
    applied, rs, cs = o.scale_factors()
    assert not applied and np.all(rs == 1.0) and np.all(cs == 1.0)


@pytest.mark.parametrize("name", ["afiro", "sparse300", "netlib400"])
@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_engine_scale_factors_equal_the_oracle(built, name, mode):
    """clpgpu_scale_factors is host code in libclpgpu.so (no device needed): the factors the engine
    applies with option "scaling" are bit-identical to the oracle's restatement of the same routine."""
    from clp_amd.engine import scale_factors

    lp = make(name)
    applied, rs, cs = scale_factors(lp, mode)
    o = OracleSimplex(lp)
    o.set_option("scaling", mode)
    assert o.dual() == 0
    applied_o, rs_o, cs_o = o.scale_factors()
    assert applied == applied_o is True or (applied == applied_o)
    assert np.array_equal(rs, rs_o) and np.array_equal(cs, cs_o)

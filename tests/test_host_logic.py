"""CPU tests of host-side logic: MPS reader, instance generators, column-range sharding."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from clp_amd import problems as P
from clp_amd.mps import read_mps
from clp_amd.sharding import column_ranges, merge_candidates

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_afiro_shape(afiro):
    # Netlib AFIRO: 27 rows, 32 columns, 83 constraint nonzeros (SURVEY.md 8 header)
    assert (afiro.m, afiro.n, len(afiro.elem)) == (27, 32, 83)
    assert np.count_nonzero(afiro.obj) == 5
    assert np.sum(afiro.row_lower == afiro.row_upper) == 8  # E rows


def test_reference_example_mps_if_present():
    path = "/root/reference/examples/hello.mps"
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted (GPU box)")
    lp = read_mps(path)
    assert (lp.m, lp.n) == (21, 53)


def test_generators_are_deterministic_and_sorted():
    a, b = P.sparse_lp(200, 700, 6, seed=5), P.sparse_lp(200, 700, 6, seed=5)
    assert np.array_equal(a.row, b.row) and np.array_equal(a.elem, b.elem)
    for j in range(a.n):
        r = a.row[a.col_start[j]:a.col_start[j + 1]]
        assert np.all(np.diff(r) > 0)  # distinct, ascending
    assert np.all(np.abs(a.elem) >= 0.05)
    d = P.dense_lp(30, 40, seed=1)
    assert len(d.elem) == 1200 and np.all(d.obj > 0)


def test_synthetic_is_feasible_by_construction():
    lp = P.sparse_lp(150, 500, 5, seed=8)
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(lp.m, lp.n))
    mid = 0.5 * (lp.row_lower + lp.row_upper)
    # some x in the box reaches the middle of every row range (least squares sanity)
    assert np.all(lp.row_upper - lp.row_lower > 0)
    assert np.all(lp.col_lower == 0)
    assert A.shape == (150, 500) and np.isfinite(mid).all()


def test_column_ranges_partition():
    for n, r in ((200000, 8), (32, 3), (5, 8)):
        rng = column_ranges(n, r)
        assert rng[0][0] == 0 and rng[-1][1] == n
        assert all(rng[i][1] == rng[i + 1][0] for i in range(r - 1))
        sizes = [b - a for a, b in rng]
        assert max(sizes) - min(sizes) <= 256 * r or n < 256 * r


def test_merge_candidates_equals_unsharded(built):
    """Rank-major concatenation of per-shard candidate lists + min of upperTheta reproduces the
    single-shard result (reference reduce: src/ClpPackedMatrix.cpp:1848-1854)."""
    from oracle.oracle import OracleSimplex

    lp = P.sparse_lp(200, 900, 6, seed=4)
    o = OracleSimplex(lp)
    rng = np.random.default_rng(2)
    m, n = lp.m, lp.n
    idx = np.sort(rng.choice(m, 50, replace=False)).astype(np.int32)
    val = rng.standard_normal(50)
    status = rng.choice([1, 2, 3], size=n + m, p=[0.2, 0.3, 0.5]).astype(np.uint8)
    dj = np.where((status & 3) == 2, -1.0, 1.0) * rng.uniform(0, 2, n + m)
    full = o.price_row_fused(idx, val, status, dj)
    parts = []
    for r, (a, b) in enumerate(column_ranges(n, 4)):
        st = status.copy()
        st[:a] = 1  # columns outside the shard look basic -> skipped
        st[b:n] = 1
        if r:
            st[n:] = 1  # the row (slack) part belongs to rank 0
        parts.append(o.price_row_fused(idx, val, st, dj))
    merged = merge_candidates(parts)
    for k in range(4):
        assert np.array_equal(merged[k], full[k])
    assert merged[4] == full[4]


def test_newton_schulz_refresh_model():
    """The algebra behind the engine's verified refresh, in numpy: a drifted inverse X of a
    sparse, moderately conditioned C -- what a few hundred rank-1 updates leave -- comes back to the accuracy of a
    fresh inverse with ONE step X += X (I - C X); and the GEMM argument order the engine hands to a column-major
    BLAS for its row-major arrays (A = R, B = X for X R; A = X, B = C for C X) computes those products."""
    rng = np.random.default_rng(7)
    k, ld = 200, 208
    C = sp.random(k, k, density=0.05, random_state=3, format="csr").toarray() + np.diag(rng.uniform(1.0, 2.0, k))
    exact = np.linalg.inv(C)
    X = exact * (1.0 + 1.0e-8 * rng.standard_normal((k, k)))  # relative drift 1e-8 (max |I - C X| ~ 1e-7, the size measured on the GPU)
    before = np.abs(np.eye(k) - C @ X).max()
    R = np.eye(k) - C @ X
    X1 = X + X @ R
    after = np.abs(np.eye(k) - C @ X1).max()
    fresh = np.abs(np.eye(k) - C @ exact).max()
    # quadratic: the residual after the step is the square of the one before, down to the rounding floor of a fresh inverse
    assert before > 5.0e-8 and after <= 10.0 * k * before * before + 50.0 * fresh and after < 1.0e-4 * before

    # row-major buffers with a leading dimension, as the engine holds them
    def rowmajor(M):
        buf = np.zeros((k, ld))
        buf[:, :k] = M
        return buf

    def colmajor_gemm(a_buf, b_buf):
        # what a column-major BLAS computes from two row-major buffers passed with lda = ldb = ld and m = n = k:
        # it sees A = a_buf^T, B = b_buf^T and returns (A B) stored column-major = (A B)^T stored row-major
        A, B = a_buf[:, :k].T, b_buf[:, :k].T
        return (A @ B).T

    assert np.allclose(colmajor_gemm(rowmajor(R), rowmajor(X)), X @ R, rtol=0, atol=1e-13)
    assert np.allclose(colmajor_gemm(rowmajor(X), rowmajor(C)), C @ X, rtol=0, atol=1e-12)


def test_product_form_algebra():
    """The eta file of the LU mode (clp_amd/csrc/lu_kernels.hip) in numpy: with eta_j = (w_j - e_pj) / alpha_j, H = [eta_j],
    N[j][i] = eta_i[p_j] (i < j) and G = (I + N)^-1 grown one row per update (G[t] = -n_t^T G),
        B_t^-1 v = x0 - H (G x0[P]),   B_t^-T c = B0^-T (c - P (G^T (H^T c)))     with x0 = B0^-1 v,
    including positions replaced more than once.  This is the algebra the kernels implement without a t-step chain."""
    rng = np.random.default_rng(0)
    m = 40
    B = rng.standard_normal((m, m)) + 5 * np.eye(m)
    B0inv = np.linalg.inv(B)
    H, P, G = [], [], np.zeros((0, 0))
    for j in range(25):
        a = rng.standard_normal(m)
        p = int(rng.integers(0, m)) if (j % 3 or not P) else P[-1]  # every third update hits the previous position again

        def ftran(v):
            x0 = B0inv @ v
            return x0 if not H else x0 - np.array(H).T @ (G @ x0[P])

        def btran(c):
            cp = c.copy()
            if H:
                d = G.T @ (np.array(H) @ c)
                for jj, pp in enumerate(P):
                    cp[pp] -= d[jj]
            return B0inv.T @ cp

        w = ftran(a)
        assert np.allclose(w, np.linalg.solve(B, a))
        c = rng.standard_normal(m)
        assert np.allclose(btran(c), np.linalg.solve(B.T, c))
        e = np.zeros(m)
        e[p] = 1.0
        assert np.allclose(btran(e), np.linalg.solve(B.T, e))
        eta = (w - e) / w[p]
        t = len(H)
        n = np.array([H[i][p] for i in range(t)])
        Gn = np.zeros((t + 1, t + 1))
        Gn[:t, :t] = G
        if t:
            Gn[t, :t] = -(n @ G)
        Gn[t, t] = 1.0
        G = Gn
        H.append(eta)
        P.append(p)
        B[:, p] = a

"""The host side of the basis factorization (clp_amd/csrc/lu_front.h luFrontFactor, through the device-free hook clpgpu_test_lu_front): a
right-looking Markowitz LU of the nucleus C = A[R, K] with threshold pivoting that stops when the active block has filled in and leaves a
dense Schur complement for the matrix cores -- what stands in for CoinAbcBaseFactorization::factorSparse (src/CoinAbcBaseFactorization2.cpp:18),
the singleton pivots (src/CoinAbcBaseFactorization1.cpp:2589) and wantToGoDense (:2409-2462).  Every refactorization in LU mode goes
through it; on the GPU it is only seen through the solves it feeds (tests/test_gpu_lu.py).  Here, without a device, on nuclei of real bases:

  * C = L U + S exactly as the elimination defines it (L: unit pivot rows + multipliers, U: pivot rows, S: what is left on the tail's rows
    and columns), to rounding;
  * pivots, tail rows and tail columns partition the nucleus; every pivot passes the threshold test against the entries of its own U row;
  * the stop rule: stop density 0 leaves everything to the tail, a full elimination solves C x = b like scipy's sparse LU."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from clp_amd import problems as P


def nucleus(lp, status):
    """C = A[R, K]: rows whose slack is nonbasic x basic structural columns (square by counting)"""
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(lp.m, lp.n))
    K = np.flatnonzero((status[: lp.n] & 7) == 1)
    R = np.flatnonzero((status[lp.n:] & 7) != 1)
    assert len(K) == len(R)
    return A[R][:, K].tocsc()


def rebuild(F):
    k, nF = F["k"], F["pivots"]
    L = sp.lil_matrix((k, max(nF, 1)))
    U = sp.lil_matrix((max(nF, 1), k))
    for f in range(nF):
        L[F["frow"][f], f] = 1.0
        a, b = F["lStart"][f], F["lStart"][f + 1]
        for r, v in zip(F["lRow"][a:b], F["lVal"][a:b]):
            L[r, f] = v
        U[f, F["fcol"][f]] = F["fpiv"][f]
        a, b = F["uStart"][f], F["uStart"][f + 1]
        for c, v in zip(F["uCol"][a:b], F["uVal"][a:b]):
            U[f, c] = v
    S = sp.coo_matrix((F["sVal"], (F["tailRow"][F["sRow"]], F["tailCol"][F["sCol"]])), shape=(k, k)) if len(F["sVal"]) else sp.coo_matrix((k, k))
    return (L.tocsr() @ U.tocsr() + S.tocsr()).toarray()


def checks(C, F, threshold):
    k = C.shape[0]
    dense = C.toarray()
    err = np.max(np.abs(rebuild(F) - dense)) / max(1.0, np.max(np.abs(dense)))
    assert err < 1e-11, err
    assert F["pivots"] + F["tail"] == k
    assert sorted(np.concatenate([F["frow"], F["tailRow"]]).tolist()) == list(range(k))
    assert sorted(np.concatenate([F["fcol"], F["tailCol"]]).tolist()) == list(range(k))
    # the threshold test against what was left of the pivot's own row when it was chosen
    for f in range(F["pivots"]):
        a, b = F["uStart"][f], F["uStart"][f + 1]
        rest = np.max(np.abs(F["uVal"][a:b])) if b > a else 0.0
        assert abs(F["fpiv"][f]) >= threshold * max(rest, abs(F["fpiv"][f])) * (1 - 1e-12)
    return err


@pytest.mark.parametrize("maker,args,pivots", [("sparse_lp", (600, 2400, 8, 11), 700), ("netlib_shaped_lp", (500, 2000, 12000, 3), 500)])
def test_front_reproduces_the_nucleus(built, maker, args, pivots):
    from clp_amd.engine import lu_front
    from oracle.oracle import OracleSimplex

    lp = getattr(P, maker)(*args)
    o = OracleSimplex(lp)
    o.set_option("pivot_rule", 1)
    o.set_option("max_iterations", pivots)
    o.dual()
    C = nucleus(lp, o.status())
    k = C.shape[0]
    assert k > 100
    for stop, min_tail in ((0.03, 0), (0.3, 16), (1.0, 0)):
        F = lu_front(C, stop, min_tail, 0.1)
        err = checks(C, F, 0.1)
        print(f"{maker} k {k} nnz {C.nnz}: stop density {stop}, min tail {min_tail}: {F['pivots']} pivots, tail {F['tail']}, fill {F['fill']}, C - (L U + S) {err:.1e}")
    # stop density 0: nothing is eliminated, the tail is the nucleus
    F0 = lu_front(C, 0.0, 0, 0.1)
    assert F0["pivots"] == 0 and F0["tail"] == k and len(F0["sVal"]) == C.nnz
    assert np.max(np.abs(rebuild(F0) - C.toarray())) == 0.0


def test_full_elimination_solves_like_scipy(built):
    """stop density 1 and no minimum tail: the front runs to the end unless no acceptable pivot is left; with L, U and the (small) tail the
    system C x = b is solved by forward elimination, a dense solve on the tail and back substitution -- the order the device's solves use"""
    from clp_amd.engine import lu_front

    rng = np.random.default_rng(7)
    k = 400
    C = (sp.random(k, k, density=0.01, random_state=3, data_rvs=rng.standard_normal) + sp.diags(rng.uniform(1.0, 2.0, k))).tocsc()
    perm = rng.permutation(k)
    C = C[perm].tocsc()  # the diagonal is not where a naive elimination would look
    F = lu_front(C, 1.0, 0, 0.1)
    checks(C, F, 0.1)
    assert F["tail"] < 40
    b = rng.standard_normal(k)
    # forward: y_f = (b after the earlier pivots)[frow[f]]; rows below get mult * y_f taken off
    w = b.copy()
    y = np.zeros(F["pivots"])
    for f in range(F["pivots"]):
        y[f] = w[F["frow"][f]]
        a, e = F["lStart"][f], F["lStart"][f + 1]
        w[F["lRow"][a:e]] -= F["lVal"][a:e] * y[f]
    x = np.zeros(k)
    if F["tail"]:
        S = sp.coo_matrix((F["sVal"], (F["sRow"], F["sCol"])), shape=(F["tail"], F["tail"])).toarray()
        x[F["tailCol"]] = np.linalg.solve(S, w[F["tailRow"]])
    for f in range(F["pivots"] - 1, -1, -1):
        a, e = F["uStart"][f], F["uStart"][f + 1]
        x[F["fcol"][f]] = (y[f] - F["uVal"][a:e] @ x[F["uCol"][a:e]]) / F["fpiv"][f]
    ref = spla.spsolve(C, b)
    assert np.max(np.abs(x - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref)))
    assert np.max(np.abs(C @ x - b)) <= 1e-10 * max(1.0, np.max(np.abs(b)))


def test_front_at_the_mature_basis_of_config_4(built):
    """The nucleus the bench's timed window starts from (tests/golden/basis_sparse_30000.npy: order 10 514, 115 766 entries): the front / tail
    splits the engine reports on the GPU for it (tests/test_gpu_mature_parity.py: 4 869 + 5 645 under the default stop density 0.03,
    4 267 + 6 247 under 0.012) come out of the host code alone, and C = L U + S holds entry for entry (sparse arithmetic: the dense form would
    be 0.9 GB)."""
    import os

    from clp_amd.engine import lu_front

    lp = P.sparse_lp()
    here = os.path.dirname(os.path.abspath(__file__))
    status = np.load(os.path.join(here, "golden", "basis_sparse_30000.npy")) & 7
    C = nucleus(lp, status)
    assert C.shape == (10514, 10514) and C.nnz == 115766
    for stop, split in ((0.03, (4869, 5645)), (0.012, (4267, 6247))):
        F = lu_front(C, stop, 0, 0.1)
        assert (F["pivots"], F["tail"]) == split
        k, nF = F["k"], F["pivots"]
        lcount, ucount = np.diff(F["lStart"]), np.diff(F["uStart"])
        L = sp.csr_matrix((np.concatenate([np.ones(nF), F["lVal"]]), (np.concatenate([F["frow"], F["lRow"]]),
                                                                     np.concatenate([np.arange(nF), np.repeat(np.arange(nF), lcount)]))), shape=(k, nF))
        U = sp.csr_matrix((np.concatenate([F["fpiv"], F["uVal"]]), (np.concatenate([np.arange(nF), np.repeat(np.arange(nF), ucount)]),
                                                                    np.concatenate([F["fcol"], F["uCol"]]))), shape=(nF, k))
        S = sp.csr_matrix((F["sVal"], (F["tailRow"][F["sRow"]], F["tailCol"][F["sCol"]])), shape=(k, k))
        D = (L @ U + S - C).tocoo()
        err = float(np.max(np.abs(D.data))) if D.nnz else 0.0
        print(f"stop density {stop}: front {nF} + tail {F['tail']}, fill {F['fill']}, tail entries {len(F['sVal'])} ({len(F['sVal']) / F['tail'] ** 2:.2f} dense), max |L U + S - C| {err:.1e}")
        assert err < 1e-10

"""The host half of the LU factorization (clp_amd/csrc/lu_front.h: Markowitz front with threshold pivoting that stops at a
dense tail) on the CPU: the factors it returns must reproduce the matrix -- L (column etas in pivot order), U rows, and the
remaining active block as the Schur complement -- checked by solving with them in numpy against scipy's sparse direct solve.
No GPU involved: this is host logic (CPU factor + upload)."""
import os
import struct
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lufront") / "harness")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "lu_front_harness.cpp")])
    return exe


def read_vectors(path, dtypes):
    buf = open(path, "rb").read()
    off, out = 0, []
    for dt in dtypes:
        n = struct.unpack_from("q", buf, off)[0]
        off += 8
        out.append(np.frombuffer(buf, dtype=dt, count=n, offset=off).copy())
        off += n * np.dtype(dt).itemsize
    return out


def solve_with_factors(k, fac, b):
    frow, fcol, fpiv, lS, lR, lV, uS, uC, uV, tR, tC, sR, sC, sV = fac
    wr = b.copy()
    for f in range(len(frow)):
        e = slice(lS[f], lS[f + 1])
        wr[lR[e]] -= lV[e] * wr[frow[f]]
    x = np.zeros(k)
    if len(tR):
        S = np.zeros((len(tR), len(tR)))
        np.add.at(S, (sR, sC), sV)
        x[tC] = np.linalg.solve(S, wr[tR])
    for f in range(len(frow) - 1, -1, -1):
        e = slice(uS[f], uS[f + 1])
        x[fcol[f]] = (wr[frow[f]] - np.dot(uV[e], x[uC[e]])) / fpiv[f]
    return x


@pytest.mark.parametrize("k,per_col,density,min_tail", [(400, 3, 0.05, 8), (1500, 6, 0.02, 16), (1500, 6, 1.0, 0), (300, 40, 0.001, 0)])
def test_front_factors_reproduce_the_matrix(harness, tmp_path, k, per_col, density, min_tail):
    rng = np.random.default_rng(k + per_col)
    rows = np.concatenate([rng.choice(k, per_col, replace=False) for _ in range(k)])
    cols = np.repeat(np.arange(k), per_col)
    vals = rng.uniform(0.05, 1.0, len(rows)) * rng.choice([-1.0, 1.0], len(rows))
    C = sp.csc_matrix((vals, (rows, cols)), shape=(k, k)) + sp.identity(k, format="csc") * 3.0  # comfortably nonsingular
    C = C.tocsc()
    C.sort_indices()
    src, dst = str(tmp_path / "C.bin"), str(tmp_path / "F.bin")
    with open(src, "wb") as o:
        o.write(struct.pack("qq", k, C.nnz))
        o.write(C.indptr.astype(np.int32).tobytes())
        o.write(C.indices.astype(np.int32).tobytes())
        o.write(C.data.tobytes())
    out = subprocess.run([harness, src, str(density), str(min_tail), dst], capture_output=True, text=True, check=True).stdout.split()
    nF, k2 = int(out[0]), int(out[1])
    assert nF + k2 == k
    if density >= 1.0:
        assert k2 <= min_tail + 1 or nF >= k - 64  # no density stop: the front runs (nearly) to the end
    if density <= 0.001:
        assert nF == 0  # already denser than the stop rule: everything goes to the dense tail
    fac = read_vectors(dst, [np.int32, np.int32, np.float64, np.int32, np.int32, np.float64, np.int32, np.int32, np.float64, np.int32, np.int32,
                             np.int32, np.int32, np.float64])
    assert sorted(np.concatenate([fac[0], fac[9]]).tolist()) == list(range(k))  # every row is a front pivot row or a tail row
    assert sorted(np.concatenate([fac[1], fac[10]]).tolist()) == list(range(k))
    if nF:
        assert np.min(np.abs(fac[2])) > 1e-11
    lu = sla.splu(C)
    for _ in range(3):
        b = rng.standard_normal(k)
        x = solve_with_factors(k, fac, b)
        ref = lu.solve(b)
        assert np.max(np.abs(x - ref)) <= 1e-9 * (1.0 + np.max(np.abs(ref)))


def test_front_leaves_dependent_columns_to_the_tail(harness, tmp_path):
    """A structurally singular nucleus (an empty column, two identical columns, an empty row): the front must not crash or
    pivot on them; they stay in the active block, where the dense inversion reports the singularity (engine: repair path)."""
    rng = np.random.default_rng(7)
    k = 200
    rows = np.concatenate([rng.choice(k, 4, replace=False) for _ in range(k)])
    cols = np.repeat(np.arange(k), 4)
    vals = rng.uniform(0.1, 1.0, len(rows))
    C = (sp.csc_matrix((vals, (rows, cols)), shape=(k, k)) + sp.identity(k, format="csc") * 2.0).tolil()
    C[:, 17] = 0.0  # empty column
    C[:, 31] = C[:, 30]  # duplicate column
    C[55, :] = 0.0  # empty row
    C = C.tocsc()
    C.eliminate_zeros()
    C.sort_indices()
    src, dst = str(tmp_path / "C.bin"), str(tmp_path / "F.bin")
    with open(src, "wb") as o:
        o.write(struct.pack("qq", k, C.nnz))
        o.write(C.indptr.astype(np.int32).tobytes())
        o.write(C.indices.astype(np.int32).tobytes())
        o.write(C.data.tobytes())
    out = subprocess.run([harness, src, "1.0", "0", dst], capture_output=True, text=True, check=True).stdout.split()
    nF, k2 = int(out[0]), int(out[1])
    assert nF + k2 == k and k2 >= 2
    fac = read_vectors(dst, [np.int32, np.int32, np.float64, np.int32, np.int32, np.float64, np.int32, np.int32, np.float64, np.int32, np.int32,
                             np.int32, np.int32, np.float64])
    assert 17 in fac[10].tolist()  # the empty column was never a pivot column
    assert 55 in fac[9].tolist()  # nor the empty row a pivot row
    assert np.min(np.abs(fac[2])) > 1e-11


def _run_front(harness, tmp_path, C, density, min_tail):
    k = C.shape[0]
    src, dst = str(tmp_path / "C.bin"), str(tmp_path / "F.bin")
    with open(src, "wb") as o:
        o.write(struct.pack("qq", k, C.nnz))
        o.write(C.indptr.astype(np.int32).tobytes())
        o.write(C.indices.astype(np.int32).tobytes())
        o.write(C.data.tobytes())
    out = subprocess.run([harness, src, str(density), str(min_tail), dst], capture_output=True, text=True, check=True).stdout.split()
    fac = read_vectors(dst, [np.int32, np.int32, np.float64, np.int32, np.int32, np.float64, np.int32, np.int32, np.float64, np.int32, np.int32,
                             np.int32, np.int32, np.float64])
    return int(out[0]), int(out[1]), fac


@pytest.mark.parametrize("seed", range(12))
def test_front_on_varied_structures(harness, tmp_path, seed):
    """Shapes a simplex nucleus takes: slack-like singletons mixed in, a band, power-law column lengths, a few long rows, values
    spread over six orders of magnitude.  Whatever the front decides (how far it gets, which pivots the threshold refuses), the
    factors must solve with the matrix to 1e-8 relative to SuperLU -- the growth the threshold 0.1 allows stays small."""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.integers(150, 1200))
    kind = seed % 4
    rows, cols, vals = [], [], []
    for j in range(k):
        if kind == 0:  # band + random
            r = np.unique(np.clip(j + rng.integers(-6, 7, 5), 0, k - 1))
        elif kind == 1:  # power-law column lengths
            length = min(k, max(1, int(1.0 / rng.uniform(0.01, 1.0))))
            r = rng.choice(k, length, replace=False)
        elif kind == 2:  # singletons (slack-like) among short columns
            r = np.array([rng.integers(k)]) if rng.uniform() < 0.4 else rng.choice(k, 3, replace=False)
        else:  # a few long rows
            r = np.unique(np.concatenate([rng.choice(k, 3, replace=False), rng.choice(5, 2)]))
        v = rng.choice([-1.0, 1.0], len(r)) * 10.0 ** rng.uniform(-3.0, 3.0, len(r))
        rows.append(r)
        cols.append(np.full(len(r), j))
        vals.append(v)
    C = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(k, k))
    C = (C + sp.diags(rng.choice([-1.0, 1.0], k) * 10.0 ** rng.uniform(0.0, 2.0, k))).tocsc()  # structurally nonsingular
    C.sum_duplicates()
    C.sort_indices()
    lu = sla.splu(C)
    if np.min(np.abs(lu.U.diagonal())) < 1e-8 * np.max(np.abs(lu.U.diagonal())):
        pytest.skip("numerically singular draw")
    for density, min_tail in ((0.012, 16), (1.0, 0)):
        nF, k2, fac = _run_front(harness, tmp_path, C, density, min_tail)
        assert nF + k2 == k
        for _ in range(2):
            b = rng.standard_normal(k)
            ref = lu.solve(b)
            x = solve_with_factors(k, fac, b)
            assert np.max(np.abs(x - ref)) <= 1e-8 * (1.0 + np.max(np.abs(ref))), (seed, kind, density, nF, k2)

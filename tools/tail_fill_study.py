#!/usr/bin/env python3
"""CPU only.  How sparse is the mature basis of config 4 really?  Takes the committed basis (tests/golden/basis_sparse_30000.npy: pivot 30 000 of
the 50 000 x 200 000 LP), forms the nucleus (basic structurals x rows whose slack is nonbasic), runs the engine's own host Markowitz front
(clp_amd/csrc/lu_front.h through tests/host/lu_front_harness.cpp) at several stop densities, and factors the remaining tail with SuperLU under
three orderings to see what a sparse factorization of it would cost in fill.  Output kept as profiles/r04_tail_fill_study.txt.

    python tools/tail_fill_study.py [--full]       # --full also runs the front with no density stop (about 7 minutes of host time)
    python tools/tail_fill_study.py --netlib <basis.npy>   # the same study on the Netlib-shaped variant (power-law column counts, tools/dump_basis.py netlib N)"""
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from clp_amd import problems as P  # noqa: E402
from test_lu_front_host import read_vectors  # noqa: E402


def main():
    work = tempfile.mkdtemp(prefix="tailfill")
    exe = os.path.join(work, "harness")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "lu_front_harness.cpp")])
    if "--netlib" in sys.argv:
        lp = P.netlib_shaped_lp()
        status = np.load(sys.argv[sys.argv.index("--netlib") + 1]).astype(np.uint8) & 7
    else:
        lp = P.sparse_lp(50000, 200000, 50)
        status = np.load(os.path.join(ROOT, "tests", "golden", "basis_sparse_30000.npy")).astype(np.uint8) & 7
    m, n = lp.m, lp.n
    A = sp.csc_matrix((lp.elem, lp.row, lp.col_start), shape=(m, n))
    columns, rows = np.flatnonzero(status[:n] == 1), np.flatnonzero(status[n:] != 1)
    k = len(columns)
    C = A[:, columns][rows, :].tocsc()
    C.sort_indices()
    print(f"basis ({'Netlib-shaped variant' if '--netlib' in sys.argv else 'config 4 after 30 000 pivots'}): {k} basic structurals, {m - k} basic slacks; nucleus {k} x {k}, {C.nnz} nonzeros ({C.nnz / k:.1f} per column)")
    src, dst = os.path.join(work, "C.bin"), os.path.join(work, "F.bin")
    with open(src, "wb") as o:
        o.write(struct.pack("qq", k, C.nnz))
        o.write(C.indptr.astype(np.int32).tobytes())
        o.write(C.indices.astype(np.int32).tobytes())
        o.write(C.data.tobytes())
    print("host Markowitz front (threshold 0.1), stopped when the active block reaches the given density:")
    print("stop_density  front  tail  L_nnz  U_nnz  S_nnz  dense_tail_entries  host_seconds")
    for density in (0.012, 0.05, 0.2) + ((1.0,) if "--full" in sys.argv else ()):
        t = time.time()
        out = subprocess.run([exe, src, str(density), "16", dst], capture_output=True, text=True, check=True).stdout.split()
        nF, k2 = int(out[0]), int(out[1])
        fac = read_vectors(dst, [np.int32, np.int32, np.float64, np.int32, np.int32, np.float64, np.int32, np.int32, np.float64, np.int32, np.int32,
                                 np.int32, np.int32, np.float64])
        print(f"{density:12g}  {nF:5d}  {k2:4d}  {len(fac[4]):5d}  {len(fac[7]):5d}  {len(fac[13]):5d}  {k2 * k2:18d}  {time.time() - t:.2f}")
        if density == 0.012:
            S = sp.csc_matrix((fac[13], (fac[11], fac[12])), shape=(k2, k2))
            tail = (k2, S)
    k2, S = tail
    print(f"the tail the engine inverts densely ({k2} x {k2}, {S.nnz} nonzeros = {S.nnz / k2 / k2:.4f} dense) under a sparse LU (SuperLU, threshold 0.1):")
    print("ordering  L_nnz  U_nnz  (L+U)/k2^2  seconds")
    for perm in ("COLAMD", "MMD_AT_PLUS_A", "NATURAL"):
        t = time.time()
        lu = sla.splu(S, permc_spec=perm, diag_pivot_thresh=0.1)
        print(f"{perm:14s}  {lu.L.nnz}  {lu.U.nnz}  {(lu.L.nnz + lu.U.nnz) / k2 / k2:.3f}  {time.time() - t:.1f}")


if __name__ == "__main__":
    main()

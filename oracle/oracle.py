"""ctypes binding of the CPU oracle (oracle/clp_dual_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py.  Nothing under clp_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class PivotRecord(C.Structure):
    _fields_ = [("iteration", C.c_int), ("sequenceIn", C.c_int), ("sequenceOut", C.c_int),
                ("pivotRow", C.c_int), ("numberFlipped", C.c_int), ("reserved", C.c_int),
                ("theta", C.c_double), ("alpha", C.c_double), ("dualOut", C.c_double),
                ("objective", C.c_double)]


PIVOT_DTYPE = np.dtype([("iteration", "i4"), ("sequenceIn", "i4"), ("sequenceOut", "i4"), ("pivotRow", "i4"),
                        ("numberFlipped", "i4"), ("reserved", "i4"), ("theta", "f8"), ("alpha", "f8"),
                        ("dualOut", "f8"), ("objective", "f8")])


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libclporacle.so")
    src = os.path.join(_HERE, "clp_dual_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libclporacle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        p = C.c_void_p
        ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        up = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
        L.orc_create.restype = p
        L.orc_create.argtypes = [C.c_int, C.c_int, ip, ip, dp, dp, dp, dp, dp, dp]
        L.orc_destroy.argtypes = [p]
        L.orc_set_option.argtypes = [p, C.c_char_p, C.c_double]
        L.orc_set_status.argtypes = [p, up]
        L.orc_dual.argtypes = [p]
        L.orc_number_iterations.argtypes = [p]
        L.orc_number_refactorizations.argtypes = [p]
        L.orc_number_perturbations.argtypes = [p]
        L.orc_number_backwards.argtypes = [p]
        L.orc_number_loop_flags.argtypes = [p]
        L.orc_number_accuracy_restores.argtypes = [p]
        L.orc_number_singular_restores.argtypes = [p]
        L.orc_test_perturb.argtypes = [p, C.c_int, C.c_int, up, dp]
        L.orc_objective_value.argtypes = [p]
        L.orc_objective_value.restype = C.c_double
        L.orc_iteration_seconds.argtypes = [p]
        L.orc_iteration_seconds.restype = C.c_double
        for f in ("orc_get_solution", "orc_get_reduced_costs", "orc_get_row_duals"):
            getattr(L, f).argtypes = [p, dp]
        L.orc_get_status.argtypes = [p, up]
        L.orc_get_pivot_variable.argtypes = [p, ip]
        L.orc_get_pivot_log.argtypes = [p, C.c_void_p, C.c_int]
        L.orc_get_row_weights.argtypes = [p, dp, dp]
        L.orc_get_scale_factors.argtypes = [p, dp, dp]
        L.orc_times.argtypes = [p, C.c_double, dp, dp]
        L.orc_transpose_times.argtypes = [p, C.c_double, dp, dp]
        L.orc_price_row_fused.argtypes = [p, C.c_int, ip, dp, up, dp, C.c_double, C.c_double, C.c_double, ip, dp,
                                          C.POINTER(C.c_int), ip, dp, C.POINTER(C.c_double)]
        L.orc_factorize.argtypes = [p, up, ip]
        L.orc_ftran.argtypes = [p, dp]
        L.orc_btran.argtypes = [p, dp]
        L.orc_replace_column.argtypes = [p, dp, C.c_int, C.c_double]
        _LIB = L
    return _LIB


class OracleSimplex:
    """CPU oracle with a ClpSimplex-shaped surface (loadProblem / dual / getters)."""

    def __init__(self, lp):
        L = lib()
        self.lp = lp
        self.m, self.n = int(lp.m), int(lp.n)
        c = np.ascontiguousarray
        self._h = L.orc_create(self.m, self.n, c(lp.col_start, dtype=np.int32), c(lp.row, dtype=np.int32),
                               c(lp.elem, dtype=np.float64), c(lp.col_lower, dtype=np.float64),
                               c(lp.col_upper, dtype=np.float64), c(lp.obj, dtype=np.float64),
                               c(lp.row_lower, dtype=np.float64), c(lp.row_upper, dtype=np.float64))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def set_option(self, name, value):
        if lib().orc_set_option(self._h, name.encode(), float(value)) != 0:
            raise KeyError(name)

    def set_status(self, status):
        lib().orc_set_status(self._h, np.ascontiguousarray(status, dtype=np.uint8))

    def dual(self):
        return lib().orc_dual(self._h)

    @property
    def iterations(self):
        return lib().orc_number_iterations(self._h)

    @property
    def refactorizations(self):
        return lib().orc_number_refactorizations(self._h)

    def test_perturb(self, perturbation, iterations, status):
        """(return code, perturbation_ afterwards, perturbed costs) of ClpSimplexDual::perturb on a fresh rim."""
        cost = np.zeros(self.m + self.n)
        r = lib().orc_test_perturb(self._h, int(perturbation), int(iterations), np.ascontiguousarray(status, dtype=np.uint8), cost)
        return r // 1000, r % 1000, cost

    @property
    def backwards(self):
        return lib().orc_number_backwards(self._h)

    @property
    def singular_restores(self):
        return lib().orc_number_singular_restores(self._h)

    @property
    def accuracy_restores(self):
        return lib().orc_number_accuracy_restores(self._h)

    @property
    def loop_flags(self):
        return lib().orc_number_loop_flags(self._h)

    @property
    def perturbations(self):
        return lib().orc_number_perturbations(self._h)

    @property
    def objective(self):
        return lib().orc_objective_value(self._h)

    @property
    def seconds(self):
        return lib().orc_iteration_seconds(self._h)

    def _vec(self, fn, dtype=np.float64, size=None):
        out = np.zeros(size or (self.m + self.n), dtype=dtype)
        getattr(lib(), fn)(self._h, out)
        return out

    def solution(self):
        return self._vec("orc_get_solution")

    def reduced_costs(self):
        return self._vec("orc_get_reduced_costs")

    def status(self):
        return self._vec("orc_get_status", np.uint8)

    def pivot_variable(self):
        return self._vec("orc_get_pivot_variable", np.int32, self.m)

    def row_duals(self):
        return self._vec("orc_get_row_duals", size=self.m)

    def row_weights(self):
        w, inf = np.zeros(self.m), np.zeros(self.m)
        lib().orc_get_row_weights(self._h, w, inf)
        return w, inf

    def scale_factors(self):
        """(scaled?, rowScale[m], columnScale[n]) of the last dual(): ClpPackedMatrix::scale factors"""
        rs, cs = np.empty(self.m), np.empty(self.n)
        applied = lib().orc_get_scale_factors(self._h, rs, cs)
        return bool(applied), rs, cs

    def pivot_log(self):
        count = lib().orc_get_pivot_log(self._h, None, 0)
        out = np.zeros(count, dtype=PIVOT_DTYPE)
        if count:
            lib().orc_get_pivot_log(self._h, out.ctypes.data_as(C.c_void_p), count)
        return out

    # ---- unit-level ----
    def times(self, scalar, x, y):
        y = np.array(y, dtype=np.float64)
        lib().orc_times(self._h, scalar, np.ascontiguousarray(x, dtype=np.float64), y)
        return y

    def transpose_times(self, scalar, x, y):
        y = np.array(y, dtype=np.float64)
        lib().orc_transpose_times(self._h, scalar, np.ascontiguousarray(x, dtype=np.float64), y)
        return y

    def price_row_fused(self, pi_index, pi_value, status, dj, zero_tol=1e-13, dual_tol=1e-7, acceptable_pivot=1e-9):
        n, m = self.n, self.m
        out_i = np.zeros(n, np.int32)
        out_v = np.zeros(n)
        cand_i = np.zeros(n + m, np.int32)
        cand_v = np.zeros(n + m)
        ncand = C.c_int(0)
        upper = C.c_double(0.0)
        pi_index = np.ascontiguousarray(pi_index, dtype=np.int32)
        nnz = lib().orc_price_row_fused(self._h, len(pi_index), pi_index, np.ascontiguousarray(pi_value, dtype=np.float64),
                                        np.ascontiguousarray(status, dtype=np.uint8), np.ascontiguousarray(dj, dtype=np.float64),
                                        zero_tol, dual_tol, acceptable_pivot, out_i, out_v, C.byref(ncand), cand_i, cand_v,
                                        C.byref(upper))
        return (out_i[:nnz].copy(), out_v[:nnz].copy(), cand_i[:ncand.value].copy(), cand_v[:ncand.value].copy(),
                upper.value)

    def factorize(self, status):
        pv = np.zeros(self.m, np.int32)
        rc = lib().orc_factorize(self._h, np.ascontiguousarray(status, dtype=np.uint8), pv)
        return rc, pv

    def ftran(self, v):
        v = np.array(v, dtype=np.float64)
        lib().orc_ftran(self._h, v)
        return v

    def btran(self, v):
        v = np.array(v, dtype=np.float64)
        lib().orc_btran(self._h, v)
        return v

    def replace_column(self, w, pivot_row, alpha):
        return lib().orc_replace_column(self._h, np.ascontiguousarray(w, dtype=np.float64), int(pivot_row), float(alpha))


def _looping_call(f, has_restype, objective, infeasibility, count, iteration, flag_bits, newest):
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    n = len(objective)
    d = [np.ascontiguousarray(a, dtype=np.float64) for a in (objective, infeasibility)]
    i = [np.ascontiguousarray(a, dtype=np.int32) for a in (count, iteration, flag_bits, newest)]
    code, force, flagged = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    tol, bound = np.zeros(n), np.zeros(n)
    f.argtypes = [C.c_int, dp, dp, ip, ip, ip, ip, ip, dp, dp, ip, ip]
    f.restype = C.c_int if has_restype else None
    r = f(n, d[0], d[1], i[0], i[1], i[2], i[3], code, tol, bound, force, flagged)
    assert not has_restype or r == 0
    return code, tol, bound, force, flagged


def test_looping(objective, infeasibility, count, iteration, flag_bits, newest):
    """ClpSimplexProgress::looping of the oracle over a sequence of status checks (orc_test_looping)"""
    return _looping_call(lib().orc_test_looping, False, objective, infeasibility, count, iteration, flag_bits, newest)


def test_cycle(seq_in, seq_out, way_in, way_out):
    """ClpSimplexProgress::cycle of the oracle over a sequence of pivots (orc_test_cycle)"""
    arr = [np.ascontiguousarray(a, dtype=np.int32) for a in (seq_in, seq_out, way_in, way_out)]
    out = np.zeros(len(arr[0]), np.int32)
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f = lib().orc_test_cycle
    f.restype = None
    f.argtypes = [C.c_int, ip, ip, ip, ip, ip]
    f(len(out), *arr, out)
    return out

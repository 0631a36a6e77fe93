"""ctypes binding of the CPU oracle (oracle/clp_dual_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py.  Nothing under clp_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# ---- committed results of the oracle's LONG solves (tests/golden/oracle_cache/, written by tests/golden/make_oracle_cache.py) ----
# The GPU parity tests compare the engine with finished oracle solves that take one CPU core minutes (1500 x 6000 under Dantzig: 19 484
# pivots): inside a GPU lease that is wasted GPU time.  A solve whose inputs hash to a committed record -- the oracle's SOURCE, the LP's
# arrays, every option, the starting statuses -- returns that record (status code, counters, solution, reduced costs, statuses,
# pivotVariable, weights, scale factors, pivot log) instead of running; anything the record cannot answer (a second dual() on the same
# model, the unit-level calls) first replays the real solve.  A changed oracle source misses every record and runs live.
_CACHE_DIR = os.environ.get("CLP_ORACLE_CACHE", os.path.join(os.path.dirname(_HERE), "tests", "golden", "oracle_cache"))
_CACHE_WRITE = os.environ.get("CLP_ORACLE_CACHE_WRITE", "")  # a directory: solves longer than _CACHE_MIN_S are recorded there
_CACHE_MIN_S = float(os.environ.get("CLP_ORACLE_CACHE_MIN_S", "2.0"))
# Carrying records across a source change that cannot change them.  tests/golden/make_oracle_cache.py --rekey-from HASH: a first dual() that
# finds no record under the current source hash but one under HASH copies it to its new name instead of solving again (the mature-basis
# record alone is twenty minutes of one core).  Only for a change confined to the SCALED path of the oracle -- the guard below refuses any
# record whose solve applied scale factors or asked for scaling -- and the change must be named in the commit that uses it.
_REKEY_FROM = os.environ.get("CLP_ORACLE_REKEY_FROM", "")
_SRC_HASH = None


def _source_hash():
    global _SRC_HASH
    if _SRC_HASH is None:
        h = hashlib.sha1()
        for name in ("clp_dual_oracle.c", "clp_dual_oracle.h", "Makefile"):
            with open(os.path.join(_HERE, name), "rb") as f:
                h.update(f.read())
        _SRC_HASH = h.hexdigest()
    return _SRC_HASH


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
        L.orc_number_try_primal.argtypes = [p]
        L.orc_number_partial_scans.argtypes = [p]
        L.orc_number_chuzr_recalls.argtypes = [p]
        L.orc_factor_elements.argtypes = [p]
        L.orc_factor_elements.restype = C.c_long
        L.orc_number_free_first_rows.argtypes = [p]
        L.orc_number_free_entered.argtypes = [p]
        L.orc_test_perturb.argtypes = [p, C.c_int, C.c_int, up, dp]
        L.orc_objective_value.argtypes = [p]
        L.orc_objective_value.restype = C.c_double
        L.orc_iteration_seconds.argtypes = [p]
        L.orc_iteration_seconds.restype = C.c_double
        L.orc_startup_seconds.argtypes = [p]
        L.orc_startup_seconds.restype = C.c_double
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
        self._opts = {}        # final value of every option set (the key of a recorded solve)
        self._start = None     # starting statuses, if any
        self._duals = 0        # dual() calls so far
        self._rec = None       # the committed record standing in for the first dual(), until something needs the live model
        self._lp_hash = None

    # ---- recorded solves ----
    def _key(self, source=None):
        if self._lp_hash is None:
            h = hashlib.sha1()
            lp = self.lp
            h.update(np.array([self.m, self.n], dtype=np.int64).tobytes())
            for a, dt in ((lp.col_start, np.int32), (lp.row, np.int32), (lp.elem, np.float64), (lp.col_lower, np.float64),
                          (lp.col_upper, np.float64), (lp.obj, np.float64), (lp.row_lower, np.float64), (lp.row_upper, np.float64)):
                h.update(np.ascontiguousarray(a, dtype=dt).tobytes())
            self._lp_hash = h.hexdigest()
        h = hashlib.sha1()
        h.update((source or _source_hash()).encode())
        h.update(self._lp_hash.encode())
        h.update(repr(sorted(self._opts.items())).encode())
        h.update(b"none" if self._start is None else self._start.tobytes())
        return h.hexdigest()

    def _rekey(self, path):
        """copy the record this solve had under the source hash _REKEY_FROM to `path`; True if there was one and it may be carried over"""
        if not (_REKEY_FROM and _CACHE_WRITE) or self._opts.get("scaling", 0.0) != 0.0:
            return False
        old = os.path.join(_CACHE_DIR, self._key(_REKEY_FROM) + ".npz")
        if not os.path.exists(old):
            return False
        with np.load(old) as z:
            if int(z["counters"][8]) != 0:  # the recorded solve applied scale factors: not carried over
                return False
        import shutil

        shutil.copyfile(old, path)
        print(f"oracle record {os.path.basename(old)[:12]} -> {os.path.basename(path)[:12]} (m {self.m}, n {self.n})", flush=True)
        return True

    def _live(self):
        """bring the C model to the state the record describes (replays the recorded solve)"""
        if self._rec is not None:
            self._rec = None
            lib().orc_dual(self._h)

    def _snapshot(self, code):
        L = lib()
        w, inf = np.zeros(self.m), np.zeros(self.m)
        L.orc_get_row_weights(self._h, w, inf)
        rs, cs = np.empty(self.m), np.empty(self.n)
        applied = L.orc_get_scale_factors(self._h, rs, cs)
        count = L.orc_get_pivot_log(self._h, None, 0)
        log = np.zeros(count, dtype=PIVOT_DTYPE)
        if count:
            L.orc_get_pivot_log(self._h, log.ctypes.data_as(C.c_void_p), count)
        counters = np.array([code, L.orc_number_iterations(self._h), L.orc_number_refactorizations(self._h), L.orc_number_perturbations(self._h),
                             L.orc_number_backwards(self._h), L.orc_number_loop_flags(self._h), L.orc_number_accuracy_restores(self._h),
                             L.orc_number_singular_restores(self._h), applied, L.orc_number_free_first_rows(self._h),
                             L.orc_number_free_entered(self._h), L.orc_number_try_primal(self._h), L.orc_number_partial_scans(self._h),
                             L.orc_number_chuzr_recalls(self._h), L.orc_factor_elements(self._h)], dtype=np.int64)
        return dict(counters=counters, scalars=np.array([L.orc_objective_value(self._h), L.orc_iteration_seconds(self._h)]),
                    solution=self._vec("orc_get_solution"), reduced_costs=self._vec("orc_get_reduced_costs"),
                    status=self._vec("orc_get_status", np.uint8), pivot_variable=self._vec("orc_get_pivot_variable", np.int32, self.m),
                    row_duals=self._vec("orc_get_row_duals", size=self.m), weights=w, infeas=inf, row_scale=rs, col_scale=cs, pivot_log=log)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def set_option(self, name, value):
        self._live()
        if lib().orc_set_option(self._h, name.encode(), float(value)) != 0:
            raise KeyError(name)
        self._opts[name] = float(value)

    def set_status(self, status):
        self._live()
        self._start = np.ascontiguousarray(status, dtype=np.uint8).copy()
        lib().orc_set_status(self._h, self._start)

    def has_record(self):
        """True when the next first dual() of this model would be answered from a committed record"""
        return self._duals == 0 and os.path.exists(os.path.join(_CACHE_DIR, self._key() + ".npz"))

    def dual(self, live=False):
        """live=True: never answered from a committed record (a timing must be this machine's: bench.py's cpu_baseline)"""
        self._live()
        self._duals += 1
        first = self._duals == 1
        if first and not live and os.path.isdir(_CACHE_DIR):
            path = os.path.join(_CACHE_DIR, self._key() + ".npz")
            if os.path.exists(path) or self._rekey(path):
                try:
                    with np.load(path) as z:
                        self._rec = {k: z[k] for k in z.files}
                    if os.environ.get("CLP_ORACLE_TRACE"):
                        print(f"oracle record {os.path.basename(path)[:12]} answers a solve (m {self.m}, n {self.n})", flush=True)
                    return int(self._rec["counters"][0])
                except Exception:  # noqa: BLE001 -- an unreadable record is no record: solve live
                    self._rec = None
        t0 = time.time()
        code = lib().orc_dual(self._h)
        if first and _CACHE_WRITE and time.time() - t0 >= _CACHE_MIN_S:
            os.makedirs(_CACHE_WRITE, exist_ok=True)
            tmp = os.path.join(_CACHE_WRITE, f".{os.getpid()}.tmp.npz")
            np.savez_compressed(tmp, **self._snapshot(code))
            os.replace(tmp, os.path.join(_CACHE_WRITE, self._key() + ".npz"))  # a record is either whole or absent
        return code

    def _counter(self, index, fn):
        if self._rec is not None:
            return int(self._rec["counters"][index])
        return getattr(lib(), fn)(self._h)

    @property
    def iterations(self):
        return self._counter(1, "orc_number_iterations")

    @property
    def refactorizations(self):
        return self._counter(2, "orc_number_refactorizations")

    def test_perturb(self, perturbation, iterations, status):
        """(return code, perturbation_ afterwards, perturbed costs) of ClpSimplexDual::perturb on a fresh rim."""
        self._live()
        cost = np.zeros(self.m + self.n)
        r = lib().orc_test_perturb(self._h, int(perturbation), int(iterations), np.ascontiguousarray(status, dtype=np.uint8), cost)
        return r // 1000, r % 1000, cost

    @property
    def backwards(self):
        return self._counter(4, "orc_number_backwards")

    @property
    def singular_restores(self):
        return self._counter(7, "orc_number_singular_restores")

    @property
    def free_first_rows(self):
        """pivot rows chosen by dualRow's free-first entry (option free_nonbasic 1)"""
        return self._counter(9, "orc_number_free_first_rows")

    @property
    def free_entered(self):
        """pivots whose incoming variable was a free one taken by the general branch of dualColumn0 (option free_nonbasic 1)"""
        return self._counter(10, "orc_number_free_entered")

    @property
    def try_primal(self):
        """times gutsOfDual's "problems - try primal" exit ended the solve with status 10 (src/ClpSimplexDual.cpp:540-547)"""
        return self._counter(11, "orc_number_try_primal")

    @property
    def partial_scans(self):
        """ClpDualRowSteepest::pivotRow calls that stopped on numberWanted (src/ClpDualRowSteepest.cpp:258-278, :329-335)"""
        return self._counter(12, "orc_number_partial_scans")

    chuzr_partial_scans = partial_scans

    @property
    def chuzr_recalls(self):
        """second calls of pivotRow with largestDualError 0 (:338-346)"""
        return self._counter(13, "orc_number_chuzr_recalls")

    @property
    def factor_elements(self):
        """what stood for factorization()->numberElements() at the last factorization (option steepest_elements)"""
        return self._counter(14, "orc_factor_elements")

    @property
    def accuracy_restores(self):
        return self._counter(6, "orc_number_accuracy_restores")

    @property
    def loop_flags(self):
        return self._counter(5, "orc_number_loop_flags")

    @property
    def perturbations(self):
        return self._counter(3, "orc_number_perturbations")

    @property
    def objective(self):
        if self._rec is not None:
            return float(self._rec["scalars"][0])
        return lib().orc_objective_value(self._h)

    @property
    def startup_seconds(self):
        """the part of `seconds` spent before the first status check (start-up factorization + resync), this machine's clock only"""
        if self._rec is not None:
            raise RuntimeError("a solve answered from a committed record has no clock of this machine (dual(live=True) times it here)")
        return lib().orc_startup_seconds(self._h)

    @property
    def seconds(self):
        """wall time of the last dual() ON THIS MACHINE; a solve answered from a committed record has none (the record's clock is
        the authoring box's: `recorded_seconds` says so explicitly)"""
        if self._rec is not None:
            raise RuntimeError("OracleSimplex.seconds: this solve was answered from a committed record; use dual(live=True) to time it here, "
                               "or recorded_seconds for the authoring box's clock")
        return lib().orc_iteration_seconds(self._h)

    @property
    def recorded_seconds(self):
        """the authoring box's wall time of a solve answered from a committed record, else None"""
        return float(self._rec["scalars"][1]) if self._rec is not None else None

    def _vec(self, fn, dtype=np.float64, size=None):
        out = np.zeros(size or (self.m + self.n), dtype=dtype)
        getattr(lib(), fn)(self._h, out)
        return out

    def _recorded(self, name):
        return None if self._rec is None else self._rec[name].copy()

    def solution(self):
        r = self._recorded("solution")
        return r if r is not None else self._vec("orc_get_solution")

    def reduced_costs(self):
        r = self._recorded("reduced_costs")
        return r if r is not None else self._vec("orc_get_reduced_costs")

    def status(self):
        r = self._recorded("status")
        return r if r is not None else self._vec("orc_get_status", np.uint8)

    def pivot_variable(self):
        r = self._recorded("pivot_variable")
        return r if r is not None else self._vec("orc_get_pivot_variable", np.int32, self.m)

    def row_duals(self):
        r = self._recorded("row_duals")
        return r if r is not None else self._vec("orc_get_row_duals", size=self.m)

    def row_weights(self):
        if self._rec is not None:
            return self._rec["weights"].copy(), self._rec["infeas"].copy()
        w, inf = np.zeros(self.m), np.zeros(self.m)
        lib().orc_get_row_weights(self._h, w, inf)
        return w, inf

    def scale_factors(self):
        """(scaled?, rowScale[m], columnScale[n]) of the last dual(): ClpPackedMatrix::scale factors"""
        if self._rec is not None:
            return bool(self._rec["counters"][8]), self._rec["row_scale"].copy(), self._rec["col_scale"].copy()
        rs, cs = np.empty(self.m), np.empty(self.n)
        applied = lib().orc_get_scale_factors(self._h, rs, cs)
        return bool(applied), rs, cs

    def pivot_log(self):
        if self._rec is not None:
            return self._rec["pivot_log"].copy()
        count = lib().orc_get_pivot_log(self._h, None, 0)
        out = np.zeros(count, dtype=PIVOT_DTYPE)
        if count:
            lib().orc_get_pivot_log(self._h, out.ctypes.data_as(C.c_void_p), count)
        return out

    # ---- unit-level ----
    def times(self, scalar, x, y):
        self._live()
        y = np.array(y, dtype=np.float64)
        lib().orc_times(self._h, scalar, np.ascontiguousarray(x, dtype=np.float64), y)
        return y

    def transpose_times(self, scalar, x, y):
        self._live()
        y = np.array(y, dtype=np.float64)
        lib().orc_transpose_times(self._h, scalar, np.ascontiguousarray(x, dtype=np.float64), y)
        return y

    def price_row_fused(self, pi_index, pi_value, status, dj, zero_tol=1e-13, dual_tol=1e-7, acceptable_pivot=1e-9):
        self._live()
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
        self._live()
        pv = np.zeros(self.m, np.int32)
        rc = lib().orc_factorize(self._h, np.ascontiguousarray(status, dtype=np.uint8), pv)
        return rc, pv

    def ftran(self, v):
        self._live()
        v = np.array(v, dtype=np.float64)
        lib().orc_ftran(self._h, v)
        return v

    def btran(self, v):
        self._live()
        v = np.array(v, dtype=np.float64)
        lib().orc_btran(self._h, v)
        return v

    def replace_column(self, w, pivot_row, alpha):
        self._live()
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

"""ctypes binding of libclpgpu.so (include/clpgpu.h) with a ClpSimplex-shaped surface.

This is plumbing: every numerical operation happens in the HIP library.  There is no CPU fallback --
importing works without a GPU (so the C ABI can be inspected), but constructing a
:class:`ClpGpuSimplex` raises when the library or a HIP device is missing.

Method names follow the reference's ClpSimplex / ClpModel API (src/ClpSimplex.hpp, src/ClpModel.hpp):
``loadProblem``, ``dual``, ``setMaximumIterations``, ``objectiveValue``, ``primalColumnSolution``,
``primalRowSolution``, ``dualColumnSolution``, ``dualRowSolution``, ``statusArray``, ``numberIterations``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libclpgpu.so")
_LIB = None

PIVOT_DTYPE = np.dtype([("iteration", "i4"), ("sequenceIn", "i4"), ("sequenceOut", "i4"), ("pivotRow", "i4"),
                        ("numberFlipped", "i4"), ("reserved", "i4"), ("theta", "f8"), ("alpha", "f8"),
                        ("dualOut", "f8"), ("objective", "f8")])


class Stats(C.Structure):
    _fields_ = [("price_ms", C.c_double), ("price_launches", C.c_long), ("price_bytes", C.c_double),
                ("total_ms", C.c_double), ("iterations", C.c_long), ("refactorizations", C.c_long),
                ("row_ms", C.c_double), ("row_launches", C.c_long), ("row_bytes", C.c_double),
                ("nucleus", C.c_long), ("nucleus_capacity", C.c_long), ("refreshes", C.c_long), ("refreshes_rejected", C.c_long),
                ("lu_active", C.c_long), ("lu_front", C.c_long), ("lu_tail", C.c_long), ("lu_factorizations", C.c_long),
                ("lu_front_ms", C.c_double), ("lu_invert_ms", C.c_double), ("lu_build_ms", C.c_double), ("eta_count", C.c_long),
                ("perturbations", C.c_long), ("backwards_restores", C.c_long), ("loop_flags", C.c_long),
                ("accuracy_restores", C.c_long), ("singular_restores", C.c_long),
                ("price_form", C.c_long), ("dense_pi_launches", C.c_long), ("price_form_switches", C.c_long),
                ("exits_scheduled", C.c_long), ("exits_alpha_check", C.c_long), ("exits_backwards", C.c_long),
                ("exits_bad_update", C.c_long), ("comm_mode", C.c_long), ("shard_cand_cap", C.c_long),
                ("free_first_rows", C.c_long), ("free_entered", C.c_long), ("try_primal_exits", C.c_long),
                ("chuzr_partial_scans", C.c_long), ("chuzr_recalls", C.c_long),
                ("chuzr_ordered_walks", C.c_long), ("dc_wide_timeouts", C.c_long),
                ("eta_compact_slots", C.c_long), ("factor_elements", C.c_long)]


# every symbol include/clpgpu.h declares (tests/test_abi.py checks the library exports all of them)
ABI_SYMBOLS = [
    "clpgpu_create", "clpgpu_destroy", "clpgpu_last_error", "clpgpu_stream", "clpgpu_load_problem",
    "clpgpu_set_column_range", "clpgpu_comm_unique_id", "clpgpu_comm_init", "clpgpu_times", "clpgpu_transpose_times", "clpgpu_price_row", "clpgpu_factorize",
    "clpgpu_ftran", "clpgpu_btran", "clpgpu_replace_column", "clpgpu_pivots", "clpgpu_set_option",
    "clpgpu_set_status", "clpgpu_dual", "clpgpu_dual_steps", "clpgpu_fast_dual", "clpgpu_strong_branching", "clpgpu_problem_status", "clpgpu_number_iterations", "clpgpu_objective_value",
    "clpgpu_get_solution", "clpgpu_get_reduced_costs", "clpgpu_get_status", "clpgpu_get_pivot_variable",
    "clpgpu_get_pivot_log", "clpgpu_get_row_weights", "clpgpu_get_stats",
    "clpgpu_chg_row_lower", "clpgpu_chg_row_upper", "clpgpu_chg_column_lower", "clpgpu_chg_column_upper",
    "clpgpu_chg_obj_coefficients", "clpgpu_scale_factors",
    "clpgpu_clone", "clpgpu_set_scales", "clpgpu_ftran_ft", "clpgpu_ftran_two_ft", "clpgpu_bind_rim", "clpgpu_pivot_row",
    "clpgpu_update_weights", "clpgpu_update_primal", "clpgpu_save_weights", "clpgpu_unroll_weights",
    "clpgpu_get_kernel_times", "clpgpu_dgemm", "clpgpu_test_cycle", "clpgpu_debug_price_bench", "clpgpu_test_looping",
    "clpgpu_test_jds_layout", "clpgpu_test_free_first_row", "clpgpu_test_lu_front",
    "clpgpu_virtual_group_create", "clpgpu_virtual_group_destroy", "clpgpu_virtual_attach", "clpgpu_virtual_dual_steps",
]


def build(force: bool = False) -> str:
    """Compile libclpgpu.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in ("engine.hip", "kernels.hip", "lu_kernels.hip", "lu_host.hip", "lu_front.h", "perturb_host.h", "gemm_kernel.hip",
                                               "device_state.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "clpgpu.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", src_dir, "-s"])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP extension is the product; there is no CPU fallback)")
        # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process do not see the
        # same devices.  Importing torch first makes the loader bind libclpgpu.so to the runtime torch
        # already mapped (same SONAME), so torch.cuda / torch.distributed and the engine share it.
        try:
            import torch  # noqa: F401
        except Exception:  # torch is plumbing, not a dependency of the engine
            pass
        L = C.CDLL(LIB_PATH)
        p = C.c_void_p
        ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        up = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
        L.clpgpu_create.restype = p
        L.clpgpu_create.argtypes = [C.c_int]
        L.clpgpu_destroy.argtypes = [p]
        L.clpgpu_last_error.restype = C.c_char_p
        L.clpgpu_last_error.argtypes = [p]
        L.clpgpu_stream.restype = p
        L.clpgpu_stream.argtypes = [p]
        L.clpgpu_load_problem.argtypes = [p, C.c_int, C.c_int, ip, ip, dp, dp, dp, dp, dp, dp]
        L.clpgpu_set_column_range.argtypes = [p, C.c_int, C.c_int]
        L.clpgpu_comm_unique_id.argtypes = [C.c_char_p]
        L.clpgpu_comm_init.argtypes = [p, C.c_int, C.c_int, C.c_char_p]
        L.clpgpu_times.argtypes = [p, C.c_double, dp, dp]
        L.clpgpu_transpose_times.argtypes = [p, C.c_double, dp, dp]
        L.clpgpu_price_row.argtypes = [p, C.c_int, ip, dp, up, dp, C.c_double, C.c_double, C.c_double,
                                       C.POINTER(C.c_int), ip, dp, C.POINTER(C.c_int), ip, dp, C.POINTER(C.c_double)]
        L.clpgpu_factorize.argtypes = [p, up, ip]
        L.clpgpu_ftran.argtypes = [p, dp]
        L.clpgpu_btran.argtypes = [p, dp]
        L.clpgpu_replace_column.argtypes = [p, C.c_int, C.c_int, C.c_double, C.c_double]
        L.clpgpu_pivots.argtypes = [p]
        L.clpgpu_set_option.argtypes = [p, C.c_char_p, C.c_double]
        L.clpgpu_set_status.argtypes = [p, up]
        L.clpgpu_scale_factors.argtypes = [C.c_int, C.c_int, ip, ip, dp, dp, dp, dp, dp, C.c_int, C.c_double, dp, dp]
        for name in ("row_lower", "row_upper", "column_lower", "column_upper", "obj_coefficients"):
            getattr(L, "clpgpu_chg_" + name).argtypes = [p, dp]
        L.clpgpu_dual.argtypes = [p]
        L.clpgpu_dual_steps.argtypes = [p, C.c_int]
        L.clpgpu_fast_dual.argtypes = [p, C.c_int]
        L.clpgpu_problem_status.argtypes = [p]
        L.clpgpu_strong_branching.argtypes = [p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                              C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int]
        L.clpgpu_number_iterations.argtypes = [p]
        L.clpgpu_objective_value.argtypes = [p]
        L.clpgpu_objective_value.restype = C.c_double
        L.clpgpu_get_solution.argtypes = [p, dp]
        L.clpgpu_get_reduced_costs.argtypes = [p, dp]
        L.clpgpu_get_status.argtypes = [p, up]
        L.clpgpu_get_pivot_variable.argtypes = [p, ip]
        L.clpgpu_get_pivot_log.argtypes = [p, C.c_void_p, C.c_int]
        L.clpgpu_get_row_weights.argtypes = [p, dp, dp]
        L.clpgpu_get_stats.argtypes = [p, C.POINTER(Stats)]
        L.clpgpu_clone.restype = p
        L.clpgpu_clone.argtypes = [p]
        L.clpgpu_set_scales.argtypes = [p, C.c_void_p, C.c_void_p]
        L.clpgpu_ftran_ft.argtypes = [p, dp]
        L.clpgpu_ftran_two_ft.argtypes = [p, dp, dp]
        L.clpgpu_bind_rim.argtypes = [p] + [C.c_void_p] * 6
        L.clpgpu_pivot_row.argtypes = [p]
        L.clpgpu_update_weights.argtypes = [p, C.c_int, ip, dp, C.c_int, C.c_int, C.c_double, dp, C.POINTER(C.c_double)]
        L.clpgpu_update_primal.argtypes = [p, C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_double)]
        L.clpgpu_save_weights.argtypes = [p, C.c_int]
        L.clpgpu_unroll_weights.argtypes = [p]
        L.clpgpu_get_kernel_times.argtypes = [p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_long)]
        _LIB = L
    return _LIB


def scale_factors(lp, mode=3, primal_tolerance=1.0e-7):
    """ClpPackedMatrix::scale factors as the engine computes them (host code, no GPU needed):
    (scaled?, rowScale[m], columnScale[n])."""
    rs, cs = np.empty(lp.m), np.empty(lp.n)
    rc = lib().clpgpu_scale_factors(int(lp.m), int(lp.n), np.ascontiguousarray(lp.col_start, dtype=np.int32),
                                    np.ascontiguousarray(lp.row, dtype=np.int32), np.ascontiguousarray(lp.elem, dtype=np.float64),
                                    np.ascontiguousarray(lp.col_lower, dtype=np.float64), np.ascontiguousarray(lp.col_upper, dtype=np.float64),
                                    np.ascontiguousarray(lp.row_lower, dtype=np.float64), np.ascontiguousarray(lp.row_upper, dtype=np.float64),
                                    int(mode), float(primal_tolerance), rs, cs)
    if rc < 0:
        raise ValueError("clpgpu_scale_factors: bad input")
    return rc == 0, rs, cs


class ClpGpuSimplex:
    """Engine-mode driver: ClpSimplex::dual() on one MI355X (precedent: ClpSimplex::dealWithAbc)."""

    def __init__(self, device: int = 0):
        self._h = lib().clpgpu_create(int(device))
        if not self._h:
            raise RuntimeError("clpgpu_create failed: no usable HIP device (gfx950) -- libclpgpu has no CPU fallback")
        self.m = self.n = 0

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _LIB is not None:
            _LIB.clpgpu_destroy(h)
            self._h = None

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {lib().clpgpu_last_error(self._h).decode()}")

    # ---- ClpModel / ClpSimplex surface --------------------------------------------------------
    def lastError(self):
        return lib().clpgpu_last_error(self._h).decode()

    def loadProblem(self, lp):
        c = np.ascontiguousarray
        self.m, self.n = int(lp.m), int(lp.n)
        self._check(lib().clpgpu_load_problem(
            self._h, self.m, self.n, c(lp.col_start, dtype=np.int32), c(lp.row, dtype=np.int32),
            c(lp.elem, dtype=np.float64), c(lp.col_lower, dtype=np.float64), c(lp.col_upper, dtype=np.float64),
            c(lp.obj, dtype=np.float64), c(lp.row_lower, dtype=np.float64), c(lp.row_upper, dtype=np.float64)),
            "clpgpu_load_problem")
        return self

    def set_option(self, name, value):
        if lib().clpgpu_set_option(self._h, name.encode(), float(value)) != 0:
            raise KeyError(name)

    def setMaximumIterations(self, value):
        self.set_option("max_iterations", value)

    def setDualRowPivotAlgorithm(self, name):
        self.set_option("pivot_rule", {"dantzig": 0, "steepest": 1}[name])

    def setColumnRange(self, first, last):
        self._check(lib().clpgpu_set_column_range(self._h, int(first), int(last)), "clpgpu_set_column_range")

    def setStatusArray(self, status):
        self._check(lib().clpgpu_set_status(self._h, np.ascontiguousarray(status, dtype=np.uint8)), "clpgpu_set_status")

    # ClpModel::chgRowLower ... chgObjCoefficients: new bounds / costs, matrix stays on the device
    def chgRowLower(self, values):
        self._check(lib().clpgpu_chg_row_lower(self._h, np.ascontiguousarray(values, dtype=np.float64)), "clpgpu_chg_row_lower")

    def chgRowUpper(self, values):
        self._check(lib().clpgpu_chg_row_upper(self._h, np.ascontiguousarray(values, dtype=np.float64)), "clpgpu_chg_row_upper")

    def chgColumnLower(self, values):
        self._check(lib().clpgpu_chg_column_lower(self._h, np.ascontiguousarray(values, dtype=np.float64)), "clpgpu_chg_column_lower")

    def chgColumnUpper(self, values):
        self._check(lib().clpgpu_chg_column_upper(self._h, np.ascontiguousarray(values, dtype=np.float64)), "clpgpu_chg_column_upper")

    def chgObjCoefficients(self, values):
        self._check(lib().clpgpu_chg_obj_coefficients(self._h, np.ascontiguousarray(values, dtype=np.float64)), "clpgpu_chg_obj_coefficients")

    def dual(self):
        return lib().clpgpu_dual(self._h)

    def dual_steps(self, iterations):
        return lib().clpgpu_dual_steps(self._h, int(iterations))

    def problemStatus(self):
        return lib().clpgpu_problem_status(self._h)

    def fastDual(self, alwaysFinish=False):
        """ClpSimplexDual::fastDual: 0 finished, 1 stopped"""
        rc = lib().clpgpu_fast_dual(self._h, int(bool(alwaysFinish)))
        if rc < 0:
            self._check(rc, "clpgpu_fast_dual")
        return rc

    def strongBranching(self, variables, newLower, newUpper, stopOnFirstInfeasible=True, alwaysFinish=False, solutions=True):
        """ClpSimplexDual::strongBranching.  Returns (returnCode, downChange, upChange, status[2k], iterations[2k],
        solutions[2k, n] or None); status / iterations / solutions: even = down, odd = up."""
        k = len(variables)
        var = (C.c_int * k)(*[int(v) for v in variables])
        lo = (C.c_double * k)(*[float(v) for v in newLower])
        up = (C.c_double * k)(*[float(v) for v in newUpper])
        st, it = (C.c_int * (2 * k))(), (C.c_int * (2 * k))()
        sol = np.zeros((2 * k, self.n)) if solutions else None
        ptrs = None
        if solutions:
            ptrs = (C.POINTER(C.c_double) * (2 * k))(*[sol[i].ctypes.data_as(C.POINTER(C.c_double)) for i in range(2 * k)])
        rc = lib().clpgpu_strong_branching(self._h, k, var, lo, up, ptrs, st, it, int(bool(stopOnFirstInfeasible)), int(bool(alwaysFinish)))
        if rc in (-2, -99):
            self._check(rc, "clpgpu_strong_branching")
        return rc, np.array(up[:]), np.array(lo[:]), np.array(st[:]), np.array(it[:]), sol

    def numberIterations(self):
        return lib().clpgpu_number_iterations(self._h)

    def objectiveValue(self):
        return lib().clpgpu_objective_value(self._h)

    def _vec(self, fn, dtype=np.float64, size=None):
        out = np.zeros(size or (self.m + self.n), dtype=dtype)
        self._check(getattr(lib(), fn)(self._h, out), fn)
        return out

    def solution(self):
        return self._vec("clpgpu_get_solution")

    def primalColumnSolution(self):
        return self.solution()[: self.n]

    def primalRowSolution(self):
        return self.solution()[self.n:]

    def reducedCosts(self):
        return self._vec("clpgpu_get_reduced_costs")

    def dualColumnSolution(self):
        return self.reducedCosts()[: self.n]

    def dualRowSolution(self):
        return self.reducedCosts()[self.n:]

    def statusArray(self):
        return self._vec("clpgpu_get_status", np.uint8)

    def pivotVariable(self):
        return self._vec("clpgpu_get_pivot_variable", np.int32, self.m)

    def pivotLog(self):
        total = lib().clpgpu_get_pivot_log(self._h, None, 0)
        out = np.zeros(max(total, 0), dtype=PIVOT_DTYPE)
        if total > 0:
            lib().clpgpu_get_pivot_log(self._h, out.ctypes.data_as(C.c_void_p), total)
        return out

    def rowWeights(self):
        w, inf = np.zeros(self.m), np.zeros(self.m)
        self._check(lib().clpgpu_get_row_weights(self._h, w, inf), "clpgpu_get_row_weights")
        return w, inf

    def stats(self):
        s = Stats()
        self._check(lib().clpgpu_get_stats(self._h, C.byref(s)), "clpgpu_get_stats")
        return {f: getattr(s, f) for f, _ in Stats._fields_}

    def debugPriceBench(self, masks, reps=50):
        """mean microseconds per launch of the by-column pricing kernel under each debug mask (clpgpu_debug_price_bench)"""
        mk = np.ascontiguousarray(masks, dtype=np.int32)
        out = np.zeros(len(mk))
        f = lib().clpgpu_debug_price_bench
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS"),
                      np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")]
        self._check(f(self._h, int(reps), len(mk), mk, out), "clpgpu_debug_price_bench")
        return out

    def kernelTimes(self):
        """{kernel: (total ms, launches)} gathered with option timing = 2"""
        cap = 64
        names, ms, cnt = (C.c_char_p * cap)(), (C.c_double * cap)(), (C.c_long * cap)()
        k = lib().clpgpu_get_kernel_times(self._h, cap, names, ms, cnt)
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(max(0, min(k, cap)))}

    def stream(self):
        return lib().clpgpu_stream(self._h)

    # ---- plug-in level calls (ClpMatrixBase / CoinOtherFactorization surfaces) -------------------
    def times(self, scalar, x, y):
        y = np.array(y, dtype=np.float64)
        self._check(lib().clpgpu_times(self._h, float(scalar), np.ascontiguousarray(x, dtype=np.float64), y), "clpgpu_times")
        return y

    def transposeTimes(self, scalar, x, y):
        y = np.array(y, dtype=np.float64)
        self._check(lib().clpgpu_transpose_times(self._h, float(scalar), np.ascontiguousarray(x, dtype=np.float64), y),
                    "clpgpu_transpose_times")
        return y

    def priceRow(self, pi_index, pi_value, status, dj, zero_tol=1e-13, dual_tol=1e-7, acceptable_pivot=1e-9):
        n, m = self.n, self.m
        out_i = np.zeros(n, np.int32)
        out_v = np.zeros(n)
        cand_i = np.zeros(n + m, np.int32)
        cand_v = np.zeros(n + m)
        nout, ncand, upper = C.c_int(0), C.c_int(0), C.c_double(0.0)
        pi_index = np.ascontiguousarray(pi_index, dtype=np.int32)
        self._check(lib().clpgpu_price_row(self._h, len(pi_index), pi_index, np.ascontiguousarray(pi_value, dtype=np.float64),
                                           np.ascontiguousarray(status, dtype=np.uint8),
                                           np.ascontiguousarray(dj, dtype=np.float64), zero_tol, dual_tol, acceptable_pivot,
                                           C.byref(nout), out_i, out_v, C.byref(ncand), cand_i, cand_v, C.byref(upper)),
                    "clpgpu_price_row")
        return (out_i[:nout.value].copy(), out_v[:nout.value].copy(), cand_i[:ncand.value].copy(),
                cand_v[:ncand.value].copy(), upper.value)

    def factorize(self, status):
        pv = np.zeros(self.m, np.int32)
        rc = lib().clpgpu_factorize(self._h, np.ascontiguousarray(status, dtype=np.uint8), pv)
        return rc, pv

    def ftran(self, v):
        v = np.array(v, dtype=np.float64)
        self._check(lib().clpgpu_ftran(self._h, v), "clpgpu_ftran")
        return v

    def btran(self, v):
        v = np.array(v, dtype=np.float64)
        self._check(lib().clpgpu_btran(self._h, v), "clpgpu_btran")
        return v

    def replaceColumn(self, pivot_row, sequence_in):
        return lib().clpgpu_replace_column(self._h, int(pivot_row), int(sequence_in), 0.0, 1e-8)

    def testCycle(self, seq_in, seq_out, way_in, way_out):
        """the device cycle detector over a sequence of pivots (see clpgpu_test_cycle)"""
        arr = [np.ascontiguousarray(a, dtype=np.int32) for a in (seq_in, seq_out, way_in, way_out)]
        out = np.zeros(len(arr[0]), np.int32)
        ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        f = lib().clpgpu_test_cycle
        f.argtypes = [C.c_void_p, C.c_int, ip, ip, ip, ip, ip]
        self._check(f(self._h, len(out), *arr, out), "clpgpu_test_cycle")
        return out

    def dgemm(self, alpha, a, b, beta, c):
        """c = beta c + alpha a b on the engine's own MFMA f64 GEMM (square row-major arrays)"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        c = np.array(c, dtype=np.float64, order="C")
        dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        f = lib().clpgpu_dgemm
        f.argtypes = [C.c_void_p, C.c_int, C.c_double, dp, dp, C.c_double, dp]
        self._check(f(self._h, int(a.shape[0]), float(alpha), a, b, float(beta), c), "clpgpu_dgemm")
        return c

    def pivots(self):
        return lib().clpgpu_pivots(self._h)

    def ftranFT(self, v):
        v = np.array(v, dtype=np.float64)
        rc = lib().clpgpu_ftran_ft(self._h, v)
        if rc < 0:
            self._check(rc, "clpgpu_ftran_ft")
        return v

    def ftranTwoFT(self, v_ft, v2):
        a, b = np.array(v_ft, dtype=np.float64), np.array(v2, dtype=np.float64)
        rc = lib().clpgpu_ftran_two_ft(self._h, a, b)
        if rc < 0:
            self._check(rc, "clpgpu_ftran_two_ft")
        return a, b

    # ---- ClpDualRowPivot surface ----------------------------------------------------------------
    @staticmethod
    def _opt(a, dtype):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=dtype)
        return a, a.ctypes.data_as(C.c_void_p)

    def bindRim(self, cost=None, lower=None, upper=None, dj=None, solution=None, status=None):
        keep, ptrs = [], []
        for a, dt in ((cost, np.float64), (lower, np.float64), (upper, np.float64), (dj, np.float64),
                      (solution, np.float64), (status, np.uint8)):
            arr, ptr = self._opt(a, dt)
            keep.append(arr)
            ptrs.append(ptr)
        self._check(lib().clpgpu_bind_rim(self._h, *ptrs), "clpgpu_bind_rim")

    def pivotRow(self):
        return lib().clpgpu_pivot_row(self._h)

    def updateWeights(self, pi_index, pi_value, pivot_row, sequence_in, model_alpha):
        w, alpha = np.zeros(self.m), C.c_double(0.0)
        pi_index = np.ascontiguousarray(pi_index, dtype=np.int32)
        self._check(lib().clpgpu_update_weights(self._h, len(pi_index), pi_index, np.ascontiguousarray(pi_value, dtype=np.float64),
                                                int(pivot_row), int(sequence_in), float(model_alpha), w, C.byref(alpha)),
                    "clpgpu_update_weights")
        return alpha.value, w

    def updatePrimalSolution(self, pivot_row, theta, updated_column=None):
        change = C.c_double(0.0)
        arr, ptr = self._opt(updated_column, np.float64)
        self._check(lib().clpgpu_update_primal(self._h, ptr, int(pivot_row), float(theta), C.byref(change)), "clpgpu_update_primal")
        return change.value

    def saveWeights(self, mode):
        self._check(lib().clpgpu_save_weights(self._h, int(mode)), "clpgpu_save_weights")

    def unrollWeights(self):
        self._check(lib().clpgpu_unroll_weights(self._h), "clpgpu_unroll_weights")

    def clone(self):
        other = object.__new__(ClpGpuSimplex)
        other._h = lib().clpgpu_clone(self._h)
        if not other._h:
            raise RuntimeError("clpgpu_clone failed")
        other.m, other.n = self.m, self.n
        return other

    def setScales(self, row_scale, column_scale):
        ra, rp = self._opt(row_scale, np.float64)
        ca, cp = self._opt(column_scale, np.float64)
        self._check(lib().clpgpu_set_scales(self._h, rp, cp), "clpgpu_set_scales")


def test_looping(objective, infeasibility, count, iteration, flag_bits, newest):
    """The engine's host restatement of ClpSimplexProgress::looping over a sequence of status checks (clpgpu_test_looping);
    host code only, runs without a GPU.  Returns (code, dualTolerance, dualBound, forceFactorization, flagged) arrays."""
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    n = len(objective)
    d = [np.ascontiguousarray(a, dtype=np.float64) for a in (objective, infeasibility)]
    i = [np.ascontiguousarray(a, dtype=np.int32) for a in (count, iteration, flag_bits, newest)]
    code, force, flagged = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    tol, bound = np.zeros(n), np.zeros(n)
    f = lib().clpgpu_test_looping
    f.argtypes = [C.c_int, dp, dp, ip, ip, ip, ip, ip, dp, dp, ip, ip]
    f.restype = C.c_int
    if f(n, d[0], d[1], i[0], i[1], i[2], i[3], code, tol, bound, force, flagged) != 0:
        raise RuntimeError("clpgpu_test_looping failed")
    return code, tol, bound, force, flagged


def lu_front(matrix, stop_density=0.03, min_tail=0, threshold=0.1):
    """The engine's host Markowitz LU of a square sparse matrix (scipy sparse or dense) up to its dense tail
    (clpgpu_test_lu_front; host code only, runs without a GPU).  Returns a dict: pivots (frow, fcol, fpiv), L and U by
    pivot (CSR-like starts), the tail's rows / columns and entries, fill."""
    import scipy.sparse as sp

    M = sp.csc_matrix(matrix)
    M.sort_indices()
    k = M.shape[0]
    assert M.shape == (k, k)
    cs = np.ascontiguousarray(M.indptr, dtype=np.int32)
    cr = np.ascontiguousarray(M.indices, dtype=np.int32)
    cv = np.ascontiguousarray(M.data, dtype=np.float64)
    f = lib().clpgpu_test_lu_front
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    lp_ = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
    f.argtypes = [C.c_int, ip, ip, dp, C.c_double, C.c_int, C.c_double, lp_, C.c_int] + [C.c_void_p] * 14
    f.restype = C.c_int
    counts = np.zeros(6, np.int64)
    if f(k, cs, cr, cv, float(stop_density), int(min_tail), float(threshold), counts, 0, *([None] * 14)) != 0:
        raise ValueError("clpgpu_test_lu_front: bad input")
    nF, nL, nU, k2, nS, fill = (int(x) for x in counts)
    i32, f64 = (lambda n: np.zeros(max(n, 1), np.int32)), (lambda n: np.zeros(max(n, 1), np.float64))
    frow, fcol, fpiv = i32(nF), i32(nF), f64(nF)
    lStart, lRow, lVal = i32(nF + 1), i32(nL), f64(nL)
    uStart, uCol, uVal = i32(nF + 1), i32(nU), f64(nU)
    tailRow, tailCol = i32(k2), i32(k2)
    sRow, sCol, sVal = i32(nS), i32(nS), f64(nS)
    arrs = [frow, fcol, fpiv, lStart, lRow, lVal, uStart, uCol, uVal, tailRow, tailCol, sRow, sCol, sVal]
    if f(k, cs, cr, cv, float(stop_density), int(min_tail), float(threshold), counts, 1, *[a.ctypes.data for a in arrs]) != 0:
        raise ValueError("clpgpu_test_lu_front: bad input")
    return {"k": k, "pivots": nF, "tail": k2, "fill": fill, "frow": frow[:nF], "fcol": fcol[:nF], "fpiv": fpiv[:nF], "lStart": lStart[: nF + 1],
            "lRow": lRow[:nL], "lVal": lVal[:nL], "uStart": uStart[: nF + 1], "uCol": uCol[:nU], "uVal": uVal[:nU], "tailRow": tailRow[:k2],
            "tailCol": tailCol[:k2], "sRow": sRow[:nS], "sCol": sCol[:nS], "sVal": sVal[:nS]}


def free_first_row(work, pivot_variable, solution, lower, upper, status):
    """The row a free column should pivot on, as the engine's host code chooses it for dualRow's free-first entry
    (clpgpu_test_free_first_row; host code only, runs without a GPU).  -1: none."""
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    bp = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
    f = lib().clpgpu_test_free_first_row
    f.argtypes = [C.c_int, C.c_int, dp, ip, dp, dp, dp, bp]
    f.restype = C.c_int
    work = np.ascontiguousarray(work, dtype=np.float64)
    rc = f(len(work), len(solution), work, np.ascontiguousarray(pivot_variable, dtype=np.int32), np.ascontiguousarray(solution, dtype=np.float64),
           np.ascontiguousarray(lower, dtype=np.float64), np.ascontiguousarray(upper, dtype=np.float64), np.ascontiguousarray(status, dtype=np.uint8))
    if rc == -99:
        raise ValueError("clpgpu_test_free_first_row: bad input")
    return rc


def jds_layout(lp, order):
    """The jagged row-tiled pricing layout k_price_lds reads (clpgpu_test_jds_layout), built by the engine's host code without a
    GPU.  `order` = column keys in home order, 64 per slice (-1 none).  Returns a dict of the arrays, or None where the layout
    refuses the matrix."""
    ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    bp = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
    up = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
    order = np.ascontiguousarray(order, dtype=np.int32)
    assert order.size % 64 == 0 and order.size
    slices = order.size // 64
    cs = np.ascontiguousarray(lp.col_start, dtype=np.int32)
    row = np.ascontiguousarray(lp.row, dtype=np.int32)
    elem = np.ascontiguousarray(lp.elem, dtype=np.float64)
    f = lib().clpgpu_test_jds_layout
    f.argtypes = [C.c_int, C.c_int, ip, ip, dp, C.c_int, ip, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.c_longlong,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f.restype = C.c_int
    tiles, tile_rows, records = C.c_int(0), C.c_int(0), C.c_longlong(0)
    rc = f(int(lp.m), int(lp.n), cs, row, elem, slices, order, C.byref(tiles), C.byref(tile_rows), C.byref(records), 0, None, None, None, None, None, None)
    if rc == 1:
        return None
    if rc != 0:
        raise RuntimeError(f"clpgpu_test_jds_layout failed ({rc})")
    T, R = tiles.value, records.value
    seg = np.zeros(slices, np.int32)
    cnt, src = np.zeros(slices * T * 64, np.uint8), np.zeros(slices * T * 64, np.uint8)
    home = np.zeros(slices * 64, np.uint8)
    rp, ep = np.zeros(max(R, 1), np.uint32), np.zeros(2 * max(R, 1), np.float64)
    rc = f(int(lp.m), int(lp.n), cs, row, elem, slices, order, C.byref(tiles), C.byref(tile_rows), C.byref(records), max(R, 1),
           seg.ctypes.data, cnt.ctypes.data, src.ctypes.data, home.ctypes.data, rp.ctypes.data, ep.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"clpgpu_test_jds_layout failed ({rc})")
    return {"tiles": T, "tile_rows": tile_rows.value, "records": R, "seg_start": seg, "cnt": cnt.reshape(slices, T, 64),
            "src": src.reshape(slices, T, 64), "home": home.reshape(slices, 64), "row_pair": rp[:R], "elem_pair": ep[:2 * R].reshape(R, 2)}


class VirtualRanks:
    """N loopback ranks on one GPU (include/clpgpu.h, clpgpu_virtual_*): the column-sharded engine with real rank
    offsets, the exchanges done by device-to-device copies.  `engines[r]` is rank r's ClpGpuSimplex."""

    def __init__(self, lp, nranks, configure=None, device=0, preconfigure=None):
        """configure(e) runs on every rank's engine after its LP is loaded, preconfigure(e) before (layout options)"""
        L = lib()
        L.clpgpu_virtual_group_create.restype = C.c_void_p
        L.clpgpu_virtual_group_create.argtypes = [C.c_int]
        L.clpgpu_virtual_group_destroy.argtypes = [C.c_void_p]
        L.clpgpu_virtual_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.clpgpu_virtual_dual_steps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        self._g = L.clpgpu_virtual_group_create(nranks)
        if not self._g:
            raise RuntimeError("clpgpu_virtual_group_create failed")
        self.engines = []
        for r in range(nranks):
            e = ClpGpuSimplex(device)
            if preconfigure:
                preconfigure(e)
            e.loadProblem(lp)
            if configure:
                configure(e)
            rc = L.clpgpu_virtual_attach(e._h, self._g, r)
            if rc:
                raise RuntimeError(f"clpgpu_virtual_attach failed ({rc}): {e.lastError()}")
            self.engines.append(e)

    def dual_steps(self, iterations=-1):
        st = (C.c_int * len(self.engines))()
        rc = lib().clpgpu_virtual_dual_steps(self._g, int(iterations), st)
        if rc:
            raise RuntimeError(f"clpgpu_virtual_dual_steps failed ({rc}): the ranks lost step")
        return list(st)

    def __del__(self):
        self.engines = []
        if getattr(self, "_g", None) and lib is not None:
            try:
                lib().clpgpu_virtual_group_destroy(self._g)
            except Exception:  # noqa: BLE001 -- interpreter shutdown
                pass
            self._g = None

/*
 * clpgpu.h -- C ABI of libclpgpu.so, the MI355X (gfx950) dual-simplex iteration engine.
 *
 * Drop-in boundary for the per-iteration hot path of coin-or/Clp's revised dual simplex
 * (SURVEY.md section 8b).  Plain C: POD pointers and sizes only, no C++/torch types, no exceptions,
 * every call synchronous at return, `int` status codes mirroring the reference's.  One context per
 * ClpSimplex; contexts are independent.  Host arrays are the caller's; the engine keeps device copies.
 *
 * Each entry point cites the reference interface it stands in for (file:line under the reference
 * tree).  INTEGRATION.md shows the C++ adapter classes (ClpPackedMatrix / ClpDualRowPivot /
 * CoinOtherFactorization subclasses and the dealWithAbc-style engine swap) a Clp maintainer would
 * add to bind them.
 *
 * Conventions (identical to ClpSimplex): sequences [0,n) = structural columns, [n,n+m) = row
 * slacks, slack column = -e_i (src/ClpSimplex.cpp:3442-3474); rim arrays are [columns | rows]
 * (src/ClpSimplex.hpp:1864-1922); status byte = ClpSimplex::Status in the low 3 bits
 * (src/ClpSimplex.hpp:119-133), bits 3-4 fake-bound flags, bit 6 "flagged".
 */
#ifndef CLPGPU_H
#define CLPGPU_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct clpgpu_context clpgpu_context;

/* per-iteration record, the shape of CLP_SIMPLEX_HOUSE2 (src/ClpMessage.cpp:48,
 * emitted src/ClpSimplex.cpp:2227-2236) */
typedef struct {
  int iteration;
  int sequenceIn;
  int sequenceOut;
  int pivotRow;
  int numberFlipped;
  int reserved;
  double theta;
  double alpha;
  double dualOut;
  double objective;
} clpgpu_pivot_record;

/* per-kernel-class device timings gathered with hipEvents on the engine's stream */
typedef struct {
  double price_ms;      /* sum over launches of the row-pricing kernel */
  long price_launches;
  double price_bytes;   /* algorithmic bytes moved by those launches (SURVEY 8d formula) */
  double total_ms;      /* whole clpgpu_dual wall time */
  long iterations;
  long refactorizations;
  double row_ms;        /* the same three for pricing launches that went BY ROW (sparse pi) */
  long row_launches;
  double row_bytes;     /* B_row of SURVEY 8d: 12 B per visited row entry and pi nonzero, 8 per touched column, 20 per emitted one */
  long nucleus;         /* k: basic structurals = order of the nucleus inverse right now */
  long nucleus_capacity; /* rows allocated for it (3 k x k f64 matrices) */
  long refreshes;        /* scheduled refactorizations at which the inverse was kept (verified refresh, option refresh_min_k) */
  long refreshes_rejected; /* ... and those where the residual check sent it to a re-inversion after all */
  /* LU factorization mode (option factor_mode): sparse Markowitz front + dense tail + product-form eta file */
  long lu_active;          /* 1 when the factorization now on the device is the LU form */
  long lu_front;           /* pivots of the sparse front */
  long lu_tail;            /* order of the dense tail (inverted on the matrix cores) */
  long lu_factorizations;  /* LU factorizations so far */
  double lu_front_ms;      /* host time spent in the Markowitz front, all factorizations */
  double lu_invert_ms;     /* time of the tail inversions (device, measured on the host clock) */
  double lu_build_ms;      /* level schedules + uploads */
  long eta_count;          /* basis updates since the last factorization (length of the eta file in LU mode) */
  long perturbations;      /* times ClpSimplexDual::perturb changed the costs in the last solve (option perturbation) */
  long backwards_restores; /* times statusOfProblemInDual went back to the last good basis because the objective fell (:5395-5476) */
  long loop_flags;         /* times ClpSimplexProgress::looping found a repeat over status checks and acted (ClpSolve.cpp:4553) */
  long accuracy_restores;  /* times errors beyond 1e15 sent statusOfProblemInDual back to the last good basis (:5237-5318) */
  long singular_restores;  /* times a singular refactorization did (:5060-5125) */
  /* row pricing form of the iteration chain (option price_lds) */
  long price_form;          /* 1: the captured chain now prices with pi tiles in LDS (dense pi), 0: k_price_sell + the by-row form */
  long dense_pi_launches;   /* pricing launches whose pi was dense (12 nnz >= m) */
  long price_form_switches; /* times the host changed the chain's pricing form */
  /* what ended the batches of whileIterating that led to a status check (src/ClpSimplexDual.cpp:1849 / :1451 / :1574 / :1618) */
  long exits_scheduled;     /* housekeeping asked for the refactorization (eta-file length, forced factorization) */
  long exits_alpha_check;   /* btran / ftran alpha disagreed */
  long exits_backwards;     /* objective going backwards */
  long exits_bad_update;    /* the basis update reported a singular pivot */
  /* column-sharded runs */
  long comm_mode;           /* 0 not sharded, 2 candidate / flip lists exchanged, 1 dense row slices (the fall-back) */
  long shard_cand_cap;      /* candidates per rank the exchange buffer holds now (grown once on overflow) */
  long free_first_rows;     /* option free_nonbasic: pivots whose row came from dualRow's free-first entry (src/ClpSimplexDual.cpp:3005-3055) */
  long free_entered;        /* ... pivots that brought a free / superbasic variable in through dualColumn0's general branch (:4058-4179) */
  long try_primal_exits;    /* 1 if the solve ended in gutsOfDual's "problems - try primal" exit (status 10, src/ClpSimplexDual.cpp:540-547) */
  /* ClpDualRowSteepest::pivotRow (option steepest_mode, src/ClpDualRowSteepest.cpp:258-346) */
  long chuzr_partial_scans; /* calls that looked at numberWanted entries of the infeasibility list instead of all of them (:258-278, :329-335) */
  long chuzr_recalls;       /* second calls with largestDualError 0 after a changed tolerance found no row (:338-346) */
  long chuzr_ordered_walks; /* partial scans that walked the list in order in one workgroup (a flagged candidate or the last pivot row in the scanned part) */
  long dc_wide_timeouts;    /* times k_dual_column_wide's grid barrier gave up (the context then keeps long lists in one workgroup, option dc_wide 0) */
  long eta_compact_slots;   /* LU mode, option lu_compact_eta: positions the chain's FTRAN reads the eta file over (structurals of the refactorization + positions whose slack left since); 0 when the full file is read */
  long factor_elements;     /* what stands for factorization()->numberElements() since the last factorization (option steepest_elements) */
} clpgpu_stats;

/* ---- lifetime ------------------------------------------------------------------------- */
/* device = HIP device ordinal.  Returns NULL when no gfx950 device / HIP runtime is usable:
 * there is no CPU fallback. */
clpgpu_context *clpgpu_create(int device);
void clpgpu_destroy(clpgpu_context *ctx);
/* ClpMatrixBase::clone (src/ClpMatrixBase.hpp:96), ClpDualRowPivot::clone (src/ClpDualRowPivot.hpp:76),
 * CoinOtherFactorization::clone: Clp copies models and their plug-ins freely (ClpModel.cpp:822, :885,
 * ClpSimplex.cpp:2851).  An independent context on the same device with the same problem, options,
 * changed bounds/costs and warm-start status; the factorization is rebuilt by the clone's first
 * factorize / dual.  NULL on failure. */
clpgpu_context *clpgpu_clone(const clpgpu_context *ctx);
const char *clpgpu_last_error(const clpgpu_context *ctx);
/* HIP stream (hipStream_t as void*) the engine launches on; for event timing by callers */
void *clpgpu_stream(clpgpu_context *ctx);

/* ---- ClpMatrixBase / ClpPackedMatrix surface (src/ClpMatrixBase.hpp:38-546) ------------ */
/* ClpSimplex::loadProblem(const ClpMatrixBase&, collb, colub, obj, rowlb, rowub)
 * (src/ClpSimplex.hpp:237): CSC A as CoinPackedMatrix stores it + original bounds/costs. */
int clpgpu_load_problem(clpgpu_context *ctx, int numberRows, int numberColumns, const int *columnStart,
                        const int *rowIndex, const double *element, const double *columnLower,
                        const double *columnUpper, const double *objective, const double *rowLower,
                        const double *rowUpper);
/* column-range shard for multi-GPU pricing (ABOCA_LITE chunking,
 * src/ClpPackedMatrix.cpp:1823-1854): this context prices only columns [first, last). */
int clpgpu_set_column_range(clpgpu_context *ctx, int firstColumn, int lastColumn);

/* Multi-GPU engine mode: one process per GPU; rank 0 obtains an id with clpgpu_comm_unique_id
 * (128 bytes, ncclUniqueId), every rank calls clpgpu_comm_init with the same id.  The context then
 * prices columns [rank*chunk, (rank+1)*chunk) only and exchanges its slice of the tableau row with an
 * RCCL all-gather each pivot (the reduce of src/ClpPackedMatrix.cpp:1848-1854 / the per-block
 * combine of src/AbcSimplexDual.cpp:1623-1634, done as a gather because everything downstream of
 * the row is replicated).  RCCL is resolved with dlopen, so single-GPU use has no dependency on it. */
int clpgpu_comm_unique_id(void *id128);
int clpgpu_comm_init(clpgpu_context *ctx, int rank, int nranks, const void *id128);

/* ClpModel::rowScale_ / columnScale_ supplied by the caller (what ClpPackedMatrix::scale left in the
 * model, src/ClpPackedMatrix.cpp:4120, src/ClpSimplex.cpp:3701) instead of option "scaling": the engine
 * keeps the LP in those units, getters return unscaled values.  Call after clpgpu_load_problem (the
 * sizes come from it; the device copy is rebuilt).  NULL, NULL drops them.  0 OK, -1 bad factors,
 * -2 nothing loaded. */
int clpgpu_set_scales(clpgpu_context *ctx, const double *rowScale, const double *columnScale);

/* The plug-in level calls below (times ... replace_column, ftran_ft, update_weights) work in the
 * caller's units and return -3 on a context that scales internally. */
/* ClpMatrixBase::times(scalar, x, y) (:275; ClpPackedMatrix.cpp:296): y += scalar*A*x */
int clpgpu_times(clpgpu_context *ctx, double scalar, const double *x, double *y);
/* ClpMatrixBase::transposeTimes(scalar, x, y) (:287; ClpPackedMatrix.cpp:362): y += scalar*A^T*x */
int clpgpu_transpose_times(clpgpu_context *ctx, double scalar, const double *x, double *y);

/* ClpMatrixBase::transposeTimes(model, scalar=-1, x, y, z) (:308; ClpPackedMatrix.cpp:706) on the
 * by-column path with the fused first ratio-test pass (ClpPackedMatrix.cpp:1799-1993, requested
 * through spareIntArray_[0]==1, ClpSimplexDual.cpp:1290-1308).
 * in : packed pi (numberPi, piIndex, piValue), status[n+m], dj[n+m] (the current rim state)
 * out: tableau row column part (outIndex ascending, outValue; returns count via *numberOut),
 *      candidate list (candIndex = sequence numbers, rows first then columns ascending; candValue =
 *      alpha with sign), *upperTheta (spareDoubleArray_[0]). */
int clpgpu_price_row(clpgpu_context *ctx, int numberPi, const int *piIndex, const double *piValue,
                     const unsigned char *status, const double *dj, double zeroTolerance,
                     double dualTolerance, double acceptablePivot, int *numberOut, int *outIndex,
                     double *outValue, int *numberCandidates, int *candIndex, double *candValue,
                     double *upperTheta);

/* ---- ClpFactorization / CoinOtherFactorization surface (src/ClpFactorization.hpp:34-551) - */
/* ClpFactorization::factorize (src/ClpFactorization.cpp:1649): status[n+m] gives the basic set.
 * Fills pivotVariable[m].  Returns 0 OK, -1 singular, -2 wrong number of basics, -99 memory
 * (src/ClpFactorization.hpp:53). */
int clpgpu_factorize(clpgpu_context *ctx, const unsigned char *status, int *pivotVariable);
/* updateColumn (:2803) FTRAN, dense region of length m, in place (row space -> basis positions) */
int clpgpu_ftran(clpgpu_context *ctx, double *region);
/* updateColumnTranspose (:2993) BTRAN, dense length m in place (positions -> row space) */
int clpgpu_btran(clpgpu_context *ctx, double *region);
/* replaceColumn (:2584): basis position pivotRow leaves, `sequenceIn` enters.  The engine reruns
 * the two solves it needs on the device.  Returns 0 OK, 2 singular, 3 no room, 5 max pivots
 * (src/ClpFactorization.hpp:83). */
int clpgpu_replace_column(clpgpu_context *ctx, int pivotRow, int sequenceIn, double pivotCheck,
                          double acceptablePivot);
int clpgpu_pivots(const clpgpu_context *ctx);
/* ClpFactorization::updateColumnFT (src/ClpFactorization.cpp:2723; in-tree twin
 * CoinAbcBaseFactorization3.cpp:2173): FTRAN of the entering column that also keeps what the basis
 * update needs.  Here the result itself stays on the device as the "updated column" later calls refer
 * to.  Returns the number of nonzeros (the reference: negative = no room, never here), -99 on error. */
int clpgpu_ftran_ft(clpgpu_context *ctx, double *region);
/* ClpFactorization::updateTwoColumnsFT (:2889; twin CoinAbcBaseFactorization3.cpp:1155): the FT solve of
 * the entering column and the plain FTRAN of the DSE vector in one sweep.  Returns nonzeros of regionFT. */
int clpgpu_ftran_two_ft(clpgpu_context *ctx, double *regionFT, double *region2);

/* ---- ClpDualRowPivot surface (src/ClpDualRowPivot.hpp:30-76) ------------------------------ */
/* Clp owns the rim arrays; the ones given (NULL = unchanged) are copied to the engine's mirrors,
 * n+m entries each, [columns | rows] (src/ClpSimplex.hpp:1864-1922). */
int clpgpu_bind_rim(clpgpu_context *ctx, const double *cost, const double *lower, const double *upper,
                    const double *dj, const double *solution, const unsigned char *status);
/* pivotRow() (:30; ClpDualRowSteepest.cpp:179, ClpDualRowDantzig.cpp:56): leaving row, or -1 when
 * nothing is primal infeasible (the reference's "looks optimal" answer). */
int clpgpu_pivot_row(clpgpu_context *ctx);
/* updateWeights(input, spare, spare2, updatedColumn) (:33; ClpDualRowSteepest.cpp:375): input = packed
 * pi; does the FT FTRAN of column sequenceIn and the FTRAN of pi, updates the DSE weights with
 * modelAlpha = the ratio test's alpha (model_->alpha(), :509); updatedColumn[m] (by basis position)
 * and *alpha (its pivotRow entry, the value the reference returns) are outputs. */
int clpgpu_update_weights(clpgpu_context *ctx, int numberPi, const int *piIndex, const double *piValue,
                          int pivotRow, int sequenceIn, double modelAlpha, double *updatedColumn,
                          double *alpha);
/* updatePrimalSolution(input, theta, changeInObjective) (:40; ClpDualRowSteepest.cpp:630): basic
 * solution -= theta * updatedColumn (NULL = the column the last clpgpu_update_weights /
 * clpgpu_ftran_ft produced), infeasibility list refreshed; read the result with clpgpu_get_solution. */
int clpgpu_update_primal(clpgpu_context *ctx, const double *updatedColumn, int pivotRow, double theta,
                         double *changeInObjective);
/* saveWeights(model, mode) (:53; ClpDualRowSteepest.cpp:773): 1 before a factorization, 2 / 4 after,
 * 3 redo the infeasibilities only, 5 / 7 strong-branching initialisation, 6 scale back. */
int clpgpu_save_weights(clpgpu_context *ctx, int mode);
/* unrollWeights() (:60; ClpDualRowSteepest.cpp:1022): undo the last clpgpu_update_weights */
int clpgpu_unroll_weights(clpgpu_context *ctx);

/* ---- whole-engine mode (precedent: ClpSimplex::dealWithAbc, src/ClpSolve.cpp:555-833) ---- */
/* options by name (ClpSimplex setters): "pivot_rule" 0 Dantzig (ClpDualRowDantzig) / 1 steepest
 * (ClpDualRowSteepest); "max_iterations" (setMaximumIterations); "max_pivots"
 * (factorization maximumPivots); "dual_bound"; "primal_tolerance"; "dual_tolerance"; "zero_tolerance";
 * "acceptable_pivot";
 * "perturbation" (ClpSimplex::setPerturbation, the value dual() is entered with: 102 = never, the default of a bare
 * context because a perturbed solve can end as status 10 "needs primal clean-up" and the engine holds no primal;
 * 100 = the ClpSimplex constructor's value: nothing at start-up, the kick after 2(m+n) iterations,
 * src/ClpSimplexDual.cpp:488; 50 = the clp command's: perturb at start-up when at most a quarter of the |costs| are
 * distinct; 51-69 fixed maximum fractions; < 50 = 10^value; ClpSimplexDual::perturb :6533; the clpGpuDual adapter
 * passes the model's value and runs primal on status 10 as ClpSimplex::dual does);
 * "random_seed"; "log_level"; "check_every" (host polls the device control block every N
 * iterations).  Engine tuning / test knobs (no counterpart in the reference): "timing" (HIP events
 * around every pricing launch), "price_kernel" (inner-loop variant of the pricing kernel, default
 * 6), "use_graph", "blocked_refactor", "register_panel" (re-inversion variants), "fork_update" (basis
 * update on a second stream),
 * "flip_list_cap" (size of the bound-flip append buffer; small values force its overflow path),
 * "flip_scatter" (1 default: the waves that detect a bound flip scatter its column into per-row slots,
 * sparse LPs whose heaviest row holds <= n/256 entries; 2: any sparse LP; 0: the flip right-hand side is
 * assembled by one workgroup from the flip records), "flip_slot_cap" (contributions a row keeps
 * individually, <= 16; small values force the many-contributors path),
 * "scaling" (0 off; 1/2/3/4 = ClpModel::scaling modes; must be set BEFORE clpgpu_load_problem: the
 * device then holds the scaled LP, getters return unscaled values),
 * "row_price_frac" (row pricing goes by row when nnz(pi) <= frac * m, ClpPackedMatrix.cpp:727-754; 0 = always
 * by column; also selects the form of clpgpu_price_row), "refactor_mode" (-1 auto / 1 one-level / 2
 * two-level vector / 3 two-level MFMA re-inversion), "refactor_min_k" (auto: two-level MFMA from this
 * many basic structurals on, default 1024), "refresh_min_k" / "refresh_max" / "refresh_tolerance" /
 * "refresh_refine" (verified refresh: from this nucleus order on -- default 6144, LPs with long rows "refresh_min_k_dense" = 2048; 0 = never -- a scheduled
 * refactorization keeps the explicit inverse, improved by one Newton-Schulz step X += X (I - C X) unless
 * refresh_refine is 0 (a sparse residual kernel + one f64 GEMM from rocBLAS), when the solutions recomputed with
 * it leave max |A x - s| and max basic |dj| below the tolerance, default 1e-6; it re-inverts otherwise and every
 * refresh_max-th time, default 15),
 * "factor_mode" (-1 auto: the LU form from "lu_min_k" = 3072 basic structurals on, for sparse LPs;
 * 0 explicit inverse of the nucleus; 1 LU form: host Markowitz front + dense tail inverted on the matrix cores +
 * product-form eta file) with "lu_stop_density", "lu_min_tail", "lu_threshold",
 * "lu_inverse_fill_cap" (front / explicit-inverse controls), "lu_max_pivots", "lu_min_pivots", "lu_adaptive" (eta-file
 * length: from a cost model of the refactorization, the same on every run), "lu_polish", "lu_polish_tolerance" (Newton-Schulz steps on the tail inverse),
 * "gemm_backend" (0 the engine's own MFMA f64 GEMM, 1 rocBLAS), "solution_refinements" / "refine_above" (iterative
 * refinement of the recomputed primal and dual solutions), "price_lds" (1 default: dense tableau rows priced with pi tiles staged in LDS, the chain's form chosen per batch; 2: on every pivot; 0: never; "price_lds_min_windows", "price_lds_grid": its layout knobs),
 * "sell_windows" (1 default: the pricing copy of A sorted by column length inside windows of 256 keys, one compaction block
 * per workgroup -- coalesced tableau-row stores, one candidate count per workgroup; 0: sorted globally, per-candidate atomics),
 * "fake_bound_cleanup" (1: "infeasible" reached with nonbasic variables still at fake bounds is reported as 10, "clean up in
 * primal", as ClpSimplex::dual does, src/ClpSimplex.cpp:5800-5803 -- for callers that hold a primal, the clpGpuDual adapter sets
 * it; 0 default: a bare context reports the 1 it found), "free_nonbasic" (1: nonbasic free columns keep the status isFree as in the
 * reference -- ClpSimplex::allSlackBasis src/ClpSimplex.cpp:7846, createRim's clean-up of a caller's basis :4317-4338, dualRow's
 * free-first entry src/ClpSimplexDual.cpp:3005-3055 with nextSuperBasic :8285, the general branch of dualColumn0 :4058-4179, "primal
 * feasible and only free dual infeasibilities: 10" :5619-5622 -- the clpGpuDual adapter sets it; 0 default: they are given bothFake
 * bounds at start, which serves a context without a primal better; not available in column-sharded runs),
 * "try_primal" (1: gutsOfDual's "problems - try primal" exit, src/ClpSimplexDual.cpp:533-547 -- primal infeasibilities that grew
 * 1e5-fold while the objective stood still, the signature of a runaway dual-bound escalation, end the solve with status 10; the
 * clpGpuDual adapter sets it; 0 default: a bare context has no primal to hand over to and carries on).
 * "factor_mode" -1 takes the LU form in column-sharded runs too.
 * Experiment knobs (profiles/r04_objective_race.md): "dse_reset_every" (uniform steepest-edge weights again at every N-th
 * refactorization), "debug_reset_weights_at" (once, from this iteration on).
 * Fault injection for the tests: "debug_backwards_at" (the first two status checks at or after this iteration see the
 * objective fall: drives the restore of src/ClpSimplexDual.cpp:5326-5488), "debug_bad_accuracy_at" (the first status
 * check at or after this iteration finds a primal error of 1e16: the restore of :5237-5318), "debug_singular_at" (the
 * refactorization of that check is taken as singular: the restore of :5060-5125), "debug_poison_inverse_at"
 * (a NaN in the kept inverse right before the next verified refresh). */
int clpgpu_set_option(clpgpu_context *ctx, const char *name, double value);
/* Whole-array replacement of bounds / costs with the matrix left resident
 * (ClpModel::chgRowLower ... chgObjCoefficients, src/ClpModel.hpp:254-262, src/ClpModel.cpp:2669-2770;
 * C interface Clp_chgRowLower ... src/Clp_C_Interface.h:150-158).  NULL means "no bound" / zero cost
 * as in the reference.  The next clpgpu_dual starts from these and from the status passed with
 * clpgpu_set_status: the warm re-solve of branch and bound. */
int clpgpu_chg_row_lower(clpgpu_context *ctx, const double *rowLower);
int clpgpu_chg_row_upper(clpgpu_context *ctx, const double *rowUpper);
int clpgpu_chg_column_lower(clpgpu_context *ctx, const double *columnLower);
int clpgpu_chg_column_upper(clpgpu_context *ctx, const double *columnUpper);
int clpgpu_chg_obj_coefficients(clpgpu_context *ctx, const double *objIn);
/* ClpPackedMatrix::scale (src/ClpPackedMatrix.cpp:4120-4760; ClpModel::scaling modes 1 equilibrium,
 * 2 geometric, 3/4 auto) as a stand-alone host computation: the row / column factors the engine
 * applies when option "scaling" is set before clpgpu_load_problem.  Needs no context and no device.
 * Returns 1 if the matrix is left unscaled (all factors 1), 0 if scaled, -1 on bad input. */
int clpgpu_scale_factors(int numberRows, int numberColumns, const int *columnStart, const int *rowIndex,
                         const double *element, const double *columnLower, const double *columnUpper,
                         const double *rowLower, const double *rowUpper, int mode, double primalTolerance,
                         double *rowScale, double *columnScale);
/* optional warm start (ClpSimplex::statusArray) */
int clpgpu_set_status(clpgpu_context *ctx, const unsigned char *status);
/* ClpSimplex::dual() (src/ClpSimplex.cpp:5631 -> ClpSimplexDual::dual :637): returns problemStatus
 * 0 optimal, 1 primal infeasible, 2 dual infeasible, 3 iteration limit, 4 numerical trouble,
 * 10 "needs primal clean-up" (src/ClpSimplex.cpp:5808). */
int clpgpu_dual(clpgpu_context *ctx);
/* Run at most `iterations` further pivots from the current device state (bench stepping);
 * the first call performs startup.  Returns problemStatus, or -1 if still iterating. */
int clpgpu_dual_steps(clpgpu_context *ctx, int iterations);
/* ClpSimplexDual::fastDual (src/ClpSimplexDual.hpp:276, src/ClpSimplexDual.cpp:7227-7480): the dual from the
 * basis at hand (clpgpu_set_status, or the one the last solve ended with when none is given), iteration count
 * restarted, bounds / costs as last changed with clpgpu_chg_*.  Returns 0 when the run came to a conclusion
 * (clpgpu_problem_status then says which: 0 optimal, 1 infeasible, 2 unbounded), 1 when it was stopped
 * (problem status 3): iteration limit (option "max_iterations"), or -- alwaysFinish == 0 -- the first time the
 * iteration loop asks for a refactorization (:7422-7431).  -99 on errors. */
int clpgpu_fast_dual(clpgpu_context *ctx, int alwaysFinish);
/* ClpSimplexDual::strongBranching (src/ClpSimplexDual.hpp:125-131, src/ClpSimplexDual.cpp:6965-7226), same
 * arguments and meaning: for variables[i] the "down" branch (column upper bound newUpper[i]) and the "up"
 * branch (column lower bound newLower[i]) are each solved with fastDual from the basis of the finished solve
 * the context holds; on return newUpper[i] / newLower[i] hold the change in objective of the down / up branch
 * (1e100 = infeasible), outputStatus[2i], [2i+1] = 0 finished / 1 infeasible / 2 unfinished,
 * outputIterations[2i], [2i+1] the iteration counts, outputSolution[2i], [2i+1] (each numberColumns long;
 * the array or single entries may be NULL) the column solutions -- even down, odd up.  The context is put
 * back (bounds, basis, solution, objective) before returning.  Return 0 nothing interesting, 1 some column
 * infeasible one way, -1 a column infeasible both ways, -2 error.  (startFinishOptions of the reference has
 * no meaning here: the matrix and rim live on the device throughout.) */
int clpgpu_strong_branching(clpgpu_context *ctx, int numberVariables, const int *variables, double *newLower,
                            double *newUpper, double **outputSolution, int *outputStatus, int *outputIterations,
                            int stopOnFirstInfeasible, int alwaysFinish);

/* ClpModel::problemStatus (src/ClpModel.hpp:441): -1 not finished, else as clpgpu_dual returns it */
int clpgpu_problem_status(const clpgpu_context *ctx);
int clpgpu_number_iterations(const clpgpu_context *ctx);
double clpgpu_objective_value(const clpgpu_context *ctx);
/* n+m doubles each, [columns | rows] */
int clpgpu_get_solution(clpgpu_context *ctx, double *solution);
int clpgpu_get_reduced_costs(clpgpu_context *ctx, double *dj);
int clpgpu_get_status(clpgpu_context *ctx, unsigned char *status);
int clpgpu_get_pivot_variable(clpgpu_context *ctx, int *pivotVariable);
/* returns total number of records; copies min(total, maxRecords) */
int clpgpu_get_pivot_log(clpgpu_context *ctx, clpgpu_pivot_record *out, int maxRecords);
/* dual steepest-edge reference weights and squared infeasibilities by basis position
 * (ClpDualRowSteepest::weights_, infeasible_; src/ClpDualRowSteepest.hpp) -- diagnostics */
int clpgpu_get_row_weights(clpgpu_context *ctx, double *weights, double *infeasibility);
/* ---- loopback ranks: the column-sharded path (AbcSimplexDual.cpp:1623-1634, ClpPackedMatrix.cpp:1823-1854)
 * with 2 / 4 / 8 ranks on ONE GPU.  N contexts of one process, one host thread and stream each; what the RCCL
 * all-gathers carry between GPUs travels by device-to-device copies.  For tests of the sharded code path on a
 * one-GPU box: rank offsets of the pack / merge kernels, owned reduced costs, the overflow fallback. */
typedef struct clpgpu_virtual_group clpgpu_virtual_group;
clpgpu_virtual_group *clpgpu_virtual_group_create(int nranks);
void clpgpu_virtual_group_destroy(clpgpu_virtual_group *group);
/* after clpgpu_load_problem and the options, instead of clpgpu_comm_init */
int clpgpu_virtual_attach(clpgpu_context *ctx, clpgpu_virtual_group *group, int rank);
/* clpgpu_dual_steps (iterations < 0: clpgpu_dual) on every rank at once; status[r] = what rank r returned.
 * 0, or -2 when the ranks lost step (an exchange timed out). */
int clpgpu_virtual_dual_steps(clpgpu_virtual_group *group, int iterations, int *status);

/* ClpSimplexProgress::cycle (src/ClpSolve.cpp:4726-4825; called from ClpSimplex::housekeeping, src/ClpSimplex.cpp:2397-2431):
 * the device housekeeping's cycle detector fed with a sequence of pivots (entering, leaving, their directions), from
 * empty history; matched[i] = its verdict at pivot i (0 none, k a cycle of length k, 100 irregular repeats).  A parity hook:
 * no LP here runs into a cycle on its own. */
int clpgpu_test_cycle(clpgpu_context *ctx, int count, const int *in, const int *out, const int *wayIn, const int *wayOut, int *matched);
/* Development hook (no Clp counterpart): the by-column row-pricing kernel -- ClpPackedMatrix::transposeTimesByColumn + the fused
 * first ratio pass, src/ClpPackedMatrix.cpp:961, :1799 -- launched `reps` times with a dense pi between two pivots of a started
 * context and timed with HIP events; masks[v] switches parts of the kernel off (1 candidate-count atomics, 2 by-column scatter of the
 * tableau row, 4 status / dj gathers, 8 the matrix sweep, 16 the gather of pi) so that their cost is measured apart;
 * microseconds[v] = mean launch time under masks[v].  The per-pivot scratch it overwrites is rewritten by the next pivot. */
int clpgpu_debug_price_bench(clpgpu_context *ctx, int reps, int numberMasks, const int *masks, double *microseconds);
/* Parity hook for the engine's host restatement of ClpSimplexProgress::looping (src/ClpSolve.cpp:4438-4611: the loop detector
 * over status checks that statusOfProblemInDual consults, src/ClpSimplexDual.cpp:5506-5536).  Host code only -- needs no device.
 * Check i sees objective[i] / infeasibility[i] / numberInfeasibilities[i] at iteration[i] with progressFlag_ & 3 = flagBits[i]
 * and newestIncoming[i] as the last incoming variable of the small-cycle list; code[i] = looping()'s return (-1 carry on, -2
 * something changed, 0 declare victory, 3 / 4 give up); dualTolerance / dualBound / forceFactorization = the solver's values
 * afterwards (1e-7, 1e10, -1 to begin with); flagged[i] = the sequence (< 64) it flagged, or -1.  Returns 0, -99 on bad arguments. */
int clpgpu_test_looping(int count, const double *objective, const double *infeasibility, const int *numberInfeasibilities, const int *iteration,
                        const int *flagBits, const int *newestIncoming, int *code, double *dualTolerance, double *dualBound, int *forceFactorization,
                        int *flagged);

/* Test hook (no Clp counterpart; the reference prices a dense pi by column straight from the CSC copy,
 * ClpPackedMatrix::transposeTimesByColumn, src/ClpPackedMatrix.cpp:961): the jagged row-tiled layout k_price_lds reads, built on
 * the host without a device.  order = numSlices * 64 column keys (-1 none); with capRecords = 0 only *tiles, *tileRows and
 * *records come back; with room, segStart[numSlices], cnt / src [numSlices * tiles * 64], home[numSlices * 64],
 * rowPair[records], elemPair[2 * records].  Returns 0, 1 = the layout refuses the matrix, -99 on bad arguments. */
int clpgpu_test_jds_layout(int m, int n, const int *colStart, const int *row, const double *elem, int numSlices, const int *order, int *tiles,
                           int *tileRows, long long *records, long long capRecords, int *segStart, unsigned char *cnt, unsigned char *src,
                           unsigned char *home, unsigned *rowPair, double *elemPair);

/* Test hook for the host side of the basis factorization (clp_amd/csrc/lu_front.h: the Markowitz LU of the nucleus that stops at a
 * dense tail -- what stands in for CoinAbcBaseFactorization::factorSparse, src/CoinAbcBaseFactorization2.cpp:18, and wantToGoDense,
 * src/CoinAbcBaseFactorization1.cpp:2409-2462), host code only.  C is k x k by columns; counts[6] = pivots, entries of L, entries of U off
 * the pivots, order of the tail, entries of the tail, fill-in.  have = 0: counts only; 1: the arrays too (sizes from a first call):
 * frow / fcol / fpiv [pivots], lStart / uStart [pivots + 1], lRow / lVal, uCol / uVal, tailRow / tailCol [tail], sRow / sCol / sVal.
 * Returns 0, -99 on bad arguments. */
int clpgpu_test_lu_front(int k, const int *cStart, const int *cRow, const double *cVal, double stopDensity, int minTail, double threshold,
                         long long *counts, int have, int *frow, int *fcol, double *fpiv, int *lStart, int *lRow, double *lVal, int *uStart,
                         int *uCol, double *uVal, int *tailRow, int *tailCol, int *sRow, int *sCol, double *sVal);

/* Test hook for the row choice of ClpSimplexDual::dualRow's free-first entry (src/ClpSimplexDual.cpp:3016-3049; option "free_nonbasic"),
 * host code only: work[m] = the FTRANned free column by basis position, pivotVariable[m], and solution / lower / upper / status by
 * sequence (numberSequences of them).  Returns the row the column should pivot on, -1 if none qualifies, -99 on bad arguments. */
int clpgpu_test_free_first_row(int m, int numberSequences, const double *work, const int *pivotVariable, const double *solution, const double *lower,
                               const double *upper, const unsigned char *status);

/* CoinAbcDgemm (src/CoinAbcHelperFunctions.cpp:1658; used by CoinAbcDgetrf, src/AbcSimplexParallel.cpp:2491-2534):
 * the engine's own f64 GEMM on the matrix cores, c = beta c + alpha a b for row-major n x n host arrays.  The
 * kernel behind the Newton-Schulz steps on the explicit (tail) inverse; exposed so that tests hold it to numpy. */
int clpgpu_dgemm(clpgpu_context *ctx, int n, double alpha, const double *a, const double *b, double beta, double *c);
int clpgpu_get_stats(clpgpu_context *ctx, clpgpu_stats *stats);
/* option "timing" = 2 (eager launches, a HIP event after every launch of the chain): accumulated time per
 * kernel of the pivots run so far -- kernel plus the launch gap before it.  names[i] point at static
 * strings.  Returns the number of kernels seen (fills at most maxKernels). */
int clpgpu_get_kernel_times(clpgpu_context *ctx, int maxKernels, const char **names, double *milliseconds,
                            long *launches);

#ifdef __cplusplus
}
#endif
#endif

// engine.hip -- host side of libclpgpu.so: owns the device state, drives the kernel chain of one
// pivot, and restates the *control plane* of ClpSimplexDual (statusOfProblemInDual, the exit
// handling of whileIterating) on host mirrors that are synced only at refactorization boundaries.
// There is no CPU compute fallback: every function that does arithmetic on the problem calls a
// HIP kernel of kernels.hip, and clpgpu_create() fails when no HIP device is present.
#include "kernels.hip"

#include <dlfcn.h>
#include <float.h>
#include <pthread.h>
#include <time.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/clpgpu.h"
#include "lu_front.h"
#include "perturb_host.h"

using namespace clpgpu;

#define HIPCHECK(expr)                                                                             \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      setError("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);         \
      return -99;                                                                                  \
    }                                                                                              \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// launch on the iteration chain; with option "timing" = 2 (eager launches) an event mark follows every
// launch on the main stream, so the per-kernel times of exactly the pivots being benchmarked can be read
#define KL(name, kernel, grid, block, lds, strm, ...)                                              \
  do {                                                                                             \
    hipLaunchKernelGGL(kernel, grid, block, lds, strm, __VA_ARGS__);                               \
    if (ktOn && (strm) == stream)                                                                  \
      ktMark(name);                                                                                \
  } while (0)
#define KT_MAX 32  // marks per pivot (the chain has 11-15 launches)

struct clpgpu_context;
struct LuTriHost;
// Loopback "communicator": N contexts of ONE process on one GPU, each driven by its own host thread and stream,
// exchange what the RCCL all-gathers would carry through device-to-device copies (clpgpu_virtual_*).  It exists so
// that the column-sharded code path -- rank offsets in k_shard_pack_* / k_shard_merge_*, owned reduced costs, the
// overflow fallback agreeing across ranks -- runs with 2 / 4 / 8 ranks on a one-GPU box.
struct clpgpu_virtual_group {
  int nranks = 0;
  clpgpu_context *ctx[8] = {};
  const void *sendPtr[8] = {};
  pthread_mutex_t mutex = PTHREAD_MUTEX_INITIALIZER;
  pthread_cond_t cond = PTHREAD_COND_INITIALIZER;
  int waiting = 0, generation = 0;
  bool failed = false;
  // barrier with a time limit: a rank that never arrives (diverged control flow) must not hang the box
  bool barrier()
  {
    pthread_mutex_lock(&mutex);
    if (failed) {
      pthread_mutex_unlock(&mutex);
      return false;
    }
    const int gen = generation;
    if (++waiting == nranks) {
      waiting = 0;
      generation++;
      pthread_cond_broadcast(&cond);
    } else {
      struct timespec ts;
      clock_gettime(CLOCK_REALTIME, &ts);
      ts.tv_sec += 20;
      while (gen == generation && !failed)
        if (pthread_cond_timedwait(&cond, &mutex, &ts) != 0) {
          failed = true;
          pthread_cond_broadcast(&cond);
          break;
        }
    }
    const bool ok = !failed;
    pthread_mutex_unlock(&mutex);
    return ok;
  }
};
// growable device buffer of the LU factorization (sizes change from one refactorization to the next)
struct DBuf {
  void *p = nullptr;
  size_t cap = 0;
  int need(clpgpu_context *ctx, size_t bytes, void *&out);
  template <typename T> int put(clpgpu_context *ctx, const std::vector<T> &v, T *&out);
};
enum {
  LB_SROW, LB_SCOL, LB_SVAL, LB_ROWOFLOCAL, LB_POSOFCOL, LB_TAILROW, LB_TAILCOL, LB_SROWINDEX, LB_SROWSTART, LB_SROWCOL, LB_SROWVAL,
  LB_SCOLSTART, LB_SCOLROW, LB_SCOLVAL, LB_WR, LB_XC, LB_TCV, LB_X0, LB_CP, LB_Y, LB_H, LB_G, LB_GT, LB_P, LB_PREV, LB_NEXT, LB_S, LB_GV, LB_DV,
  LB_LASTOFPOS, LB_HC, LB_XK, LB_SROWOF, LB_CSLOT, LB_POSOFCSLOT, LB_POSOFBASICCOL, LB_TRI, LB_COUNT = LB_TRI + 35
};

struct clpgpu_context {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string error;
  // ---- problem (host copies)
  int m = 0, n = 0, N = 0;
  long nnz = 0;
  std::vector<int> colStart, row;
  std::vector<double> elem;
  std::vector<double> colLower, colUpper, obj, rowLower, rowUpper;
  // the caller's (unscaled) arrays: what clone / set_scales reload from and what scaling reads
  std::vector<double> origElem, origColLower, origColUpper, origObj, origRowLower, origRowUpper;
  bool haveExternalScales = false;  // clpgpu_set_scales: factors supplied by the caller
  std::vector<int> rowStart;
  // ---- host mirrors of the rim (valid after pull())
  std::vector<double> lower, upper, cost, dj, sol, origLower, origUpper;
  std::vector<unsigned char> status;
  std::vector<int> pivotVariable;
  bool haveStatus = false;
  std::vector<unsigned char> userStatus;
  // ---- options / ClpSimplex scalars
  double primalTolerance = 1.0e-7, dualTolerance = 1.0e-7, dualToleranceBase = 1.0e-7, dualBound = 1.0e10;
  double zeroTolerance = 1.0e-13, acceptablePivot = 1.0e-8, largeValue = 1.0e15;
  // the values the caller asked for (clpgpu_set_option): a scaled load may loosen the tolerances
  // (ClpPackedMatrix::scale :4268, :4303) and a solve changes dualBound / acceptablePivot
  // (changeBounds, the "no incoming" exit); every load / startup starts again from these, as
  // ClpSimplex::dual does through ClpDataSave (saveData / restoreData, src/ClpSimplex.cpp:5690)
  double optPrimalTolerance = 1.0e-7, optDualTolerance = 1.0e-7, optZeroTolerance = 1.0e-13, optDualBound = 1.0e10,
         optAcceptablePivot = 1.0e-8;
  int maximumIterations = 2147483647, pivotRule = 1, maximumPivots = 200, logLevel = 0, checkEvery = 1;
  unsigned int seed = 1234567u;
  int timing = 0;
  // ---- state
  int problemStatus = -1, numberIterations = 0, numberRefactorizations = 0;
  double objectiveValue = 0.0;
  double largestPrimalError = 0.0, largestDualError = 0.0;
  double sumPrimalInfeasibilities = 0.0, sumDualInfeasibilities = 0.0, sumOfRelaxedPrimalInfeasibilities = 0.0,
         sumOfRelaxedDualInfeasibilities = 0.0;
  int numberPrimalInfeasibilities = 0, numberDualInfeasibilities = 0;
  int numberFake = 0, numberChanged = 0, numberTimesOptimal = 0, forceFactorization = -1, lastBadIteration = -999999;
  int lastCleaned = 0, factorType = 0;
  // ClpSimplex::perturbation_: the value every solve is entered with (option "perturbation": 102 never, 100 only the
  // kick after 2(m+n) iterations, 50 the clp command's default, 51-69 fixed fractions) and the running state
  // (101 = the costs are perturbed now, 102 = no more perturbing in this solve)
  int perturbationOption = 102, perturbation = 102, numberPerturbations = 0;
  // ClpSimplexProgress (src/ClpSolve.cpp:4289-4725), the part the dual uses: objective (less the possible improvement),
  // sum and number of primal infeasibilities and iteration count of the last five status checks, newest last
  static const int PROGRESS = 5;
  double progObjective[PROGRESS], progInfeasibility[PROGRESS];
  int progNumberInfeasibilities[PROGRESS], progIteration[PROGRESS];
  int progTimes = 0, progBadTimes = 0, progReallyBadTimes = 0, progTimesFlagged = 0;
  int progressFlag = 0;  // ClpSimplex::progressFlag_: 1 / 2 from the device (housekeeping), 4 costs copied, 8 has looked optimal
  std::vector<double> costCopy;          // the second half of cost_ once progressFlag_ & 4 (:5381-5390)
  double bestPossibleImprovement = 0.0;  // ClpSimplex::checkDualSolution :3087
  // gutsOfDual's "problems - try primal" exit (src/ClpSimplexDual.cpp:533-547; option "try_primal": 0 default -- a bare context has no primal to hand
  // over to -- the clpGpuDual adapter sets 1): state across the status checks
  int tryPrimal = 0, numberTryPrimal = 0;
  // ClpDualRowSteepest::mode_ (src/ClpDualRowSteepest.hpp:118: the constructor's default is 3) and what stands for
  // factorization()->numberElements() in its mode 3 (src/ClpDualRowSteepest.cpp:262): 0 = entries of the basic structural columns (the
  // count of an LU without fill; what the oracle computes too; the default), 1 = what the factorization on the device holds in
  // CoinFactorization's terms: in LU mode front L + U + the dense tail + the frozen slack part; under the explicit inverse -- nuclei
  // below lu_min_k, which triangularize or nearly so -- the entries an LU of that nucleus holds, i.e. the same count as 0 (k^2, the
  // inverse's own storage, would send a 224-column nucleus of a 50 000-row LP past ratio 1 where CoinFactorization's LU of it holds
  // 2 500 entries).  Why 0 is the default although a real LU of a mature basis holds far more than the structural columns' entries:
  // measured on the ladder (round 6), full scans in LU mode stall this LP family -- rung 7 000 x 28 000: 102 000 pivots / 35 s with the
  // partial scans of 0 against > 2.7 M pivots / 900 s unfinished with 1; rung 5 000: 64-80 000 against 105 000 pivots
  int steepestMode = 3, steepestElements = 0, chuzrFloor = 2000, debugLastBadIteration = -999999;
  long long pendingFactorElements = 0, luOwnElements = 0;
  double debugToleranceFactor = 0.0;
  double smallestPrimalInfeasibility = DBL_MAX, lastObjectiveValueGuts = -1.0e100;
  int numberBackwards = 0, numberLoopFlags = 0;  // statistics: backwards-objective restores, loops acted upon
  int debugBackwardsAt = -1;  // fault injection (option debug_backwards_at), as in the oracle
  int debugPoisonInverseAt = -1, numberPoisoned = 0;  // fault injection (option debug_poison_inverse_at)
  // option "free_nonbasic" 1: nonbasic free columns stay isFree as in the reference instead of being given bothFake bounds at start
  // (DESIGN section 2): free-first row choice (dualRow, src/ClpSimplexDual.cpp:3005-3055; host-assisted, freeFirstRow below), the
  // general branch of dualColumn0 (:4058-4179; k_free_scan), firstFree_ and the free counters of the status checks
  int freeNonbasic = 0;
  int firstFree = -1;      // ClpSimplex::firstFree_
  int noFreeOrSuper = 1;   // moreSpecialOptions_ & 8 as the last status check left it ("no free or super basic")
  int numberDualInfeasibilitiesWithoutFree = 0;
  int numberFreeFirstRows = 0, numberFreeEntered = 0;  // diagnostics: pivots whose row came from the free-first entry / that brought a free column in
  std::vector<int> freeListHost;
  int *dFreeList = nullptr;
  size_t freeListCap = 0;
  int freeFirstRow(int &chosenRow);
  int pushFreeList();
  int checkBoth = 1;  // option "check_both": gutsOfSolution ends in checkBothSolutions (1, this reference version) or in the older pair (0)
  int debugResetWeightsAt = -1;  // option debug_reset_weights_at (experiment)
  int dseResetEvery = 0, dseResetCounter = 0, numberWeightResets = 0;  // option dse_reset_every (experiment): uniform weights again every N-th refactorization
  int debugBadAccuracyAt = -1, numberAccuracyRestores = 0;  // fault injection (option debug_bad_accuracy_at) and its count
  int debugSingularAt = -1, numberSingularRestores = 0;     // fault injection (option debug_singular_at) and the count of such restores
  void progressReset();
  void progressStartCheck();
  int progressLooping();
  bool looksOptimal() const;
  void resetFakeBoundsToOriginal();
  std::vector<double> perturbationArray;  // ClpSimplex::perturbationArray_: 2n uniform numbers, drawn once per problem
  bool started = false, needStatus = true, weightsInitialized = false;
  bool rimInfeasible = false;  // ClpSimplex::sanityCheck found lower > upper at start-up: status 1 without a solve
  bool rebuildRowCopy = true;  // the device keeps the [basic|nonbasic] row partition current between refactorizations
  // basis / solution at the last good refactorization (ClpSimplex::saveStatus_, savedSolution_): what a
  // singular refactorization falls back to (ClpSimplexDual.cpp:5060-5125)
  std::vector<unsigned char> saveStatus;
  std::vector<double> savedSolution;
  bool haveSnapshot = false;
  // Verified refresh (large nuclei): at a scheduled refactorization the explicit inverse, kept current by the
  // rank-1 / bordering updates, is KEPT when the solutions recomputed with it leave residuals below
  // refreshTolerance (max |A x - s|, max |dj| over the basics: the reference's own measures of a factorization,
  // largestPrimalError_ / largestDualError_); otherwise -- and every refreshMax-th time anyway -- the nucleus is
  // re-inverted.  The reference re-factorizes because its eta file grows and errors accumulate; with an explicit
  // inverse only the second reason is left, and it is measured.  Kept as it is, the inverse of the bench LP
  // drifts to residuals of 1e-7..1e-5 within the ~475 pivots between refactorizations (10-100x what a
  // re-inversion leaves, profiles/r02_refresh_errors.txt); with one Newton-Schulz step first (refineInverse) it
  // comes out at 1e-9..4e-8, below the re-inversion's own 4e-9..6e-7, at a fifth of the cost
  // (profiles/r02_refresh_newton.txt).  Options "refresh_min_k" (nuclei of at least this order, default 6144;
  // 0 = never), "refresh_max", "refresh_tolerance", "refresh_refine".
  int refreshMinK = 6144, refreshMax = 15, consecutiveRefreshes = 0, numberRefreshes = 0, numberRefreshesRejected = 0;
  double refreshTolerance = 1.0e-6;
  int lastExitState = EXIT_REFACTOR;  // why the iteration loop last stopped (whileIterating)
  long exitScheduled = 0, exitAlphaCheck = 0, exitBackwards = 0, exitBadUpdate = 0, priceFormSwitches = 0;  // what sent the loop to a status check
  bool stepPendingRefactor = false;   // a stepped run stopped on a pivot whose housekeeping asked for a refactorization
  bool refreshEligible();
  int refreshFactor();
  // option "refresh_refine" (default 1): before the check, one Newton-Schulz step X += X (I - C X) on the kept
  // inverse -- a sparse residual kernel and one k^3 f64 GEMM (rocBLAS, loaded on first use) instead of the
  // latency-bound elimination; refreshResidualMax: no step (re-invert) when max |I - C X| exceeds it
  int refreshRefine = 1, numberRefines = 0, refreshMinKDense = 2048;
  double refreshResidualMax = 1.0e-2, lastResidual = 0.0;
  void *blasHandle = nullptr;
  // tailFromS: the matrix is the LU mode's dense tail (its entries are still on the device as triplets);
  // goodBelow: return 3 without a step when max |I - C X| is already below it
  int refineInverse(bool tailFromS = false, double goodBelow = 0.0);
  // option "gemm_backend": 0 = the engine's own MFMA f64 GEMM (gemm_kernel.hip, default), 1 = rocBLAS dgemm loaded at
  // run time (kept as the comparison point SURVEY section 7 allows)
  int gemmBackend = 0;
  int dgemmDevice(int nn, double alpha, const double *A, const double *B, double beta, double *C);
  int luPolish = 2, numberPolishSteps = 0;
  double luPolishTolerance = 1.0e-9, luLastResidual = 0.0;
  int numberThrownOut = 0;  // structurals replaced by slacks by the singular-basis repair, whole solve
  void resetFakeBounds();
  int pivots = 0, kNucleus = 0;
  // option "solution_refinements": refinement passes of the primal / dual solves at a refactorization, taken
  // only while the residual exceeds "refine_above" (ClpSimplex::numberRefinements_, src/ClpSimplex.hpp)
  int solutionRefinements = 2, numberPrimalRefinements = 0, numberDualRefinements = 0;
  double refineAbove = 1.0e-9;
  // ---- LU factorization mode (lu_front.h, lu_host.hip, lu_kernels.hip): the nucleus as a sparse
  // Markowitz front (host, at the refactorization) + a dense tail inverted on the matrix cores, with a
  // product-form eta file between refactorizations.  Option "factor_mode": 0 = explicit inverse of the whole
  // nucleus (rank-1 updated), 1 = LU, -1 (default) = LU from "lu_min_k" basic structurals on (sparse LPs, one GPU).
  int factorMode = -1, luMinK = 3072, luMaxPivots = 2000, luMinTail = 16;
  double luStopDensity = 0.03, luThreshold = 0.1;  // (0.012 until round 5: profiles/r05_stop_density_probe.txt)
  bool luActive = false, luSlotsCleared = false;
  LuFront luF;
  LuDev hLu = {};
  LuDev *dLu = nullptr;
  DBuf luBuf[LB_COUNT];
  double luFrontSeconds = 0.0, luInvertSeconds = 0.0, luBuildSeconds = 0.0;
  long luFactorizations = 0;
  int luLastFront = 0, luLastTail = 0;
  long luLastInverseFill = 0;
  // option "lu_adaptive" (default 1): the eta file's length follows the measured refactorization time
  int luAdaptive = 1, luMinPivots = 200, luEtaLimit = 1000;
  double luRefactorSeconds = 0.0;
  int fakeBoundCleanup = 0;  // option "fake_bound_cleanup": status 1 with fake bounds active becomes 10 (ClpSimplex::dual :5800-5803)
  int luFillSkip = 0, luFillSkipBelowK = 0;  // refactorizations that skip the LU attempt after a fill-cap fallback
  double luInverseFillCap = 6.0e6;  // option "lu_inverse_fill_cap"
  int luUploadTri(const LuTriHost &h, LuTri &d, int slot);
  int luFtran(const double *v0, const double *v1, double *o0, double *o1);
  int luBtran(const double *cPos, double *yRow);
  void luLaunchBtran();
  void luLaunchFtran(int gm, int parity);
  // ---- device
  Dev D;
  Ctrl *hCtrl = nullptr;  // pinned
  std::vector<void *> allocations;
  int kcap = 0, ld = 0;
  int *dKcol = nullptr, *dLocalOfRow = nullptr, *dInfo = nullptr;
  int nLongBlocks = 0;
  int nSellBlocks = 0, nChzBlocks = 0, priceKernel = 6, useGraph = 1, nWideBlocks = 1;
  bool widePricing = false;
  int maxColumnLength = 1;
  bool denseColumns = false;  // every column holds all m rows in ascending order
  // no row holds more than n/256 entries: a row then collects more than FLIP_SLOTS flip contributions only
  // in freak pivots (the scatter form of the flip right-hand side is kept for such LPs)
  bool lightRows = false;
  bool wideRows = false;  // mean row length >= 256 (dense LPs): wave-per-row / split-k variants of the row-wise stages
  int blockedRefactor = 1;
  // option "refactor_mode": -1 auto (two-level in-place re-inversion with the MFMA update from
  // refactorMinK basic structurals on, the one-level exact form below), 1 one-level, 2 two-level with the
  // vector update (same bits as 1), 3 two-level with the MFMA update
  int refactorMode = -1, refactorMinK = 1024;
  int registerPanel = 1;  // option "register_panel": 0 forces the global-memory panel kernel (used for k > 4096)
  // basis update (rank-1 sweep + fix-ups of Minv) on a second stream beside primal update,
  // housekeeping and the next CHUZR; joined before the next BTRAN reads Minv
  // option "row_price_frac": row pricing goes BY ROW when nnz(pi) <= frac * m (the reference's switch,
  // src/ClpPackedMatrix.cpp:727-754, with the crossover measured on the MI355X); 0 = always by column
  double rowPriceFrac = 0.02;
  // option "price_lds" (default 1; before the load): keep a jagged, row-tiled copy of the SELL windows and price DENSE tableau
  // rows with pi in LDS (k_price_lds, kernels.hip).  Bit-identical to k_price_sell (tests).  The chain carries one pricing
  // form per captured graph: priceMode 1 = k_price_lds alone, 0 = k_price_sell + the by-row form; the host switches between
  // them from the share of dense-pi pivots in the last batch (Ctrl::statDensePi) -- either form prices any pi correctly, so
  // the choice changes the speed of a pivot, never its result.
  int numberDcWideTimeouts = 0, debugDcTimeoutAt = -1;
  int commGraph = 0;  // option "comm_graph": hipGraph batches in column-sharded runs over a real communicator (see clpgpu_comm_init)
  int luFold = 1;  // option "lu_fold", bits: 1 = the primal update made by k_lu_pf_append's position workgroups (one launch fewer per LU-mode pivot,
                   // 28.5 -> 17.0 us for the pair between HIP events); 2 = c' built by the last workgroup of k_lu_pf_d (measured SLOWER: 29.3 against
                   // 12.6 + 10.0 us -- the chain walk of one workgroup behind coherent loads; off)
  int panel68 = 1;  // option "panel_6x8" (tuning): tails of 4097 .. 6144 rows take inner panels of 8 columns (k_gj_panel_reg<6, 8, 1024>)
  int luPfsBlocks = 256;  // option "lu_pfs_blocks" (tuning): workgroups of k_lu_pf_s (each stages x0[P] once)
  int luGemvThreads = 128;  // option "lu_gemv_threads" (tuning): workgroup size of the row-dot streams k_lu_gemv3 / k_lu_gemvT / k_lu_eta_apply
  int luScatterPpb = 200;  // option "lu_scatter_ppb" (tuning): positions per workgroup of k_ftran_scatter3_lu with the compact eta file
  int luCompactEta = 1;  // option "lu_compact_eta": the chain's FTRAN reads the eta file over the structural positions only (device_state.h, LuDev::Hc)
  int dcWide = 1;  // option "dc_wide": 1 (default) lists beyond one workgroup's registers go to k_dual_column_wide, 2 every list does (tests), 0 the single-workgroup walk
  int priceLds = 1;
  int priceLdsMinWindows = 256;  // option "price_lds_min_windows": narrower LPs keep k_price_sell (a one-workgroup-per-CU launch needs work for every CU)
  int priceLdsGridCap = 256;     // option "price_lds_grid" (test knob): fewer workgroups than CUs, so that a workgroup takes several rounds of windows
  int priceMode = 0, graphPriceMode = 0;
  bool freeActive = false, graphFreeActive = false;  // the chain carries k_free_scan (option free_nonbasic and free / superbasic nonbasics about)
  double modePriced = 0.0, modeDense = 0.0;
  bool jdsReady = false;
  size_t priceLdsBytes = 0;
  int priceLdsGrid = 0;
  std::vector<void *> sellBuffers;  // device buffers of the current SELL / jagged copies (freed when buildSell runs again)
  int sellWindows = 1;  // option "sell_windows": the SELL copy sorted by length inside compaction-block windows (buildSell)
  // option "flip_scatter": the waves that detect a bound flip scatter its column into per-row slots
  // (1: sparse LPs with light rows only; 2: any sparse LP -- tests); 0 = the single-workgroup assembly
  // from the flip records
  int flipScatter = 1;
  int flipSlotCap = FLIP_SLOTS;  // option "flip_slot_cap" (test knob: small values force the many-contributors path)
  // option "scaling" (0 off, default; 1/2/3/4 as ClpModel::scaling): set BEFORE clpgpu_load_problem.  The
  // device then holds the scaled LP; solution getters return unscaled values, clpgpu_chg_* take unscaled ones.
  int scalingMode = 0;
  bool scaled = false;
  std::vector<double> rowScale, colScale;
  int flipListCap = FLIP_LIST_CAP;  // option "flip_list_cap": smaller values force the overflow path (tests)
  int forkUpdate = 0;  // measured: 217 us/pivot forked vs 200 us single-stream (cross-stream graph edges cost more than they hide)
  hipStream_t stream2 = nullptr;
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  bool sidePending = false;
  // ---- multi-GPU (RCCL resolved at run time; a single-GPU build has no link dependency on it)
  int rank = 0, nranks = 1, shardChunk = 0;
  bool commActive = false;
  // commMode 2 (default): every rank handles only its own column range and the ranks all-gather the
  // per-rank candidate lists and flip records (k_shard_*); 1: the round-1 form (dense tableau-row
  // slices all-gathered, everything downstream replicated) -- also what a run falls back to when a
  // rank's records outgrow the exchange buffer
  int commMode = 2;
  int shardCandCap = 2048, shardFlipCap = 512;  // options "shard_cand_cap", "shard_flip_cap"
  int shardGrow = 1;  // option "shard_grow": 0 = an overflow goes straight to the dense row exchange (tests)
  double *dCandSend = nullptr, *dCandRecv = nullptr, *dFlipSend = nullptr, *dFlipRecv = nullptr;
  int allocShardBuffers();
  bool shardBuffersOwned = false;
  void *comm = nullptr;
  int (*ncclAllGatherFn)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  clpgpu_virtual_group *virtualGroup = nullptr;  // loopback exchange instead of RCCL (clpgpu_virtual_attach)
  bool commFailed = false;
  // every rank contributes `count` elements; recv holds them in rank order (send may alias its slice of recv)
  void allGather(const void *send, void *recv, size_t count, int dtype);
  int (*ncclCommDestroyFn)(void *) = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graphExec = nullptr;
  int graphIterations = 0;
  // shorter batches (1, 2, 4, ... pivots) for the tail of a stepped run (clpgpu_dual_steps): a batch of
  // check_every pivots whose step limit falls inside it replays every kernel of the remaining pivots as a no-op
  // (~20 us per pivot); captured the first time a tail of that size is asked for
  static const int TAIL_SIZES = 8;
  hipGraph_t tailGraph[TAIL_SIZES] = {};
  hipGraphExec_t tailExec[TAIL_SIZES] = {};
  int captureBatch(int count, hipGraph_t &g, hipGraphExec_t &exec);
  int logCapacity = 0;
  // ---- stats
  clpgpu_stats stats;
  std::vector<hipEvent_t> evStart, evStop, evMid;  // evMid: behind the by-column kernel, ahead of the by-row form's second pass
  int evUsed = 0;
  double seconds = 0.0;
  // per-kernel event marks (timing == 2)
  bool ktOn = false;
  int ktPivot = 0;
  std::vector<hipEvent_t> ktEvents;        // [checkEvery * KT_MAX]
  std::vector<int> ktSlotOfMark, ktMarks;  // slot of every mark; marks recorded per pivot of the batch
  std::vector<const char *> ktNames;
  std::vector<double> ktMs;
  std::vector<long> ktCount;
  void ktBegin(int pivotInBatch);
  void ktMark(const char *name);
  void ktCollect(int livePivots);

  void setError(const char *fmt, ...)
  {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    error = buf;
    if (logLevel > 0)
      fprintf(stderr, "clpgpu: %s\n", buf);
  }

  template <typename T> int dalloc(T *&p, size_t count)
  {
    void *q = nullptr;
    size_t bytes = sizeof(T) * (count ? count : 1);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
      setError("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
      return -99;
    }
    (void)hipMemsetAsync(q, 0, bytes, stream);
    allocations.push_back(q);
    p = (T *)q;
    return 0;
  }
  template <typename T> int h2d(T *dst, const T *src, size_t count)
  {
    hipError_t e = hipMemcpyAsync((void *)dst, src, sizeof(T) * count, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) {
      setError("h2d failed: %s", hipGetErrorString(e));
      return -99;
    }
    return 0;
  }
  template <typename T> int d2h(T *dst, const T *src, size_t count)
  {
    hipError_t e = hipMemcpyAsync(dst, (const void *)src, sizeof(T) * count, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess)
      e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
      setError("d2h failed: %s", hipGetErrorString(e));
      return -99;
    }
    return 0;
  }
  int sync()
  {
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
      setError("stream sync failed: %s", hipGetErrorString(e));
      return -99;
    }
    return 0;
  }

  int loadProblem(int m_, int n_, const int *cs, const int *ri, const double *el, const double *cl, const double *cu,
                  const double *ob, const double *rl, const double *ru);
  void releaseProblem();
  void applyShard();
  int checkLaunches(const char *where);
  int allocNucleus(int kNeeded);
  int buildSell();
  int buildJds(const std::vector<int> &order, int numSlices);
  void dropGraph();
  int pushCtrl();
  int pullCtrl();
  int pushRim();
  int pullRim(bool all);
  int startup();
  int factorize(bool repair = false);
  int factorizeOnce();
  int rebuildRowCopyIfNeeded();
  int prepareWork(int k);
  int invertWork(int k, std::vector<int> &perm, int info[4]);
  int factorizeLu(const std::vector<int> &kcol, const std::vector<int> &rrows, const std::vector<int> &localOfRow);
  int lastSingularColumn = -1, lastSingularRow = -1;
  int ftranDevice(const double *vRow, double *xPos);
  int ftranDevice2(const double *v1Row, const double *v2Row, double *x1Pos, double *x2Pos);
  void preparePlugin();
  bool rejectScaled(const char *what);
  int btranDevice(const double *cPos, double *yRow);
  int gutsOfSolution();
  void checkPrimalSolution();
  void checkDualSolution();
  void checkBothSolutions();
  int changeBounds(int initialize, double &changeCost);
  int perturb();
  void restoreCosts();
  int numberAtFakeBound() const;
  int updateDualsFullRecompute();
  int saveWeights(int mode);
  int statusOfProblemInDual(int type);
  int launchIteration(bool firstOfBatch, int parity);
  void joinUpdateBranch();
  int launchBatch(int count = -1);
  bool capturing = false;
  int whileIterating(int stepTarget);
  int debugPriceBench(int reps, int numberMasks, const int *masks, double *microseconds);
  // ClpSimplexDual::fastDual (src/ClpSimplexDual.cpp:7227): 0 = run() as clpgpu_dual does; 1 = in fastDual with
  // alwaysFinish, 2 = in fastDual without (stop at the first exit of the iteration loop that asks for a
  // refactorization: "can't say anything interesting - might as well return", :7422-7431)
  int fastDualMode = 0;
  int lastReturnCode = -1;  // whileIterating's return code in the reference's numbering (:7352-7358)
  int fastDual(bool alwaysFinish);
  int run(int maxSteps);
  void finish();
  int priceRow(int numberPi, const int *piIndex, const double *piValue, const unsigned char *st, const double *djv,
               double zeroTol, double dualTol, double accPivot, int *numberOut, int *outIndex, double *outValue,
               int *numberCand, int *candIndex, double *candValue, double *upperTheta);
};

int DBuf::need(clpgpu_context *ctx, size_t bytes, void *&out)
{
  if (bytes > cap || !p) {
    if (p) {
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipFree(p);
      p = nullptr;
    }
    const size_t want = bytes + bytes / 4 + 256;
    if (hipMalloc(&p, want) != hipSuccess) {
      p = nullptr;
      cap = 0;
      ctx->setError("hipMalloc(%zu) failed (LU factorization)", want);
      return -99;
    }
    cap = want;
    (void)hipMemsetAsync(p, 0, want, ctx->stream);
  }
  out = p;
  return 0;
}
template <typename T> int DBuf::put(clpgpu_context *ctx, const std::vector<T> &v, T *&out)
{
  void *q = nullptr;
  int rc = need(ctx, sizeof(T) * (v.size() ? v.size() : 1), q);
  out = (T *)q;
  if (!rc && !v.empty())
    rc = ctx->h2d(out, v.data(), v.size());
  return rc;
}

#include "lu_host.hip"

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Scaling -- ClpPackedMatrix::scale (src/ClpPackedMatrix.cpp:4120-4760) as ClpSimplex::createRim
// calls it (src/ClpSimplex.cpp:3701): row and column factors for mode 1 (equilibrium), 2 (geometric),
// 3/4 (auto: equilibrium first, geometric kept when its spread is more than twice better), then the
// final column pass.  Host code (the reference does this once per solve on the CPU as well).  May
// tighten the zero tolerance / loosen the feasibility tolerances exactly as the reference does.
// Returns 1 when the matrix is left unscaled (every |a_ij| within [0.5, 2], :4273), else 0.
// ---------------------------------------------------------------------------------------------
static int computeScaleFactors(int m, int n, const int *colStart, const int *row, const double *elem, const double *colLower,
                               const double *colUpper, const double *rowLower, const double *rowUpper, int mode,
                               double &primalTolerance, double &dualTolerance, double &zeroTolerance, double *rowScale,
                               double *columnScale)
{
  // row-ordered copy for the row passes
  std::vector<int> rowStart(m + 1, 0), rcol(colStart[n] ? colStart[n] : 1);
  std::vector<double> relem(colStart[n] ? colStart[n] : 1);
  for (int p = 0; p < colStart[n]; p++)
    rowStart[row[p] + 1]++;
  for (int i = 0; i < m; i++)
    rowStart[i + 1] += rowStart[i];
  {
    std::vector<int> fill(rowStart.begin(), rowStart.end() - 1);
    for (int j = 0; j < n; j++)
      for (int p = colStart[j]; p < colStart[j + 1]; p++) {
        int q = fill[row[p]]++;
        rcol[q] = j;
        relem[q] = elem[p];
      }
  }
  std::vector<char> usefulColumn(n, 0), usedRow(m, 0);
  double largest = 0.0, smallest = 1.0e50;
  for (int j = 0; j < n; j++) {
    char useful = 0;
    if (colUpper[j] > colLower[j] + 1.0e-12) {
      for (int p = colStart[j]; p < colStart[j + 1]; p++) {
        double value = fabs(elem[p]);
        if (value > 1.0e-20) {
          useful = 1;
          largest = std::max(largest, value);
          smallest = std::min(smallest, value);
        }
      }
    }
    usefulColumn[j] = useful;
  }
  if (smallest * 1.0e12 < largest) {  // :4268 increase tolerances
    dualTolerance = std::max(dualTolerance, 5.0e-7);
    primalTolerance = std::max(primalTolerance, 5.0e-7);
  }
  if (smallest >= 0.5 && largest <= 2.0)  // :4273 don't bother scaling
    return 1;
  if (largest > 1.0e13 * smallest)  // :4303 safer to have a smaller zero tolerance
    zeroTolerance = std::min(zeroTolerance, std::max(smallest / largest * 0.5, 1.0e-18));
  int scalingMethod = mode == 4 ? 3 : mode;
  double savedOverallRatio = 0.0;
  const double tolerance = 5.0 * primalTolerance;
  double overallLargest, overallSmallest = 1.0e20;
  bool finished = false;
  while (!finished) {
    int numberPass = 3;
    std::fill(rowScale, rowScale + m, 1.0);
    std::fill(columnScale, columnScale + n, 1.0);
    if (scalingMethod == 1 || scalingMethod == 3) {
      for (int i = 0; i < m; i++) {  // maximum in each row (:4340)
        largest = 1.0e-10;
        for (int q = rowStart[i]; q < rowStart[i + 1]; q++)
          if (usefulColumn[rcol[q]])
            largest = std::max(largest, fabs(relem[q]));
        rowScale[i] = 1.0 / largest;
      }
    } else {
      while (numberPass) {  // geometric mean: rows, columns, rows (:4365-4445)
        numberPass--;
        for (int i = 0; i < m; i++) {
          largest = 1.0e-50;
          smallest = 1.0e50;
          for (int q = rowStart[i]; q < rowStart[i + 1]; q++) {
            int j = rcol[q];
            if (usefulColumn[j]) {
              double value = fabs(relem[q]) * columnScale[j];
              largest = std::max(largest, value);
              smallest = std::min(smallest, value);
            }
          }
          rowScale[i] = 1.0 / sqrt(smallest * largest);
        }
        if (numberPass == 1)
          break;
        for (int j = 0; j < n; j++)
          if (usefulColumn[j]) {
            largest = 1.0e-50;
            smallest = 1.0e50;
            for (int p = colStart[j]; p < colStart[j + 1]; p++) {
              double value = fabs(elem[p]) * rowScale[row[p]];
              largest = std::max(largest, value);
              smallest = std::min(smallest, value);
            }
            columnScale[j] = 1.0 / sqrt(smallest * largest);
          }
      }
    }
    for (int i = 0; i < m; i++) {  // if ranges will make horrid then scale (:4451)
      double difference = rowUpper[i] - rowLower[i];
      double scaledDifference = difference * rowScale[i];
      if (scaledDifference > tolerance && scaledDifference < 1.0e-4) {
        rowScale[i] *= 1.0e-4 / scaledDifference;
        rowScale[i] = std::max(1.0e-10, std::min(1.0e10, rowScale[i]));
      }
    }
    overallSmallest = 1.0e50;  // what the smallest would be if every column's largest were 1.0 (:4465)
    for (int j = 0; j < n; j++)
      if (usefulColumn[j]) {
        largest = 1.0e-20;
        smallest = 1.0e50;
        for (int p = colStart[j]; p < colStart[j + 1]; p++) {
          double value = fabs(elem[p] * rowScale[row[p]]);
          largest = std::max(largest, value);
          smallest = std::min(smallest, value);
        }
        if (overallSmallest * largest > smallest)
          overallSmallest = smallest / largest;
      }
    if (scalingMethod == 1 || scalingMethod == 2) {
      finished = true;
    } else if (savedOverallRatio == 0.0 && scalingMethod != 4) {
      savedOverallRatio = overallSmallest;
      scalingMethod = 4;
    } else if (overallSmallest > 2.0 * savedOverallRatio) {
      finished = true;  // geometric was better
    } else {
      scalingMethod = 1;  // redo equilibrium
    }
  }
  // final pass: columns scaled so that their largest entry is reasonable (:4528-4590)
  overallLargest = 1.0;
  if (overallSmallest < 1.0e-1)
    overallLargest = 1.0 / sqrt(overallSmallest);
  overallLargest = std::min(100.0, overallLargest);
  overallSmallest = 1.0e50;
  for (int j = 0; j < n; j++) {
    if (colUpper[j] > colLower[j] + 1.0e-12 && colStart[j + 1] > colStart[j]) {
      largest = 1.0e-20;
      smallest = 1.0e50;
      for (int p = colStart[j]; p < colStart[j + 1]; p++) {
        int i = row[p];
        usedRow[i] = 1;
        double value = fabs(elem[p] * rowScale[i]);
        largest = std::max(largest, value);
        smallest = std::min(smallest, value);
      }
      columnScale[j] = overallLargest / largest;
      double difference = colUpper[j] - colLower[j];
      if (difference < 1.0e-5 * columnScale[j])
        columnScale[j] = difference / 1.0e-5;  // make gap larger
      overallSmallest = std::min(overallSmallest, smallest * columnScale[j]);
    } else {
      columnScale[j] = 1.0;
    }
  }
  for (int i = 0; i < m; i++)
    if (!usedRow[i])
      rowScale[i] = 1.0;
  if (overallSmallest < 1.0e-13)  // :4601
    zeroTolerance = std::max(overallSmallest * 0.5, 1.0e-18);
  return 0;
}

// a bound pair in scaled units as createRim builds it (src/ClpSimplex.cpp:3920-3980): infinities
// stay infinite, gaps below the primal tolerance are closed
static void scaleBoundPair(double lowerValue, double upperValue, double multiplier, double primalTolerance, double &lo, double &up)
{
  const double INF = 1.0e30;
  if (lowerValue > -1.0e20) {
    lo = lowerValue * multiplier;
    if (upperValue >= 1.0e20) {
      up = INF;
    } else {
      up = upperValue * multiplier;
      if (fabs(up - lo) <= primalTolerance) {
        if (lo >= 0.0)
          up = lo;
        else if (up <= 0.0)
          lo = up;
        else
          lo = up = 0.0;
      }
    }
  } else if (upperValue < 1.0e20) {
    lo = -INF;
    up = upperValue * multiplier;
  } else {
    lo = -INF;
    up = INF;
  }
}

int clpgpu_context::loadProblem(int m_, int n_, const int *cs, const int *ri, const double *el, const double *cl,
                                const double *cu, const double *ob, const double *rl, const double *ru)
{
  // a second load on the same context (ClpSimplex::loadProblem on a live model) starts from scratch:
  // the inputs are copied first, they may be this context's own host copies (clone, set_scales)
  {
    const long nz = cs[n_];
    std::vector<int> cs2(cs, cs + n_ + 1), ri2(ri, ri + nz);
    std::vector<double> el2(el, el + nz), cl2(cl, cl + n_), cu2(cu, cu + n_), ob2(ob, ob + n_), rl2(rl, rl + m_), ru2(ru, ru + m_);
    releaseProblem();
    colStart.swap(cs2);
    row.swap(ri2);
    elem.swap(el2);
    colLower.swap(cl2);
    colUpper.swap(cu2);
    obj.swap(ob2);
    rowLower.swap(rl2);
    rowUpper.swap(ru2);
  }
  m = m_;
  n = n_;
  N = m + n;
  nnz = colStart[n];
  origElem = elem;
  origColLower = colLower;
  origColUpper = colUpper;
  origObj = obj;
  origRowLower = rowLower;
  origRowUpper = rowUpper;
  cl = origColLower.data();
  cu = origColUpper.data();
  rl = origRowLower.data();
  ru = origRowUpper.data();
  scaled = false;
  if (haveExternalScales) {
    // clpgpu_set_scales: the caller's factors (Clp's own rowScale_ / columnScale_) instead of computed ones
    scaled = true;
    for (int j = 0; j < n; j++) {
      for (int p = colStart[j]; p < colStart[j + 1]; p++)
        elem[p] = elem[p] * (colScale[j] * rowScale[row[p]]);  // element *= scale * rowScale[iRow], src/ClpPackedMatrix.cpp:4789-4794
      obj[j] *= colScale[j];
      scaleBoundPair(cl[j], cu[j], 1.0 / colScale[j], primalTolerance, colLower[j], colUpper[j]);
    }
    for (int i = 0; i < m; i++)
      scaleBoundPair(rl[i], ru[i], rowScale[i], primalTolerance, rowLower[i], rowUpper[i]);
  } else if (scalingMode > 0) {
    // ClpSimplex::createRim with scalingFlag_ > 0 (src/ClpSimplex.cpp:3682-3702, :3880-3980): factors
    // from the caller's matrix, then everything the engine keeps is in scaled units
    rowScale.assign(m, 1.0);
    colScale.assign(n, 1.0);
    if (!computeScaleFactors(m, n, colStart.data(), row.data(), elem.data(), colLower.data(), colUpper.data(), rowLower.data(),
                             rowUpper.data(), scalingMode, primalTolerance, dualTolerance, zeroTolerance, rowScale.data(),
                             colScale.data())) {
      scaled = true;
      dualToleranceBase = dualTolerance;
      for (int j = 0; j < n; j++) {
        for (int p = colStart[j]; p < colStart[j + 1]; p++)
          elem[p] = elem[p] * (colScale[j] * rowScale[row[p]]);  // element *= scale * rowScale[iRow], src/ClpPackedMatrix.cpp:4789-4794
        obj[j] *= colScale[j];
        scaleBoundPair(cl[j], cu[j], 1.0 / colScale[j], primalTolerance, colLower[j], colUpper[j]);
      }
      for (int i = 0; i < m; i++)
        scaleBoundPair(rl[i], ru[i], rowScale[i], primalTolerance, rowLower[i], rowUpper[i]);
    }
  }
  // row copy (ClpSimplex::createRim builds rowCopy_, src/ClpSimplex.cpp:3648) with cross indices
  rowStart.assign(m + 1, 0);
  for (long p = 0; p < nnz; p++)
    rowStart[row[p] + 1]++;
  for (int i = 0; i < m; i++)
    rowStart[i + 1] += rowStart[i];
  std::vector<int> ccol(nnz), csrToCsc(nnz), cscToCsr(nnz), fill(rowStart.begin(), rowStart.end() - 1);
  std::vector<double> relem(nnz);
  for (int j = 0; j < n; j++)
    for (int p = colStart[j]; p < colStart[j + 1]; p++) {
      int q = fill[row[p]]++;
      ccol[q] = j;
      relem[q] = elem[p];
      csrToCsc[q] = p;
      cscToCsr[p] = q;
    }
  memset(&D, 0, sizeof(D));
  D.m = m;
  D.n = n;
  D.N = N;
  D.firstColumn = 0;
  D.lastColumn = n;
  D.priceFirst = 0;
  D.priceLast = n;
  int *dColStart, *dRow, *dRowStart;
  double *dElem;
  int rc = 0;
  rc |= dalloc(dColStart, n + 1);
  rc |= dalloc(dRow, nnz);
  rc |= dalloc(dElem, nnz);
  rc |= dalloc(dRowStart, m + 1);
  rc |= dalloc(D.ccol, nnz);
  rc |= dalloc(D.cslot, nnz);
  rc |= dalloc(D.relem, nnz);
  rc |= dalloc(D.csrToCsc, nnz);
  rc |= dalloc(D.cscToCsr, nnz);
  rc |= dalloc(D.basicCount, m);
  if (rc)
    return rc;
  rc |= h2d(dColStart, colStart.data(), n + 1);
  rc |= h2d(dRow, row.data(), nnz);
  rc |= h2d(dElem, elem.data(), nnz);
  rc |= h2d(dRowStart, rowStart.data(), m + 1);
  rc |= h2d(D.ccol, ccol.data(), nnz);
  rc |= h2d(D.relem, relem.data(), nnz);
  rc |= h2d(D.csrToCsc, csrToCsc.data(), nnz);
  rc |= h2d(D.cscToCsr, cscToCsr.data(), nnz);
  D.colStart = dColStart;
  D.row = dRow;
  D.elem = dElem;
  D.rowStart = dRowStart;
  double *dOrigLower, *dOrigUpper;
  rc |= dalloc(D.lower, N);
  rc |= dalloc(D.upper, N);
  rc |= dalloc(D.cost, N);
  rc |= dalloc(D.dj, N);
  rc |= dalloc(D.sol, N);
  rc |= dalloc(dOrigLower, N);
  rc |= dalloc(dOrigUpper, N);
  rc |= dalloc(D.status, N);
  rc |= dalloc(D.pivotVariable, m);
  rc |= dalloc(D.posOfSlack, m);
  rc |= dalloc(D.slotOfRow, m);
  rc |= dalloc(D.slotOfCol, n);
  rc |= dalloc(D.vecC, m);
  rc |= dalloc(D.rho, m);
  rc |= dalloc(D.piNeg, (size_t)m + 2 * PL_MAX_TILE_ROWS);  // zero-padded: k_price_lds reads whole row tiles
  rc |= dalloc(D.piBits, (size_t)(m + 63) / 64 + 8);
  rc |= dalloc(D.tIndex, (size_t)n + 1);
  rc |= dalloc(D.tValue, (size_t)n + 1);
  rc |= dalloc(D.alphaCol, (size_t)n + 8 * 256 + 64);
  rc |= dalloc(D.vecV1, m);
  rc |= dalloc(D.vecV2, m);
  rc |= dalloc(D.w, m);
  rc |= dalloc(D.tau, m);
  rc |= dalloc(D.x3, m);
  rc |= dalloc(D.flipRhs, m);
  rc |= dalloc(D.weights, m);
  rc |= dalloc(D.altWeights, m);
  rc |= dalloc(D.infeas, m);
  rc |= dalloc(D.weightBySeq, N);
  rc |= dalloc(D.savedWeightBySeq, N);
  rc |= dalloc(D.infIndex, m);
  rc |= dalloc(D.candFlag, (size_t)N + 8 * 256 + 64);
  rc |= dalloc(D.candSeq, N);
  rc |= dalloc(D.candAlpha, N);
  rc |= dalloc(D.candTag, N);
  rc |= dalloc(D.candLive, N);
  rc |= dalloc(D.touchCol, (size_t)n + 1);
  rc |= dalloc(D.touchRow, (size_t)n * ROW_SLOTS + 1);
  rc |= dalloc(D.touchVal, (size_t)n * ROW_SLOTS + 1);
  rc |= dalloc(D.candDj, N);
  rc |= dalloc(D.candRange, N);
  rc |= dalloc(D.candBlk, N);
  rc |= dalloc(D.candRk, N);
  rc |= dalloc(D.wsIdxG, DC_WS_CAP);
  rc |= dalloc(D.dcPart, 2 * (size_t)DCW_BLOCKS * DCW_PART);
  rc |= dalloc(D.flipRecMv, FLIP_LIST_CAP);
  rc |= dalloc(D.flipRecObj, FLIP_LIST_CAP);
  rc |= dalloc(D.flipRecStart, FLIP_LIST_CAP);
  rc |= dalloc(D.flipRecLen, FLIP_LIST_CAP);
  // per-workgroup slots of the N-wide kernels (one per 256 keys) -- and of k_ftran_scatter3_lu, whose workgroups hold as few as
  // 64 basis positions each (luLaunchFtran): with n < ~3 m that launch has more workgroups than there are 256-key blocks
  int nb = std::max(cdiv(m, PRICE_BLOCK) + cdiv(n, PRICE_BLOCK), cdiv(m, 64)) + 2;
  rc |= dalloc(D.blockCount, nb);
  rc |= dalloc(D.blockOffset, nb);
  rc |= dalloc(D.classBlock, 3 * (size_t)nb);
  rc |= dalloc(D.blockMin, nb);
  rc |= dalloc(D.blockSum, nb);
  rc |= dalloc(D.flipSeq, N);
  rc |= dalloc(D.flipKey, FLIP_LIST_CAP);
  rc |= dalloc(D.rowDot, 3 * (size_t)m);
  rc |= dalloc(D.flipMv, FLIP_MAX_FLIPS);
  rc |= dalloc(D.appendFlag, m);
  rc |= dalloc(D.appendFlag1, m);
  rc |= dalloc(D.blockOffset1, cdiv(m, 256) + 2);
  rc |= dalloc(D.flipTouch, m);
  rc |= dalloc(D.flipHot, FLIP_HOT_CAP);
  rc |= dalloc(D.flipRowKey, (size_t)m * FLIP_SLOTS);
  rc |= dalloc(D.flipRowVal, (size_t)m * FLIP_SLOTS);
  rc |= dalloc(D.ctrl, 1);
  nChzBlocks = cdiv(m, 256 * CHZ_ITEMS);
  rc |= dalloc(D.chzBest, nChzBlocks);
  rc |= dalloc(D.chzKey, nChzBlocks);
  rc |= dalloc(D.chzRow, nChzBlocks);
  rc |= dalloc(D.chzCnt, nChzBlocks);
  rc |= dalloc(D.normPartial, cdiv(m, 4) + 1);
  rc |= dalloc(dLocalOfRow, m);
  rc |= dalloc(dInfo, 4);
  if (rc)
    return rc;
  D.origLower = dOrigLower;
  D.origUpper = dOrigUpper;
  origLower.resize(N);
  origUpper.resize(N);
  for (int j = 0; j < n; j++) {
    origLower[j] = colLower[j];
    origUpper[j] = colUpper[j];
  }
  for (int i = 0; i < m; i++) {
    origLower[n + i] = rowLower[i];
    origUpper[n + i] = rowUpper[i];
  }
  rc |= h2d(dOrigLower, origLower.data(), N);
  rc |= h2d(dOrigUpper, origUpper.data(), N);
  lower.resize(N);
  upper.resize(N);
  cost.resize(N);
  dj.assign(N, 0.0);
  sol.assign(N, 0.0);
  status.assign(N, 0);
  pivotVariable.assign(m, 0);
  hipError_t e = hipHostMalloc((void **)&hCtrl, sizeof(Ctrl), hipHostMallocDefault);
  if (e != hipSuccess) {
    setError("hipHostMalloc failed: %s", hipGetErrorString(e));
    return -99;
  }
  memset(hCtrl, 0, sizeof(Ctrl));
  hCtrl->pivotRow = hCtrl->sequenceIn = hCtrl->sequenceOut = -1;  // model_->pivotRow() before the first pivot
  if (commActive) {
    applyShard();  // a reload keeps the communicator; the shard follows the new column count
    rc |= allocShardBuffers();
  }
  rc |= buildSell();
  rc |= sync();
  started = false;
  luFillSkip = 0;
  stepPendingRefactor = false;
  return rc;
}

// Everything a load creates is dropped again: device allocations, the pinned control block, the
// captured graph, the solve state.  Options set through clpgpu_set_option survive, so does the
// communicator.
void clpgpu_context::releaseProblem()
{
  if (stream)
    (void)hipStreamSynchronize(stream);
  dropGraph();
  for (void *p : allocations)
    (void)hipFree(p);
  allocations.clear();
  sellBuffers.clear();  // (they were among the allocations; a stale entry could name a later allocation's address)
  jdsReady = false;
  priceMode = 0;
  modePriced = modeDense = 0.0;
  dFreeList = nullptr;  // (was in `allocations`)
  freeActive = graphFreeActive = false;
  for (DBuf &b : luBuf) {
    if (b.p)
      (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
  }
  dLu = nullptr;  // (was in `allocations`)
  hLu = LuDev();
  luActive = luSlotsCleared = false;
  if (hCtrl)
    (void)hipHostFree(hCtrl);
  hCtrl = nullptr;
  memset(&D, 0, sizeof(D));
  kcap = ld = 0;
  logCapacity = 0;
  dKcol = dLocalOfRow = dInfo = nullptr;
  dCandSend = dCandRecv = dFlipSend = dFlipRecv = nullptr;
  shardBuffersOwned = false;
  nLongBlocks = nSellBlocks = nChzBlocks = 0;
  weightsInitialized = false;
  haveStatus = false;
  userStatus.clear();
  perturbationArray.clear();
  started = false;
  needStatus = true;
  rebuildRowCopy = true;
  problemStatus = -1;
  numberIterations = numberRefactorizations = 0;
  pivots = kNucleus = 0;
  objectiveValue = 0.0;
  primalTolerance = optPrimalTolerance;
  dualTolerance = dualToleranceBase = optDualTolerance;
  zeroTolerance = optZeroTolerance;
  dualBound = optDualBound;
  acceptablePivot = optAcceptablePivot;
  scaled = false;
}

// equal, 256-aligned column shards so that rank-major concatenation is the by-column order
// (ABOCA_LITE chunking, src/ClpPackedMatrix.cpp:1823-1854); clp_amd/sharding.py mirrors the formula
void clpgpu_context::applyShard()
{
  int chunk = (n + nranks - 1) / nranks;
  chunk = (chunk + PRICE_BLOCK - 1) / PRICE_BLOCK * PRICE_BLOCK;
  shardChunk = chunk;
  D.priceFirst = std::min(rank * chunk, n);
  D.priceLast = std::min((rank + 1) * chunk, n);
  if (commMode == 2) {
    D.firstColumn = D.priceFirst;  // candidates, dual update and flip test on the own range only
    D.lastColumn = D.priceLast;
  } else {
    D.firstColumn = 0;
    D.lastColumn = n;
  }
}

int clpgpu_context::allocShardBuffers()
{
  // (called again when the exchange buffers grow, EXIT_SHARD_OVERFLOW: the previous ones are freed once the new set is complete;
  // a failure half-way leaves the old pointers and sizes in place for the dense-row fall-back -- ADVICE round 5)
  int *newClass = nullptr;
  double *newCandSend = nullptr, *newCandRecv = nullptr, *newFlipSend = nullptr, *newFlipRecv = nullptr;
  const size_t candRec = SHARD_HDR + 4 * (size_t)shardCandCap, flipRec = SHARD_HDR + 5 * (size_t)shardFlipCap;
  int rc = 0;
  rc |= dalloc(newClass, 3 * (size_t)(cdiv(m, PRICE_BLOCK) + cdiv(n, PRICE_BLOCK) + cdiv(nranks * shardCandCap, PRICE_BLOCK) + 4));
  if (!rc)
    rc |= dalloc(newCandSend, candRec);
  if (!rc)
    rc |= dalloc(newCandRecv, candRec * (size_t)nranks);
  if (!rc)
    rc |= dalloc(newFlipSend, flipRec);
  if (!rc)
    rc |= dalloc(newFlipRecv, flipRec * (size_t)nranks);
  auto release = [&](void *q) {
    if (!q)
      return;
    auto it = std::find(allocations.begin(), allocations.end(), q);
    if (it != allocations.end()) {
      allocations.erase(it);
      (void)hipStreamSynchronize(stream);
      (void)hipFree(q);
    }
  };
  if (rc) {
    release(newClass);
    release(newCandSend);
    release(newCandRecv);
    release(newFlipSend);
    release(newFlipRecv);
    return -99;
  }
  if (shardBuffersOwned) {
    release(D.classBlock);
    release(dCandSend);
    release(dCandRecv);
    release(dFlipSend);
    release(dFlipRecv);
  } else {
    release(D.classBlock);  // the load-time array is replaced by the wider one
  }
  D.classBlock = newClass;
  dCandSend = newCandSend;
  dCandRecv = newCandRecv;
  dFlipSend = newFlipSend;
  dFlipRecv = newFlipRecv;
  shardBuffersOwned = true;
  return 0;
}

int clpgpu_context::checkLaunches(const char *where)
{
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    setError("kernel launch failed in %s: %s", where, hipGetErrorString(e));
    return -99;
  }
  return 0;
}

// Sliced-ELL copy of the priced column range (the coalescing-friendly layout of SURVEY.md 8f.2,
// cf. ClpPackedMatrix3's blocks of equal-length columns, src/ClpPackedMatrix.cpp:6732-7235).
// Columns are sorted by length (stable) so a 64-column slice is padded by ~1% only; the order of
// the entries inside a column is untouched, so the summation order per column is the CSC order.
int clpgpu_context::buildSell()
{
  // (called again by set_option("sell_windows") after the load and by the shard fallback: the previous copies go first)
  for (void *q : sellBuffers) {
    auto it = std::find(allocations.begin(), allocations.end(), q);
    if (it != allocations.end()) {
      allocations.erase(it);
      (void)hipFree(q);
    }
  }
  sellBuffers.clear();
  const int first = D.priceFirst, last = D.priceLast;
  int count = last - first;
  std::vector<int> order(count);
  for (int i = 0; i < count; i++)
    order[i] = first + i;
  // Windowed copy (default; option "sell_windows"): the columns are sorted by length only INSIDE windows of PRICE_BLOCK
  // consecutive keys aligned with the compaction blocks of the N-wide kernels, four slices per window, so that one workgroup
  // of the pricing kernel owns one compaction block: coalesced tableau-row / flag stores and one candidate count per workgroup
  // (priceSellBody).  Costs ~9 % padding on Poisson column counts (the global sort pads ~0 %).
  const bool windowed = sellWindows && first >= D.firstColumn && (first - D.firstColumn) % PRICE_BLOCK == 0;
  const int winBase = windowed ? (first - D.firstColumn) / PRICE_BLOCK : 0;
  std::vector<int> windowLong;
  if (windowed) {
    const int nWin = count > 0 ? cdiv(last - D.firstColumn, PRICE_BLOCK) - winBase : 0;
    std::vector<int> placed((size_t)nWin * PRICE_BLOCK, -1), cols;
    for (int w = 0; w < nWin; w++) {
      const int jlo = std::max(first, D.firstColumn + (winBase + w) * PRICE_BLOCK);
      const int jhi = std::min(last, D.firstColumn + (winBase + w + 1) * PRICE_BLOCK);
      cols.clear();
      for (int j = jlo; j < jhi; j++) {
        if (colStart[j + 1] - colStart[j] > SELL_LONG)
          windowLong.push_back(j);
        else
          cols.push_back(j);
      }
      std::stable_sort(cols.begin(), cols.end(), [&](int a, int b) { return colStart[a + 1] - colStart[a] > colStart[b + 1] - colStart[b]; });
      for (size_t i = 0; i < cols.size(); i++)
        placed[(size_t)w * PRICE_BLOCK + i] = cols[i];
    }
    order.swap(placed);
  }
  // columns by decreasing length
  if (!windowed)
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return colStart[a + 1] - colStart[a] > colStart[b + 1] - colStart[b]; });
  // columns longer than SELL_LONG entries would keep one lane busy for len/8 dependent trips while
  // the rest of the chip waits (power-law column counts): they leave the SELL copy and are priced by
  // a wave each (priceLongBody).  `order` is sorted by decreasing length, so they are its prefix.
  int nLong = 0;
  std::vector<int> longCols;
  const int countAll = count;
  if (windowed) {
    longCols = windowLong;
    nLong = (int)longCols.size();
    count = (int)order.size();  // positions, -1 where a window has fewer than 256 columns of its own
  } else {
    while (nLong < count && colStart[order[nLong] + 1] - colStart[order[nLong]] > SELL_LONG)
      nLong++;
    longCols.assign(order.begin(), order.begin() + nLong);
    order.erase(order.begin(), order.begin() + nLong);
    count = countAll - nLong;
  }
  const int numSlices = cdiv(count, 64);
  std::vector<int> sellStart(numSlices + 1, 0), sellCol((size_t)numSlices * 64, -1), sellLen((size_t)numSlices * 64, 0);
  for (int s = 0; s < numSlices; s++) {
    int maxLen = 0;
    for (int l = 0; l < 64; l++) {
      int i = s * 64 + l;
      if (i < count && order[i] >= 0) {
        int j = order[i];
        sellCol[i] = j;
        sellLen[i] = colStart[j + 1] - colStart[j];
        maxLen = std::max(maxLen, sellLen[i]);
      }
    }
    maxLen = (maxLen + SELL_U - 1) / SELL_U * SELL_U;
    sellStart[s + 1] = sellStart[s] + maxLen * 64;
  }
  const size_t total = (size_t)sellStart[numSlices];
  std::vector<int> sellRow(total ? total : 1, 0);
  std::vector<double> sellElem(total ? total : 1, 0.0);
  for (int s = 0; s < numSlices; s++)
    for (int l = 0; l < 64; l++) {
      int i = s * 64 + l;
      if (i >= count || order[i] < 0)
        continue;
      int j = order[i];
      size_t base = (size_t)sellStart[s] + l;
      for (int p = colStart[j], t = 0; p < colStart[j + 1]; p++, t++) {
        sellRow[base + (size_t)t * 64] = row[p];
        sellElem[base + (size_t)t * 64] = elem[p];
      }
    }
  int *dStart, *dCol, *dLen, *dRow, *dLong;
  double *dElem;
  int rc = 0;
  rc |= dalloc(dLong, (size_t)nLong + 1);
  rc |= dalloc(dStart, numSlices + 1);
  rc |= dalloc(dCol, (size_t)numSlices * 64);
  rc |= dalloc(dLen, (size_t)numSlices * 64);
  rc |= dalloc(dRow, sellRow.size());
  rc |= dalloc(dElem, sellElem.size());
  nSellBlocks = cdiv(numSlices, 4);
  nLongBlocks = nLong;  // one workgroup per long column
  widePricing = countAll > 0 && (double)nnz / (double)n >= 256.0;
  // (whole-matrix figures, so that every rank of a column-sharded run takes the same variants)
  wideRows = m > 0 && (double)colStart[n] / (double)m >= 256.0;
  maxColumnLength = 1;
  for (int j = 0; j < n; j++)
    maxColumnLength = std::max(maxColumnLength, colStart[j + 1] - colStart[j]);
  {
    std::vector<int> rowCount(m, 0);
    for (int p = 0; p < colStart[n]; p++)
      rowCount[row[p]]++;
    int maxRow = 0;
    for (int i = 0; i < m; i++)
      maxRow = std::max(maxRow, rowCount[i]);
    lightRows = (long long)maxRow * 256 <= (long long)n;
  }
  denseColumns = wideRows && (size_t)colStart[n] == (size_t)m * (size_t)n;
  for (int j = 0; j < n && denseColumns; j++)
    for (int p = colStart[j], i = 0; p < colStart[j + 1]; p++, i++)
      if (row[p] != i) {
        denseColumns = false;
        break;
      }
  nWideBlocks = std::min(WIDE_BLOCKS, std::max(1, cdiv(countAll, 4)));
  rc |= dalloc(D.sellMin, std::max(nSellBlocks + nLongBlocks, WIDE_BLOCKS));
  rc |= dalloc(D.sellBytes, std::max(nSellBlocks + nLongBlocks, WIDE_BLOCKS));
  if (rc)
    return rc;
  if (nLong)
    rc |= h2d(dLong, longCols.data(), (size_t)nLong);
  rc |= h2d(dStart, sellStart.data(), numSlices + 1);
  rc |= h2d(dCol, sellCol.data(), sellCol.size());
  rc |= h2d(dLen, sellLen.data(), sellLen.size());
  rc |= h2d(dRow, sellRow.data(), sellRow.size());
  rc |= h2d(dElem, sellElem.data(), sellElem.size());
  rc |= sync();
  D.sellStart = dStart;
  D.sellCol = dCol;
  D.sellLen = dLen;
  D.sellRow = dRow;
  D.sellElem = dElem;
  D.numSlices = numSlices;
  D.sellWindowed = windowed ? 1 : 0;
  D.sellWinBase = winBase;
  D.longCol = dLong;
  D.numLong = nLong;
  for (void *q : {(void *)dLong, (void *)dStart, (void *)dCol, (void *)dLen, (void *)dRow, (void *)dElem})
    sellBuffers.push_back(q);
  jdsReady = false;
  D.jdsWindows = 0;
  if (!rc && windowed && priceLds)
    rc |= buildJds(order, numSlices);
  dropGraph();
  return rc;
}

// The jagged row-tiled layout itself (device_state.h; consumed by k_price_lds), host arithmetic only: `order` holds, per slice
// of 64 positions, the column keys in the slice's home order (-1 none).  false: a column has more than 255 entries in a tile or
// the stream does not fit 31-bit record indices.  Also behind clpgpu_test_jds_layout, which lets a CPU test walk the layout the
// way the kernel does (tests/test_host_logic.py).
static bool jdsLayout(int m, const int *colStart, const int *row, const double *elem, const std::vector<int> &order, int numSlices, int T,
                      int tileRows, std::vector<int> &segStart, std::vector<int> &col, std::vector<unsigned char> &cnt, std::vector<unsigned char> &src,
                      std::vector<unsigned char> &home, std::vector<unsigned> &rowPair, std::vector<double2> &elemPair)
{
  (void)m;
  segStart.assign(numSlices, 0);
  col.assign((size_t)numSlices * 64, -1);
  cnt.assign((size_t)numSlices * T * 64, 0);
  src.assign((size_t)numSlices * T * 64, 0);
  home.assign((size_t)numSlices * 64, 0);
  rowPair.clear();
  elemPair.clear();
  std::vector<int> ord(64), prevPos(64), pos(64), ptr(64), cntHome((size_t)T * 64);
  for (int slice = 0; slice < numSlices; slice++) {
    int h[64];
    for (int l = 0; l < 64; l++) {
      h[l] = order[(size_t)slice * 64 + l];
      col[(size_t)slice * 64 + l] = h[l];
    }
    std::fill(cntHome.begin(), cntHome.end(), 0);
    for (int l = 0; l < 64; l++)
      if (h[l] >= 0)
        for (int p = colStart[h[l]]; p < colStart[h[l] + 1]; p++)
          cntHome[(size_t)(row[p] / tileRows) * 64 + l]++;
    for (int l = 0; l < 64; l++) {
      prevPos[l] = l;
      ptr[l] = h[l] >= 0 ? colStart[h[l]] : 0;
    }
    if (rowPair.size() > 2000000000u)
      return false;
    segStart[slice] = (int)rowPair.size();
    for (int tau = 0; tau < T; tau++) {
      for (int l = 0; l < 64; l++)
        ord[l] = l;
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return cntHome[(size_t)tau * 64 + a] > cntHome[(size_t)tau * 64 + b]; });
      int maxc = 0;
      for (int q = 0; q < 64; q++) {
        const int c = cntHome[(size_t)tau * 64 + ord[q]];
        if (c > 255)
          return false;  // (columns longer than SELL_LONG never get here)
        cnt[((size_t)slice * T + tau) * 64 + q] = (unsigned char)c;
        src[((size_t)slice * T + tau) * 64 + q] = (unsigned char)prevPos[ord[q]];
        pos[ord[q]] = q;
        maxc = std::max(maxc, c);
      }
      for (int t = 0; t < maxc; t += 2)
        for (int q = 0; q < 64; q++) {
          const int l = ord[q], c = cntHome[(size_t)tau * 64 + l];
          if (c > t) {
            const int e = ptr[l] + t;
            unsigned rr = (unsigned)(row[e] - tau * tileRows);
            double2 ee = make_double2(elem[e], 0.0);
            if (c > t + 1) {
              rr |= (unsigned)(row[e + 1] - tau * tileRows) << 16;
              ee.y = elem[e + 1];
            }
            rowPair.push_back(rr);
            elemPair.push_back(ee);
          }
        }
      for (int l = 0; l < 64; l++) {
        ptr[l] += cntHome[(size_t)tau * 64 + l];
        prevPos[l] = pos[l];
      }
    }
    for (int l = 0; l < 64; l++)
      home[(size_t)slice * 64 + l] = (unsigned char)prevPos[l];
  }
  return true;
}
static void jdsTiling(int m, int &T, int &tileRows)
{
  T = cdiv(m, PL_MAX_TILE_ROWS);
  tileRows = (cdiv(m, T) + 127) & ~127;
}

// Jagged row-tiled copy of the windowed SELL slices for k_price_lds (layout: device_state.h; kernel: kernels.hip).
// `order` is buildSell's placement: position w * 256 + i holds the i-th longest column of window w (or -1).
int clpgpu_context::buildJds(const std::vector<int> &order, int numSlices)
{
  const int nWin = (int)(order.size() / PRICE_BLOCK);
  // worth a one-workgroup-per-CU launch only on wide LPs; needs every column's rows in ascending order (tile after tile
  // is then the column's own entry order) and 16-bit tile-local row indices
  if (nWin < priceLdsMinWindows || nWin * 4 != numSlices || m < 4096 || widePricing)
    return 0;
  int T, tileRows;
  jdsTiling(m, T, tileRows);
  if (tileRows > PL_MAX_TILE_ROWS || (size_t)T * tileRows > (size_t)m + 2 * PL_MAX_TILE_ROWS)
    return 0;
  for (int j : order)
    if (j >= 0)
      for (int p = colStart[j] + 1; p < colStart[j + 1]; p++)
        if (row[p] <= row[p - 1])
          return 0;
  std::vector<int> segStart, col;
  std::vector<unsigned char> cnt, src, home;
  std::vector<unsigned> rowPair;
  std::vector<double2> elemPair;
  rowPair.reserve((size_t)nnz / 2 + (size_t)nnz / 16 + 64);
  elemPair.reserve((size_t)nnz / 2 + (size_t)nnz / 16 + 64);
  if (!jdsLayout(m, colStart.data(), row.data(), elem.data(), order, numSlices, T, tileRows, segStart, col, cnt, src, home, rowPair, elemPair))
    return 0;
  rowPair.resize(rowPair.size() + 64, 0u);  // a pair with no active lane reads the record behind the stream
  elemPair.resize(elemPair.size() + 64, make_double2(0.0, 0.0));
  int *dSeg, *dColJ;
  unsigned char *dCnt, *dSrc, *dHome;
  unsigned *dRp;
  double2 *dEp;
  int rc = 0;
  rc |= dalloc(dSeg, segStart.size());
  rc |= dalloc(dColJ, col.size());
  rc |= dalloc(dCnt, cnt.size());
  rc |= dalloc(dSrc, src.size());
  rc |= dalloc(dHome, home.size());
  rc |= dalloc(dRp, rowPair.size());
  rc |= dalloc(dEp, elemPair.size());
  if (rc)
    return rc;
  for (void *q : {(void *)dSeg, (void *)dColJ, (void *)dCnt, (void *)dSrc, (void *)dHome, (void *)dRp, (void *)dEp})
    sellBuffers.push_back(q);
  rc |= h2d(dSeg, segStart.data(), segStart.size());
  rc |= h2d(dColJ, col.data(), col.size());
  rc |= h2d(dCnt, cnt.data(), cnt.size());
  rc |= h2d(dSrc, src.data(), src.size());
  rc |= h2d(dHome, home.data(), home.size());
  rc |= h2d(dRp, rowPair.data(), rowPair.size());
  rc |= h2d(dEp, elemPair.data(), elemPair.size());
  rc |= sync();
  priceLdsBytes = ((size_t)tileRows * 8 + 16 + (PL_THREADS / 256) * 256 * 9 + (PL_THREADS / 64) * 20 + 15) & ~(size_t)15;
  // (the attribute belongs to the kernel, not to this context: always the largest tile any context can ask for, so that a
  // second, smaller LP loaded in the same process does not lower it under a first one's launches)
  const size_t ldsMax = ((size_t)PL_MAX_TILE_ROWS * 8 + 16 + (PL_THREADS / 256) * 256 * 9 + (PL_THREADS / 64) * 20 + 15) & ~(size_t)15;
  if (rc || hipFuncSetAttribute((const void *)k_price_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsMax) != hipSuccess) {
    (void)hipGetLastError();
    return rc;
  }
  D.jdsSegStart = dSeg;
  D.jdsCol = dColJ;
  D.jdsCnt = dCnt;
  D.jdsSrc = dSrc;
  D.jdsHome = dHome;
  D.jdsRowPair = dRp;
  D.jdsElemPair = dEp;
  D.jdsTiles = T;
  D.jdsTileRows = tileRows;
  D.jdsWindows = nWin;
  priceLdsGrid = std::max(1, std::min(priceLdsGridCap, cdiv(nWin, PL_THREADS / 256)));
  jdsReady = true;
  return 0;
}

void clpgpu_context::dropGraph()
{
  if (graphExec)
    (void)hipGraphExecDestroy(graphExec);
  if (graph)
    (void)hipGraphDestroy(graph);
  graphExec = nullptr;
  graph = nullptr;
  for (int i = 0; i < TAIL_SIZES; i++) {
    if (tailExec[i])
      (void)hipGraphExecDestroy(tailExec[i]);
    if (tailGraph[i])
      (void)hipGraphDestroy(tailGraph[i]);
    tailExec[i] = nullptr;
    tailGraph[i] = nullptr;
  }
}

int clpgpu_context::allocNucleus(int kNeeded)
{
  int want = kNeeded + maximumPivots + 16;
  if (want <= kcap)
    return 0;
  // growth never happens inside the iteration loop: k grows by at most one per pivot and a
  // refactorization comes at least every maximumPivots pivots
  kcap = want + want / 4;
  ld = (kcap + 63) & ~63;
  D.ld = ld;
  int rc = 0;
  size_t mat = (size_t)kcap * ld;
  rc |= dalloc(D.Minv, mat);
  rc |= dalloc(D.workW, mat);
  rc |= dalloc(D.workX, mat);
  rc |= dalloc(D.partial, (size_t)(kcap / 64 + 2) * ld);
  rc |= dalloc(D.slotRow, kcap);
  rc |= dalloc(D.slotCol, kcap);
  rc |= dalloc(D.slotPos, kcap);
  rc |= dalloc(D.slotA, (size_t)kcap + 2);  // (+ 2: the tail GEMVs read these as 16-byte pairs)
  rc |= dalloc(D.slotB, kcap);
  rc |= dalloc(D.slotC, kcap);
  rc |= dalloc(D.slotD, kcap);
  rc |= dalloc(D.slotE, kcap);
  rc |= dalloc(D.slotF, kcap);
  rc |= dalloc(D.rhoSlot, kcap);
  rc |= dalloc(D.slotV1, (size_t)kcap + 2);  // (+ 2: the tail GEMVs read these as 16-byte pairs)
  rc |= dalloc(D.rhoSlotF, (size_t)kcap + 2);  // (+ 2: the tail GEMVs read these as 16-byte pairs)
  rc |= dalloc(D.flipSlot, (size_t)kcap + 2);  // (+ 2: the tail GEMVs read these as 16-byte pairs)
  rc |= dalloc(D.perm, kcap);
  rc |= dalloc(D.gjL, (size_t)kcap * GJ_B);
  rc |= dalloc(D.gjU, (size_t)GJ_B * 2 * ld);
  rc |= dalloc(D.gjPiv, GJ_NB);
  rc |= dalloc(D.gjL2, (size_t)kcap * GJ_NB);
  rc |= dalloc(D.gjU2, (size_t)GJ_NB * ld);
  rc |= dalloc(dKcol, kcap);
  dropGraph();
  return rc;
}

int clpgpu_context::pushCtrl()
{
  return h2d(D.ctrl, hCtrl, 1);
}
int clpgpu_context::pullCtrl()
{
  return d2h(hCtrl, D.ctrl, 1);
}
int clpgpu_context::pushRim()
{
  int rc = 0;
  rc |= h2d(D.lower, lower.data(), N);
  rc |= h2d(D.upper, upper.data(), N);
  rc |= h2d(D.cost, cost.data(), N);
  rc |= h2d(D.dj, dj.data(), N);
  rc |= h2d(D.sol, sol.data(), N);
  rc |= h2d(D.status, status.data(), N);
  return rc;
}
int clpgpu_context::pullRim(bool all)
{
  int rc = 0;
  if (all) {
    rc |= d2h(lower.data(), D.lower, N);
    rc |= d2h(upper.data(), D.upper, N);
    rc |= d2h(cost.data(), D.cost, N);
    rc |= d2h(status.data(), D.status, N);
    rc |= d2h(pivotVariable.data(), D.pivotVariable, m);
  }
  rc |= d2h(dj.data(), D.dj, N);
  rc |= d2h(sol.data(), D.sol, N);
  return rc;
}

static inline int getFake(unsigned char s) { return (s >> 3) & 3; }
static inline unsigned char withFake(unsigned char s, int f) { return (unsigned char)((s & ~24) | (f << 3)); }
static inline unsigned char withStatus(unsigned char s, int st) { return (unsigned char)((s & ~7) | st); }

// ---------------------------------------------------------------------------------------------
// Refactorization: ClpFactorization::factorize (src/ClpFactorization.cpp:1649) -- collect basic
// rows then columns, nucleus inversion on the device, pivotVariable from the pivot order.
// ---------------------------------------------------------------------------------------------
// repair = true: a dependent basic structural is replaced by the slack of a row no step pivoted on and
// the factorization is repeated (the loop of ClpFactorization::factorize, src/ClpFactorization.cpp:2382-2532:
// "take out" at the nearer bound, "put in slack"); returns the number of structurals thrown out (>= 0).
// repair = false: -1 on a singular basis, as CoinOtherFactorization::factor reports it.
int clpgpu_context::factorize(bool repair)
{
  int thrownOut = 0;
  for (;;) {
    int rcOnce = factorizeOnce();
    if (rcOnce != -1 || !repair)
      return rcOnce < 0 ? rcOnce : thrownOut;
    // singular at step s: column s of the nucleus (structural lastSingularColumn) depends on the columns
    // before it; lastSingularRow was never a pivot row
    const int seq = lastSingularColumn, iRow = lastSingularRow;
    if (seq < 0 || iRow < 0 || thrownOut > m)
      return -1;
    const double lo = lower[seq], up = upper[seq], value = sol[seq];
    if (lo > -largeValue || up < largeValue) {
      if (fabs(value - lo) < fabs(value - up)) {
        status[seq] = withStatus(status[seq], ST_LOWER);
        sol[seq] = lo;
      } else {
        status[seq] = withStatus(status[seq], ST_UPPER);
        sol[seq] = up;
      }
    } else {
      status[seq] = withStatus(status[seq], ST_UPPER);  // free: fake bounds follow in changeBounds (see startup)
    }
    if (lo == up)
      status[seq] = withStatus(status[seq], ST_FIXED);
    status[n + iRow] = withStatus(withFake(status[n + iRow], FAKE_NONE), ST_BASIC);
    thrownOut++;
    numberThrownOut++;
    rebuildRowCopy = true;
    if (logLevel > 0)
      fprintf(stderr, "clpgpu: singular basis: structural %d out, slack of row %d in\n", seq, iRow);
  }
}

// row copy partition [basic | nonbasic]: rebuilt on the host (refactorization boundary only)
int clpgpu_context::rebuildRowCopyIfNeeded()
{
  int rc = 0;
  if (rebuildRowCopy) {
  // row copy partition [basic | nonbasic]: rebuilt on the host (refactorization boundary only)
    {
      std::vector<int> ccol(nnz), csrToCsc(nnz), cscToCsr(nnz), basicCount(m, 0);
      std::vector<double> relem(nnz);
      std::vector<int> head(rowStart.begin(), rowStart.end() - 1), tail(rowStart.begin() + 1, rowStart.end());
      for (int j = 0; j < n; j++) {
        bool basic = (status[j] & 7) == ST_BASIC;
        for (int p = colStart[j]; p < colStart[j + 1]; p++) {
          int i = row[p];
          int q = basic ? head[i]++ : --tail[i];
          ccol[q] = j;
          relem[q] = elem[p];
          csrToCsc[q] = p;
          cscToCsr[p] = q;
          if (basic)
            basicCount[i]++;
        }
      }
      rc |= h2d(D.ccol, ccol.data(), nnz);
      rc |= h2d(D.relem, relem.data(), nnz);
      rc |= h2d(D.csrToCsc, csrToCsc.data(), nnz);
      rc |= h2d(D.cscToCsr, cscToCsr.data(), nnz);
      rc |= h2d(D.basicCount, basicCount.data(), m);
      rc |= sync();
    }
  
  rebuildRowCopy = false;
  }
  return rc;
}

// the dense inversion shared by the two factorization forms: D.workW (k x k, filled by the caller after
// prepareWork) -> D.Minv, the pivot row of every column in perm; info[0] != 0: singular at step info[0]-1
int clpgpu_context::prepareWork(int k)
{
  int rc = 0;
  std::vector<int> ident(k);
  for (int i = 0; i < k; i++)
    ident[i] = i;
  rc |= h2d(D.perm, ident.data(), k);
  int zero4[4] = { 0, 0, 0, 0 };
  rc |= h2d(dInfo, zero4, 4);
  size_t mat = (size_t)k * ld;
  // (byte counts are size_t: k * ld passes 2^31 doubles' worth of bytes long before it passes INT_MAX entries)
  if (hipMemsetAsync(D.workW, 0, mat * sizeof(double), stream) != hipSuccess ||
      hipMemsetAsync(D.workX, 0, mat * sizeof(double), stream) != hipSuccess) {
    setError("factorize: clearing the work matrices failed");
    return -99;
  }
  return rc;
}
int clpgpu_context::invertWork(int k, std::vector<int> &perm, int info[4])
{
  int rc = 0;
  {
    hipLaunchKernelGGL(k_identity, dim3(cdiv(k, 256)), dim3(256), 0, stream, D, k);
    dim3 g2(cdiv(k, 256), k < 1024 ? k : 1024);
    int mode = refactorMode;
    if (mode < 0)
      mode = (blockedRefactor && k >= refactorMinK) ? 3 : 1;
    if (mode >= 2 && (!blockedRefactor || k > 16384))
      mode = 1;  // the register panels of the two-level form hold up to 16384 rows
    if (mode >= 2) {
      // two-level in-place form (see k_gj2_*): M = workW, identity side implicit
      // (inner panel width: rows per thread x width doubles live in registers -- 8 x 8 at 512 threads up to 4096 rows, 6 x 8 at 1024
      // threads up to 6144 (96 of the 128 VGPRs a wave of a 16-wave workgroup may hold), 8 x 4 up to 8192, 16 x 2 beyond)
      const int bIn = k <= (panel68 ? 6144 : 4096) ? 8 : (k <= 8192 ? 4 : 2);
      for (int I0 = 0; I0 < k; I0 += GJ_NB) {
        const int nb = std::min(GJ_NB, k - I0);
        for (int j0 = 0; j0 < nb; j0 += bIn) {
          const int i0 = I0 + j0, b = std::min(bIn, nb - j0);
          GjOut out{ D.gjL2, GJ_NB, j0, j0 };
          if (k <= 1024)
            hipLaunchKernelGGL((k_gj_panel_reg<2, 8, 512>), dim3(1), dim3(512), 0, stream, D, i0, b, k, dInfo, out);
          else if (k <= 2048)
            hipLaunchKernelGGL((k_gj_panel_reg<4, 8, 512>), dim3(1), dim3(512), 0, stream, D, i0, b, k, dInfo, out);
          else if (k <= 4096)
            hipLaunchKernelGGL((k_gj_panel_reg<8, 8, 512>), dim3(1), dim3(512), 0, stream, D, i0, b, k, dInfo, out);
          else if (k <= 6144 && panel68)
            hipLaunchKernelGGL((k_gj_panel_reg<6, 8, 1024>), dim3(1), dim3(1024), 0, stream, D, i0, b, k, dInfo, out);
          else if (k <= 8192)
            hipLaunchKernelGGL((k_gj_panel_reg<8, 4, 1024>), dim3(1), dim3(1024), 0, stream, D, i0, b, k, dInfo, out);
          else
            hipLaunchKernelGGL((k_gj_panel_reg<16, 2, 1024>), dim3(1), dim3(1024), 0, stream, D, i0, b, k, dInfo, out);
          const int c0 = i0 + b, c1 = I0 + nb;
          if (c1 > c0) {
            // the rest of the outer panel: this inner panel's swaps, pivot-row values, rank-b update
            hipLaunchKernelGGL(k_gj2_rowswaps, dim3(cdiv(c1 - c0, 64)), dim3(64), 0, stream, D, i0, b, dInfo, c0, c1, 0, 0, j0);
            hipLaunchKernelGGL((k_gj2_upanel<8>), dim3(cdiv(c1 - c0, 64)), dim3(64), 0, stream, D, i0, b, dInfo, c0, c1,
                               (const double *)D.gjL2, GJ_NB, j0, D.gjU, GJ_NB);
            hipLaunchKernelGGL((k_gj2_trail_vec<8>), dim3(cdiv(c1 - c0, 64), cdiv(k, GJ_ROWS)), dim3(64), 0, stream, D, i0, b, k, dInfo,
                               c0, c1, (const double *)D.gjL2, GJ_NB, j0, (const double *)D.gjU, GJ_NB);
          }
        }
        // everything outside the outer panel, once per outer block
        hipLaunchKernelGGL(k_gj2_rowswaps, dim3(cdiv(k, 256)), dim3(256), 0, stream, D, I0, nb, dInfo, 0, k, I0, I0 + nb, 0);
        hipLaunchKernelGGL(k_gj2_unit, dim3(cdiv(k * nb, 256)), dim3(256), 0, stream, D, I0, nb, k, dInfo);
        hipLaunchKernelGGL((k_gj2_upanel<GJ_NB>), dim3(cdiv(k, 256)), dim3(256), 0, stream, D, I0, nb, dInfo, 0, k, (const double *)D.gjL2,
                           GJ_NB, 0, D.gjU2, ld);
        if (mode == 3)
          hipLaunchKernelGGL(k_gj2_trail_mfma, dim3(cdiv(k, 64), cdiv(k, 64)), dim3(256), 0, stream, D, I0, nb, k, dInfo,
                             (const double *)D.gjL2, GJ_NB, (const double *)D.gjU2, ld);
        else
          hipLaunchKernelGGL((k_gj2_trail_vec<GJ_NB>), dim3(cdiv(k, 256), cdiv(k, GJ_ROWS)), dim3(256), 0, stream, D, I0, nb, k, dInfo, 0, k,
                             (const double *)D.gjL2, GJ_NB, 0, (const double *)D.gjU2, ld);
      }
      hipLaunchKernelGGL(k_gj2_finish, g2, dim3(256), 0, stream, D, k);
    } else {
    if (blockedRefactor) {
      // panel width: the register-resident panel kernel holds rows-per-thread x width doubles
      const int kp = registerPanel ? k : 1 << 30;  // which panel kernel (and width) this k gets
      const int bs = kp <= 1024 ? 32 : (kp <= 3072 ? 16 : (kp <= 4096 ? 8 : GJ_B));
      for (int i0 = 0; i0 < k; i0 += bs) {
        const int b = std::min(bs, k - i0);
        const int ncols = (k - (i0 + b)) + k;
        if (kp <= 1024)
          hipLaunchKernelGGL((k_gj_panel_reg<2, 32, 512>), dim3(1), dim3(512), 0, stream, D, i0, b, k, dInfo);
        else if (kp <= 2048)
          hipLaunchKernelGGL((k_gj_panel_reg<4, 16, 512>), dim3(1), dim3(512), 0, stream, D, i0, b, k, dInfo);
        else if (kp <= 3072)
          hipLaunchKernelGGL((k_gj_panel_reg<6, 16, 512>), dim3(1), dim3(512), 0, stream, D, i0, b, k, dInfo);
        else if (kp <= 4096)
          hipLaunchKernelGGL((k_gj_panel_reg<8, 8, 512>), dim3(1), dim3(512), 0, stream, D, i0, b, k, dInfo);
        else
          hipLaunchKernelGGL(k_gj_panel, dim3(1), dim3(1024), 0, stream, D, i0, b, k, dInfo);
        hipLaunchKernelGGL(k_gj_rowswaps, dim3(cdiv(ncols, 256)), dim3(256), 0, stream, D, i0, b, k, dInfo);
        hipLaunchKernelGGL(k_gj_upanel, dim3(cdiv(ncols, 256)), dim3(256), 0, stream, D, i0, b, k, dInfo);
        hipLaunchKernelGGL(k_gj_trail, dim3(cdiv(ncols, 256), cdiv(k, GJ_ROWS)), dim3(256), 0, stream, D, i0, b, k, dInfo);
      }
    } else {
      for (int i = 0; i < k; i++) {
        hipLaunchKernelGGL(k_gj_step, dim3(1), dim3(1024), 0, stream, D, i, k, dInfo);
        hipLaunchKernelGGL(k_gj_elim, g2, dim3(256), 0, stream, D, i, k, dInfo);
      }
    }
    hipLaunchKernelGGL(k_gj_finish, g2, dim3(256), 0, stream, D, k);
    }
    if (checkLaunches("factorize"))
      return -99;
    rc |= d2h(info, dInfo, 4);
    if (rc)
      return rc;
    if (!info[0])
      rc |= d2h(perm.data(), D.perm, k);
  }
  return rc;
}

int clpgpu_context::factorizeOnce()
{
  std::vector<int> kcol, rrows, localOfRow(m, -1);
  int numberBasic = 0;
  for (int i = 0; i < m; i++) {
    if ((status[n + i] & 7) == ST_BASIC)
      numberBasic++;
    else {
      localOfRow[i] = (int)rrows.size();
      rrows.push_back(i);
    }
  }
  for (int j = 0; j < n; j++)
    if ((status[j] & 7) == ST_BASIC) {
      kcol.push_back(j);
      numberBasic++;
    }
  if (numberBasic != m || kcol.size() != rrows.size()) {
    setError("factorize: %d basic variables for %d rows", numberBasic, m);
    return -2;
  }
  const int k = (int)kcol.size();
  numberRefactorizations++;
  // what ClpDualRowSteepest::pivotRow will read as factorization()->numberElements() (option steepest_elements 0; 1: set below from
  // the form that was built)
  pendingFactorElements = 0;
  for (int c = 0; c < k; c++)
    pendingFactorElements += colStart[kcol[c] + 1] - colStart[kcol[c]];
  // (column-sharded runs too: the factorization, both solves and the eta file live in row space and are replicated on every rank --
  // only pricing, the candidate and flip lists and the reduced costs of the columns are sharded, and the two exchanges sit between
  // kernels the LU chain does not touch; tests/test_gpu_virtual_ranks.py runs 2 and 4 loopback ranks in LU mode against the
  // unsharded LU-mode engine)
  bool wantLu = k > 0 && (factorMode == 1 || (factorMode < 0 && !wideRows && k >= luMinK));
  // an LP whose front inverses exceeded lu_inverse_fill_cap (factorizeLu -> -7) is not asked again at every refactorization
  // (the front, the tail inversion and the polish would all run before the cap is seen): the next few refactorizations at a
  // similar nucleus size go straight to the explicit inverse
  if (wantLu && luFillSkip > 0 && k < luFillSkipBelowK && factorMode < 0) {
    luFillSkip--;
    wantLu = false;
  }
  if (wantLu != luActive)
    dropGraph();  // the chain of a pivot differs between the two forms
  if (wantLu) {
    const int lrc = factorizeLu(kcol, rrows, localOfRow);
    if (lrc == 0)
      hCtrl->factorElements = steepestElements ? luOwnElements : pendingFactorElements;
    if (lrc != -7)
      return lrc;
    luFillSkip = 8;
    luFillSkipBelowK = k + k / 4 + 64;
    if (luActive)
      dropGraph();
  }
  luActive = false;
  D.luMode = 0;
  luSlotsCleared = false;
  int rc = allocNucleus(k);
  if (rc)
    return rc;
  std::vector<int> perm(k);
  if (k) {
    rc |= h2d(dKcol, kcol.data(), k);
    rc |= h2d(dLocalOfRow, localOfRow.data(), m);
    int info[4];
    rc |= prepareWork(k);
    if (rc)
      return rc;
    hipLaunchKernelGGL(k_gather_nucleus, dim3(k), dim3(64), 0, stream, D, dKcol, dLocalOfRow, k);
    rc = invertWork(k, perm, info);
    if (rc)
      return rc;
    if (info[0]) {
      setError("factorize: singular nucleus at step %d of %d", info[0] - 1, k);
      lastSingularColumn = kcol[info[0] - 1];
      lastSingularRow = (info[2] >= 0 && info[2] < k) ? rrows[info[2]] : -1;
      return -1;
    }
  }
  // bookkeeping arrays
  std::vector<int> posOfSlack(m, -1), slotOfRow(m, -1), slotOfCol(n, -1), slotRow(k), slotCol(k), slotPos(k);
  for (int i = 0; i < m; i++)
    if (localOfRow[i] < 0) {
      posOfSlack[i] = i;
      pivotVariable[i] = n + i;
    }
  for (int c = 0; c < k; c++) {
    int pos = rrows[perm[c]];
    slotCol[c] = kcol[c];
    slotPos[c] = pos;
    slotOfCol[kcol[c]] = c;
    pivotVariable[pos] = kcol[c];
    slotRow[c] = rrows[c];
    slotOfRow[rrows[c]] = c;
  }
  rc |= h2d(D.posOfSlack, posOfSlack.data(), m);
  rc |= h2d(D.slotOfRow, slotOfRow.data(), m);
  rc |= h2d(D.slotOfCol, slotOfCol.data(), n);
  if (k) {
    rc |= h2d(D.slotRow, slotRow.data(), k);
    rc |= h2d(D.slotCol, slotCol.data(), k);
    rc |= h2d(D.slotPos, slotPos.data(), k);
  }
  rc |= h2d(D.pivotVariable, pivotVariable.data(), m);
  rc |= rebuildRowCopyIfNeeded();
  // the slots were renumbered: the basic entries of the row copy carry their column's slot
  hipLaunchKernelGGL(k_cslot_rebuild, dim3(cdiv(m, 256)), dim3(256), 0, stream, D);
  kNucleus = k;
  pivots = 0;
  hCtrl->k = k;
  hCtrl->pivots = 0;
  hCtrl->kcap = kcap;
  hCtrl->factorElements = pendingFactorElements;  // (both settings of steepest_elements: see the option)
  return rc;
}

// generic dense solves (used at refactorization boundaries and by the C-ABI plug-in calls)
int clpgpu_context::ftranDevice(const double *vRow, double *xPos)
{
  if (luActive)
    return luFtran(vRow, nullptr, xPos, nullptr);
  const int k = hCtrl->k;
  if (k) {
    hipLaunchKernelGGL(k_ftran_gather, dim3(cdiv(k, 256)), dim3(256), 0, stream, D, vRow, (const double *)nullptr, D.slotA,
                       (double *)nullptr, 0);
    hipLaunchKernelGGL(k_gemv2, dim3(cdiv(k, 4)), dim3(256), 0, stream, D, (const double *)D.slotA, (const double *)nullptr,
                       D.slotC, (double *)nullptr, 0);
  }
  hipLaunchKernelGGL(k_ftran_scatter, dim3(cdiv(m + k, 256)), dim3(256), 0, stream, D, vRow, (const double *)nullptr,
                     (const double *)D.slotC, (const double *)nullptr, xPos, (double *)nullptr, 0);
  return 0;
}
int clpgpu_context::btranDevice(const double *cPos, double *yRow)
{
  if (luActive)
    return luBtran(cPos, yRow);
  const int k = hCtrl->k;
  hipLaunchKernelGGL(k_btran_slack, dim3(cdiv(m, 256)), dim3(256), 0, stream, D, cPos, yRow, 0);
  if (k) {
    hipLaunchKernelGGL(k_btran_t, dim3(widePricing ? cdiv(k, 4) : cdiv(k, 256)), dim3(256), 0, stream, D, cPos, (const double *)yRow, D.slotA, 0,
                       widePricing ? 1 : 0);
    hipLaunchKernelGGL(k_gemvT_partial, dim3(cdiv(k, 256), cdiv(k, 64)), dim3(256), 0, stream, D, (const double *)D.slotA, 0);
    hipLaunchKernelGGL(k_gemvT_final, dim3(cdiv(k, 256)), dim3(256), 0, stream, D, yRow, 0, 0);
  }
  return 0;
}

// two right-hand sides in one sweep (updateTwoColumnsFT, src/ClpFactorization.cpp:2889)
int clpgpu_context::ftranDevice2(const double *v1Row, const double *v2Row, double *x1Pos, double *x2Pos)
{
  if (luActive)
    return luFtran(v1Row, v2Row, x1Pos, x2Pos);
  const int k = hCtrl->k;
  if (k) {
    hipLaunchKernelGGL(k_ftran_gather, dim3(cdiv(k, 256)), dim3(256), 0, stream, D, v1Row, v2Row, D.slotA, D.slotB, 0);
    hipLaunchKernelGGL(k_gemv2, dim3(cdiv(k, 4)), dim3(256), 0, stream, D, (const double *)D.slotA, (const double *)D.slotB, D.slotC,
                       D.slotD, 0);
  }
  hipLaunchKernelGGL(k_ftran_scatter, dim3(cdiv(m + k, 256)), dim3(256), 0, stream, D, v1Row, v2Row, (const double *)D.slotC,
                     (const double *)D.slotD, x1Pos, x2Pos, 0);
  return 0;
}

// plug-in level calls run single kernels of the iteration chain outside clpgpu_dual: the control block
// must hold the scalars those kernels read (a context that never ran startup() has them at zero)
void clpgpu_context::preparePlugin()
{
  Ctrl *h = hCtrl;
  h->state = RUN;
  h->stepLimit = -1;
  h->pivotRule = pivotRule;
  h->primalTolerance = primalTolerance;
  h->dualTolerance = dualTolerance;
  if (!h->zeroTolerance)
    h->zeroTolerance = zeroTolerance;
  h->dualBound = dualBound;
  h->largeValue = largeValue;
  h->largestPrimalError = largestPrimalError;
  h->largestDualError = largestDualError;
  h->maximumPivots = maximumPivots;
  h->maximumIterations = 2147483647;
  h->lastBadIteration = lastBadIteration;
  h->steepestMode = steepestMode;
  h->chuzrFloor = chuzrFloor;
  h->debugToleranceFactor = debugToleranceFactor;
  h->debugDcTimeoutAt = -1;
  h->acceptablePivotBase = acceptablePivot;
  h->kcap = kcap;
  if (!started) {
    h->seed = seed;
    h->logCapacity = 0;
  }
}

// the plug-in calls take and return vectors in the caller's units; a context that scales internally
// (option "scaling" / clpgpu_set_scales) holds a different matrix, so they are refused there rather
// than answered in the wrong units (a Clp adapter passes its scaledMatrix_ as the problem instead)
bool clpgpu_context::rejectScaled(const char *what)
{
  if (!scaled)
    return false;
  setError("%s: not available on a context that scales internally (plug-in calls work in the caller's units)", what);
  return true;
}

// ClpSimplex::gutsOfSolution (src/ClpSimplex.cpp:574): computePrimals :914, computeDuals :1164 on
// the device, then the infeasibility sums on the host mirrors.
int clpgpu_context::gutsOfSolution()
{
  int rc = pushCtrl();
  const int g = cdiv(m, 256);
  const int gr = wideRows ? cdiv(m, 4) : g;
  // NaN-propagating maximum: a non-finite residual must read as "bad", not as zero
  auto worst = [](double a, double b) { return !(b <= a) ? b : a; };
  auto primalError = [&](double &largest) {
    // largestPrimalError: max |(A x - s)_i| over all rows, reduced per block on the device
    hipLaunchKernelGGL(k_primal_residual, dim3(gr), dim3(256), 0, stream, D, wideRows ? 1 : 0);
    std::vector<double> part(gr);
    int r = d2h(part.data(), D.normPartial, gr);
    largest = 0.0;
    for (int b = 0; b < gr; b++)
      largest = worst(largest, part[b]);
    return r;
  };
  // diagnostics (log_level >= 2): what the incremental updates of the last stretch had drifted to, against the resync below
  std::vector<double> driftDj, driftSol;
  const bool drift = logLevel > 1 && started && numberIterations > 0;
  if (drift) {
    driftDj.resize(N);
    driftSol.resize(N);
    rc |= d2h(driftDj.data(), D.dj, N);
    rc |= d2h(driftSol.data(), D.sol, N);
  }
  hipLaunchKernelGGL(k_zero_basic, dim3(g), dim3(256), 0, stream, D);
  hipLaunchKernelGGL(k_primal_rhs, dim3(gr), dim3(256), 0, stream, D, D.vecV2, wideRows ? 1 : 0);
  ftranDevice(D.vecV2, D.x3);
  hipLaunchKernelGGL(k_store_basic, dim3(g), dim3(256), 0, stream, D, (const double *)D.x3);
  rc |= primalError(largestPrimalError);
  // iterative refinement (ClpSimplex::computePrimals :1040-1130 does the same under numberRefinements_): with the
  // basics in place, s - A x is the residual of B x_B = rhs; one more FTRAN of it is the correction.  Only when the
  // factorization left a residual that matters (ill-conditioned bases); never in the parity tests' regime.
  for (int pass = 0; pass < solutionRefinements && largestPrimalError > refineAbove; pass++) {
    hipLaunchKernelGGL(k_primal_rhs, dim3(gr), dim3(256), 0, stream, D, D.vecV2, wideRows ? 1 : 0);
    ftranDevice(D.vecV2, D.x3);
    hipLaunchKernelGGL(k_add_basic, dim3(g), dim3(256), 0, stream, D, (const double *)D.x3);
    double after = 0.0;
    rc |= primalError(after);
    numberPrimalRefinements++;
    if (!(after < largestPrimalError)) {
      // no better: take the correction back
      hipLaunchKernelGGL(k_sub_basic, dim3(g), dim3(256), 0, stream, D, (const double *)D.x3);
      break;
    }
    largestPrimalError = after;
  }
  hipLaunchKernelGGL(k_basic_costs, dim3(g), dim3(256), 0, stream, D, D.tau);
  btranDevice(D.tau, D.vecV2);
  auto djs = [&]() {
    if (widePricing)
      hipLaunchKernelGGL(k_djs, dim3(cdiv(n, 4) + cdiv(m, 256)), dim3(256), 0, stream, D, (const double *)D.vecV2, 1);
    else
      hipLaunchKernelGGL(k_djs, dim3(cdiv(N, 256)), dim3(256), 0, stream, D, (const double *)D.vecV2, 0);
  };
  djs();
  rc |= pullRim(false);
  auto dualError = [&]() {
    // largestDualError: max over basics of |dj| (should be zero)
    double largest = 0.0;
    for (int i = 0; i < m; i++)
      largest = worst(largest, fabs(dj[pivotVariable[i]]));
    return largest;
  };
  largestDualError = dualError();
  for (int pass = 0; pass < solutionRefinements && largestDualError > refineAbove; pass++) {
    // the basic reduced costs are the residual of B^T y = c_B: y += B^-T (that residual)
    hipLaunchKernelGGL(k_basic_djs, dim3(g), dim3(256), 0, stream, D, D.tau);
    btranDevice(D.tau, D.x3);
    hipLaunchKernelGGL(k_axpy, dim3(g), dim3(256), 0, stream, D.vecV2, (const double *)D.x3, 1.0, m);
    djs();
    rc |= pullRim(false);
    const double after = dualError();
    numberDualRefinements++;
    if (!(after < largestDualError)) {
      hipLaunchKernelGGL(k_axpy, dim3(g), dim3(256), 0, stream, D.vecV2, (const double *)D.x3, -1.0, m);
      djs();
      rc |= pullRim(false);
      break;
    }
    largestDualError = after;
  }
  hipLaunchKernelGGL(k_zero, dim3(g), dim3(256), 0, stream, D.vecV2, m);
  hipLaunchKernelGGL(k_zero, dim3(g), dim3(256), 0, stream, D.tau, m);
  hipLaunchKernelGGL(k_zero, dim3(g), dim3(256), 0, stream, D.x3, m);
  if (drift) {
    // nonbasic reduced costs: largest change, and how many sit on the wrong side of the tolerance only after the resync
    // (those are the ones the next status check flips or shifts); basic primal values: largest absolute and relative change
    double maxDj = 0.0, maxX = 0.0, maxXRel = 0.0, maxAbsX = 0.0;
    int newlyBad = 0;
    for (int i = 0; i < N; i++) {
      const int st = status[i] & 7;
      if (st == ST_BASIC) {
        const double d = fabs(sol[i] - driftSol[i]);
        maxX = worst(maxX, d);
        maxXRel = worst(maxXRel, d / (1.0 + fabs(sol[i])));
        maxAbsX = worst(maxAbsX, fabs(sol[i]));
      } else {
        maxDj = worst(maxDj, fabs(dj[i] - driftDj[i]));
        const double sgn = (st == ST_UPPER) ? -1.0 : ((st == ST_LOWER) ? 1.0 : 0.0);
        if (sgn != 0.0 && sgn * dj[i] < -dualTolerance && !(sgn * driftDj[i] < -dualTolerance))
          newlyBad++;
      }
    }
    fprintf(stderr, "clpgpu: drift at iteration %d: nonbasic dj max %g, %d newly dual infeasible; basic x max %g (relative %g, largest |x| %g)\n",
            numberIterations, maxDj, newlyBad, maxX, maxXRel, maxAbsX);
  }
  if (checkBoth) {
    checkBothSolutions();
  } else {
    if (freeNonbasic)
      noFreeOrSuper = 0;  // "say may be free or superbasic", the old way of src/ClpSimplex.cpp:3228-3234
    checkPrimalSolution();
    checkDualSolution();
  }
  return rc;
}

// ClpSimplex::checkPrimalSolution (src/ClpSimplex.cpp:2989)
void clpgpu_context::checkPrimalSolution()
{
  double relaxedTolerance = primalTolerance + fmin(1.0e-2, largestPrimalError);
  objectiveValue = 0.0;
  sumPrimalInfeasibilities = 0.0;
  numberPrimalInfeasibilities = 0;
  sumOfRelaxedPrimalInfeasibilities = 0.0;
  for (int pass = 0; pass < 2; pass++) {
    int lo = pass ? 0 : n, hi = pass ? n : N;
    for (int i = lo; i < hi; i++) {
      double infeasibility = 0.0;
      objectiveValue += sol[i] * cost[i];
      if (sol[i] > upper[i])
        infeasibility = sol[i] - upper[i];
      else if (sol[i] < lower[i])
        infeasibility = lower[i] - sol[i];
      if (infeasibility > primalTolerance) {
        sumPrimalInfeasibilities += infeasibility - primalTolerance;
        if (infeasibility > relaxedTolerance)
          sumOfRelaxedPrimalInfeasibilities += infeasibility - relaxedTolerance;
        numberPrimalInfeasibilities++;
      }
    }
  }
}

// ClpSimplex::checkDualSolution (src/ClpSimplex.cpp:3070)
void clpgpu_context::checkDualSolution()
{
  double relaxedTolerance = dualTolerance + fmin(1.0e-2, largestDualError);
  const double possTolerance = 5.0 * relaxedTolerance;  // :3093
  bestPossibleImprovement = 0.0;
  sumDualInfeasibilities = 0.0;
  numberDualInfeasibilities = 0;
  sumOfRelaxedDualInfeasibilities = 0.0;
  numberDualInfeasibilitiesWithoutFree = 0;
  // with option free_nonbasic: the isFree branches ("free so relax a lot", :3125-3136), the count without free variables and
  // firstFree_ (:3214-3219)
  const bool withFree = freeNonbasic != 0;
  int firstFreePrimal = -1, firstFreeDual = -1, numberSuperBasicWithDj = 0;
  for (int pass = 0; pass < 2; pass++) {
    int lo = pass ? n : 0, hi = pass ? N : n;
    for (int i = lo; i < hi; i++) {
      if ((status[i] & 7) != ST_BASIC && !(status[i] & FLAGGED_BIT)) {
        double distanceUp = upper[i] - sol[i], distanceDown = sol[i] - lower[i], value = dj[i];
        const bool isFree = withFree && (status[i] & 7) == ST_FREE;
        if (distanceUp > primalTolerance) {
          if (withFree && distanceDown > primalTolerance) {  // check if "free" (:3110-3118)
            if (fabs(value) > 1.0e2 * relaxedTolerance) {
              numberSuperBasicWithDj++;
              if (firstFreeDual < 0)
                firstFreeDual = i;
            }
            if (firstFreePrimal < 0)
              firstFreePrimal = i;
          }
          if (value < 0.0) {
            double v = -value;
            if (v > dualTolerance) {
              if (!(isFree && i < n)) {  // (the row loop has no relaxed form, :3180-3190)
                if (!isFree)
                  numberDualInfeasibilitiesWithoutFree++;
                sumDualInfeasibilities += v - dualTolerance;
                if (v > possTolerance)
                  bestPossibleImprovement += fmin(distanceUp, 1.0e10) * v;
                if (v > relaxedTolerance)
                  sumOfRelaxedDualInfeasibilities += v - relaxedTolerance;
                numberDualInfeasibilities++;
              } else {
                v *= 0.01;  // free so relax a lot
                if (v > dualTolerance) {
                  sumDualInfeasibilities += v - dualTolerance;
                  if (v > possTolerance)
                    bestPossibleImprovement = 1.0e100;
                  if (v > relaxedTolerance)
                    sumOfRelaxedDualInfeasibilities += v - relaxedTolerance;
                  numberDualInfeasibilities++;
                }
              }
            }
          }
        }
        if (distanceDown > primalTolerance && value > 0.0) {
          if (value > dualTolerance) {
            sumDualInfeasibilities += value - dualTolerance;
            if (value > possTolerance)
              bestPossibleImprovement += value * fmin(distanceDown, 1.0e10);
            if (value > relaxedTolerance)
              sumOfRelaxedDualInfeasibilities += value - relaxedTolerance;
            numberDualInfeasibilities++;
            if (!isFree)
              numberDualInfeasibilitiesWithoutFree++;
          }
        }
      }
    }
  }
  if (withFree) {
    if (firstFreeDual >= 0)
      firstFree = firstFreeDual;
    else if (numberSuperBasicWithDj || progIteration[PROGRESS - 1] <= 0)
      firstFree = firstFreePrimal;
  }
}


// ClpSimplex::checkBothSolutions (src/ClpSimplex.cpp:3226-3440): what gutsOfSolution ends in in this reference version (:762) -- one
// pass over [columns | rows] for the objective, the primal infeasibilities and, for feasible nonbasic unflagged variables, the dual
// ones.  Against the older pair it relaxes the dual tolerance by at least 5 x dualTolerance (max(largestDualError, 5 dualTolerance),
// :3255), puts no 1e10 cap on the possible improvement, takes a nonbasic variable strictly between its bounds as "may be free" (dj x 100,
// improvement 1e100, and `value` -- sic, :3358 -- in the relaxed sum), and adds in sequence order.  firstFree_ / the free counters are
// not kept (no free nonbasics on this path).  Option "check_both" 0 restores the pair.
void clpgpu_context::checkBothSolutions()
{
  objectiveValue = 0.0;
  sumPrimalInfeasibilities = 0.0;
  numberPrimalInfeasibilities = 0;
  const double relaxedToleranceP = primalTolerance + fmin(1.0e-2, fmax(largestPrimalError, 0.0 * primalTolerance));
  const double relaxedToleranceD = dualTolerance + fmin(1.0e-2, fmax(largestDualError, 5.0 * dualTolerance));
  const double possTolerance = 5.0 * relaxedToleranceD;
  sumOfRelaxedPrimalInfeasibilities = 0.0;
  sumDualInfeasibilities = 0.0;
  numberDualInfeasibilities = 0;
  sumOfRelaxedDualInfeasibilities = 0.0;
  bestPossibleImprovement = 0.0;
  // option free_nonbasic: moreSpecialOptions_ & 8 (:3273, :3294, :3345), the count without free variables and firstFree_ (:3414-3424)
  int numberDualInfeasibilitiesFree = 0, firstFreePrimal = -1, firstFreeDual = -1, numberSuperBasicWithDj = 0;
  int noneFreeOrSuper = 1;
  for (int i = 0; i < N; i++) {
    const double value = sol[i];
    objectiveValue += value * cost[i];
    const double distanceUp = upper[i] - value, distanceDown = value - lower[i];
    if (distanceUp < -primalTolerance) {
      const double infeasibility = -distanceUp;
      if ((status[i] & 7) != ST_BASIC)
        noneFreeOrSuper = 0;  // say superbasic variables exist (:3294)
      sumPrimalInfeasibilities += infeasibility - primalTolerance;
      if (infeasibility > relaxedToleranceP)
        sumOfRelaxedPrimalInfeasibilities += infeasibility - relaxedToleranceP;
      numberPrimalInfeasibilities++;
    } else if (distanceDown < -primalTolerance) {
      const double infeasibility = -distanceDown;
      if ((status[i] & 7) != ST_BASIC)
        noneFreeOrSuper = 0;
      sumPrimalInfeasibilities += infeasibility - primalTolerance;
      if (infeasibility > relaxedToleranceP)
        sumOfRelaxedPrimalInfeasibilities += infeasibility - relaxedToleranceP;
      numberPrimalInfeasibilities++;
    } else if ((status[i] & 7) != ST_BASIC && !(status[i] & FLAGGED_BIT)) {
      double djValue = dj[i];
      if (distanceDown < primalTolerance) {
        if (distanceUp > primalTolerance && djValue < -dualTolerance) {
          sumDualInfeasibilities -= djValue + dualTolerance;
          if (djValue < -possTolerance)
            bestPossibleImprovement -= distanceUp * djValue;
          if (djValue < -relaxedToleranceD)
            sumOfRelaxedDualInfeasibilities -= djValue + relaxedToleranceD;
          numberDualInfeasibilities++;
        }
      } else if (distanceUp < primalTolerance) {
        if (djValue > dualTolerance) {
          sumDualInfeasibilities += djValue - dualTolerance;
          if (djValue > possTolerance)
            bestPossibleImprovement += distanceDown * djValue;
          if (djValue > relaxedToleranceD)
            sumOfRelaxedDualInfeasibilities += djValue - relaxedToleranceD;
          numberDualInfeasibilities++;
        }
      } else {
        noneFreeOrSuper = 0;  // say free or superbasic (:3345)
        djValue *= 100.0;  // strictly between its bounds: may be free
        if (fabs(djValue) > dualTolerance) {
          if ((status[i] & 7) == ST_FREE)
            numberDualInfeasibilitiesFree++;
          sumDualInfeasibilities += fabs(djValue) - dualTolerance;
          bestPossibleImprovement = 1.0e100;
          numberDualInfeasibilities++;
          if (fabs(djValue) > relaxedToleranceD) {
            sumOfRelaxedDualInfeasibilities += value - relaxedToleranceD;
            numberSuperBasicWithDj++;
            if (firstFreeDual < 0)
              firstFreeDual = i;
            if (firstFreePrimal < 0)
              firstFreePrimal = i;
          }
        } else if ((status[i] & 7) == ST_SUPER && firstFreePrimal < 0) {
          firstFreePrimal = i;
        }
      }
    }
  }
  numberDualInfeasibilitiesWithoutFree = numberDualInfeasibilities;
  if (freeNonbasic) {
    noFreeOrSuper = noneFreeOrSuper;
    numberDualInfeasibilitiesWithoutFree = numberDualInfeasibilities - numberDualInfeasibilitiesFree;
    if (firstFreeDual >= 0)
      firstFree = firstFreeDual;  // dual (:3418)
    else if (numberSuperBasicWithDj || progIteration[PROGRESS - 1] <= 0)
      firstFree = firstFreePrimal;
  }
}

// ClpSimplexDual::changeBounds (src/ClpSimplexDual.cpp:3148-3512) on the host mirrors
int clpgpu_context::changeBounds(int initialize, double &changeCost)
{
  numberFake = 0;
  if (!initialize) {
    int numberInfeasibilities = 0;
    double newBound = 5.0 * dualBound;
    changeCost = 0.0;
    for (int i = 0; i < N; i++) {
      lower[i] = origLower[i];
      upper[i] = origUpper[i];
    }
    for (int i = 0; i < N; i++) {
      double value = sol[i];
      status[i] = withFake(status[i], FAKE_NONE);
      int st = status[i] & 7;
      if (st == ST_UPPER) {
        if (fabs(value - upper[i]) > primalTolerance) {
          if (fabs(dj[i]) > 1.0e-9)
            numberInfeasibilities++;
          else {
            status[i] = withStatus(status[i], ST_SUPER);
            if (freeNonbasic)
              noFreeOrSuper = 0;  // moreSpecialOptions_ &= ~8 (:3181)
          }
        }
      } else if (st == ST_LOWER) {
        if (fabs(value - lower[i]) > primalTolerance) {
          if (fabs(dj[i]) > 1.0e-9)
            numberInfeasibilities++;
          else {
            status[i] = withStatus(status[i], ST_SUPER);
            if (freeNonbasic)
              noFreeOrSuper = 0;  // (:3192)
          }
        }
      }
    }
    if (numberInfeasibilities) {
      for (int i = 0; i < N; i++) {
        double lowerValue = lower[i], upperValue = upper[i], newLowerValue, newUpperValue;
        int st = status[i] & 7;
        if (st == ST_UPPER || st == ST_LOWER) {
          double value = sol[i];
          if (value - lowerValue <= upperValue - value) {
            newLowerValue = fmax(lowerValue, value - 0.666667 * newBound);
            newUpperValue = fmin(upperValue, newLowerValue + newBound);
          } else {
            newUpperValue = fmin(upperValue, value + 0.666667 * newBound);
            newLowerValue = fmax(lowerValue, newUpperValue - newBound);
          }
          if (newLowerValue > lowerValue) {
            if (newUpperValue < upperValue) {
              status[i] = withFake(status[i], FAKE_BOTH);
              if (st == ST_LOWER) {
                newLowerValue = value;
                newUpperValue = fmin(upperValue, newLowerValue + newBound);
              } else {
                newUpperValue = value;
                newLowerValue = fmax(lowerValue, newUpperValue - newBound);
              }
              numberFake++;
            } else {
              status[i] = withFake(status[i], FAKE_LOWER);
              numberFake++;
            }
          } else if (newUpperValue < upperValue) {
            status[i] = withFake(status[i], FAKE_UPPER);
            numberFake++;
          }
          lower[i] = newLowerValue;
          upper[i] = newUpperValue;
          sol[i] = (st == ST_UPPER) ? newUpperValue : newLowerValue;
          double movement = sol[i] - value;
          if (movement)
            changeCost += movement * cost[i];
        }
      }
      dualBound = newBound;
    } else {
      numberInfeasibilities = -1;
    }
    return numberInfeasibilities;
  }
  if (initialize == 3) {
    for (int i = 0; i < N; i++)
      if (getFake(status[i]) != FAKE_NONE) {
        lower[i] = origLower[i];
        upper[i] = origUpper[i];
        status[i] = withFake(status[i], FAKE_NONE);
      }
  }
  double testBound = 0.999999 * dualBound;
  for (int i = 0; i < N; i++) {
    int st = status[i] & 7;
    if (st == ST_UPPER || st == ST_LOWER) {
      double lowerValue = lower[i], upperValue = upper[i], value = sol[i];
      if (lowerValue > -largeValue || upperValue < largeValue) {
        if (fabs(lowerValue - value) <= fabs(upperValue - value)) {
          if (upperValue > lowerValue + testBound) {
            if (getFake(status[i]) == FAKE_NONE)
              numberFake++;
            upper[i] = lowerValue + dualBound;
            status[i] = withFake(status[i], FAKE_UPPER);
          }
        } else {
          if (lowerValue < upperValue - testBound) {
            if (getFake(status[i]) == FAKE_NONE)
              numberFake++;
            lower[i] = upperValue - dualBound;
            status[i] = withFake(status[i], FAKE_LOWER);
          }
        }
        sol[i] = (st == ST_UPPER) ? upper[i] : lower[i];
      } else {
        lower[i] = -0.5 * dualBound;
        upper[i] = 0.5 * dualBound;
        status[i] = withStatus(withFake(status[i], FAKE_BOTH), ST_UPPER);
        numberFake++;
        sol[i] = 0.5 * dualBound;
      }
    } else if (st == ST_BASIC) {
      status[i] = withFake(status[i], FAKE_NONE);
      double gap = upper[i] - lower[i];
      if (gap > 0.5 * dualBound && gap < 2.0 * dualBound) {
        lower[i] = origLower[i];
        upper[i] = origUpper[i];
      }
    }
  }
  return 1;
}

// ClpSimplexDual::resetFakeBounds(1) (src/ClpSimplexDual.cpp:8505-8596): working bounds rebuilt from the
// original ones and the fake-bound flags of the status bytes
// ---------------------------------------------------------------------------------------------
// ClpSimplexProgress for the dual (src/ClpSolve.cpp:4289-4725), host side: one record per status check
// ---------------------------------------------------------------------------------------------
void clpgpu_context::progressReset()  // ::reset :4613, algorithm_ < 0
{
  for (int i = 0; i < PROGRESS; i++) {
    progObjective[i] = -DBL_MAX * 1.0e-50;
    progInfeasibility[i] = -1.0;
    progNumberInfeasibilities[i] = -1;
    progIteration[i] = -1;
  }
  progTimes = progBadTimes = progReallyBadTimes = progTimesFlagged = 0;
}

// ::startCheck :4715 -- the small-cycle detector lives in the control block (ring buffer, kernels.hip cycleStep)
void clpgpu_context::progressStartCheck()
{
  for (int i = 0; i < 12; i++) {
    hCtrl->cycIn[i] = hCtrl->cycOut[i] = -1;
    hCtrl->cycWay[i] = 0;
  }
  hCtrl->cycHead = 0;
}

// ClpSimplexDual::resetFakeBounds(0) :8303-8309: createRim1(false), then changeBounds(3)
void clpgpu_context::resetFakeBoundsToOriginal()
{
  lower = origLower;
  upper = origUpper;
  double dummy = 0.0;
  changeBounds(3, dummy);
}

// ClpDualRowSteepest::looksOptimal (src/ClpDualRowSteepest.cpp:1070); ClpDualRowPivot's own says no (Dantzig)
bool clpgpu_context::looksOptimal() const
{
  if (!pivotRule)
    return false;
  const double tolerance = fmin(1000.0, primalTolerance + fmin(1.0e-2, largestPrimalError));
  for (int iRow = 0; iRow < m; iRow++) {
    const int iPivot = pivotVariable[iRow];
    if (sol[iPivot] < lower[iPivot] - tolerance || sol[iPivot] > upper[iPivot] + tolerance)
      return false;
  }
  return true;
}

// ::looping :4438-4611 for the dual.  -1 carry on; -2 something was changed (tolerance, dual bound, a flag: the host rim
// is then ahead of the device); 0 declare victory; 3 / 4 give up.
int clpgpu_context::progressLooping()
{
  const double objective = objectiveValue - bestPossibleImprovement, infeasibility = sumPrimalInfeasibilities;
  const int count = numberPrimalInfeasibilities;
  auto sameBits = [](double a, double b) { return memcmp(&a, &b, sizeof(double)) == 0; };  // equalDouble :4424
  int numberMatched = 0, matched = 0, same = 0;
  for (int i = 0; i < PROGRESS; i++) {
    if (sameBits(objective, progObjective[i]) && sameBits(infeasibility, progInfeasibility[i]) && count == progNumberInfeasibilities[i]) {
      matched |= 1 << i;
      if (numberIterations != progIteration[i])
        numberMatched++;
      else
        same++;
    }
    if (i) {
      progObjective[i - 1] = progObjective[i];
      progInfeasibility[i - 1] = progInfeasibility[i];
      progNumberInfeasibilities[i - 1] = progNumberInfeasibilities[i];
      progIteration[i - 1] = progIteration[i];
    }
  }
  progObjective[PROGRESS - 1] = objective;
  progInfeasibility[PROGRESS - 1] = infeasibility;
  progNumberInfeasibilities[PROGRESS - 1] = count;
  progIteration[PROGRESS - 1] = numberIterations;
  if (same == PROGRESS)
    numberMatched = PROGRESS;
  if (progressFlag & 3)
    numberMatched = 0;
  progTimes++;
  if (progTimes < 10)
    numberMatched = 0;
  if (matched == (1 << (PROGRESS - 1)))
    numberMatched = 0;
  if (!numberMatched)
    return -1;
  numberLoopFlags++;
  progBadTimes++;
  if (progBadTimes >= 10)
    return infeasibility < 1.0e-4 ? 0 : 3;
  forceFactorization = 1;
  if (progBadTimes < 2) {
    progressStartCheck();
    dualTolerance *= 1.05;
    if (dualBound < 1.0e17) {
      dualBound *= 1.1;
      resetFakeBoundsToOriginal();
    }
  } else {
    if (dualBound > 1.0e14)
      dualBound = 1.0e14;
    const int newest = hCtrl->cycIn[(hCtrl->cycHead + 11) % 12];  // in_[CLP_CYCLE - 1]
    if (newest < 0)
      return 4;
    status[newest] |= FLAGGED_BIT;
    progressStartCheck();
    progBadTimes = 2;
  }
  return -2;
}

// ClpSimplexDual::perturb (src/ClpSimplexDual.cpp:6533): the arithmetic is in perturb_host.h; the caller pushes the costs
int clpgpu_context::perturb()
{
  PerturbRim rim{m, n, colStart.data(), elem.data(), lower.data(), upper.data(), status.data(), origObj.data(), dualTolerance, largeValue,
                 numberIterations};
  const int rc = perturbCosts(rim, perturbation, perturbationArray, seed, cost.data());
  if (perturbation == 101)
    numberPerturbations++;
  return rc;
}

// ClpSimplex::createRim4(false) (src/ClpSimplex.cpp:4645): the costs of the problem back in the rim
void clpgpu_context::restoreCosts()
{
  for (int j = 0; j < n; j++)
    cost[j] = obj[j];
  for (int i = 0; i < m; i++)
    cost[n + i] = 0.0;
}

void clpgpu_context::resetFakeBounds()
{
  lower = origLower;
  upper = origUpper;
  numberFake = 0;
  for (int i = 0; i < N; i++) {
    const int fake = getFake(status[i]);
    if (fake == FAKE_NONE)
      continue;
    const int st = status[i] & 7;
    if (st == ST_BASIC || st == ST_FIXED) {
      status[i] = withFake(status[i], FAKE_NONE);
      continue;
    }
    const double lowerValue = lower[i], upperValue = upper[i], value = sol[i];
    numberFake++;
    if (fake == FAKE_UPPER) {
      upper[i] = lowerValue + dualBound;
      sol[i] = (st == ST_LOWER) ? lowerValue : upper[i];
    } else if (fake == FAKE_LOWER) {
      lower[i] = upperValue - dualBound;
      sol[i] = (st == ST_LOWER) ? lower[i] : upperValue;
    } else if (st == ST_LOWER) {
      lower[i] = value;
      upper[i] = value + dualBound;
    } else if (st == ST_UPPER) {
      upper[i] = value;
      lower[i] = value - dualBound;
    } else {
      lower[i] = value - 0.5 * dualBound;
      upper[i] = value + 0.5 * dualBound;
    }
  }
}

int clpgpu_context::numberAtFakeBound() const
{
  int count = 0;
  for (int i = 0; i < N; i++) {
    int f = getFake(status[i]), st = status[i] & 7;
    if (st == ST_UPPER && (f == FAKE_UPPER || f == FAKE_BOTH))
      count++;
    else if (st == ST_LOWER && (f == FAKE_LOWER || f == FAKE_BOTH))
      count++;
  }
  return count;
}

// ClpSimplexDual::updateDualsInDual(..., fullRecompute = true) (:2648-2868) + flipBounds, on the
// host mirrors: only runs at refactorization boundaries, the basic solution is recomputed on the
// device afterwards, so the rhs movement it would produce is not needed.
int clpgpu_context::updateDualsFullRecompute()
{
  double tolerance = dualTolerance + fmin(1.0e-2, largestDualError);
  int numberFlips = 0;
  for (int i = 0; i < N; i++) {
    double value = dj[i];
    int st = status[i] & 7;
    if (st == ST_UPPER) {
      if (value > tolerance) {
        double movement = lower[i] - upper[i];
        if (fabs(movement) > dualBound && getFake(status[i]) == FAKE_NONE) {
          status[i] = withFake(status[i], FAKE_LOWER);
          lower[i] = upper[i] - dualBound;
          numberFake++;
        }
        status[i] = withStatus(status[i], ST_LOWER);
        sol[i] = lower[i];
        numberFlips++;
      } else if (value > -tolerance && getFake(status[i]) == FAKE_UPPER) {
        status[i] = withStatus(status[i], ST_LOWER);
        sol[i] = lower[i];
      }
    } else if (st == ST_LOWER) {
      if (value < -tolerance) {
        double movement = upper[i] - lower[i];
        if (fabs(movement) > dualBound && getFake(status[i]) == FAKE_NONE) {
          status[i] = withFake(status[i], FAKE_UPPER);
          upper[i] = lower[i] + dualBound;
          numberFake++;
        }
        status[i] = withStatus(status[i], ST_UPPER);
        sol[i] = upper[i];
        numberFlips++;
      } else if (value < tolerance && getFake(status[i]) == FAKE_LOWER) {
        status[i] = withStatus(status[i], ST_UPPER);
        sol[i] = upper[i];
      }
    }
  }
  return numberFlips;
}

// ClpDualRowSteepest::saveWeights (src/ClpDualRowSteepest.cpp:773): 1 = before factorization,
// 2/4 = after (restore by sequence, rebuild infeasibilities), 3 = just redo infeasibilities
int clpgpu_context::saveWeights(int mode)
{
  const int g = cdiv(m, 256);
  if (mode == 1) {
    if (pivotRule && weightsInitialized) {
      hipLaunchKernelGGL(k_fill, dim3(cdiv(N, 256)), dim3(256), 0, stream, D.weightBySeq, -1.0, N);
      hipLaunchKernelGGL(k_weights_to_seq, dim3(g), dim3(256), 0, stream, D, D.weightBySeq);
    }
    return 0;
  }
  if (!pivotRule)
    return 0;
  if (mode == 2 || mode == 4) {
    // :875-915.  Mode 2 keeps what mode 1 recorded (sequence -> weight of the basis being left) as savedWeights_ and maps it
    // onto the new basis; mode 4 maps the copy kept by the LAST mode 2 -- the weights that went with the basis a restore
    // (singular factorization, objective going backwards) comes back to -- and leaves that copy alone.
    if (!weightsInitialized) {
      hipLaunchKernelGGL(k_weights_from_seq, dim3(g), dim3(256), 0, stream, D, (const double *)D.weightBySeq, 1);
      hipLaunchKernelGGL(k_fill, dim3(cdiv(N, 256)), dim3(256), 0, stream, D.savedWeightBySeq, -1.0, N);
      hipLaunchKernelGGL(k_weights_to_seq, dim3(g), dim3(256), 0, stream, D, D.savedWeightBySeq);
    } else if (mode == 2) {
      (void)hipMemcpyAsync(D.savedWeightBySeq, D.weightBySeq, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, stream);
      hipLaunchKernelGGL(k_weights_from_seq, dim3(g), dim3(256), 0, stream, D, (const double *)D.weightBySeq, 0);
    } else {
      hipLaunchKernelGGL(k_weights_from_seq, dim3(g), dim3(256), 0, stream, D, (const double *)D.savedWeightBySeq, 0);
    }
    weightsInitialized = true;
  } else if (mode == 5 || mode == 7) {
    // strong branching entry (:786, :815-822): weights start again at 1.0 (the default steepest mode
    // 3 does not compute full norms here, `mode_ != 1`), infeasibilities rebuilt below
    hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, stream, D.weights, 1.0, m);
    weightsInitialized = true;
  } else if (mode == 6) {
    // scale back the weights as the primal error grows (:937-957)
    double allowed = largestPrimalError > 1.0e3 ? 10.0 : (largestPrimalError > 1.0e2 ? 50.0 : (largestPrimalError > 1.0e1 ? 100.0 : 1000.0));
    hipLaunchKernelGGL(k_weights_scale_back, dim3(g), dim3(256), 0, stream, D, allowed);
  }
  // rebuild the list in ascending position order
  hCtrl->numberInfeasible = 0;
  int rc = pushCtrl();
  hipLaunchKernelGGL(k_infeas_flags, dim3(g), dim3(256), 0, stream, D);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, stream, D, g, 2, 0, 0);
  hipLaunchKernelGGL(k_append_scatter, dim3(g), dim3(256), 0, stream, D, 0, 0);
  hipLaunchKernelGGL(k_infeas_finish, dim3(1), dim3(1), 0, stream, D);
  rc |= pullCtrl();
  return rc;
}

// ---------------------------------------------------------------------------------------------
// startup: ClpSimplexDual::startupSolve (:230) -> ClpSimplex::startup (src/ClpSimplex.cpp:9430):
// createRim, all-slack basis unless a status array was given, factorize, changeBounds(1),
// gutsOfSolution.
// ---------------------------------------------------------------------------------------------
int clpgpu_context::startup()
{
  for (int j = 0; j < n; j++) {
    lower[j] = colLower[j];
    upper[j] = colUpper[j];
    cost[j] = obj[j];
  }
  for (int i = 0; i < m; i++) {
    lower[n + i] = rowLower[i];
    upper[n + i] = rowUpper[i];
    cost[n + i] = 0.0;
  }
  // the bounds may have been replaced since the load (clpgpu_chg_*): the "original bound" copies
  // that changeBounds / the fake-bound logic restore from follow them, on the host and on the device
  origLower = lower;
  origUpper = upper;
  if (h2d(const_cast<double *>(D.origLower), origLower.data(), N) || h2d(const_cast<double *>(D.origUpper), origUpper.data(), N))
    return -99;
  rimInfeasible = false;
  {
    // ClpSimplex::sanityCheck (src/ClpSimplex.cpp:7645-7790), the bound part, as createRim(63) runs it at start-up (:4270):
    // bounds that cross by more than the primal tolerance make the problem infeasible before anything is solved (status 1,
    // :7773-7780 -- what a branch that empties a variable's range looks like); bounds closer than the tolerance are made
    // equal in the working copy (:7695-7700)
    double fixTolerance = primalTolerance;
    if (fixTolerance < 2.0e-8)
      fixTolerance *= 1.1;
    int numberBad = 0;
    for (int i = 0; i < N; i++) {
      const double gap = upper[i] - lower[i];
      if (gap < -primalTolerance)
        numberBad++;
      else if (gap <= fixTolerance && gap != 0.0)
        upper[i] = lower[i];
    }
    if (numberBad) {
      rimInfeasible = true;
      problemStatus = 1;
      numberIterations = numberRefactorizations = 0;
      objectiveValue = 0.0;
      return 0;
    }
  }
  if (haveStatus) {
    status = userStatus;
  } else {
    // ClpSimplex::allSlackBasis (src/ClpSimplex.cpp:7831)
    for (int i = 0; i < m; i++)
      status[n + i] = ST_BASIC;
    for (int j = 0; j < n; j++) {
      if (colLower[j] >= 0.0)
        status[j] = ST_LOWER;
      else if (colUpper[j] <= 0.0)
        status[j] = ST_UPPER;
      else if (colLower[j] < -1.0e20 && colUpper[j] > 1.0e20)
        status[j] = freeNonbasic ? ST_FREE : ST_UPPER;  // free: the reference's isFree (allSlackBasis :7846-7849) with option free_nonbasic, else
                                                        // given bothFake bounds by changeBounds(1)
      else if (fabs(colLower[j]) < fabs(colUpper[j]))
        status[j] = ST_LOWER;
      else
        status[j] = ST_UPPER;
    }
  }
  for (int i = 0; i < N; i++) {
    int st = status[i] & 7;
    status[i] = (unsigned char)st;
    if (st == ST_LOWER || st == ST_FIXED)
      sol[i] = lower[i];
    else if (st == ST_UPPER)
      sol[i] = upper[i];
    else if (st != ST_BASIC)
      sol[i] = 0.0;
    if (st != ST_BASIC && lower[i] == upper[i])
      status[i] = withStatus(status[i], ST_FIXED);
    if (st == ST_LOWER && lower[i] < -1.0e20 && upper[i] < 1.0e20) {
      status[i] = withStatus(status[i], ST_UPPER);
      sol[i] = upper[i];
    } else if (st == ST_UPPER && upper[i] > 1.0e20 && lower[i] > -1.0e20) {
      status[i] = withStatus(status[i], ST_LOWER);
      sol[i] = lower[i];
    }
    if (freeNonbasic && (st == ST_LOWER || st == ST_UPPER) && lower[i] < -1.0e20 && upper[i] > 1.0e20) {
      status[i] = withStatus(status[i], ST_FREE);  // createRim's clean-up of a caller's basis, src/ClpSimplex.cpp:4317-4338
      sol[i] = 0.0;
    }
    dj[i] = 0.0;
  }
  noFreeOrSuper = 1;
  firstFree = -1;
  numberDualInfeasibilitiesWithoutFree = 0;
  numberFreeFirstRows = numberFreeEntered = 0;
  problemStatus = -1;
  numberIterations = 0;
  numberRefactorizations = 0;
  numberRefreshes = numberRefreshesRejected = consecutiveRefreshes = 0;
  numberFake = numberChanged = numberTimesOptimal = 0;
  // ClpDataSave: every dual() is entered with the caller's perturbation_; fastDual (:7260) does not perturb at all
  perturbation = fastDualMode ? 102 : perturbationOption;
  numberPerturbations = 0;
  progressReset();  // ClpSimplex::saveData -> progress_.fillFromModel (src/ClpSimplex.cpp:9732)
  progressFlag = 0;  // :461
  bestPossibleImprovement = 0.0;
  numberBackwards = numberLoopFlags = numberAccuracyRestores = numberSingularRestores = 0;
  numberTryPrimal = 0;
  smallestPrimalInfeasibility = DBL_MAX;  // gutsOfDual :442
  lastObjectiveValueGuts = -1.0e100;      // :460
  forceFactorization = -1;
  lastBadIteration = debugLastBadIteration;  // -999999 unless option debug_last_bad_iteration (fault injection for CHUZR's changed tolerance)
  lastCleaned = 0;
  factorType = 0;
  weightsInitialized = false;
  largestPrimalError = largestDualError = 0.0;
  objectiveValue = 0.0;
  dualTolerance = dualToleranceBase;
  // a previous solve on this context may have left 5 x dualBound (changeBounds) or a relaxed
  // acceptablePivot (the "no incoming" exit): every dual() starts from the caller's values, the
  // effect of ClpDataSave around ClpSimplex::dual (src/ClpSimplex.cpp:5690-5697, :5918)
  dualBound = optDualBound;
  acceptablePivot = optAcceptablePivot;
  memset(&stats, 0, sizeof(stats));
  if (logCapacity < 65536) {
    logCapacity = 1 << 20;
    int rc = dalloc(D.log, logCapacity);
    if (rc)
      return rc;
    dropGraph();
  }
  // control block
  memset(hCtrl, 0, sizeof(Ctrl));
  hCtrl->state = EXIT_REFACTOR;
  hCtrl->pivotRow = -1;
  hCtrl->sequenceIn = hCtrl->sequenceOut = -1;
  for (int i = 0; i < 12; i++)  // progress_.startCheck() (ClpSimplexDual.cpp:452)
    hCtrl->cycIn[i] = hCtrl->cycOut[i] = -1;
  hCtrl->cycHead = 0;
  hCtrl->maximumPivots = maximumPivots;
  hCtrl->maximumIterations = maximumIterations;
  hCtrl->forceFactorization = -1;
  hCtrl->stepLimit = -1;
  hCtrl->logCapacity = logCapacity;
  hCtrl->pivotRule = pivotRule;
  hCtrl->steepestMode = steepestMode;
  hCtrl->chuzrFloor = chuzrFloor;
  hCtrl->debugToleranceFactor = debugToleranceFactor;
  hCtrl->debugDcTimeoutAt = debugDcTimeoutAt;
  hCtrl->lastBadIteration = lastBadIteration;
  hCtrl->seed = seed;
  hCtrl->acceptablePivotBase = acceptablePivot;
  hCtrl->primalTolerance = primalTolerance;
  hCtrl->dualTolerance = dualTolerance;
  hCtrl->zeroTolerance = zeroTolerance;
  hCtrl->dualBound = dualBound;
  hCtrl->largeValue = largeValue;
  rebuildRowCopy = true;
  int rc = pushCtrl();
  haveSnapshot = false;
  numberThrownOut = 0;
  {
    int frc = factorize(true);  // a singular starting basis is repaired (ClpSimplex::internalFactorize)
    if (frc < 0)
      rc |= frc;
  }
  if (rc) {
    problemStatus = 4;
    return rc;
  }
  double dummy = 0.0;
  changeBounds(1, dummy);
  rc |= pushRim();
  rc |= gutsOfSolution();
  // startupSolve :330-336: "problemStatus_ = 0" when the starting basis is already optimal comes FIRST; the costs are
  // perturbed only `if (problemStatus_ < 0 && perturbation_ < 100)` -- a warm-started optimal basis returns with no
  // iterations and untouched costs
  const bool optimalAtStart = !numberDualInfeasibilities && !numberPrimalInfeasibilities;
  if (perturbation < 100 && !optimalAtStart) {
    // startupSolve :335-341.  perturb() == 1 ("safer to use primal": every cost is zero) is only a hint to
    // callers that hold a primal; dual carries on unperturbed.
    perturb();
    rc |= h2d(D.cost, cost.data(), N);
    rc |= gutsOfSolution();
  }
  if (!numberDualInfeasibilities && !numberPrimalInfeasibilities && perturbation < 101)
    problemStatus = 0;  // ClpSimplexDual::dual :664-666: nothing to do
  started = true;
  needStatus = true;
  return rc;
}

// ---------------------------------------------------------------------------------------------
// ClpSimplexDual::statusOfProblemInDual (:4996-6343), host control plane; arithmetic on the
// device (factorize, gutsOfSolution).  Restates the branches that can be reached without
// perturbation, values pass, Cbc options or primal fall-back.
// ---------------------------------------------------------------------------------------------
// see the members: keep the inverse at a scheduled refactorization?
bool clpgpu_context::refreshEligible()
{
  if (refreshMinK <= 0 || !blockedRefactor || rebuildRowCopy || forceFactorization == 1 || !hCtrl || luActive)
    return false;
  // only a SCHEDULED refactorization may keep the inverse: an exit raised by the numerics (alpha mismatch, bad
  // update, objective going backwards, no pivot row / no entering column) asks for a clean factorization, as
  // the reference gives it (ClpSimplexDual.cpp:1451-1500, :1574, :1618); nor in column-sharded runs, where every
  // rank would have to take the same accept / reject decision
  if (lastExitState != EXIT_REFACTOR || commActive)
    return false;
  const int k = hCtrl->k;
  // (long rows: the re-inversion is a larger share of a pivot's cost already at a few thousand, and both halves
  // of the step are GEMMs there; no parity test solves a dense LP with a nucleus beyond a few hundred)
  const int minK = wideRows ? std::min(refreshMinK, refreshMinKDense) : refreshMinK;
  if (k < minK || consecutiveRefreshes >= refreshMax)
    return false;
  return k + maximumPivots + 16 <= kcap;  // room for the growth until the next one
}
// rocBLAS, loaded when a nucleus first asks for the GEMM of the Newton-Schulz step
static void *rocblasLibrary()
{
  static void *h = nullptr;
  if (!h)
    h = dlopen("librocblas.so.5", RTLD_NOW | RTLD_LOCAL);
  if (!h)
    h = dlopen("librocblas.so", RTLD_NOW | RTLD_LOCAL);
  return h;
}
// C = beta C + alpha A B (n x n, row-major, leading dimension ld) on the engine's own MFMA kernel
int clpgpu_context::dgemmDevice(int nn, double alpha, const double *A, const double *B, double beta, double *C)
{
  hipLaunchKernelGGL(k_dgemm, dim3(cdiv(nn, DG_T), cdiv(nn, DG_T)), dim3(256), 0, stream, nn, nn, nn, alpha, A, ld, B, ld, beta, C, ld);
  return 0;
}

int clpgpu_context::refineInverse(bool tailFromS, double goodBelow)
{
  if (!gemmBackend) {
    // the engine's own GEMM (default): no library involved
    const int k = tailFromS ? hLu.k2 : hCtrl->k;
    const size_t mat = (size_t)k * ld * sizeof(double);
    unsigned long long *dMax = (unsigned long long *)D.normPartial;
    if (hipMemsetAsync(dMax, 0, sizeof(unsigned long long), stream) != hipSuccess)
      return 1;
    if (wideRows || tailFromS) {
      if (hipMemsetAsync(D.workW, 0, mat, stream) != hipSuccess || hipMemsetAsync(D.workX, 0, mat, stream) != hipSuccess)
        return 1;
      if (tailFromS) {
        const int snz = (int)luF.sVal.size();
        if (snz)
          hipLaunchKernelGGL(k_lu_scatter_tail, dim3(cdiv(snz, 256)), dim3(256), 0, stream, D, (const int *)luBuf[LB_SROW].p,
                             (const int *)luBuf[LB_SCOL].p, (const double *)luBuf[LB_SVAL].p, snz);
      } else {
        hipLaunchKernelGGL(k_gather_slots, dim3(k), dim3(64), 0, stream, D, k, D.workW);
      }
      hipLaunchKernelGGL(k_identity, dim3(cdiv(k, 256)), dim3(256), 0, stream, D, k);
      dgemmDevice(k, -1.0, D.workW, D.Minv, 1.0, D.workX);  // R = I - C X
      hipLaunchKernelGGL(k_absmax_rows, dim3(k), dim3(256), 0, stream, D, k, (const double *)D.workX, dMax);
    } else {
      hipLaunchKernelGGL(k_refine_residual, dim3(k), dim3(256), 0, stream, D, k, D.workX, dMax);
    }
    unsigned long long bits = 0;
    if (checkLaunches("refineInverse") || d2h(&bits, dMax, 1))
      return 1;
    memcpy(&lastResidual, &bits, sizeof(double));
    if (!(lastResidual <= (tailFromS ? 0.5 : refreshResidualMax)))
      return 2;  // too far for one step (or not finite): re-invert
    if (lastResidual < goodBelow)
      return 3;  // nothing to gain
    // X <- X + X R  (into workW, then back)
    if (hipMemcpyAsync(D.workW, D.Minv, mat, hipMemcpyDeviceToDevice, stream) != hipSuccess)
      return 1;
    dgemmDevice(k, 1.0, D.Minv, D.workX, 1.0, D.workW);
    if (hipMemcpyAsync(D.Minv, D.workW, mat, hipMemcpyDeviceToDevice, stream) != hipSuccess)
      return 1;
    numberRefines++;
    return checkLaunches("refineInverse (step)") ? 1 : 0;
  }
  typedef int (*create_t)(void **);
  typedef int (*stream_t)(void *, hipStream_t);
  typedef int (*dgemm_t)(void *, int, int, int, int, int, const double *, const double *, int, const double *, int, const double *, double *, int);
  void *lib = rocblasLibrary();
  if (!lib)
    return 1;
  static create_t createFn = (create_t)dlsym(lib, "rocblas_create_handle");
  static stream_t streamFn = (stream_t)dlsym(lib, "rocblas_set_stream");
  static dgemm_t dgemmFn = (dgemm_t)dlsym(lib, "rocblas_dgemm");
  if (!createFn || !streamFn || !dgemmFn)
    return 1;
  if (!blasHandle) {
    if (createFn(&blasHandle) != 0 || streamFn(blasHandle, stream) != 0) {
      blasHandle = nullptr;
      return 1;
    }
  }
  const int k = tailFromS ? hLu.k2 : hCtrl->k;
  const size_t mat = (size_t)k * ld * sizeof(double);
  unsigned long long *dMax = (unsigned long long *)D.normPartial;
  if (hipMemsetAsync(dMax, 0, sizeof(unsigned long long), stream) != hipSuccess)
    return 1;
  const double one = 1.0, minusOne = -1.0;
  const int none = 111;  // rocblas_operation_none
  if (wideRows || tailFromS) {
    // long rows: C gathered dense, R = I - C X as a GEMM (row-major C X = column-major X^T C^T: A = X, B = C)
    if (hipMemsetAsync(D.workW, 0, mat, stream) != hipSuccess || hipMemsetAsync(D.workX, 0, mat, stream) != hipSuccess)
      return 1;
    if (tailFromS) {
      const int snz = (int)luF.sVal.size();
      if (snz)
        hipLaunchKernelGGL(k_lu_scatter_tail, dim3(cdiv(snz, 256)), dim3(256), 0, stream, D, (const int *)luBuf[LB_SROW].p,
                           (const int *)luBuf[LB_SCOL].p, (const double *)luBuf[LB_SVAL].p, snz);
    } else {
      hipLaunchKernelGGL(k_gather_slots, dim3(k), dim3(64), 0, stream, D, k, D.workW);
    }
    hipLaunchKernelGGL(k_identity, dim3(cdiv(k, 256)), dim3(256), 0, stream, D, k);
    if (checkLaunches("refineInverse (gather)"))
      return 1;
    if (dgemmFn(blasHandle, none, none, k, k, k, &minusOne, D.Minv, ld, D.workW, ld, &one, D.workX, ld) != 0)
      return 1;
    hipLaunchKernelGGL(k_absmax_rows, dim3(k), dim3(256), 0, stream, D, k, (const double *)D.workX, dMax);
  } else {
    hipLaunchKernelGGL(k_refine_residual, dim3(k), dim3(256), 0, stream, D, k, D.workX, dMax);
  }
  unsigned long long bits = 0;
  if (d2h(&bits, dMax, 1))
    return 1;
  memcpy(&lastResidual, &bits, sizeof(double));
  if (!(lastResidual <= (tailFromS ? 0.5 : refreshResidualMax)))
    return 2;  // too far for one step (or not finite): re-invert
  if (lastResidual < goodBelow)
    return 3;  // nothing to gain
  // workW = X; workW += X R  (row-major X R = column-major R^T X^T: A = R, B = X in rocBLAS terms)
  if (hipMemcpyAsync(D.workW, D.Minv, mat, hipMemcpyDeviceToDevice, stream) != hipSuccess)
    return 1;
  if (dgemmFn(blasHandle, none, none, k, k, k, &one, D.workX, ld, D.Minv, ld, &one, D.workW, ld) != 0)
    return 1;
  if (hipMemcpyAsync(D.Minv, D.workW, mat, hipMemcpyDeviceToDevice, stream) != hipSuccess)
    return 1;
  numberRefines++;
  return 0;
}

int clpgpu_context::refreshFactor()
{
  // what factorizeOnce leaves behind, minus the re-inversion: the slot arrays, the row-copy partition and
  // pivotVariable are current on the device (houseBody keeps them); the host mirror of pivotVariable follows
  int rc = d2h(pivotVariable.data(), D.pivotVariable, m);
  kNucleus = hCtrl->k;
  pivots = 0;
  hCtrl->pivots = 0;
  consecutiveRefreshes++;
  numberRefreshes++;
  return rc;
}

int clpgpu_context::statusOfProblemInDual(int type)
{
  int rc = 0;
  int numberPivots = pivots;
  int tentativeStatus = problemStatus;
  bool weightsSaved = false;
  bool gutsDone = false;  // a verified refresh has already recomputed the solutions
  bool unflagVariables = true, reallyBadProblems = false;
  double changeCost = 0.0;
  if (problemStatus > -3 || numberPivots > 0) {
    rc |= saveWeights(1);
    weightsSaved = true;
    if (type && debugPoisonInverseAt >= 0 && numberIterations >= debugPoisonInverseAt && refreshEligible()) {
      // fault injection (option debug_poison_inverse_at): a NaN in the kept inverse right before a verified refresh,
      // which then has to be refused (the maxima of the check chain propagate NaN) and followed by a re-inversion
      const double poison = nan("");
      rc |= h2d(D.Minv + (size_t)(hCtrl->k / 2) * ld + hCtrl->k / 3, &poison, 1);
      debugPoisonInverseAt = -1;
      numberPoisoned++;
    }
    if (type && refreshEligible() && (!refreshRefine || refineInverse() == 0)) {
      rc |= pullRim(true);
      rc |= refreshFactor();
      rc |= gutsOfSolution();
      if (logLevel > 1)
        fprintf(stderr, "clpgpu: iteration %d, nucleus %d kept%s (max |I - C X| before %g): errors %g %g\n", numberIterations, kNucleus,
                refreshRefine ? " + Newton step" : "", lastResidual, largestPrimalError, largestDualError);
      if (largestPrimalError <= refreshTolerance && largestDualError <= refreshTolerance) {
        gutsDone = true;
      } else {
        // the inverse has drifted: re-invert below
        numberRefreshes--;
        numberRefreshesRejected++;
        consecutiveRefreshes = refreshMax;
        if (logLevel > 0)
          fprintf(stderr, "clpgpu: refresh rejected at iteration %d (errors %g %g), re-inverting\n", numberIterations, largestPrimalError, largestDualError);
      }
    }
    if (type && !gutsDone) {
      rc |= pullRim(true);
      int frc = factorize(false);
      consecutiveRefreshes = 0;
      if (!frc && debugSingularAt >= 0 && numberIterations >= debugSingularAt && numberIterations > 0) {
        frc = -1;  // fault injection (option debug_singular_at): this refactorization is taken as singular
        debugSingularAt = -1;
      }
      if (frc == -1) {
        // singular (ClpSimplexDual.cpp:5060-5125): back to the basis of the last good factorization with
        // the leaving variable flagged and a refactorization forced after every pivot; if that basis is
        // singular too, the repaired basis (dependent structurals out, slacks in)
        if (haveSnapshot) {
          numberSingularRestores++;
          unflagVariables = false;
          for (int i = 0; i < N; i++)
            if (status[i] & FLAGGED_BIT)
              saveStatus[i] |= FLAGGED_BIT;
          status = saveStatus;
          sol = savedSolution;
          resetFakeBounds();
          if (hCtrl->sequenceOut >= 0 && hCtrl->sequenceOut < N)
            status[hCtrl->sequenceOut] |= FLAGGED_BIT;
          progBadTimes = 0;  // progress_.clearBadTimes()
          forceFactorization = 1;
          rebuildRowCopy = true;
          frc = factorize(false);
        }
        if (frc == -1) {
          frc = factorize(true);
          if (frc >= 0)
            resetFakeBounds();
        }
        if (frc >= 0) {
          frc = 0;
          type = 2;
          rc |= pushRim();
        }
      }
      if (frc) {
        problemStatus = 4;
        return frc;
      }
    }
    if (problemStatus != -4 || numberPivots > 10)
      problemStatus = -3;
  }
  const bool adaptTolerance = progInfeasibility[0] < 1.0e-1 && primalTolerance == 1.0e-7 && progIteration[0] > 0
                              && progIteration[PROGRESS - 1] - progIteration[0] > 25;
  if (adaptTolerance) {
    // the default primal tolerance is loosened when the last five checks all show tiny infeasibilities (:5136-5160)
    int iP;
    double minAverage = DBL_MAX, maxAverage = 0.0;
    for (iP = 0; iP < PROGRESS; iP++) {
      const int count = progNumberInfeasibilities[iP];
      if (!count)
        break;
      double average = progInfeasibility[iP];
      if (average > 0.1)
        break;
      average /= (double)count;
      minAverage = std::min(minAverage, average);
      maxAverage = std::max(maxAverage, average);
    }
    if (iP == PROGRESS && minAverage < 1.0e-5 && maxAverage < 1.0e-3)
      primalTolerance = optPrimalTolerance = 1.0e-6;  // dblParam_[ClpPrimalTolerance] too
  }
  if (type && !gutsDone) {
    rc |= gutsOfSolution();
    if (logLevel > 1)
      fprintf(stderr, "clpgpu: iteration %d, nucleus %d re-inverted: errors %g %g\n", numberIterations, kNucleus, largestPrimalError, largestDualError);
  } else if (type && adaptTolerance) {
    checkPrimalSolution();  // the refresh recomputed the solutions before the tolerance moved
    checkDualSolution();
  }
  if (debugBadAccuracyAt >= 0 && numberIterations >= debugBadAccuracyAt && numberIterations > 0) {
    largestPrimalError = 1.0e16;  // fault injection
    debugBadAccuracyAt = -1;
  }
  if ((!(largestPrimalError <= 1.0e15) || !(largestDualError <= 1.0e15)) && numberIterations && haveSnapshot) {
    // bad accuracy (or not a number at all): treat as singular -- back to the previous basis with a variable rejected (:5237-5318)
    numberAccuracyRestores++;
    unflagVariables = false;
    for (int i = 0; i < N; i++)
      if (status[i] & FLAGGED_BIT)
        saveStatus[i] |= FLAGGED_BIT;  // keep any flagged variables
    status = saveStatus;
    sol = savedSolution;
    resetFakeBounds();  // resetFakeBounds(1): correct bounds on all variables
    int rejectedVariable = hCtrl->sequenceOut;
    if (rejectedVariable < 0 || rejectedVariable >= N || (status[rejectedVariable] & FLAGGED_BIT)) {
      rejectedVariable = -1;
      for (int i = 0; i < m && rejectedVariable < 0; i++)
        if (!(status[pivotVariable[i]] & FLAGGED_BIT))
          rejectedVariable = pivotVariable[i];
      if (rejectedVariable < 0) {
        problemStatus = 10;  // real trouble
        return rc;
      }
    }
    status[rejectedVariable] |= FLAGGED_BIT;
    progBadTimes = 0;
    forceFactorization = 1;  // a bit drastic but ..
    type = 2;
    rebuildRowCopy = true;
    const int frc = factorize(false);
    consecutiveRefreshes = 0;
    if (frc) {
      problemStatus = 4;
      return frc;
    }
    rc |= pushRim();
    rc |= gutsOfSolution();
    if (logLevel > 0)
      fprintf(stderr, "clpgpu: bad accuracy at iteration %d: back to the last good basis, variable %d flagged\n", numberIterations, rejectedVariable);
  }
  if (progIteration[PROGRESS - 1] == numberIterations) {
    // double check infeasibility if no action (:5326-5330)
    if (looksOptimal()) {
      numberPrimalInfeasibilities = 0;
      sumPrimalInfeasibilities = 0.0;
    }
  } else {
    // has the objective gone backwards since the last check? (:5332-5488)
    const double thisObj = objectiveValue - bestPossibleImprovement;
    double lastObj = progObjective[PROGRESS - 1];
    double testTol = 5.0e-3;
    if (progTimesFlagged > 10)
      testTol *= pow(2.0, progTimesFlagged - 8);
    else if (progTimesFlagged > 5)
      testTol *= 5.0;
    if (debugBackwardsAt >= 0 && numberIterations >= debugBackwardsAt && numberIterations > 0) {
      // fault injection, as in the oracle: two checks in a row see a drop, the first small, the second large
      lastObj = thisObj + ((progressFlag & 4) ? 2.0e4 : 1.0) * (testTol * 4.0 * (fabs(thisObj) + 1.0) + 1.0);
      if (progressFlag & 4)
        debugBackwardsAt = -1;
    }
    if (firstFree < 0 /* :5360 */ && lastObj > thisObj + testTol * (fabs(thisObj) + fabs(lastObj)) + testTol) {
      if (progTimesFlagged > 10)
        progReallyBadTimes++;
      if (fastDualMode) {
        problemStatus = 3;  // in fast dual give up (:5478-5483)
      } else if (maximumPivots > 1) {
        if ((progressFlag & 4) == 0 && lastObj < thisObj + 1.0e4 && largestPrimalError < 1.0e2) {
          costCopy = cost;  // just save costs
          progressFlag |= 4;
        } else if (!haveSnapshot) {
          setError("objective going backwards at iteration %d with no saved basis", numberIterations);
        } else {
          // back to the basis of the last good check; refactorize after every pivot for a while
          numberBackwards++;
          forceFactorization = 1;
          unflagVariables = false;
          status = saveStatus;
          sol = savedSolution;
          if ((progressFlag & 4) == 0) {
            costCopy = cost;
            progressFlag |= 4;
          } else {
            cost = costCopy;
          }
          rebuildRowCopy = true;
          const int frc = factorize(false);  // the saved basis factorized before
          consecutiveRefreshes = 0;
          if (frc) {
            problemStatus = 4;
            return frc;
          }
          resetFakeBoundsToOriginal();
          type = 2;  // so will restore weights
          rc |= pushRim();
          rc |= gutsOfSolution();
          if (numberPivots < 2 && hCtrl->sequenceOut >= 0 && hCtrl->sequenceOut < N) {
            // need to reject something
            status[hCtrl->sequenceOut] |= FLAGGED_BIT;
            rc |= h2d(D.status, status.data(), N);
            progBadTimes = 0;
            progTimesFlagged++;
          }
          if (numberPivots < 10)
            reallyBadProblems = true;
          progObjective[PROGRESS - 1] = objectiveValue - bestPossibleImprovement;
          if (logLevel > 0)
            fprintf(stderr, "clpgpu: objective went backwards at iteration %d: back to the last good basis\n", numberIterations);
        }
      }
    } else if (lastObj < thisObj - 1.0e-5 * std::max(fabs(thisObj), fabs(lastObj)) - 1.0e-3) {
      numberTimesOptimal = 0;
    }
  }
  // check if looping (:5506-5536)
  const int loop = (type != 2) ? progressLooping() : -1;
  if (progReallyBadTimes > 10)
    problemStatus = 10;
  int situationChanged = 0;
  if (loop >= 0) {
    problemStatus = loop;
    if (!problemStatus) {
      numberPrimalInfeasibilities = 0;  // declaring victory
      sumPrimalInfeasibilities = 0.0;
    } else if (problemStatus != 3) {
      problemStatus = 10;
    }
    return rc;
  } else if (loop < -1) {
    rc |= pushRim();  // something may have changed
    rc |= gutsOfSolution();
    situationChanged = 1;
  }
  if (progressFlag & 2)
    situationChanged = 2;
  progressFlag &= ~3;
  if (progressFlag & 4)
    costCopy = cost;  // :5543-5547
  if (!numberPrimalInfeasibilities && !numberDualInfeasibilities)
    progressFlag |= 8;
  // if we are primal feasible and any dual infeasibilities are on free variables then it is better to go to primal (:5619-5622)
  if (freeNonbasic && !numberPrimalInfeasibilities && !numberDualInfeasibilitiesWithoutFree && numberDualInfeasibilities) {
    problemStatus = 10;
    if (logLevel > 3)
      fprintf(stderr, "clpgpu: iteration %d primal feasible, the %d dual infeasibilities are all on free variables: 10\n", numberIterations,
              numberDualInfeasibilities);
  }
  bool needCleanFake = false, dirty = false;
  double saveDualBound = dualBound;
  while (problemStatus <= -3 && saveDualBound == dualBound) {
    int cleanDuals = 0;
    if (situationChanged != 0)
      cleanDuals = 1;
    int numberChangedBounds = 0;
    int doOriginalTolerance = 0;
    if (lastCleaned == numberIterations)
      doOriginalTolerance = 1;
    if (sumOfRelaxedDualInfeasibilities == 0.0 && sumOfRelaxedPrimalInfeasibilities == 0.0) {
      numberDualInfeasibilities = 0;
      sumDualInfeasibilities = 0.0;
      numberPrimalInfeasibilities = 0;
      sumPrimalInfeasibilities = 0.0;
    }
    if (numberDualInfeasibilities == 0 || problemStatus == -4) {
      progObjective[PROGRESS - 1] = objectiveValue - bestPossibleImprovement;  // progress_.modifyObjective, :5645
      if (numberPrimalInfeasibilities == 0) {
        numberChangedBounds = (dualBound < 1.0e20) ? changeBounds(0, changeCost) : 0;
        dirty = true;
        if (numberChangedBounds <= 0 && !numberDualInfeasibilities) {
          if (perturbation == 101) {
            // looks optimal for the perturbed costs: the true costs back and look again (:5708-5733)
            perturbation = 102;
            cleanDuals = 1;
            changeBounds(1, changeCost);  // make sure fake bounds are back
            restoreCosts();
            rc |= pushRim();
            rc |= gutsOfSolution();  // computeDuals + checkDualSolution with the true costs
            progObjective[PROGRESS - 1] = -DBL_MAX;
            if (numberDualInfeasibilities) {
              numberChanged = 1;  // force something to happen
              lastCleaned = numberIterations - 1;
            }
          }
          if (lastCleaned < numberIterations && numberTimesOptimal < 4) {
            doOriginalTolerance = 2;
            numberTimesOptimal++;
            if (numberTimesOptimal == 1)
              dualTolerance = dualToleranceBase;
            else
              dualTolerance = dualToleranceBase * pow(2.0, numberTimesOptimal - 1);
            cleanDuals = 2;
          } else {
            problemStatus = 0;  // optimal
          }
        } else {
          cleanDuals = 1;
          if (doOriginalTolerance == 1) {
            if (dualBound > 1.0e17)
              problemStatus = 2;
            else
              problemStatus = -3;
            if (problemStatus == 2 && perturbation == 101) {
              // unbounded only for the perturbed costs? (:5814-5820)
              perturbation = 102;
              cleanDuals = 1;
              restoreCosts();
              rc |= h2d(D.cost, cost.data(), N);
              progObjective[PROGRESS - 1] = -DBL_MAX;
              problemStatus = -1;
            }
          } else {
            doOriginalTolerance = 2;
          }
        }
      }
      if (problemStatus == -4 || problemStatus == -5) {
        numberChangedBounds = changeBounds(0, changeCost);
        needCleanFake = true;
        dirty = true;
        if ((numberChangedBounds <= 0 || dualBound > 1.0e20 || (largestPrimalError > 1.0 && dualBound > 1.0e17))
            && (numberPivots < 4 || sumPrimalInfeasibilities > 1.0e-6)) {
          problemStatus = 1;
          if (perturbation == 101)
            perturbation = 102;  // :5860
          if (!numberPrimalInfeasibilities) {
            problemStatus = -1;
            doOriginalTolerance = 2;
          }
        } else {
          problemStatus = -1;
          cleanDuals = 1;
          if (numberChangedBounds <= 0)
            doOriginalTolerance = 2;
        }
      }
    } else {
      cleanDuals = 1;
    }
    if (problemStatus < 0) {
      if (doOriginalTolerance == 2) {
        lastCleaned = numberIterations;
        numberChanged = 0;
        perturbation = 102;  // stop any perturbations (:5891)
        restoreCosts();
        progObjective[PROGRESS - 1] = -DBL_MAX;
        // computeDuals with the original costs, on the device
        rc |= pushRim();
        rc |= gutsOfSolution();
        if (cleanDuals != 2) {
          changeBounds(3, changeCost);
          needCleanFake = true;
          cleanDuals = 2;
          dirty = true;
        }
      }
      if (cleanDuals == 1 || (cleanDuals == 2 && !numberDualInfeasibilities)) {
        updateDualsFullRecompute();
        rc |= pushRim();
        rc |= gutsOfSolution();
        if (updateDualsFullRecompute()) {
          rc |= pushRim();
          rc |= gutsOfSolution();
        }
        dirty = false;
        if (numberDualInfeasibilities) {
          if ((numberPrimalInfeasibilities || numberPivots) && problemStatus != 10)
            problemStatus = -1;
          else
            problemStatus = 10;
        } else if (situationChanged == 2) {
          problemStatus = -1;
          changeBounds(3, changeCost);
          dirty = true;
        }
        situationChanged = 0;
      } else {
        if (cleanDuals != 2)
          problemStatus = -1;
        else
          problemStatus = 10;
      }
    }
  }
  if (tentativeStatus != -2 && tentativeStatus != -1 && unflagVariables) {
    int numberFlagged = 0;
    for (int iRow = 0; iRow < m; iRow++) {
      int iPivot = pivotVariable[iRow];
      if (status[iPivot] & FLAGGED_BIT) {
        numberFlagged++;
        status[iPivot] &= (unsigned char)~FLAGGED_BIT;
        dirty = true;
      }
    }
    if (numberFlagged && !numberPivots) {
      if (numberTimesOptimal < 3) {
        numberTimesOptimal++;
        problemStatus = -1;
      } else {
        problemStatus = 10;
      }
    }
  }
  if (problemStatus < 0) {
    if (needCleanFake) {
      double dummy = 0.0;
      changeBounds(3, dummy);
      dirty = true;
    }
    if (dirty)
      rc |= pushRim();
    if (type == 0 || type == 1) {
      // the basis just factorized is the one to come back to (saveStatus_ / savedSolution_, :6160-6175)
      saveStatus = status;
      savedSolution = sol;
      haveSnapshot = true;
    }
    if (weightsSaved) {
      if (!reallyBadProblems && (largestPrimalError < 100.0 || numberPivots > 10)) {
        if (tentativeStatus > -3)
          rc |= saveWeights((type < 2) ? 2 : 4);
        else
          rc |= saveWeights(3);
      } else {
        rc |= saveWeights(6);  // reset weights or scale back
      }
      // experiment knobs (profiles/r04_objective_race.md): what a mature config-4 run gains from fresh steepest-edge weights, or from
      // its true costs, alone -- once, at the first status check from the given iteration on
      if (dseResetEvery > 0 && type && ++dseResetCounter >= dseResetEvery) {
        dseResetCounter = 0;
        rc |= saveWeights(6);
        numberWeightResets++;
      }
      if (debugResetWeightsAt >= 0 && numberIterations >= debugResetWeightsAt) {
        debugResetWeightsAt = -1;
        rc |= saveWeights(6);
        if (logLevel > 0)
          fprintf(stderr, "clpgpu: iteration %d: steepest-edge weights reset (debug_reset_weights_at)\n", numberIterations);
      }
    }
  } else if (dirty) {
    rc |= pushRim();
  }
  {
    // refactorize more often when the recorded objective fell between the last two checks (:6316-6328)
    const double thisObj = progObjective[PROGRESS - 1], lastObj = progObjective[PROGRESS - 2];
    if (lastObj > thisObj + 1.0e-4 * std::max(fabs(thisObj), fabs(lastObj)) + 1.0e-4 && firstFree < 0 /* :6319 */ && maximumPivots > 10) {
      if (forceFactorization < 0)
        forceFactorization = maximumPivots;
      forceFactorization = std::max(1, forceFactorization >> 1);
    }
  }
  if (problemStatus == 1 && (progressFlag & 8) != 0 && fabs(objectiveValue) > 1.0e10)
    problemStatus = 10;  // infeasible - but has looked feasible (:6338)
  if (logLevel > 2)
    fprintf(stderr, "STATUS it %d type %d -> problemStatus %d; primal %d (%.10g) dual %d (%.10g) relaxed %.6g %.6g; dualBound %g dualTol %g fake %d objective %.10g errors %.3g %.3g progressFlag %d\n",
            numberIterations, type, problemStatus, numberPrimalInfeasibilities, sumPrimalInfeasibilities, numberDualInfeasibilities, sumDualInfeasibilities,
            sumOfRelaxedPrimalInfeasibilities, sumOfRelaxedDualInfeasibilities, dualBound, dualTolerance, numberFake, objectiveValue, largestPrimalError, largestDualError, progressFlag);
  return rc;
}

// ---------------------------------------------------------------------------------------------
// one pivot = this fixed chain of launches (graph-capturable: every decision is on the device)
// ---------------------------------------------------------------------------------------------
void clpgpu_context::joinUpdateBranch()
{
  if (sidePending) {
    (void)hipStreamWaitEvent(stream, evJoin, 0);
    sidePending = false;
  }
}

void clpgpu_context::ktBegin(int pivotInBatch)
{
  ktPivot = pivotInBatch;
  if ((int)ktEvents.size() < checkEvery * KT_MAX) {
    size_t old = ktEvents.size();
    ktEvents.resize((size_t)checkEvery * KT_MAX);
    for (size_t i = old; i < ktEvents.size(); i++)
      (void)hipEventCreate(&ktEvents[i]);
    ktSlotOfMark.assign((size_t)checkEvery * KT_MAX, -1);
    ktMarks.assign(checkEvery, 0);
  }
  ktMarks[pivotInBatch] = 0;
  ktMark(nullptr);  // start-of-pivot mark
}
void clpgpu_context::ktMark(const char *name)
{
  int &nm = ktMarks[ktPivot];
  if (nm >= KT_MAX)
    return;
  int slot = -1;
  if (name) {
    for (size_t i = 0; i < ktNames.size(); i++)
      if (ktNames[i] == name || !strcmp(ktNames[i], name))
        slot = (int)i;
    if (slot < 0) {
      slot = (int)ktNames.size();
      ktNames.push_back(name);
      ktMs.push_back(0.0);
      ktCount.push_back(0);
    }
  }
  const size_t idx = (size_t)ktPivot * KT_MAX + nm;
  ktSlotOfMark[idx] = slot;
  (void)hipEventRecord(ktEvents[idx], stream);
  nm++;
}
// after the batch has been synchronised: elapsed time between consecutive marks goes to the kernel
// the later mark follows (kernel + the launch gap before it); only pivots the device really ran
void clpgpu_context::ktCollect(int livePivots)
{
  for (int b = 0; b < livePivots && b < (int)ktMarks.size(); b++)
    for (int i = 1; i < ktMarks[b]; i++) {
      const size_t idx = (size_t)b * KT_MAX + i;
      float ms = 0.0f;
      if (ktSlotOfMark[idx] >= 0 && hipEventElapsedTime(&ms, ktEvents[idx - 1], ktEvents[idx]) == hipSuccess) {
        ktMs[ktSlotOfMark[idx]] += ms;
        ktCount[ktSlotOfMark[idx]]++;
      }
    }
}

void clpgpu_context::allGather(const void *send, void *recv, size_t count, int dtype)
{
  if (!virtualGroup) {
    ncclAllGatherFn(send, recv, count, dtype, comm, stream);
    return;
  }
  // loopback: my contribution is complete once my stream has drained; then every rank copies every slice
  clpgpu_virtual_group *g = virtualGroup;
  const size_t bytes = count * (dtype == 1 ? 1 : 8);
  if (commFailed || hipStreamSynchronize(stream) != hipSuccess) {
    commFailed = true;
    return;
  }
  g->sendPtr[rank] = send;
  if (!g->barrier()) {
    commFailed = true;
    return;
  }
  for (int r = 0; r < nranks; r++) {
    char *dst = (char *)recv + (size_t)r * bytes;
    if (g->sendPtr[r] != (const void *)dst)
      (void)hipMemcpyAsync(dst, g->sendPtr[r], bytes, hipMemcpyDeviceToDevice, stream);
  }
  // nobody may overwrite its send buffer before every copy out of it has executed
  if (hipStreamSynchronize(stream) != hipSuccess || !g->barrier())
    commFailed = true;
}

// The row a free column should pivot on (src/ClpSimplexDual.cpp:3016-3049): `work` = the FTRANned column by basis position.  Among the
// positions with |alpha| > 1e-3: the unflagged one with the largest infeasibility x |alpha| (and |alpha| > 0.1), else -- if its |alpha|
// passes 0.01 -- the one with the largest |alpha| whose variable has a bound at all; strict maxima, the first of equals.  -1: none.
// Host arithmetic only; also behind clpgpu_test_free_first_row (tests/test_free_first_row.py).
static int freeFirstChoice(int m, const double *work, const int *pivotVariable, const double *sol, const double *lower, const double *upper,
                           const unsigned char *status)
{
  double bestFeasibleAlpha = 0.0, bestInfeasibleAlpha = 0.0;
  int bestFeasibleRow = -1, bestInfeasibleRow = -1;
  for (int iRow = 0; iRow < m; iRow++) {
    const double alpha = fabs(work[iRow]);
    if (alpha > 1.0e-3) {
      const int iSequence = pivotVariable[iRow];
      const double value = sol[iSequence], lo = lower[iSequence], up = upper[iSequence];
      double infeasibility = 0.0;
      if (value > up)
        infeasibility = value - up;
      else if (value < lo)
        infeasibility = lo - value;
      if (infeasibility * alpha > bestInfeasibleAlpha && alpha > 1.0e-1) {
        if (!(status[iSequence] & FLAGGED_BIT)) {
          bestInfeasibleAlpha = infeasibility * alpha;
          bestInfeasibleRow = iRow;
        }
      }
      if (alpha > bestFeasibleAlpha && (lo > -1.0e20 || up < 1.0e20)) {
        bestFeasibleAlpha = alpha;
        bestFeasibleRow = iRow;
      }
    }
  }
  if (bestInfeasibleRow >= 0)
    return bestInfeasibleRow;
  if (bestFeasibleAlpha > 1.0e-2)
    return bestFeasibleRow;
  return -1;
}

// option free_nonbasic: the sequences that are isFree / superBasic now, rows first and then columns -- the order the general branch
// of dualColumn0 walks the tableau row in (src/ClpSimplexDual.cpp:4066-4070) -- for k_free_scan.  Nothing becomes free or superbasic
// between two status checks (a leaving variable goes to a bound, :2068-2094 of ClpSimplex.cpp), so the list stays a superset; the
// kernel looks at the status again.  freeCount is 0 while moreSpecialOptions_ & 8 ("no free or super basic") holds: the fast
// branch, which is what the pricing kernels fuse.
int clpgpu_context::pushFreeList()
{
  freeListHost.clear();
  for (int i = n; i < N; i++) {
    const int st = status[i] & 7;
    if (st == ST_FREE || st == ST_SUPER)
      freeListHost.push_back(i);
  }
  for (int i = 0; i < n; i++) {
    const int st = status[i] & 7;
    if (st == ST_FREE || st == ST_SUPER)
      freeListHost.push_back(i);
  }
  const bool active = !noFreeOrSuper && !freeListHost.empty();
  hCtrl->freeCount = active ? (int)freeListHost.size() : 0;
  freeActive = active;
  if (!active)
    return 0;
  int rc = 0;
  if (!dFreeList) {
    rc |= dalloc(dFreeList, (size_t)N);
    if (rc)
      return rc;
    D.freeList = dFreeList;
    dropGraph();  // (the captured chains carry Dev by value)
  }
  rc |= h2d(dFreeList, freeListHost.data(), freeListHost.size());
  return rc;
}

// ClpSimplexDual::dualRow's free-first entry (src/ClpSimplexDual.cpp:3005-3055) with nextSuperBasic (:8285-8302), on the host
// between two pivots: chosenRow = the row the next free column should pivot on, or -1 (the pivot-rule object chooses).
int clpgpu_context::freeFirstRow(int &chosenRow)
{
  chosenRow = -1;
  int rc = sync();
  rc |= pullRim(true);
  // nextSuperBasic: hand out firstFree_, move it on to the next free column with a reduced cost
  int nextFree = -1;
  if (firstFree >= 0) {
    nextFree = firstFree;
    int iColumn = firstFree + 1;
    for (; iColumn < N; iColumn++)
      if ((status[iColumn] & 7) == ST_FREE && fabs(dj[iColumn]) > 1.0e2 * dualTolerance)
        break;
    firstFree = iColumn == N ? -1 : iColumn;
  }
  if (nextFree < 0 || rc)
    return rc;
  // unpack vector and find a good pivot
  std::vector<double> work(m, 0.0);
  if (nextFree >= n) {
    work[nextFree - n] = -1.0;
  } else {
    for (int p = colStart[nextFree]; p < colStart[nextFree + 1]; p++)
      work[row[p]] = elem[p];
  }
  rc |= h2d(D.vecV2, work.data(), m);
  ftranDevice(D.vecV2, D.x3);
  rc |= d2h(work.data(), D.x3, m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(m, 256)), dim3(256), 0, stream, D.vecV2, m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(m, 256)), dim3(256), 0, stream, D.x3, m);
  rc |= checkLaunches("freeFirstRow");
  rc |= sync();
  if (rc)
    return rc;
  chosenRow = freeFirstChoice(m, work.data(), pivotVariable.data(), sol.data(), lower.data(), upper.data(), status.data());
  return 0;
}

int clpgpu_context::launchIteration(bool firstOfBatch, int parity)
{
  ktOn = timing >= 2 && !capturing;
  const int nbRows = cdiv(m, PRICE_BLOCK);
  const int nbCols = cdiv(D.lastColumn - D.firstColumn, PRICE_BLOCK);
  const int nb = nbRows + nbCols;
  const int gm = cdiv(m, 256);
  const int kc = kcap;  // launch extents use the capacity; kernels read the live k from ctrl
  const bool ev = timing && !capturing && evUsed < (int)evStart.size();
  int selfScanSell = -1;
  // CHUZR (+ the analytic front end of the BTRAN)
  if (firstOfBatch)
    KL("k_chuzr_pre", k_chuzr_pre, dim3(1), dim3(64), 0, stream, D);
  if (nChzBlocks <= 256) {
    // the last workgroup of the scan makes the final selection (no separate launch)
    KL("k_chuzr_scan", k_chuzr_scan, dim3(nChzBlocks), dim3(256), 0, stream, D, wideRows ? 1 : 0);
  } else {
    KL("k_chuzr_scan", k_chuzr_scan, dim3(nChzBlocks), dim3(256), 0, stream, D, -1);
    KL("k_chuzr_final_btran", k_chuzr_final_btran, dim3(1), dim3(256), 0, stream, D, nChzBlocks, wideRows ? 1 : 0);
  }
  // BTRAN (reads Minv: the previous pivot's basis-update branch must have finished)
  joinUpdateBranch();
  if (luActive)
    luLaunchBtran();
  if (wideRows)
    KL("k_gemvT_partial2", k_gemvT_partial2, dim3(cdiv(kc, 256), cdiv(kc, 64)), dim3(256), 0, stream, D, (const double *)D.slotA, 1);
  // (not column-sharded: the candidate counts come from this kernel and the pricing kernel, there
  // is no separate counting launch)
  const bool gatherRows = commActive && commMode == 1;  // round-1 exchange: dense row slices, candidates recomputed from them
  const bool shardLists = commActive && commMode == 2;  // candidate / flip lists exchanged, reduced costs owned by the shard
  const bool countInPrice = priceKernel >= 1 && !gatherRows && nb > 256;
  // by-row pricing for sparse pi (needs the bitmap and the count-in-price compaction scheme)
  const int nSlots = nSellBlocks + nLongBlocks;
  // (the dense chain needs the count-in-price compaction scheme, as the by-row form does)
  const bool denseChain = priceMode == 1 && jdsReady && countInPrice && !widePricing && priceKernel >= 2;
  const int rowMax = (!denseChain && rowPriceFrac > 0.0 && countInPrice && !widePricing && priceKernel >= 2 && m <= 64 * SELL_BITS_MAX && nSlots > 0)
                         ? std::max(1, (int)(rowPriceFrac * m))
                         : 0;
  KL("k_rho_finish3", k_rho_finish3, dim3(gm), dim3(256), 0, stream, D, wideRows ? 1 : 0, countInPrice ? nbCols : -1, rowMax > 0 ? nSlots : 0);
  // PRICE + first ratio pass
  if (ev)
    (void)hipEventRecord(evStart[evUsed], stream);
  bool midDone = false;
  if (priceKernel >= 1) {
    if (widePricing && priceKernel != 1)
      KL("k_price_wide", k_price_wide, dim3(nWideBlocks), dim3(256), 0, stream, D, denseColumns ? 1 : 0, countInPrice ? 1 : 0);
    else if (nSlots > 0 && denseChain) {
      // pi dense on (nearly) every pivot of late: the tile-by-tile sweep with pi in LDS; it prices a sparse pi correctly too
      KL("k_price_lds", k_price_lds, dim3(priceLdsGrid), dim3(PL_THREADS), priceLdsBytes, stream, D, countInPrice ? 1 : 0);
      if (nLongBlocks > 0)  // columns too long for a SELL lane keep their own workgroups
        KL("k_price_sell", k_price_sell, dim3(nSlots), dim3(256), 0, stream, D, 1, countInPrice ? 1 : 0, nSellBlocks, nSlots, 0, 0, 32);
    } else if (nSlots > 0) {
      KL("k_price_sell", k_price_sell, dim3(nSlots + (rowMax > 0 ? gm : 0)), dim3(256),
         (m > 64 * SELL_BITS_MAX || priceKernel < 2) ? 0 : (size_t)((m + 63) / 64) * 8, stream, D,
         (m > 64 * SELL_BITS_MAX && priceKernel > 1) ? 1 : priceKernel, countInPrice ? 1 : 0, nSellBlocks, nSlots, rowMax);
      if (ev) {
        (void)hipEventRecord(evMid[evUsed], stream);  // a pivot priced by column ends here (k_price_row_finish returns at once)
        midDone = true;
      }
      if (rowMax > 0)
        KL("k_price_row_finish", k_price_row_finish, dim3(nbCols), dim3(PRICE_BLOCK), 0, stream, D, nbRows, rowMax);
    }
    if (ev && !midDone)
      (void)hipEventRecord(evMid[evUsed], stream);
    if (ev)
      (void)hipEventRecord(evStop[evUsed++], stream);
    if (gatherRows) {
      // round-1 exchange (also the fallback of the list exchange): every rank contributes its slice of
      // the tableau row and of the first-pass flags, in place; everything after is replicated
      const size_t chunk = (size_t)shardChunk;
      allGather(D.alphaCol + (size_t)rank * chunk, D.alphaCol, chunk, 8 /* ncclFloat64 */);
      allGather(D.candFlag + D.m + (size_t)rank * chunk, D.candFlag + D.m, chunk, 1 /* ncclUint8 */);
    }
    {
      const int nSell = (widePricing && priceKernel != 1) ? nWideBlocks : nSellBlocks + nLongBlocks;
      const bool fuse = nb <= 256;  // small grids: the last workgroup scans the counts itself
      if (!countInPrice)
        KL("k_cand_count", k_cand_count, dim3(nb), dim3(PRICE_BLOCK), 0, stream, D, nbRows, gatherRows ? 1 : 0, fuse ? nSell : -1);
      selfScanSell = fuse ? -1 : nSell;  // large grids: k_cand_scatter scans for itself, no scan launch
    }
  } else {
    KL("k_price", k_price, dim3(nb), dim3(PRICE_BLOCK), 0, stream, D, nbRows);
    if (ev) {
      (void)hipEventRecord(evMid[evUsed], stream);
      (void)hipEventRecord(evStop[evUsed++], stream);
    }
    KL("k_scan_blocks", k_scan_blocks, dim3(1), dim3(1024), 0, stream, D, nb, 0, 1, 0);
  }
  KL("k_cand_scatter", k_cand_scatter, dim3(nb), dim3(PRICE_BLOCK), 0, stream, D, nbRows, selfScanSell);
  if (shardLists) {
    // the local list is [row candidates (replicated) | this rank's column candidates]: the column part
    // of every rank, gathered in rank order behind the rows, is the single-GPU list
    KL("k_shard_pack_cands", k_shard_pack_cands, dim3(std::max(1, std::min(128, cdiv(shardCandCap, 1024)))), dim3(256), 0, stream, D, nbRows, dCandSend, shardCandCap);
    allGather(dCandSend, dCandRecv, SHARD_HDR + 4 * (size_t)shardCandCap, 8 /* ncclFloat64 */);
    KL("k_shard_merge_cands", k_shard_merge_cands, dim3(cdiv(shardCandCap, 256), nranks), dim3(256), 0, stream, D, (const double *)dCandRecv,
       nranks, shardCandCap);
  }
  // CHUZC; in sharded runs the classes of the working-set shortcut are those of the merged list
  int nbClass = nb;
  if (shardLists) {
    nbClass = cdiv(m + nranks * shardCandCap, PRICE_BLOCK);
    KL("k_shard_classes", k_shard_classes, dim3(nbClass), dim3(PRICE_BLOCK), 0, stream, D);
  }
  if (freeActive)  // option free_nonbasic: the isFree / superBasic part of dualColumn0's general branch (may bring a free variable in)
    KL("k_free_scan", k_free_scan, dim3(1), dim3(256), 0, stream, D);
  KL("k_dc_working_set", k_dc_working_set, dim3(128), dim3(WS_THREADS), 0, stream, D, nbClass);
  KL("k_dual_column", k_dual_column, dim3(1), dim3(DC_THREADS), 0, stream, D, nbClass, dcWide);
  if (dcWide)  // lists too long for one workgroup's registers: the ratio test over the chip (returns at once otherwise)
    KL("k_dual_column_wide", k_dual_column_wide, dim3(DCW_BLOCKS), dim3(DCW_THREADS), 0, stream, D);
  // dual update + flip detection (needs only theta), flip list, flip right-hand side
  // (+ 1: the extra workgroup unpacks the entering column)
  // sparse LPs, columns owned by this GPU or replicated: the waves that find a flip scatter its column
  // (the flip right-hand side then needs no single-workgroup pass over the flipped columns' entries)
  const int scatterFlips = ((flipScatter == 2 || (flipScatter == 1 && lightRows)) && !wideRows && !denseColumns && !shardLists && !gatherRows) ? 1 : 0;
  KL("k_dj_flags", k_dj_flags, dim3(nb + 1), dim3(PRICE_BLOCK), 0, stream, D, nbRows, flipListCap, scatterFlips, flipSlotCap);
  if (shardLists) {
    KL("k_shard_pack_flips", k_shard_pack_flips, dim3(1), dim3(256), 0, stream, D, dFlipSend, shardFlipCap, flipListCap);
    allGather(dFlipSend, dFlipRecv, SHARD_HDR + 5 * (size_t)shardFlipCap, 8 /* ncclFloat64 */);
    KL("k_shard_merge_flips", k_shard_merge_flips, dim3(1), dim3(256), 0, stream, D, (const double *)dFlipRecv, rank, nranks, shardFlipCap,
       flipListCap);
  }
  KL("k_flip_apply2", k_flip_apply2, dim3(1 + (scatterFlips ? cdiv(m, 1024) : 0)), dim3(1024), 0, stream, D, gm, denseColumns ? 1 : 0, flipListCap,
     scatterFlips, flipSlotCap);
  if (denseColumns)
    KL("k_flip_dense", k_flip_dense, dim3(gm), dim3(256), 0, stream, D);
  // one FTRAN sweep for the entering column, rho (DSE) and the flip rhs; the back end also applies
  // the flip part of the primal update
  if (luActive) {
    luLaunchFtran(gm, parity);
  } else {
    KL("k_gemv3g", k_gemv3g, dim3(cdiv(kc, GEMV_RPB)), dim3(1024), 0, stream, D);
    if (wideRows)
      KL("k_slack_dots", k_slack_dots, dim3(cdiv(m, 4)), dim3(256), 0, stream, D);
    KL("k_ftran_scatter3", k_ftran_scatter3, dim3(cdiv(m + kc, 256)), dim3(256), 0, stream, D, gm, parity, wideRows ? 1 : 0);
  }
  // basis update of the nucleus inverse: needs only what the FTRAN tail left (w and rho by slot, the
  // update scalars), nothing downstream needs Minv before the next BTRAN -> its own branch
  {
    const int gx = cdiv(kc, 256), gy = kc < 512 ? kc : 512;
    if (luActive) {
      // primal update as usual; the basis update is one more eta (column of H, row of G)
      if (luFold & 1) {
        KL("k_lu_pf_append", k_lu_pf_append, dim3(gm + cdiv(hLu.tcap, 4)), dim3(256), 0, stream, D, 2, gm);  // + the primal update
      } else {
        KL("k_primal_update", k_primal_update, dim3(gm), dim3(256), 0, stream, D, 0);
        KL("k_lu_pf_append", k_lu_pf_append, dim3(gm + cdiv(hLu.tcap, 4)), dim3(256), 0, stream, D, 1, gm);
      }
    } else if (forkUpdate && stream2) {
      (void)hipEventRecord(evFork, stream);
      (void)hipStreamWaitEvent(stream2, evFork, 0);
      KL("k_rank1", k_rank1, dim3(gx, gy), dim3(256), 0, stream2, D, parity);
      KL("k_minv_fix", k_minv_fix, dim3(1), dim3(256), 0, stream2, D, parity);
      (void)hipEventRecord(evJoin, stream2);
      sidePending = true;
      KL("k_primal_update", k_primal_update, dim3(gm), dim3(256), 0, stream, D, 0);
    } else {
      // one launch: primal update with the entering column + rank-1 sweep of the nucleus inverse
      KL("k_primal_rank1", k_primal_rank1, dim3(gm + gx * gy), dim3(256), 0, stream, D, parity, gm, gx, gy);
    }
  }
  // workgroup 0: fix-ups of the basis update, housekeeping, head of the next CHUZR; the others
  // scatter this pivot's new primal infeasibilities into the list
  if (wideRows) {
    // long columns: the row-copy partition moves of the leaving and the entering column, one thread
    // per entry over the whole chip (workgroup 0 of k_fix_house then skips them)
    KL("k_house_col", k_house_col, dim3(cdiv(maxColumnLength, 256)), dim3(256), 0, stream, D, 0);
    KL("k_house_col", k_house_col, dim3(cdiv(maxColumnLength, 256)), dim3(256), 0, stream, D, 1);
  }
  KL("k_fix_house", k_fix_house, dim3(2 + gm), dim3(256), 0, stream, D, parity, ((forkUpdate && stream2) || luActive) ? 0 : 1, wideRows ? 1 : 0);
  return 0;
}

// `checkEvery` pivots as one hipGraph (the chain has no host-visible decisions), replayed until
// the device control block leaves RUN.  Re-captured whenever a captured pointer or extent changes.
// capture `count` pivots of the chain into a graph; 0 on success (exec valid), 1 when capture / instantiation failed
int clpgpu_context::captureBatch(int count, hipGraph_t &g, hipGraphExec_t &exec)
{
  capturing = true;
  evUsed = 0;
  hipError_t e = hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal);
  if (e == hipSuccess) {
    for (int b = 0; b < count; b++)
      launchIteration(b == 0, b & 1);
    joinUpdateBranch();  // every forked branch rejoins the origin stream before the capture ends
    e = hipStreamEndCapture(stream, &g);
  }
  capturing = false;
  if (e == hipSuccess)
    e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (g)
      (void)hipGraphDestroy(g);
    g = nullptr;
    exec = nullptr;
    return 1;
  }
  return 0;
}

// one batch of pivots: check_every of them, or -- count in 1 .. check_every-1, a power of two -- a tail batch
int clpgpu_context::launchBatch(int count)
{
  if (count < 0 || count > checkEvery)
    count = checkEvery;
  if (!useGraph || timing) {
    for (int b = 0; b < count; b++) {
      if (timing >= 2)
        ktBegin(b);
      launchIteration(b == 0, b & 1);
    }
    joinUpdateBranch();
    return checkLaunches("launchIteration");
  }
  if (!graphExec || graphIterations != checkEvery || graphPriceMode != priceMode || graphFreeActive != freeActive) {
    // (the full-size graph is always built first, also when a tail batch is what runs now: a stepped run
    // that warms up with a few pivots must not pay for its instantiation later)
    dropGraph();
    if (captureBatch(checkEvery, graph, graphExec)) {
      dropGraph();
      useGraph = 0;  // fall back to eager launches of the same chain
      if (logLevel > 0)
        fprintf(stderr, "clpgpu: hipGraph capture of the pivot chain failed%s: eager launches from here on\n", commActive ? " (column-sharded run)" : "");
      sidePending = false;
      for (int b = 0; b < count; b++)
        launchIteration(b == 0, b & 1);
      joinUpdateBranch();
      return checkLaunches("launchIteration");
    }
    graphIterations = checkEvery;
    graphPriceMode = priceMode;
    graphFreeActive = freeActive;
  }
  hipGraphExec_t exec = graphExec;
  if (count < checkEvery) {
    int slot = 0;
    while ((1 << slot) < count)
      slot++;
    if ((1 << slot) != count || slot >= TAIL_SIZES) {
      setError("launchBatch: tail batch of %d pivots", count);
      return -99;
    }
    if (!tailExec[slot] && captureBatch(count, tailGraph[slot], tailExec[slot])) {
      // no graph for this size: the same pivots eagerly
      sidePending = false;
      for (int b = 0; b < count; b++)
        launchIteration(b == 0, b & 1);
      joinUpdateBranch();
      return checkLaunches("launchIteration");
    }
    exec = tailExec[slot];
  }
  hipError_t e = hipGraphLaunch(exec, stream);
  if (e != hipSuccess) {
    setError("hipGraphLaunch failed: %s", hipGetErrorString(e));
    return -99;
  }
  if (timing)
    evUsed = count;  // the event-record nodes of every captured pivot were replayed
  return 0;
}

int clpgpu_context::whileIterating(int stepTarget)
{
  // push the scalars the device needs for this run of iterations
  hCtrl->state = RUN;
  hCtrl->pendingState = RUN;
  hCtrl->stepLimit = stepTarget;
  hCtrl->saveSumDual = sumDualInfeasibilities;
  hCtrl->largestPrimalError = largestPrimalError;
  hCtrl->largestDualError = largestDualError;
  hCtrl->dualTolerance = dualTolerance;
  hCtrl->dualBound = dualBound;
  hCtrl->objectiveValue = objectiveValue;
  hCtrl->numberIterations = numberIterations;
  hCtrl->forceFactorization = forceFactorization;
  hCtrl->numberChanged = numberChanged;
  hCtrl->seed = seed;  // perturb() draws from the same generator on the host
  hCtrl->progressFlag = progressFlag & 3;
  hCtrl->primalTolerance = primalTolerance;
  hCtrl->lastBadIteration = lastBadIteration;
  hCtrl->maximumPivots = luActive ? luEtaLimit : maximumPivots;
  hCtrl->maximumIterations = maximumIterations;
  int rc = 0;
  hCtrl->presetRowPlus1 = 0;
  hCtrl->freeHold = 0;
  hCtrl->freeChosen = 0;
  hCtrl->badFree = 0.0;
  hCtrl->freeCount = 0;
  hCtrl->freeEntered = 0;
  int freeEnteredSeen = 0;
  if (freeNonbasic) {
    if (nranks > 1) {
      setError("option free_nonbasic is not available in column-sharded runs");
      return -99;
    }
    rc |= pushFreeList();
  } else {
    freeActive = false;
  }
  rc |= pushCtrl();
  if (timing && evStart.empty()) {
    evStart.resize(checkEvery);
    evStop.resize(checkEvery);
    evMid.resize(checkEvery);
    for (int i = 0; i < checkEvery; i++) {
      (void)hipEventCreate(&evStart[i]);
      (void)hipEventCreate(&evStop[i]);
      (void)hipEventCreate(&evMid[i]);
    }
  }
  while (!rc) {
    evUsed = 0;
    double launchesBefore = hCtrl->statPriceLaunches;
    const double denseBefore = hCtrl->statDensePi;
    const int logBefore = hCtrl->logCount;
    // a stepped run (clpgpu_dual_steps) ends on the pivot asked for: the last batches are 8, 4, 2, 1 pivots
    // long instead of a full batch that idles through the pivots behind the limit
    int count = checkEvery;
    if (stepTarget >= 0) {
      const int remaining = stepTarget - hCtrl->numberIterations;
      if (remaining <= 0) {
        hCtrl->state = EXIT_STEP_LIMIT;
        break;
      }
      if (remaining < checkEvery) {
        count = 1;
        while (count * 2 <= remaining && count * 2 < (1 << clpgpu_context::TAIL_SIZES))
          count *= 2;
      }
    }
    if (freeNonbasic && firstFree >= 0) {
      // dualRow's free-first entry (src/ClpSimplexDual.cpp:3005-3055) runs on the host, one pivot at a time: the next free column
      // with a reduced cost worth having is FTRANned and the row it should pivot on is handed to CHUZR (k_chuzr_pre, chuzrFinalBody);
      // the pivot's tail leaves the head of the next CHUZR alone (freeHold) until firstFree_ is used up
      int chosenRow = -1;
      rc |= freeFirstRow(chosenRow);
      hCtrl->presetRowPlus1 = chosenRow + 1;
      hCtrl->freeHold = 1;
      numberFreeFirstRows += chosenRow >= 0;
      rc |= pushCtrl();
      count = 1;
    } else if (hCtrl->freeHold) {
      hCtrl->freeHold = 0;
      rc |= pushCtrl();
    }
    rc |= launchBatch(count);
    rc |= pullCtrl();
    numberFreeEntered += hCtrl->freeEntered - freeEnteredSeen;  // (the device's count since whileIterating began)
    freeEnteredSeen = hCtrl->freeEntered;
    if (!rc && hCtrl->state == RUN && stepTarget >= 0 && hCtrl->numberIterations >= stepTarget)
      hCtrl->state = EXIT_STEP_LIMIT;  // the batch ended exactly on the limit: the device never saw a pivot beyond it
    if (!rc && hCtrl->dcWide < 0) {
      // k_dual_column_wide's grid barrier needs its 128 workgroups resident together; when another context of this process holds
      // the CUs (clones in threads, the forked update branch) a launch can starve until the bounded spin gives up.  The pivot was
      // abandoned before anything was written (no pivot, no flips): from here on this context keeps long lists in the
      // single-workgroup walk (dc_wide 0) and the solve goes on through a status check, as after any abandoned pivot.
      if (logLevel > 0)
        fprintf(stderr, "clpgpu: k_dual_column_wide: grid barrier timed out at iteration %d -- dc_wide 0 from here on\n", hCtrl->numberIterations);
      numberDcWideTimeouts++;
      dcWide = 0;
      dropGraph();
      hCtrl->dcWide = 0;
      hCtrl->pivotRow = -1;
      hCtrl->state = EXIT_REFACTOR;
      rc |= pushCtrl();
    }
    if (!rc && jdsReady && priceLds) {
      // which pricing form the next batch's chain carries.  A dense-pi pivot costs ~52 us in k_price_sell and ~32 in k_price_lds,
      // a sparse-pi one ~12 us by row and ~32 in k_price_lds: the forms break even at half the pivots dense.  k_price_lds from
      // 55 % dense on, back below 30 % (both forms give the same bits for any pi: the choice never changes a pivot)
      // (counted over at least 16 pricing launches: a stepped or timed run may come in batches of one pivot)
      modePriced += hCtrl->statPriceLaunches - launchesBefore;
      modeDense += hCtrl->statDensePi - denseBefore;
      if (priceLds >= 2)
        priceMode = 1;
      else if (modePriced >= 16.0) {
        const int before = priceMode;
        if (priceMode == 0 && modeDense >= 0.55 * modePriced)
          priceMode = 1;
        else if (priceMode == 1 && modeDense <= 0.30 * modePriced)
          priceMode = 0;
        modePriced = modeDense = 0.0;
        priceFormSwitches += priceMode != before;
      }
    }
    if (timing) {
      // the device counts a pricing launch only while the loop is live; those are the first ones
      int timed = (int)(hCtrl->statPriceLaunches - launchesBefore);
      if (timed > evUsed)
        timed = evUsed;
      // which form each of those pivots priced with: bit 30 of its log record
      std::vector<PivotRecord> recs(timed > 0 ? timed : 1);
      const int firstRec = logBefore;
      const bool haveRecs = timed > 0 && firstRec + timed <= logCapacity && !d2h(recs.data(), D.log + firstRec, timed);
      for (int i = 0; i < timed; i++) {
        float ms = 0.0f;
        const bool byRow = haveRecs && ((recs[i].reserved >> 30) & 1);
        // by column: the pricing kernel alone (what rocprofv3's kernel duration is compared with); by row: both passes
        if (hipEventElapsedTime(&ms, evStart[i], byRow ? evStop[i] : evMid[i]) == hipSuccess) {
          if (byRow) {
            stats.row_ms += ms;
            stats.row_launches++;
          } else {
            stats.price_ms += ms;
            stats.price_launches++;
          }
        }
      }
      if (timing >= 2)
        ktCollect((int)(hCtrl->statPriceLaunches - launchesBefore));
    }
    if (hCtrl->state != RUN)
      break;
  }
  if (rc)
    return rc;
  // pull scalars back
  numberIterations = hCtrl->numberIterations;
  objectiveValue = hCtrl->objectiveValue;
  pivots = hCtrl->pivots;
  forceFactorization = hCtrl->forceFactorization;
  numberChanged = hCtrl->numberChanged;
  acceptablePivot = hCtrl->acceptablePivotBase;
  seed = hCtrl->seed;
  progressFlag = (progressFlag & ~3) | (hCtrl->progressFlag & 3);
  const int state = hCtrl->state;
  lastExitState = state;
  if (logLevel > 2)
    fprintf(stderr, "clpgpu: loop left with state %d at iteration %d: pivotRow %d sequenceOut %d sequenceIn %d, %d pivots since the factorization, %d candidates, best possible pivot %g, acceptable %g\n",
            state, numberIterations, hCtrl->pivotRow, hCtrl->sequenceOut, hCtrl->sequenceIn, pivots, hCtrl->numberCandidates, hCtrl->bestPossible, hCtrl->acceptablePivot);
  const int g = cdiv(m, 256);
  if (state != EXIT_REFACTOR && state != EXIT_STEP_LIMIT && state != EXIT_MAX_ITERATIONS) {
    // the pivot was abandoned part-way: k_house did not clear the sparse work vectors
    hipLaunchKernelGGL(k_zero, dim3(g), dim3(256), 0, stream, D.vecC, m);
    hipLaunchKernelGGL(k_zero, dim3(g), dim3(256), 0, stream, D.vecV1, m);
    hipLaunchKernelGGL(k_zero, dim3(g), dim3(256), 0, stream, D.flipRhs, m);
    hipLaunchKernelGGL(k_zero, dim3(cdiv(kcap, 256)), dim3(256), 0, stream, D.slotV1, kcap);
    hipLaunchKernelGGL(k_zero, dim3(cdiv(kcap, 256)), dim3(256), 0, stream, D.flipSlot, kcap);
    hipLaunchKernelGGL(k_zero, dim3(cdiv(n, 256)), dim3(256), 0, stream, D.alphaCol, n);
  }
  lastReturnCode = -1;
  exitScheduled += state == EXIT_REFACTOR;
  exitAlphaCheck += state == EXIT_ALPHA_CHECK;
  exitBackwards += state == EXIT_BACKWARDS;
  exitBadUpdate += state == EXIT_BAD_UPDATE;
  switch (state) {
  case EXIT_STEP_LIMIT:
    // the pivot the run stopped on may have asked for a refactorization (houseBody): it is done when the run resumes,
    // so that the basis handed back is the un-refactorized one (ClpSimplex::housekeeping returns on the iteration limit
    // first, src/ClpSimplex.cpp:2391) and a resumed run continues exactly like an unstepped one
    stepPendingRefactor = hCtrl->pendingState == EXIT_REFACTOR;
    return 1;
  case EXIT_REFACTOR:
    problemStatus = -2;
    break;
  case EXIT_MAX_ITERATIONS:
    problemStatus = 3;
    lastReturnCode = 3;
    break;
  case EXIT_ALPHA_CHECK: {
    // :1451-1500
    hipLaunchKernelGGL(k_unroll_weights, dim3(g), dim3(256), 0, stream, D);
    if (pivots) {
      problemStatus = -2;
    } else {
      rc |= pullRim(true);
      status[hCtrl->sequenceOut] |= FLAGGED_BIT;
      progBadTimes = 0;  // progress_.clearBadTimes(), :1487
      rc |= pushRim();
      lastBadIteration = numberIterations;
      if (fabs(hCtrl->alpha) < 1.0e-10 && fabs(hCtrl->btranAlpha) < 1.0e-8 && numberIterations > 100)
        problemStatus = 1;
      else
        problemStatus = -2;  // the reference `continue`s with the same factorization; a refresh is equivalent here
    }
    break;
  }
  case EXIT_BACKWARDS:
    hipLaunchKernelGGL(k_unroll_weights, dim3(g), dim3(256), 0, stream, D);
    problemStatus = -2;
    break;
  case EXIT_SHARD_OVERFLOW: {
    // a rank's candidate or flip records outgrew the exchange buffer (every rank sees the same headers, so all of them arrive
    // here at the same pivot): the pivot is abandoned and the refactorization + resync that follows recomputes every reduced
    // cost (the non-owned ones were stale) and the basic solution.  First the buffers grow -- once, to what a shard can ever
    // need: every column of the shard a candidate (in the mature regime of config 4 half of them are: 12 500 per rank at
    // eight ranks against the start-up buffer of 2048), a quarter of them flips -- and the list exchange carries on; only
    // if the full-size buffers overflow too (they cannot) or cannot be had does the run continue with the dense row exchange.
    const int fullCand = std::max(shardChunk, 256), fullFlip = std::max(shardChunk / 4, 512);
    if (commMode == 2 && shardGrow && (shardCandCap < fullCand || shardFlipCap < fullFlip)) {
      shardCandCap = std::max(shardCandCap, fullCand);
      shardFlipCap = std::max(shardFlipCap, fullFlip);
      dropGraph();
      if (!allocShardBuffers()) {
        if (logLevel > 0)
          fprintf(stderr, "clpgpu: rank %d: exchange buffer overflow at iteration %d, buffers grown to %d candidates / %d flips per rank\n", rank,
                  numberIterations, shardCandCap, shardFlipCap);
        problemStatus = -2;
        break;
      }
    }
    commMode = 1;
    D.firstColumn = 0;
    D.lastColumn = n;
    if (D.sellWindowed)
      D.sellWinBase = D.priceFirst / PRICE_BLOCK;  // the windows stay aligned (shard starts are multiples of PRICE_BLOCK); their block numbers move
    dropGraph();
    if (logLevel > 0)
      fprintf(stderr, "clpgpu: rank %d: exchange buffer overflow at iteration %d, falling back to the dense row exchange\n", rank, numberIterations);
    problemStatus = -2;
    break;
  }
  case EXIT_BAD_UPDATE: {
    // updateStatus == 2 (:1618-1659)
    hipLaunchKernelGGL(k_unroll_weights, dim3(g), dim3(256), 0, stream, D);
    if (pivots) {
      problemStatus = -2;
    } else {
      rc |= pullRim(true);
      status[hCtrl->sequenceOut] |= FLAGGED_BIT;
      progBadTimes = 0;  // :1646
      rc |= pushRim();
      lastBadIteration = numberIterations;
      problemStatus = -2;
    }
    break;
  }
  case EXIT_NO_INCOMING: {
    // no incoming column is valid (:1869-2079)
    // "spareIntArray_[3] = pivotRow_; pivotRow_ = -1;" (:1874-1875): the row is NOT the "last pivot row" of the next CHUZR
    // (ClpDualRowSteepest::pivotRow makes that one a last-resort choice, src/ClpDualRowSteepest.cpp:224-232); found by the GPU fuzz
    // (tools/fuzz_gpu.py: one pivot more than the oracle on infeasible LPs)
    hCtrl->pivotRow = -1;
    problemStatus = -2;
    // "if (sequenceIn_ < 0 && acceptablePivot <= acceptablePivot_) if (!pivots) problemStatus_ = 1" (:1328)
    if (hCtrl->acceptablePivot <= acceptablePivot && !pivots)
      problemStatus = 1;
    if (pivots < 2 && acceptablePivot <= 1.0e-8 && acceptablePivot > 0.0) {
      rc |= pullRim(true);
      double dualTest = 1.0e13;
      if (!numberAtFakeBound())
        dualTest = 0.0;
      if (hCtrl->bestPossible < 1.0e-11 && dualBound > dualTest) {
        // "say infeasible ... unless primal feasible!!!!" (:1982-2027, specialOptions_ 0): the sums are those of the last status
        // check; the -4 alternative (:1999) needs more than two pivots since the factorization and cannot be taken here.
        // This 10 is whileIterating's own and reaches every caller in the reference, fastDual and strong branching included
        // (they get it from ClpSimplexDual::whileIterating as well), so it is NOT behind option fake_bound_cleanup the way
        // ClpSimplex::dual's rewrite of a finished 1 is (src/ClpSimplex.cpp:5800-5803).  The reference also
        // replaces the objective by a zero one here ("Get rid of objective", :2022-2024: the primal that follows then only
        // looks for feasibility); the engine has no primal behind it and keeps its costs -- a caller that finishes a 10
        // with Clp's primal (the clpGpuDual adapter) hands it the model's own objective, as after any other 10.
        problemStatus = 1;
        if (sumPrimalInfeasibilities < 1.0e-3 || sumDualInfeasibilities > 1.0e-5)
          problemStatus = 10;
      } else if (pivots == 0) {
        problemStatus = -4;
      }
    }
    acceptablePivot = fabs(acceptablePivot);
    if (pivots < 5 && acceptablePivot > 1.0e-8)
      acceptablePivot = 1.0e-8;
    hCtrl->acceptablePivotBase = acceptablePivot;
    lastReturnCode = (problemStatus == 1 || problemStatus == 10) ? 1 : -2;
    break;
  }
  case EXIT_NO_PIVOT_ROW: {
    // no pivot row (:2080-2331)
    lastReturnCode = pivots ? -2 : 0;
    if (!pivots) {
      rc |= pullRim(true);
      problemStatus = -1;
      if (numberPrimalInfeasibilities)
        problemStatus = -4;
      bool anyFlagged = false;
      for (int iRow = 0; iRow < m && !anyFlagged; iRow++)
        anyFlagged = (status[pivotVariable[iRow]] & FLAGGED_BIT) != 0;
      int fake = 0;
      for (int i = 0; i < N; i++)
        fake += getFake(status[i]) != FAKE_NONE;
      numberFake = fake;
      if (numberFake || numberDualInfeasibilities) {
        problemStatus = -5;
      } else if (anyFlagged) {
        problemStatus = -5;
      } else {
        problemStatus = 0;
        numberPrimalInfeasibilities = 0;
        sumPrimalInfeasibilities = 0.0;
        numberDualInfeasibilities = 0;
        sumDualInfeasibilities = 0.0;
        if (perturbation == 101 || numberChanged) {
          numberChanged = 0;
          perturbation = 102;  // :2222-2224
          restoreCosts();
          progObjective[PROGRESS - 1] = -DBL_MAX;
          rc |= pushRim();
          rc |= gutsOfSolution();
          if (numberDualInfeasibilities)
            problemStatus = 10;
        }
      }
    } else {
      problemStatus = -3;
      int half = (pivots + 1) >> 1;
      if (forceFactorization < 0 || half < forceFactorization)
        forceFactorization = half;
    }
    break;
  }
  default:
    setError("unexpected device state %d", state);
    problemStatus = 4;
    break;
  }
  return rc;
}

void clpgpu_context::finish()
{
  if (rimInfeasible)
    return;  // start-up found crossing bounds: nothing was put on the device
  pullRim(true);
  if (problemStatus == 0 || problemStatus == 3 || problemStatus == 10) {  // 10: the point a primal clean-up would start from
    double objective = 0.0;
    for (int j = 0; j < n; j++)
      objective += obj[j] * sol[j];
    objectiveValue = objective;
  }
}

int clpgpu_context::run(int maxSteps)
{
  auto t0 = std::chrono::steady_clock::now();
  int rc = 0;
  if (!started) {
    stepPendingRefactor = false;
    rc = startup();
    if (rc)
      return problemStatus = 4;
  }
  int stepTarget = (maxSteps < 0) ? -1 : numberIterations + maxSteps;
  int result = -1;
  if (stepPendingRefactor) {
    // the refactorization the last stepped run stopped in front of (see whileIterating, EXIT_STEP_LIMIT)
    stepPendingRefactor = false;
    if (problemStatus < 0 && !needStatus) {
      lastExitState = EXIT_REFACTOR;
      problemStatus = -2;
      needStatus = true;
    }
  }
  while (problemStatus < 0) {
    if (needStatus && perturbation < 101 && numberIterations > 2 * (m + n) && !fastDualMode) {
      // "if getting nowhere - why not give it a kick" (gutsOfDual :488-492)
      rc = pullRim(true);
      perturb();
      rc |= h2d(D.cost, cost.data(), N);
      rc |= gutsOfSolution();
      if (rc) {
        problemStatus = 4;
        break;
      }
      if (logLevel > 0)
        fprintf(stderr, "clpgpu: iteration %d > 2(m+n): costs perturbed\n", numberIterations);
    }
    if (needStatus) {
      const auto ts0 = std::chrono::steady_clock::now();
      rc = statusOfProblemInDual(factorType);
      if (logLevel > 1) {
        (void)sync();
        fprintf(stderr, "clpgpu: status check at iteration %d took %.1f ms in all (refactorization, resync, weights, bounds)\n", numberIterations,
                1.0e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count());
      }
      factorType = 1;
      needStatus = false;
      if (rc) {
        if (problemStatus < 0)
          problemStatus = 4;
        break;
      }
      if (!fastDualMode) {
        // "problems - try primal" (gutsOfDual :533-547): the primal infeasibilities have grown 1e5-fold since the smallest sum seen while
        // the objective stood still -- and either the last two recorded objectives say the solve fell off a cliff or the growth is
        // 1e10-fold (what a runaway escalation of the dual bound looks like): status 10, the caller's primal takes over
        if (objectiveValue > 1.0e-4 + 1.0e-9 * fabs(lastObjectiveValueGuts) + lastObjectiveValueGuts)
          smallestPrimalInfeasibility = DBL_MAX;  // reset smallest
        smallestPrimalInfeasibility = std::min(smallestPrimalInfeasibility, sumPrimalInfeasibilities);
        lastObjectiveValueGuts = objectiveValue;
        if (sumPrimalInfeasibilities > 1.0e5 && sumPrimalInfeasibilities > 1.0e5 * smallestPrimalInfeasibility
            && ((progObjective[PROGRESS - 1] < -1.0e10 && -progObjective[PROGRESS - 2] > -1.0e5)
                || sumPrimalInfeasibilities > 1.0e10 * smallestPrimalInfeasibility)
            && problemStatus < 0 && tryPrimal) {
          problemStatus = 10;
          sumPrimalInfeasibilities = -123456789.0;  // mark as large infeasibility cost wanted
          numberTryPrimal++;
          if (logLevel > 0)
            fprintf(stderr, "clpgpu: iteration %d: primal infeasibilities running away, status 10 (try primal)\n", numberIterations);
        }
      }
      if (problemStatus >= 0)
        break;
    }
    problemStatus = -1;
    rc = whileIterating(stepTarget);
    if (rc == 1) {
      result = -1;
      auto t1 = std::chrono::steady_clock::now();
      seconds += std::chrono::duration<double>(t1 - t0).count();
      return result;
    }
    if (rc) {
      problemStatus = 4;
      break;
    }
    if (fastDualMode && problemStatus < 0 && ((fastDualMode == 2 && lastReturnCode < 0) || lastReturnCode == 3)) {
      problemStatus = 3;  // :7422-7431
      break;
    }
    needStatus = true;
  }
  finish();
  // ClpSimplex::dual's own second thought (src/ClpSimplex.cpp:5800-5803): an "infeasible" reached with fake bounds active is
  // "clean up in primal as fake bounds" -- status 10.  Not in fastDual / strong branching, which do not go through ClpSimplex::dual.
  // A caller with a primal behind it (the clpGpuDual adapter, which finishes a 10 with model.primal(1)) asks for this with
  // option "fake_bound_cleanup" 1; a bare context has no primal and reports the 1 it found.
  if (problemStatus == 1 && fakeBoundCleanup && !rimInfeasible && !fastDualMode && numberAtFakeBound() > 0)
    problemStatus = 10;
  auto t1 = std::chrono::steady_clock::now();
  seconds += std::chrono::duration<double>(t1 - t0).count();
  return problemStatus;
}

// ---------------------------------------------------------------------------------------------
// plug-in level pricing call (ClpMatrixBase::transposeTimes surface): host arrays in and out
// ---------------------------------------------------------------------------------------------
int clpgpu_context::priceRow(int numberPi, const int *piIndex, const double *piValue, const unsigned char *st,
                             const double *djv, double zeroTol, double dualTol, double accPivot, int *numberOut,
                             int *outIndex, double *outValue, int *numberCand, int *candIndex, double *candValue,
                             double *upperTheta)
{
  std::vector<double> rho(m, 0.0), piNeg(m, 0.0);
  for (int i = 0; i < numberPi; i++) {
    rho[piIndex[i]] = piValue[i];
    piNeg[piIndex[i]] = -piValue[i];
  }
  int rc = 0;
  rc |= h2d(D.rho, rho.data(), m);
  rc |= h2d(D.piNeg, piNeg.data(), m);
  rc |= h2d(D.status, st, N);
  rc |= h2d(D.dj, djv, N);
  // only the scalars the pricing kernels read are set; the factorization state kept in the control
  // block (k, kcap, pivots, zeroTolerance: a ClpGpuPackedMatrix and a CoinGpuFactorization adapter
  // share one context) is left alone and the borrowed fields are put back afterwards
  const int saveState = hCtrl->state;
  const double saveDualTol = hCtrl->dualTolerance, saveZeroTol = hCtrl->zeroTolerance, saveAcc = hCtrl->acceptablePivot;
  hCtrl->state = RUN;
  hCtrl->dualTolerance = dualTol;
  hCtrl->zeroTolerance = zeroTol;
  hCtrl->acceptablePivot = accPivot;
  hCtrl->numberCandidates = 0;
  hCtrl->upperTheta = 1.0e31;
  rc |= pushCtrl();
  const int nbRows = cdiv(m, PRICE_BLOCK);
  const int nbCols = cdiv(D.lastColumn - D.firstColumn, PRICE_BLOCK);
  const int nb = nbRows + nbCols;
  hipLaunchKernelGGL(k_zero, dim3(cdiv(n, 256)), dim3(256), 0, stream, D.alphaCol, n);
  const int nSlots = nSellBlocks + nLongBlocks;
  // the reference's switch (src/ClpPackedMatrix.cpp:727-754) with this engine's crossover: option row_price_frac
  const bool byRow = priceKernel >= 1 && rowPriceFrac > 0.0 && (double)numberPi <= rowPriceFrac * m && nSlots > 0 && !widePricing;
  if (byRow) {
    // by row: row part + bitmap + neutral per-block values, the two passes, then the self-scanning compaction
    hipLaunchKernelGGL(k_price_row_init, dim3(nbRows), dim3(PRICE_BLOCK), 0, stream, D, nbCols, nSlots);
    hipLaunchKernelGGL(k_price_sell, dim3(nSlots + nbRows), dim3(256), 0, stream, D, 1, 1, nSellBlocks, nSlots, m + 1, 1);
    hipLaunchKernelGGL(k_price_row_finish, dim3(nbCols), dim3(PRICE_BLOCK), 0, stream, D, nbRows, m + 1);
    hipLaunchKernelGGL(k_cand_scatter, dim3(nb), dim3(PRICE_BLOCK), 0, stream, D, nbRows, nSlots);
  } else if (priceKernel >= 1) {
    if (nSlots > 0) {
      if (jdsReady && priceLds && 12LL * numberPi >= (long long)m) {
        // dense pi: the form the iteration chain takes then (pi tiles in LDS); long columns keep their own workgroups
        std::vector<unsigned long long> bitsHost((size_t)((m + 63) / 64) + 8, 0ull);
        for (int i = 0; i < numberPi; i++)
          if (piValue[i] != 0.0)
            bitsHost[piIndex[i] >> 6] |= 1ull << (piIndex[i] & 63);
        rc |= h2d(D.piBits, bitsHost.data(), bitsHost.size());
        hipLaunchKernelGGL(k_price_lds, dim3(priceLdsGrid), dim3(PL_THREADS), priceLdsBytes, stream, D, 0);
        if (nLongBlocks > 0)
          hipLaunchKernelGGL(k_price_sell, dim3(nSlots), dim3(256), 0, stream, D, 1, 0, nSellBlocks, nSlots, 0, 0, 32);
      } else {
        hipLaunchKernelGGL(k_price_sell, dim3(nSlots), dim3(256), 0, stream, D, 1, 0, nSellBlocks);
      }
    }
    hipLaunchKernelGGL(k_cand_count, dim3(nb), dim3(PRICE_BLOCK), 0, stream, D, nbRows, 0, nSlots);
  } else {
    hipLaunchKernelGGL(k_price, dim3(nb), dim3(PRICE_BLOCK), 0, stream, D, nbRows);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, stream, D, nb, 0, 1, 0);
  }
  if (!byRow)
    hipLaunchKernelGGL(k_cand_scatter, dim3(nb), dim3(PRICE_BLOCK), 0, stream, D, nbRows);
  rc |= checkLaunches("clpgpu_price_row");
  rc |= pullCtrl();
  std::vector<double> alpha(n);
  rc |= d2h(alpha.data(), D.alphaCol, n);
  int count = 0;
  for (int j = 0; j < n; j++)
    if (alpha[j] != 0.0) {
      outIndex[count] = j;
      outValue[count++] = alpha[j];
    }
  *numberOut = count;
  *numberCand = hCtrl->numberCandidates;
  *upperTheta = hCtrl->upperTheta;
  if (hCtrl->numberCandidates) {
    rc |= d2h(candIndex, D.candSeq, hCtrl->numberCandidates);
    rc |= d2h(candValue, D.candAlpha, hCtrl->numberCandidates);
  }
  // leave the work vectors clean
  hipLaunchKernelGGL(k_zero, dim3(cdiv(m, 256)), dim3(256), 0, stream, D.rho, m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(m, 256)), dim3(256), 0, stream, D.piNeg, m);
  hCtrl->state = saveState;
  hCtrl->dualTolerance = saveDualTol;
  hCtrl->zeroTolerance = saveZeroTol;
  hCtrl->acceptablePivot = saveAcc;
  rc |= pushCtrl();
  rc |= sync();
  return rc;
}

// Development hook: the by-column pricing kernel as the chain launches it (variant 6, dense pi), timed with HIP events on the
// engine's stream between two pivots, with parts of the kernel switched off by `masks` (see priceSellBody).  Touches only the
// per-pivot scratch pricing rewrites on every pivot (tableau row, candidate flags, block counts) and pi, which it clears again.
int clpgpu_context::debugPriceBench(int reps, int numberMasks, const int *masks, double *microseconds)
{
  if (!started || !D.colStart || nSellBlocks <= 0 || reps <= 0)
    return -1;
  int rc = sync();
  const int nbRows = cdiv(m, PRICE_BLOCK);
  const int nbCols = cdiv(D.lastColumn - D.firstColumn, PRICE_BLOCK);
  const int nb = nbRows + nbCols;
  const int nSlots = nSellBlocks + nLongBlocks;
  const bool countInPrice = nb > 256;
  const size_t words = (size_t)(m + 63) / 64 + 8;
  std::vector<unsigned long long> ones(words, ~0ull);
  rc |= h2d(D.piBits, ones.data(), words);
  hipLaunchKernelGGL(k_fill, dim3(cdiv(m, 256)), dim3(256), 0, stream, D.piNeg, 0.37, m);
  const int saveState = hCtrl->state;
  hCtrl->state = RUN;
  rc |= pushCtrl();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const size_t lds = (m > 64 * SELL_BITS_MAX) ? 0 : (size_t)((m + 63) / 64) * 8;
  for (int v = 0; v < numberMasks && !rc; v++) {
    for (int r = 0; r < reps + 2; r++) {
      if (r == 2)
        (void)hipEventRecord(e0, stream);
      if (masks[v] & (1 << 20)) {
        // bit 20: the dense-pi form of the chain (pi tiles in LDS) on this context's column range, when it was laid out
        if (!jdsReady) {
          microseconds[v] = -1.0;
          break;
        }
        hipLaunchKernelGGL(k_price_lds, dim3(priceLdsGrid), dim3(PL_THREADS), priceLdsBytes, stream, D, countInPrice ? 1 : 0);
      } else {
        hipLaunchKernelGGL(k_price_sell, dim3(nSlots), dim3(256), lds, stream, D, (m > 64 * SELL_BITS_MAX) ? 1 : 6, countInPrice ? 1 : 0, nSellBlocks,
                           nSlots, 0, 0, masks[v]);
      }
    }
    if ((masks[v] & (1 << 20)) && !jdsReady)
      continue;
    (void)hipEventRecord(e1, stream);
    rc |= sync();
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    microseconds[v] = 1.0e3 * ms / reps;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  std::vector<unsigned long long> zeros(words, 0ull);
  rc |= h2d(D.piBits, zeros.data(), words);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(m, 256)), dim3(256), 0, stream, D.piNeg, m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(n, 256)), dim3(256), 0, stream, D.alphaCol, n);
  (void)hipMemsetAsync(D.blockCount, 0, sizeof(int) * (size_t)nb, stream);
  (void)hipMemsetAsync(D.candFlag, 0, (size_t)N, stream);
  hCtrl->state = saveState;
  rc |= pushCtrl();
  rc |= sync();
  return rc;
}

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int clpgpu_debug_price_bench(clpgpu_context *ctx, int reps, int numberMasks, const int *masks, double *microseconds)
{
  if (!ctx || !masks || !microseconds)
    return -1;
  return ctx->debugPriceBench(reps, numberMasks, masks, microseconds);
}

clpgpu_context *clpgpu_create(int device)
{
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    fprintf(stderr, "clpgpu_create: no HIP device available (this library has no CPU fallback)\n");
    return nullptr;
  }
  if (device < 0 || device >= count)
    return nullptr;
  if (hipSetDevice(device) != hipSuccess)
    return nullptr;
  clpgpu_context *ctx = new clpgpu_context();
  ctx->device = device;
  if (hipStreamCreate(&ctx->stream) != hipSuccess) {
    delete ctx;
    return nullptr;
  }
  if (hipStreamCreate(&ctx->stream2) != hipSuccess || hipEventCreateWithFlags(&ctx->evFork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->evJoin, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    ctx->stream2 = nullptr;  // the basis update then stays on the main stream
  }
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  memset(&ctx->D, 0, sizeof(ctx->D));
  return ctx;
}

void clpgpu_destroy(clpgpu_context *ctx)
{
  if (!ctx)
    return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->dropGraph();
  if (ctx->stream2) {
    (void)hipStreamSynchronize(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream2);
  }
  if (ctx->evFork)
    (void)hipEventDestroy(ctx->evFork);
  if (ctx->evJoin)
    (void)hipEventDestroy(ctx->evJoin);
  if (ctx->comm && ctx->ncclCommDestroyFn)
    ctx->ncclCommDestroyFn(ctx->comm);
  if (ctx->blasHandle) {
    typedef int (*destroy_t)(void *);
    void *lib = rocblasLibrary();
    destroy_t destroyFn = lib ? (destroy_t)dlsym(lib, "rocblas_destroy_handle") : nullptr;
    if (destroyFn)
      destroyFn(ctx->blasHandle);
  }
  for (void *p : ctx->allocations)
    (void)hipFree(p);
  if (ctx->hCtrl)
    (void)hipHostFree(ctx->hCtrl);
  for (auto &e : ctx->evStart)
    (void)hipEventDestroy(e);
  for (auto &e : ctx->evStop)
    (void)hipEventDestroy(e);
  for (auto &e : ctx->evMid)
    (void)hipEventDestroy(e);
  for (auto &e : ctx->ktEvents)
    (void)hipEventDestroy(e);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *clpgpu_last_error(const clpgpu_context *ctx) { return ctx ? ctx->error.c_str() : "null context"; }
void *clpgpu_stream(clpgpu_context *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int clpgpu_load_problem(clpgpu_context *ctx, int numberRows, int numberColumns, const int *columnStart, const int *rowIndex,
                        const double *element, const double *columnLower, const double *columnUpper, const double *objective,
                        const double *rowLower, const double *rowUpper)
{
  if (!ctx)
    return -99;
  (void)hipSetDevice(ctx->device);
  return ctx->loadProblem(numberRows, numberColumns, columnStart, rowIndex, element, columnLower, columnUpper, objective,
                          rowLower, rowUpper);
}

int clpgpu_set_column_range(clpgpu_context *ctx, int firstColumn, int lastColumn)
{
  if (!ctx || firstColumn < 0 || lastColumn > ctx->n || firstColumn > lastColumn)
    return -1;
  ctx->D.firstColumn = ctx->D.priceFirst = firstColumn;
  ctx->D.lastColumn = ctx->D.priceLast = lastColumn;
  int rc = ctx->buildSell();
  return rc;
}

int clpgpu_times(clpgpu_context *ctx, double scalar, const double *x, double *y)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  if (ctx->rejectScaled("clpgpu_times"))
    return -3;
  // needs the row copy in its load-time state or any partition: order inside a row does not matter
  int rc = 0;
  rc |= ctx->h2d(ctx->D.alphaCol, x, ctx->n);
  rc |= ctx->h2d(ctx->D.x3, y, ctx->m);
  hipLaunchKernelGGL(k_times, dim3(cdiv(ctx->m, 256)), dim3(256), 0, ctx->stream, ctx->D, scalar, (const double *)ctx->D.alphaCol,
                     ctx->D.x3);
  rc |= ctx->d2h(y, ctx->D.x3, ctx->m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(ctx->m, 256)), dim3(256), 0, ctx->stream, ctx->D.x3, ctx->m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(ctx->n, 256)), dim3(256), 0, ctx->stream, ctx->D.alphaCol, ctx->n);
  rc |= ctx->sync();
  return rc;
}

int clpgpu_transpose_times(clpgpu_context *ctx, double scalar, const double *x, double *y)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  if (ctx->rejectScaled("clpgpu_transpose_times"))
    return -3;
  int rc = 0;
  rc |= ctx->h2d(ctx->D.x3, x, ctx->m);
  rc |= ctx->h2d(ctx->D.alphaCol, y, ctx->n);
  hipLaunchKernelGGL(k_transpose_times, dim3(cdiv(ctx->n, 256)), dim3(256), 0, ctx->stream, ctx->D, scalar,
                     (const double *)ctx->D.x3, ctx->D.alphaCol);
  rc |= ctx->d2h(y, ctx->D.alphaCol, ctx->n);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(ctx->m, 256)), dim3(256), 0, ctx->stream, ctx->D.x3, ctx->m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(ctx->n, 256)), dim3(256), 0, ctx->stream, ctx->D.alphaCol, ctx->n);
  rc |= ctx->sync();
  return rc;
}

int clpgpu_price_row(clpgpu_context *ctx, int numberPi, const int *piIndex, const double *piValue, const unsigned char *status,
                     const double *dj, double zeroTolerance, double dualTolerance, double acceptablePivot, int *numberOut,
                     int *outIndex, double *outValue, int *numberCandidates, int *candIndex, double *candValue,
                     double *upperTheta)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  if (ctx->rejectScaled("clpgpu_price_row"))
    return -3;
  return ctx->priceRow(numberPi, piIndex, piValue, status, dj, zeroTolerance, dualTolerance, acceptablePivot, numberOut, outIndex,
                       outValue, numberCandidates, candIndex, candValue, upperTheta);
}

int clpgpu_factorize(clpgpu_context *ctx, const unsigned char *status, int *pivotVariable)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  if (ctx->rejectScaled("clpgpu_factorize"))
    return -3;
  ctx->status.assign(status, status + ctx->N);
  ctx->rebuildRowCopy = true;
  if (!ctx->hCtrl->zeroTolerance)
    ctx->hCtrl->zeroTolerance = ctx->zeroTolerance;
  int rc = ctx->pushCtrl();
  rc |= ctx->h2d(ctx->D.status, ctx->status.data(), ctx->N);
  if (rc)
    return rc;
  rc = ctx->factorize();
  if (!rc) {
    rc = ctx->pushCtrl();
    if (pivotVariable)
      memcpy(pivotVariable, ctx->pivotVariable.data(), sizeof(int) * (size_t)ctx->m);
  }
  return rc;
}

int clpgpu_ftran(clpgpu_context *ctx, double *region)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  if (ctx->rejectScaled("clpgpu_ftran"))
    return -3;
  int rc = ctx->h2d(ctx->D.vecV2, region, ctx->m);
  ctx->ftranDevice(ctx->D.vecV2, ctx->D.x3);
  rc |= ctx->d2h(region, ctx->D.x3, ctx->m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(ctx->m, 256)), dim3(256), 0, ctx->stream, ctx->D.vecV2, ctx->m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(ctx->m, 256)), dim3(256), 0, ctx->stream, ctx->D.x3, ctx->m);
  rc |= ctx->sync();
  return rc;
}

int clpgpu_btran(clpgpu_context *ctx, double *region)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  if (ctx->rejectScaled("clpgpu_btran"))
    return -3;
  int rc = ctx->h2d(ctx->D.tau, region, ctx->m);
  ctx->btranDevice(ctx->D.tau, ctx->D.vecV2);
  rc |= ctx->d2h(region, ctx->D.vecV2, ctx->m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(ctx->m, 256)), dim3(256), 0, ctx->stream, ctx->D.vecV2, ctx->m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(ctx->m, 256)), dim3(256), 0, ctx->stream, ctx->D.tau, ctx->m);
  rc |= ctx->sync();
  return rc;
}

// CoinOtherFactorization::replaceColumn surface: the rank-1 nucleus update needs w = B^-1 a_q and
// rho = B^-T e_p, both recomputed here from the device state (a Clp adapter passes the same two
// vectors it already holds; this standalone form keeps the C ABI to PODs).
int clpgpu_replace_column(clpgpu_context *ctx, int pivotRow, int sequenceIn, double pivotCheck, double acceptablePivot)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  if (ctx->rejectScaled("clpgpu_replace_column"))
    return -3;
  (void)pivotCheck;
  (void)acceptablePivot;
  const int m = ctx->m, n = ctx->n;
  Ctrl *h = ctx->hCtrl;
  if (h->pivots >= (ctx->luActive ? ctx->luEtaLimit : ctx->maximumPivots))
    return 5;
  if (ctx->luActive) {
    // LU mode: w = B^-1 a_q through the factorization and the eta file, then one more eta
    std::vector<double> v(m, 0.0);
    if (sequenceIn >= n)
      v[sequenceIn - n] = -1.0;
    else
      for (int p = ctx->colStart[sequenceIn]; p < ctx->colStart[sequenceIn + 1]; p++)
        v[ctx->row[p]] = ctx->elem[p];
    Dev &D = ctx->D;
    hipStream_t s = ctx->stream;
    const int gm = cdiv(m, 256);
    int rc = ctx->h2d(D.vecV1, v.data(), m);
    ctx->luFtran(D.vecV1, nullptr, D.w, nullptr);
    double alpha = 0.0;
    rc |= ctx->d2h(&alpha, D.w + pivotRow, 1);
    if (rc)
      return rc;
    if (fabs(alpha) < ctx->zeroTolerance) {
      hipLaunchKernelGGL(k_zero, dim3(gm), dim3(256), 0, s, D.vecV1, m);
      ctx->sync();
      return 2;
    }
    rc |= ctx->pullCtrl();
    h->state = RUN;
    h->pivotRow = pivotRow;
    h->sequenceIn = sequenceIn;
    h->sequenceOut = ctx->pivotVariable[pivotRow];
    h->directionOut = 1;
    h->alpha = alpha;
    h->theta = 0.0;
    h->dualOut = 0.0;
    h->directionIn = 1;
    h->lowerIn = h->upperIn = h->valueIn = 0.0;
    h->lowerOut = h->upperOut = 0.0;
    h->objectiveChange = 0.0;
    h->maximumIterations = 2147483647;
    h->maximumPivots = LU_TCAP_MAX;
    h->logCapacity = 0;
    rc |= ctx->pushCtrl();
    hipLaunchKernelGGL(k_lu_pf_append, dim3(gm + cdiv(ctx->hLu.tcap, 4)), dim3(256), 0, s, D, 0, gm);
    hipLaunchKernelGGL(k_house, dim3(1), dim3(256), 0, s, D);
    rc |= ctx->pullCtrl();
    rc |= ctx->d2h(ctx->pivotVariable.data(), D.pivotVariable, m);
    hipLaunchKernelGGL(k_zero, dim3(gm), dim3(256), 0, s, D.w, m);
    rc |= ctx->checkLaunches("clpgpu_replace_column");
    rc |= ctx->sync();
    ctx->pivots = h->pivots;
    return rc ? rc : 0;
  }
  if (h->k + 2 >= ctx->kcap)
    return 3;
  // stage the two solves through the same kernels the iteration uses
  std::vector<double> c(m, 0.0), v(m, 0.0);
  c[pivotRow] = 1.0;
  if (sequenceIn >= n)
    v[sequenceIn - n] = -1.0;
  else
    for (int p = ctx->colStart[sequenceIn]; p < ctx->colStart[sequenceIn + 1]; p++)
      v[ctx->row[p]] = ctx->elem[p];
  int rc = 0;
  rc |= ctx->h2d(ctx->D.vecC, c.data(), m);
  rc |= ctx->h2d(ctx->D.vecV1, v.data(), m);
  int seqOut = ctx->pivotVariable[pivotRow];
  h->state = RUN;
  h->pivotRow = pivotRow;
  h->sequenceIn = sequenceIn;
  h->sequenceOut = seqOut;
  h->directionOut = 1;
  h->zeroTolerance = 0.0;  // keep rho unpruned for the plug-in form
  rc |= ctx->pushCtrl();
  const int gm = cdiv(m, 256), kc = ctx->kcap, gk = cdiv(kc, 256);
  hipStream_t s = ctx->stream;
  Dev &D = ctx->D;
  hipLaunchKernelGGL(k_btran_slack, dim3(gm), dim3(256), 0, s, D, (const double *)D.vecC, D.rho, 1);
  hipLaunchKernelGGL(k_btran_t, dim3(gk), dim3(256), 0, s, D, (const double *)D.vecC, (const double *)D.rho, D.slotA, 1);
  hipLaunchKernelGGL(k_gemvT_partial, dim3(gk, cdiv(kc, 64)), dim3(256), 0, s, D, (const double *)D.slotA, 1);
  hipLaunchKernelGGL(k_gemvT_final, dim3(gk), dim3(256), 0, s, D, D.rho, 1, 1);
  hipLaunchKernelGGL(k_ftran_gather, dim3(gk), dim3(256), 0, s, D, (const double *)D.vecV1, (const double *)nullptr, D.slotA,
                     (double *)nullptr, 1);
  hipLaunchKernelGGL(k_gemv2, dim3(cdiv(kc, 4)), dim3(256), 0, s, D, (const double *)D.slotA, (const double *)nullptr, D.slotC,
                     (double *)nullptr, 1);
  hipLaunchKernelGGL(k_ftran_scatter, dim3(cdiv(m + kc, 256)), dim3(256), 0, s, D, (const double *)D.vecV1,
                     (const double *)nullptr, (const double *)D.slotC, (const double *)nullptr, D.w, (double *)nullptr, 1);
  double alpha = 0.0;
  rc |= ctx->d2h(&alpha, D.w + pivotRow, 1);
  if (rc)
    return rc;
  if (fabs(alpha) < ctx->zeroTolerance) {
    h->state = EXIT_BAD_UPDATE;
    hipLaunchKernelGGL(k_zero, dim3(gm), dim3(256), 0, s, D.vecC, m);
    hipLaunchKernelGGL(k_zero, dim3(gm), dim3(256), 0, s, D.vecV1, m);
    hipLaunchKernelGGL(k_zero, dim3(gm), dim3(256), 0, s, D.rho, m);
    ctx->sync();
    return 2;
  }
  rc |= ctx->pullCtrl();
  h->alpha = alpha;
  int inStruct = sequenceIn < n, outStruct = seqOut < n;
  std::vector<int> slotOfCol(n), slotOfRow(m);
  rc |= ctx->d2h(slotOfCol.data(), D.slotOfCol, n);
  rc |= ctx->d2h(slotOfRow.data(), D.slotOfRow, m);
  h->updateCase = outStruct ? (inStruct ? 0 : 2) : (inStruct ? 1 : 3);
  h->slotColOut = outStruct ? slotOfCol[seqOut] : -1;
  h->rowOfSlackOut = outStruct ? -1 : (seqOut - n);
  h->slotRowIn = inStruct ? -1 : slotOfRow[sequenceIn - n];
  h->zeroTolerance = ctx->zeroTolerance;
  // k_house also advances the iteration bookkeeping; give it neutral scalars
  h->theta = 0.0;
  h->dualOut = 0.0;
  h->directionIn = 1;
  h->lowerIn = h->upperIn = h->valueIn = 0.0;
  h->lowerOut = h->upperOut = 0.0;
  h->objectiveChange = 0.0;
  h->maximumIterations = 2147483647;
  h->maximumPivots = ctx->maximumPivots;
  h->logCapacity = 0;
  rc |= ctx->pushCtrl();
  hipLaunchKernelGGL(k_rank1, dim3(cdiv(kc, 256), kc < 512 ? kc : 512), dim3(256), 0, s, D);
  hipLaunchKernelGGL(k_rank1_fix, dim3(cdiv(kc + 1, 256)), dim3(256), 0, s, D);
  hipLaunchKernelGGL(k_rank1_fix2, dim3(gk), dim3(256), 0, s, D);
  // bookkeeping only (positions, slots, row copy partition)
  std::vector<double> saveSol(2), saveDj(2);
  hipLaunchKernelGGL(k_house, dim3(1), dim3(256), 0, s, D);
  rc |= ctx->pullCtrl();
  rc |= ctx->d2h(ctx->pivotVariable.data(), D.pivotVariable, m);
  hipLaunchKernelGGL(k_zero, dim3(gm), dim3(256), 0, s, D.rho, m);
  hipLaunchKernelGGL(k_zero, dim3(gm), dim3(256), 0, s, D.w, m);
  rc |= ctx->sync();
  ctx->pivots = h->pivots;
  return rc ? rc : 0;
}

int clpgpu_pivots(const clpgpu_context *ctx) { return (ctx && ctx->hCtrl) ? ctx->hCtrl->pivots : 0; }

// parity hook for the cycle detector of the device housekeeping (ClpSimplexProgress::cycle, src/ClpSolve.cpp:4726-4825):
// the sequence of pivots given goes through cycleStep on a scratch control block; matched[i] = its verdict at pivot i
int clpgpu_test_cycle(clpgpu_context *ctx, int count, const int *in, const int *out, const int *wayIn, const int *wayOut, int *matched)
{
  if (!ctx || count <= 0)
    return -99;
  Ctrl *scratch = nullptr;
  int *d = nullptr;
  if (hipMalloc((void **)&scratch, sizeof(Ctrl)) != hipSuccess || hipMalloc((void **)&d, sizeof(int) * 5 * (size_t)count) != hipSuccess) {
    if (scratch)
      (void)hipFree(scratch);
    return -99;
  }
  int rc = 0;
  rc |= ctx->h2d(d, in, count);
  rc |= ctx->h2d(d + count, out, count);
  rc |= ctx->h2d(d + 2 * count, wayIn, count);
  rc |= ctx->h2d(d + 3 * count, wayOut, count);
  hipLaunchKernelGGL(k_test_cycle, dim3(1), dim3(64), 0, ctx->stream, scratch, count, (const int *)d, (const int *)(d + count),
                     (const int *)(d + 2 * count), (const int *)(d + 3 * count), d + 4 * count);
  rc |= ctx->checkLaunches("clpgpu_test_cycle");
  rc |= ctx->d2h(matched, d + 4 * count, count);
  (void)hipFree(scratch);
  (void)hipFree(d);
  return rc;
}

// test hook for the jagged row-tiled pricing layout (jdsLayout above; consumed by k_price_lds), host code only: no device is
// needed or touched.  `order` = numSlices * 64 column keys in home order (-1 none).  Called with capRecords = 0 it only
// reports the sizes (tiles, tileRows, records); called again with room it fills the arrays the kernel reads.  A CPU test walks
// them the way the kernel does and holds every column's dot product to the sequential one (tests/test_host_logic.py).
// Returns 0, 1 when the layout refuses the matrix (a column with more than 255 entries in one tile), -99 on bad arguments.
int clpgpu_test_jds_layout(int m, int n, const int *colStart, const int *row, const double *elem, int numSlices, const int *order, int *tiles,
                           int *tileRows, long long *records, long long capRecords, int *segStart, unsigned char *cnt, unsigned char *src,
                           unsigned char *home, unsigned *rowPair, double *elemPair)
{
  if (m <= 0 || n <= 0 || numSlices <= 0 || !colStart || !row || !elem || !order || !tiles || !tileRows || !records)
    return -99;
  for (size_t i = 0; i < (size_t)numSlices * 64; i++)
    if (order[i] >= n)
      return -99;
  int T, tr;
  jdsTiling(m, T, tr);
  std::vector<int> ord(order, order + (size_t)numSlices * 64), seg, col;
  std::vector<unsigned char> c, sr, hm;
  std::vector<unsigned> rp;
  std::vector<double2> ep;
  if (!jdsLayout(m, colStart, row, elem, ord, numSlices, T, tr, seg, col, c, sr, hm, rp, ep))
    return 1;
  *tiles = T;
  *tileRows = tr;
  *records = (long long)rp.size();
  if (capRecords <= 0)
    return 0;
  if (capRecords < (long long)rp.size() || !segStart || !cnt || !src || !home || !rowPair || !elemPair)
    return -99;
  memcpy(segStart, seg.data(), seg.size() * sizeof(int));
  memcpy(cnt, c.data(), c.size());
  memcpy(src, sr.data(), sr.size());
  memcpy(home, hm.data(), hm.size());
  memcpy(rowPair, rp.data(), rp.size() * sizeof(unsigned));
  memcpy(elemPair, ep.data(), ep.size() * sizeof(double2));
  return 0;
}

// test hook for the host side of the basis factorization (lu_front.h luFrontFactor: Markowitz LU of the nucleus that stops at a dense
// tail), host code only: no device is needed or touched.  C (k x k) by columns; counts[6] = { pivots, entries of L, entries of U (off the
// pivots), order of the tail, entries of the tail, fill-in }.  With capacities of 0 only the counts come back; with room the pivots
// (frow, fcol, fpiv), L by pivot (lStart[pivots + 1], lRow, lVal: multipliers), U by pivot (uStart, uCol, uVal: the pivot row off the
// pivot) and the tail (tailRow, tailCol, then its entries as (slot row, slot col, value)).  A CPU test rebuilds C = L U + S from them
// (tests/test_lu_front.py).  Returns 0, -99 on bad arguments.
int clpgpu_test_lu_front(int k, const int *cStart, const int *cRow, const double *cVal, double stopDensity, int minTail, double threshold,
                         long long *counts, int have, int *frow, int *fcol, double *fpiv, int *lStart, int *lRow, double *lVal, int *uStart,
                         int *uCol, double *uVal, int *tailRow, int *tailCol, int *sRow, int *sCol, double *sVal)
{
  if (k <= 0 || !cStart || !cRow || !cVal || !counts)
    return -99;
  for (int p = 0; p < cStart[k]; p++)
    if (cRow[p] < 0 || cRow[p] >= k)
      return -99;
  LuFront F;
  luFrontFactor(k, cStart, cRow, cVal, stopDensity, minTail, threshold, 1.0e-11, F);
  counts[0] = F.nF;
  counts[1] = (long long)F.lRow.size();
  counts[2] = (long long)F.uCol.size();
  counts[3] = F.k2;
  counts[4] = (long long)F.sVal.size();
  counts[5] = F.fill;
  if (!have)
    return 0;
  if (!frow || !fcol || !fpiv || !lStart || !lRow || !lVal || !uStart || !uCol || !uVal || !tailRow || !tailCol || !sRow || !sCol || !sVal)
    return -99;
  auto put = [](auto *dst, const auto &v) {
    if (!v.empty())
      memcpy(dst, v.data(), v.size() * sizeof(v[0]));
  };
  put(frow, F.frow);
  put(fcol, F.fcol);
  put(fpiv, F.fpiv);
  put(lStart, F.lStart);
  put(lRow, F.lRow);
  put(lVal, F.lVal);
  put(uStart, F.uStart);
  put(uCol, F.uCol);
  put(uVal, F.uVal);
  put(tailRow, F.tailRow);
  put(tailCol, F.tailCol);
  put(sRow, F.sRow);
  put(sCol, F.sCol);
  put(sVal, F.sVal);
  return 0;
}

// test hook for the row choice of dualRow's free-first entry (freeFirstChoice above), host code only.  Returns the row or -1; -99 on bad
// arguments.
int clpgpu_test_free_first_row(int m, int numberSequences, const double *work, const int *pivotVariable, const double *solution, const double *lower,
                               const double *upper, const unsigned char *status)
{
  if (m <= 0 || numberSequences <= 0 || !work || !pivotVariable || !solution || !lower || !upper || !status)
    return -99;
  for (int i = 0; i < m; i++)
    if (pivotVariable[i] < 0 || pivotVariable[i] >= numberSequences)
      return -99;
  return freeFirstChoice(m, work, pivotVariable, solution, lower, upper, status);
}

// parity hook for the engine's restatement of ClpSimplexProgress::looping (src/ClpSolve.cpp:4438-4611), host code only: no
// device is needed or touched.  A sequence of status checks goes through clpgpu_context::progressLooping on a scratch context
// without rows or columns (tests/test_progress_looping.py holds it, the CPU checker's hook of the same shape and a Python
// restatement of the reference together).  Sequences must be below 64.
int clpgpu_test_looping(int count, const double *objective, const double *infeasibility, const int *numberInfeasibilities, const int *iteration,
                        const int *flagBits, const int *newestIncoming, int *code, double *dualTolerance, double *dualBound, int *forceFactorization,
                        int *flagged)
{
  if (count < 0 || (count && (!objective || !infeasibility || !numberInfeasibilities || !iteration || !flagBits || !newestIncoming || !code
                              || !dualTolerance || !dualBound || !forceFactorization || !flagged)))
    return -99;
  clpgpu_context *ctx = new clpgpu_context();
  Ctrl ctrl;
  memset(&ctrl, 0, sizeof(ctrl));
  ctx->hCtrl = &ctrl;
  ctx->status.assign(64, 0);
  ctx->dualTolerance = ctx->dualToleranceBase = 1.0e-7;
  ctx->dualBound = 1.0e10;
  ctx->forceFactorization = -1;
  ctx->progressReset();
  for (int i = 0; i < count; i++) {
    ctx->objectiveValue = objective[i];
    ctx->bestPossibleImprovement = 0.0;
    ctx->sumPrimalInfeasibilities = infeasibility[i];
    ctx->numberPrimalInfeasibilities = numberInfeasibilities[i];
    ctx->numberIterations = iteration[i];
    ctx->progressFlag = flagBits[i];
    ctx->progressStartCheck();
    ctrl.cycIn[11] = newestIncoming[i];  // with cycHead == 0 the newest entry sits in the last slot
    std::fill(ctx->status.begin(), ctx->status.end(), (unsigned char)0);
    code[i] = ctx->progressLooping();
    dualTolerance[i] = ctx->dualTolerance;
    dualBound[i] = ctx->dualBound;
    forceFactorization[i] = ctx->forceFactorization;
    flagged[i] = -1;
    for (int j = 0; j < 64; j++)
      if (ctx->status[j] & FLAGGED_BIT)
        flagged[i] = j;
  }
  ctx->hCtrl = nullptr;
  delete ctx;
  return 0;
}

// the engine's own f64 MFMA GEMM on host arrays (row-major n x n): c = beta c + alpha a b.  A parity hook for the
// kernel behind the Newton-Schulz steps (CoinAbcDgemm's role, src/CoinAbcHelperFunctions.cpp:1658).
int clpgpu_dgemm(clpgpu_context *ctx, int nn, double alpha, const double *a, const double *b, double beta, double *c)
{
  if (!ctx || nn <= 0)
    return -99;
  double *dA = nullptr, *dB = nullptr, *dC = nullptr;
  const size_t bytes = sizeof(double) * (size_t)nn * nn;
  if (hipMalloc((void **)&dA, bytes) != hipSuccess || hipMalloc((void **)&dB, bytes) != hipSuccess || hipMalloc((void **)&dC, bytes) != hipSuccess) {
    if (dA) (void)hipFree(dA);
    if (dB) (void)hipFree(dB);
    ctx->setError("clpgpu_dgemm: hipMalloc failed");
    return -99;
  }
  int rc = ctx->h2d(dA, a, (size_t)nn * nn);
  rc |= ctx->h2d(dB, b, (size_t)nn * nn);
  rc |= ctx->h2d(dC, c, (size_t)nn * nn);
  hipLaunchKernelGGL(k_dgemm, dim3(cdiv(nn, DG_T), cdiv(nn, DG_T)), dim3(256), 0, ctx->stream, nn, nn, nn, alpha, (const double *)dA, nn,
                     (const double *)dB, nn, beta, dC, nn);
  rc |= ctx->checkLaunches("clpgpu_dgemm");
  rc |= ctx->d2h(c, dC, (size_t)nn * nn);
  (void)hipFree(dA);
  (void)hipFree(dB);
  (void)hipFree(dC);
  return rc;
}

// ---- ClpFactorization::updateColumnFT / updateTwoColumnsFT (src/ClpFactorization.cpp:2723, :2889) ----
// With the explicit nucleus inverse there is no Forrest-Tomlin stash to fill: the "FT" solve is the
// plain FTRAN, and its result stays on the device (D.w) as the updated column the following
// clpgpu_update_weights / clpgpu_replace_column refer to.  Returns the number of nonzeros like the
// reference (a negative value there means "no room in the R area": never the case here).
int clpgpu_ftran_ft(clpgpu_context *ctx, double *region)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  if (ctx->rejectScaled("clpgpu_ftran_ft"))
    return -3;
  const int m = ctx->m;
  int rc = ctx->h2d(ctx->D.vecV2, region, m);
  ctx->ftranDevice(ctx->D.vecV2, ctx->D.w);
  rc |= ctx->d2h(region, ctx->D.w, m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream, ctx->D.vecV2, m);
  rc |= ctx->checkLaunches("clpgpu_ftran_ft");
  rc |= ctx->sync();
  if (rc)
    return -99;
  int count = 0;
  for (int i = 0; i < m; i++)
    count += region[i] != 0.0;
  return count;
}

// both right-hand sides in one sweep over the nucleus inverse (k_gemv2 carries two vectors):
// regionFT = the entering column (kept on the device as the updated column), region2 = the DSE vector
int clpgpu_ftran_two_ft(clpgpu_context *ctx, double *regionFT, double *region2)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  if (ctx->rejectScaled("clpgpu_ftran_two_ft"))
    return -3;
  const int m = ctx->m;
  int rc = ctx->h2d(ctx->D.vecV1, regionFT, m);
  rc |= ctx->h2d(ctx->D.vecV2, region2, m);
  ctx->ftranDevice2(ctx->D.vecV1, ctx->D.vecV2, ctx->D.w, ctx->D.tau);
  rc |= ctx->d2h(regionFT, ctx->D.w, m);
  rc |= ctx->d2h(region2, ctx->D.tau, m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream, ctx->D.vecV1, m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream, ctx->D.vecV2, m);
  hipLaunchKernelGGL(k_zero, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream, ctx->D.tau, m);
  rc |= ctx->checkLaunches("clpgpu_ftran_two_ft");
  rc |= ctx->sync();
  if (rc)
    return -99;
  int count = 0;
  for (int i = 0; i < m; i++)
    count += regionFT[i] != 0.0;
  return count;
}

// ---- ClpDualRowPivot surface (src/ClpDualRowPivot.hpp:30-76; ClpDualRowSteepest.cpp / ClpDualRowDantzig.cpp) ----
// The rim arrays are Clp's (host); clpgpu_bind_rim copies the ones given into the engine's device
// mirrors -- the "host mirrors synced at refactorization boundaries" contract of SURVEY 8b.
int clpgpu_bind_rim(clpgpu_context *ctx, const double *cost, const double *lower, const double *upper, const double *dj,
                    const double *solution, const unsigned char *status)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  const int N = ctx->N;
  int rc = 0;
  if (cost) {
    ctx->cost.assign(cost, cost + N);
    rc |= ctx->h2d(ctx->D.cost, cost, N);
  }
  if (lower) {
    ctx->lower.assign(lower, lower + N);
    rc |= ctx->h2d(ctx->D.lower, lower, N);
  }
  if (upper) {
    ctx->upper.assign(upper, upper + N);
    rc |= ctx->h2d(ctx->D.upper, upper, N);
  }
  if (dj) {
    ctx->dj.assign(dj, dj + N);
    rc |= ctx->h2d(ctx->D.dj, dj, N);
  }
  if (solution) {
    ctx->sol.assign(solution, solution + N);
    rc |= ctx->h2d(ctx->D.sol, solution, N);
  }
  if (status) {
    ctx->status.assign(status, status + N);
    rc |= ctx->h2d(ctx->D.status, status, N);
  }
  rc |= ctx->sync();
  return rc;
}

// ClpDualRowPivot::pivotRow (ClpDualRowSteepest.cpp:179, ClpDualRowDantzig.cpp:56): the leaving row on
// the current device state (bound rim, factorization's pivotVariable, weights and infeasibility list
// from clpgpu_save_weights), or -1 when nothing is primal infeasible ("looks optimal").
int clpgpu_pivot_row(clpgpu_context *ctx)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  ctx->preparePlugin();
  Ctrl *h = ctx->hCtrl;
  h->preDone = 0;
  if (ctx->pushCtrl())
    return -99;
  Dev &D = ctx->D;
  hipStream_t s = ctx->stream;
  hipLaunchKernelGGL(k_chuzr_pre, dim3(1), dim3(64), 0, s, D);
  hipLaunchKernelGGL(k_chuzr_scan, dim3(ctx->nChzBlocks), dim3(256), 0, s, D, -1);
  hipLaunchKernelGGL(k_chuzr_final_btran, dim3(1), dim3(256), 0, s, D, ctx->nChzBlocks, 0);
  // the final stage also stages the BTRAN unit vector of the engine's own loop: not wanted here
  hipLaunchKernelGGL(k_zero, dim3(cdiv(ctx->m, 256)), dim3(256), 0, s, D.vecC, ctx->m);
  if (ctx->checkLaunches("clpgpu_pivot_row") || ctx->pullCtrl())
    return -99;
  const int row = h->state == EXIT_NO_PIVOT_ROW ? -1 : h->pivotRow;
  ctx->seed = h->seed;
  h->state = RUN;
  return row;
}

// ClpDualRowSteepest::updateWeights (:375-624): FTRAN of the entering column (the FT solve) and of pi
// in one sweep, alpha = w[pivotRow], DSE weights updated on the support of w with the ratio test's
// alpha (`model_->alpha()`, :509-516); old weights kept for clpgpu_unroll_weights.  updatedColumn (m
// doubles, by basis position) receives w; the Dantzig rule only does the solve.
int clpgpu_update_weights(clpgpu_context *ctx, int numberPi, const int *piIndex, const double *piValue, int pivotRow,
                          int sequenceIn, double modelAlpha, double *updatedColumn, double *alphaOut)
{
  if (!ctx || !ctx->hCtrl || pivotRow < 0 || pivotRow >= ctx->m || sequenceIn < 0 || sequenceIn >= ctx->N)
    return -99;
  if (ctx->rejectScaled("clpgpu_update_weights"))
    return -3;
  ctx->preparePlugin();
  const int m = ctx->m, n = ctx->n, g = cdiv(m, 256);
  std::vector<double> pi(m, 0.0), col(m, 0.0);
  for (int i = 0; i < numberPi; i++)
    pi[piIndex[i]] = piValue[i];
  if (sequenceIn >= n)
    col[sequenceIn - n] = -1.0;
  else
    for (int p = ctx->colStart[sequenceIn]; p < ctx->colStart[sequenceIn + 1]; p++)
      col[ctx->row[p]] = ctx->elem[p];
  Dev &D = ctx->D;
  hipStream_t s = ctx->stream;
  int rc = ctx->h2d(D.vecV1, col.data(), m);
  rc |= ctx->h2d(D.vecV2, pi.data(), m);
  ctx->ftranDevice2(D.vecV1, D.vecV2, D.w, D.tau);
  if (ctx->pivotRule) {
    hipLaunchKernelGGL(k_plugin_norm, dim3(g), dim3(256), 0, s, D, (const double *)D.vecV2);
    hipLaunchKernelGGL(k_plugin_weights, dim3(g), dim3(256), 0, s, D, pivotRow, modelAlpha, g);
  }
  hipLaunchKernelGGL(k_zero, dim3(g), dim3(256), 0, s, D.vecV1, m);
  hipLaunchKernelGGL(k_zero, dim3(g), dim3(256), 0, s, D.vecV2, m);
  hipLaunchKernelGGL(k_zero, dim3(g), dim3(256), 0, s, D.tau, m);
  rc |= ctx->checkLaunches("clpgpu_update_weights");
  std::vector<double> w(m);
  rc |= ctx->d2h(w.data(), D.w, m);
  if (rc)
    return -99;
  if (updatedColumn)
    memcpy(updatedColumn, w.data(), sizeof(double) * (size_t)m);
  if (alphaOut)
    *alphaOut = w[pivotRow];
  ctx->hCtrl->pivotRow = pivotRow;
  return 0;
}

// ClpDualRowSteepest::updatePrimalSolution (:630-763): x_B -= theta * w over the support of the
// updated column (the one clpgpu_update_weights / clpgpu_ftran_ft left on the device, or the one given),
// squared infeasibilities refreshed, new entries appended to the list in position order; the row
// given as pivotRow keeps a tiny entry (:705).  *changeInObjective += the objective change.
int clpgpu_update_primal(clpgpu_context *ctx, const double *updatedColumn, int pivotRow, double theta, double *changeInObjective)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  ctx->preparePlugin();
  const int m = ctx->m, g = cdiv(m, 256);
  Dev &D = ctx->D;
  hipStream_t s = ctx->stream;
  Ctrl *h = ctx->hCtrl;
  int rc = 0;
  if (updatedColumn)
    rc |= ctx->h2d(D.w, updatedColumn, m);
  h->movement = theta;
  h->numberFlips = 0;
  h->pivotRow = pivotRow;
  h->objectiveChange = 0.0;
  h->numberAppend = 0;
  rc |= ctx->pushCtrl();
  hipLaunchKernelGGL(k_primal_update, dim3(g), dim3(256), 0, s, D, 0);
  hipLaunchKernelGGL(k_append_scatter_abs, dim3(g), dim3(256), 0, s, D);
  rc |= ctx->checkLaunches("clpgpu_update_primal");
  rc |= ctx->pullCtrl();
  if (rc)
    return -99;
  if (changeInObjective)
    *changeInObjective += h->objectiveChange;
  h->appendGo = 0;
  h->numberAppend = 0;
  return ctx->pushCtrl();
}

// ClpDualRowSteepest::saveWeights (:773-1014), modes 1-7
int clpgpu_save_weights(clpgpu_context *ctx, int mode)
{
  if (!ctx || !ctx->hCtrl || mode < 1 || mode > 7)
    return -99;
  ctx->preparePlugin();
  if (ctx->pushCtrl())
    return -99;
  int rc = ctx->saveWeights(mode);
  rc |= ctx->checkLaunches("clpgpu_save_weights");
  return rc;
}

// ClpDualRowSteepest::unrollWeights (:1022-1042): undo the last clpgpu_update_weights
int clpgpu_unroll_weights(clpgpu_context *ctx)
{
  if (!ctx || !ctx->hCtrl)
    return -99;
  hipLaunchKernelGGL(k_unroll_weights, dim3(cdiv(ctx->m, 256)), dim3(256), 0, ctx->stream, ctx->D);
  int rc = ctx->checkLaunches("clpgpu_unroll_weights");
  rc |= ctx->sync();
  return rc;
}

// ---- clone / external scales ------------------------------------------------------------------------
// ClpMatrixBase::clone / ClpDualRowPivot::clone / CoinOtherFactorization::clone: Clp copies models and
// their plug-ins freely (SURVEY 8b "copy/presolve caveat").  A clone is an independent context on the
// same device holding the same problem, options, bounds/costs as changed so far and warm-start status;
// the factorization is not copied (the first factorize / dual of the clone rebuilds it).
clpgpu_context *clpgpu_clone(const clpgpu_context *src)
{
  if (!src)
    return nullptr;
  clpgpu_context *ctx = clpgpu_create(src->device);
  if (!ctx)
    return nullptr;
  ctx->optPrimalTolerance = src->optPrimalTolerance;
  ctx->optDualTolerance = src->optDualTolerance;
  ctx->optZeroTolerance = src->optZeroTolerance;
  ctx->optDualBound = src->optDualBound;
  ctx->optAcceptablePivot = src->optAcceptablePivot;
  ctx->largeValue = src->largeValue;
  ctx->maximumIterations = src->maximumIterations;
  ctx->pivotRule = src->pivotRule;
  ctx->maximumPivots = src->maximumPivots;
  ctx->logLevel = src->logLevel;
  ctx->checkEvery = src->checkEvery;
  ctx->seed = src->seed;
  ctx->perturbationOption = src->perturbationOption;
  ctx->priceKernel = src->priceKernel;
  ctx->useGraph = src->useGraph;
  ctx->blockedRefactor = src->blockedRefactor;
  ctx->registerPanel = src->registerPanel;
  ctx->rowPriceFrac = src->rowPriceFrac;
  ctx->priceLds = src->priceLds;
  ctx->dcWide = src->dcWide;
  ctx->shardGrow = src->shardGrow;
  ctx->priceLdsMinWindows = src->priceLdsMinWindows;
  ctx->priceLdsGridCap = src->priceLdsGridCap;
  ctx->sellWindows = src->sellWindows;
  ctx->flipScatter = src->flipScatter;
  ctx->flipSlotCap = src->flipSlotCap;
  ctx->refreshMinK = src->refreshMinK;
  ctx->factorMode = src->factorMode;
  ctx->luMinK = src->luMinK;
  ctx->luMaxPivots = src->luMaxPivots;
  ctx->luStopDensity = src->luStopDensity;
  ctx->luMinTail = src->luMinTail;
  ctx->luThreshold = src->luThreshold;
  ctx->refreshMax = src->refreshMax;
  ctx->refreshTolerance = src->refreshTolerance;
  ctx->refreshRefine = src->refreshRefine;
  ctx->refreshMinKDense = src->refreshMinKDense;
  ctx->refreshResidualMax = src->refreshResidualMax;
  ctx->scalingMode = src->scalingMode;
  ctx->flipListCap = src->flipListCap;
  // every other user-settable option (clpgpu_set_option): a clone used for strong branching or a node re-solve runs with
  // its parent's configuration
  ctx->solutionRefinements = src->solutionRefinements;
  ctx->refineAbove = src->refineAbove;
  ctx->gemmBackend = src->gemmBackend;
  ctx->luPolish = src->luPolish;
  ctx->luPolishTolerance = src->luPolishTolerance;
  ctx->luAdaptive = src->luAdaptive;
  ctx->luMinPivots = src->luMinPivots;
  ctx->luInverseFillCap = src->luInverseFillCap;
  ctx->luCompactEta = src->luCompactEta;
  ctx->luFold = src->luFold;
  ctx->panel68 = src->panel68;
  ctx->luGemvThreads = src->luGemvThreads;
  ctx->luPfsBlocks = src->luPfsBlocks;
  ctx->luScatterPpb = src->luScatterPpb;
  ctx->fakeBoundCleanup = src->fakeBoundCleanup;
  ctx->checkBoth = src->checkBoth;
  ctx->freeNonbasic = src->freeNonbasic;
  ctx->tryPrimal = src->tryPrimal;
  ctx->steepestMode = src->steepestMode;
  ctx->steepestElements = src->steepestElements;
  ctx->chuzrFloor = src->chuzrFloor;
  ctx->debugToleranceFactor = src->debugToleranceFactor;
  ctx->debugLastBadIteration = src->debugLastBadIteration;
  ctx->refactorMode = src->refactorMode;
  ctx->refactorMinK = src->refactorMinK;
  ctx->forkUpdate = src->forkUpdate;
  ctx->timing = src->timing;
  ctx->haveExternalScales = src->haveExternalScales;
  if (src->haveExternalScales) {
    ctx->rowScale = src->rowScale;
    ctx->colScale = src->colScale;
  }
  if (src->n > 0 && !src->colStart.empty()) {
    int rc = ctx->loadProblem(src->m, src->n, src->colStart.data(), src->row.data(), src->origElem.data(), src->origColLower.data(),
                              src->origColUpper.data(), src->origObj.data(), src->origRowLower.data(), src->origRowUpper.data());
    if (rc) {
      clpgpu_destroy(ctx);
      return nullptr;
    }
    if (src->haveStatus) {
      ctx->userStatus = src->userStatus;
      ctx->haveStatus = true;
    }
  }
  return ctx;
}

// ClpModel::rowScale_/columnScale_ handed over by the caller (SURVEY 8b clpgpu_set_scales): the engine
// then keeps the LP in those units instead of computing factors itself (option "scaling").  NULL, NULL
// drops them.  A loaded problem is rebuilt from the caller's original arrays.
int clpgpu_set_scales(clpgpu_context *ctx, const double *rowScale, const double *columnScale)
{
  if (!ctx)
    return -99;
  if ((rowScale == nullptr) != (columnScale == nullptr))
    return -1;
  if (!rowScale) {
    ctx->haveExternalScales = false;
  } else {
    if (ctx->n <= 0 || ctx->colStart.empty())
      return -2;  // sizes come from the loaded problem: load first, then hand over the factors
    for (int i = 0; i < ctx->m; i++)
      if (!(rowScale[i] > 0.0))
        return -1;
    for (int j = 0; j < ctx->n; j++)
      if (!(columnScale[j] > 0.0))
        return -1;
    ctx->rowScale.assign(rowScale, rowScale + ctx->m);
    ctx->colScale.assign(columnScale, columnScale + ctx->n);
    ctx->haveExternalScales = true;
  }
  if (ctx->n > 0 && !ctx->colStart.empty()) {
    // (loadProblem copies its inputs before it releases anything, so passing the context's own arrays is fine)
    int rc = ctx->loadProblem(ctx->m, ctx->n, ctx->colStart.data(), ctx->row.data(), ctx->origElem.data(), ctx->origColLower.data(),
                              ctx->origColUpper.data(), ctx->origObj.data(), ctx->origRowLower.data(), ctx->origRowUpper.data());
    return rc;
  }
  return 0;
}

// ---- multi-GPU: RCCL communicator for the column-sharded pricing exchange -------------------
static void *rcclHandle()
{
  static void *h = nullptr;
  if (!h)
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h)
    h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  return h;
}

int clpgpu_comm_unique_id(void *id128)
{
  void *h = rcclHandle();
  if (!h)
    return -1;
  typedef int (*fn_t)(void *);
  fn_t f = (fn_t)dlsym(h, "ncclGetUniqueId");
  if (!f)
    return -1;
  return f(id128);
}

int clpgpu_comm_init(clpgpu_context *ctx, int rank, int nranks, const void *id128)
{
  if (!ctx || nranks < 1 || rank < 0 || rank >= nranks || nranks > 8)
    return -1;
  if (nranks == 1 && !getenv("CLPGPU_FORCE_COMM"))
    return 0;  // (the env knob exercises the RCCL path with a one-rank communicator on a 1-GPU box)
  void *h = rcclHandle();
  if (!h) {
    ctx->setError("librccl not found");
    return -1;
  }
  struct UniqueId { char internal[128]; } id;
  memcpy(&id, id128, sizeof(id));
  typedef int (*init_t)(void **, int, UniqueId, int);
  init_t initFn = (init_t)dlsym(h, "ncclCommInitRank");
  ctx->ncclAllGatherFn = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
  ctx->ncclCommDestroyFn = (int (*)(void *))dlsym(h, "ncclCommDestroy");
  if (!initFn || !ctx->ncclAllGatherFn) {
    ctx->setError("RCCL symbols missing");
    return -1;
  }
  (void)hipSetDevice(ctx->device);
  int rc = initFn(&ctx->comm, nranks, id, rank);
  if (rc != 0) {
    ctx->setError("ncclCommInitRank failed (%d)", rc);
    ctx->comm = nullptr;
    return -1;
  }
  ctx->rank = rank;
  ctx->nranks = nranks;
  ctx->commActive = true;
  ctx->commMode = getenv("CLPGPU_COMM_MODE") ? atoi(getenv("CLPGPU_COMM_MODE")) : ctx->commMode;
  if (ctx->commMode != 1)
    ctx->commMode = 2;
  ctx->applyShard();
  if (ctx->allocShardBuffers())
    return -99;
  // collectives are enqueued between kernels: eager chain by default.  Option comm_graph 1 keeps the hipGraph batches -- the two
  // ncclAllGather calls of a pivot are then captured with the kernels around them (RCCL supports stream capture); a failed capture
  // falls back to eager launches as everywhere (launchBatch).  Measured with a one-rank communicator only: off by default.
  if (!ctx->commGraph)
    ctx->useGraph = 0;
  rc = ctx->buildSell();
  return rc;
}

// ---- loopback ranks (see clpgpu_virtual_group) ----
clpgpu_virtual_group *clpgpu_virtual_group_create(int nranks)
{
  if (nranks < 1 || nranks > 8)
    return nullptr;
  clpgpu_virtual_group *g = new clpgpu_virtual_group();
  g->nranks = nranks;
  return g;
}
void clpgpu_virtual_group_destroy(clpgpu_virtual_group *g) { delete g; }

int clpgpu_virtual_attach(clpgpu_context *ctx, clpgpu_virtual_group *g, int rank)
{
  if (!ctx || !g || rank < 0 || rank >= g->nranks)
    return -1;
  g->ctx[rank] = ctx;
  ctx->virtualGroup = g;
  ctx->rank = rank;
  ctx->nranks = g->nranks;
  ctx->commActive = true;
  ctx->commMode = getenv("CLPGPU_COMM_MODE") ? atoi(getenv("CLPGPU_COMM_MODE")) : ctx->commMode;
  if (ctx->commMode != 1)
    ctx->commMode = 2;
  ctx->applyShard();
  if (ctx->allocShardBuffers())
    return -99;
  ctx->useGraph = 0;
  // several ranks on ONE device: the grid barrier of k_dual_column_wide assumes its 128 workgroups are resident together, which N
  // concurrent launches of it do not guarantee -- and a rank that gave up on a pivot alone would leave the others waiting in the
  // next exchange.  Loopback ranks walk long lists in one workgroup (the same decisions, bit for bit).
  ctx->dcWide = 0;
  return ctx->buildSell();
}

// clpgpu_dual_steps on every rank of the group at once, one host thread per rank (the ranks meet at the exchanges)
int clpgpu_virtual_dual_steps(clpgpu_virtual_group *g, int iterations, int *statusOut)
{
  if (!g)
    return -99;
  struct Arg {
    clpgpu_context *ctx;
    int iterations, status;
  } args[8];
  pthread_t th[8];
  for (int r = 0; r < g->nranks; r++) {
    if (!g->ctx[r])
      return -1;
    args[r] = { g->ctx[r], iterations, 0 };
  }
  auto body = [](void *p) -> void * {
    Arg *a = (Arg *)p;
    (void)hipSetDevice(a->ctx->device);
    a->status = a->ctx->run(a->iterations);
    return nullptr;
  };
  bool created[8] = {};
  int rc = 0;
  for (int r = 0; r < g->nranks; r++) {
    created[r] = pthread_create(&th[r], nullptr, body, &args[r]) == 0;
    if (!created[r]) {
      // a rank that never starts: the others must not wait for it in the exchange
      pthread_mutex_lock(&g->mutex);
      g->failed = true;
      pthread_cond_broadcast(&g->cond);
      pthread_mutex_unlock(&g->mutex);
      args[r].status = -99;
      rc = -2;
    }
  }
  for (int r = 0; r < g->nranks; r++) {
    if (created[r])
      pthread_join(th[r], nullptr);
    if (statusOut)
      statusOut[r] = args[r].status;
    if (g->ctx[r]->commFailed)
      rc = -2;
  }
  if (g->failed)
    rc = -2;
  return rc;
}

int clpgpu_set_option(clpgpu_context *ctx, const char *name, double v)
{
  if (!ctx)
    return -99;
  if (!strcmp(name, "pivot_rule")) ctx->pivotRule = (int)v;
  else if (!strcmp(name, "max_iterations")) ctx->maximumIterations = (int)v;
  else if (!strcmp(name, "max_pivots")) {
    if (v > 0) {
      ctx->maximumPivots = (int)v;
    } else {
      // ClpSimplex::defaultFactorizationFrequency (src/ClpSimplex.cpp:11401-11429, ABC_CLP_DEFAULTS 1),
      // what ClpSimplex::initialSolve installs before calling dual() (src/ClpSolve.cpp:1574)
      int mm = ctx->m, f;
      if (mm < 10000)
        f = 75 + mm / 50;
      else if (mm < 100000)
        f = 75 + 200 + (mm - 10000) / 200;
      else
        f = 1000;
      ctx->maximumPivots = f < 1000 ? f : 1000;
    }
  }
  else if (!strcmp(name, "dual_bound")) ctx->dualBound = ctx->optDualBound = v;
  else if (!strcmp(name, "primal_tolerance")) ctx->primalTolerance = ctx->optPrimalTolerance = v;
  else if (!strcmp(name, "dual_tolerance")) ctx->dualTolerance = ctx->dualToleranceBase = ctx->optDualTolerance = v;
  else if (!strcmp(name, "zero_tolerance")) ctx->zeroTolerance = ctx->optZeroTolerance = v;
  else if (!strcmp(name, "acceptable_pivot")) ctx->acceptablePivot = ctx->optAcceptablePivot = v;
  else if (!strcmp(name, "random_seed")) ctx->seed = (unsigned int)v;
  else if (!strcmp(name, "perturbation")) ctx->perturbationOption = (int)v;
  else if (!strcmp(name, "debug_backwards_at")) ctx->debugBackwardsAt = (int)v;
  else if (!strcmp(name, "debug_poison_inverse_at")) ctx->debugPoisonInverseAt = (int)v;
  else if (!strcmp(name, "debug_bad_accuracy_at")) ctx->debugBadAccuracyAt = (int)v;
  else if (!strcmp(name, "debug_reset_weights_at")) ctx->debugResetWeightsAt = (int)v;
  else if (!strcmp(name, "check_both")) ctx->checkBoth = v != 0.0;
  else if (!strcmp(name, "free_nonbasic")) ctx->freeNonbasic = v != 0.0;
  else if (!strcmp(name, "try_primal")) ctx->tryPrimal = v != 0.0;
  else if (!strcmp(name, "steepest_mode")) ctx->steepestMode = std::max(0, std::min((int)v, 3));
  else if (!strcmp(name, "steepest_elements")) ctx->steepestElements = v != 0.0;
  else if (!strcmp(name, "debug_last_bad_iteration")) ctx->debugLastBadIteration = (int)v;
  else if (!strcmp(name, "debug_tolerance_factor")) ctx->debugToleranceFactor = v;
  else if (!strcmp(name, "debug_dc_wide_timeout_at")) ctx->debugDcTimeoutAt = (int)v;
  else if (!strcmp(name, "debug_chuzr_floor")) ctx->chuzrFloor = std::max(1, (int)v);
  else if (!strcmp(name, "dse_reset_every")) ctx->dseResetEvery = std::max(0, (int)v);
  else if (!strcmp(name, "debug_singular_at")) ctx->debugSingularAt = (int)v;
  else if (!strcmp(name, "log_level")) ctx->logLevel = (int)v;
  else if (!strcmp(name, "check_every")) {
    ctx->checkEvery = (int)v < 1 ? 1 : (int)v;
    ctx->dropGraph();
    for (auto &e : ctx->evStart) (void)hipEventDestroy(e);
    for (auto &e : ctx->evStop) (void)hipEventDestroy(e);
    for (auto &e : ctx->evMid) (void)hipEventDestroy(e);
    ctx->evStart.clear();
    ctx->evStop.clear();
    ctx->evMid.clear();
    for (auto &e : ctx->ktEvents) (void)hipEventDestroy(e);
    ctx->ktEvents.clear();
  }
  else if (!strcmp(name, "timing")) { ctx->timing = (int)v; ctx->dropGraph(); }
  else if (!strcmp(name, "price_kernel")) { ctx->priceKernel = (int)v; ctx->dropGraph(); }
  else if (!strcmp(name, "use_graph")) { ctx->useGraph = (int)v; ctx->dropGraph(); }
  else if (!strcmp(name, "blocked_refactor")) ctx->blockedRefactor = (int)v;
  else if (!strcmp(name, "register_panel")) ctx->registerPanel = (int)v;
  else if (!strcmp(name, "refactor_mode")) ctx->refactorMode = (int)v;
  else if (!strcmp(name, "shard_cand_cap")) {
    if (ctx->commActive)
      return -2;  // sizes the exchange buffers: before clpgpu_comm_init
    ctx->shardCandCap = std::max(1, (int)v);
  }
  else if (!strcmp(name, "shard_flip_cap")) {
    if (ctx->commActive)
      return -2;
    ctx->shardFlipCap = std::max(1, (int)v);
  }
  else if (!strcmp(name, "comm_mode")) {
    if (ctx->commActive)
      return -2;
    ctx->commMode = (int)v == 1 ? 1 : 2;
  }
  else if (!strcmp(name, "refactor_min_k")) ctx->refactorMinK = (int)v;
  else if (!strcmp(name, "solution_refinements")) ctx->solutionRefinements = std::max(0, (int)v);
  else if (!strcmp(name, "refine_above")) ctx->refineAbove = v;
  else if (!strcmp(name, "gemm_backend")) ctx->gemmBackend = (int)v;
  else if (!strcmp(name, "lu_polish")) ctx->luPolish = std::max(0, (int)v);
  else if (!strcmp(name, "lu_polish_tolerance")) ctx->luPolishTolerance = v;
  else if (!strcmp(name, "factor_mode")) ctx->factorMode = (int)v;
  else if (!strcmp(name, "lu_min_k")) ctx->luMinK = (int)v;
  else if (!strcmp(name, "lu_max_pivots")) ctx->luMaxPivots = std::max(1, std::min((int)v, LU_TCAP_MAX - 1));
  else if (!strcmp(name, "lu_stop_density")) ctx->luStopDensity = v;
  else if (!strcmp(name, "lu_min_tail")) ctx->luMinTail = std::max(0, (int)v);
  else if (!strcmp(name, "lu_threshold")) ctx->luThreshold = v;
  else if (!strcmp(name, "lu_adaptive")) ctx->luAdaptive = (int)v;
  else if (!strcmp(name, "lu_min_pivots")) ctx->luMinPivots = std::max(1, (int)v);
  else if (!strcmp(name, "lu_inverse_fill_cap")) ctx->luInverseFillCap = v;
  else if (!strcmp(name, "lu_compact_eta")) { ctx->luCompactEta = v != 0.0; ctx->dropGraph(); }
  else if (!strcmp(name, "lu_gemv_threads")) { ctx->luGemvThreads = ((int)v >= 1024) ? 1024 : ((int)v >= 512 ? 512 : ((int)v >= 256 ? 256 : ((int)v >= 128 ? 128 : 64))); ctx->dropGraph(); }
  else if (!strcmp(name, "comm_graph")) ctx->commGraph = v != 0.0;
  else if (!strcmp(name, "lu_fold")) { ctx->luFold = (int)v & 3; ctx->dropGraph(); }
  else if (!strcmp(name, "panel_6x8")) ctx->panel68 = v != 0.0;
  else if (!strcmp(name, "lu_pfs_blocks")) { ctx->luPfsBlocks = std::max(1, std::min(1024, (int)v)); ctx->dropGraph(); }
  else if (!strcmp(name, "lu_scatter_ppb")) { ctx->luScatterPpb = std::max(64, std::min(256, ((int)v + 7) & ~7)); ctx->dropGraph(); }
  else if (!strcmp(name, "fake_bound_cleanup")) ctx->fakeBoundCleanup = v != 0.0;
  else if (!strcmp(name, "fork_update")) { ctx->forkUpdate = (int)v; ctx->dropGraph(); }
  else if (!strcmp(name, "refresh_min_k")) ctx->refreshMinK = (int)v;
  else if (!strcmp(name, "refresh_min_k_dense")) ctx->refreshMinKDense = (int)v;
  else if (!strcmp(name, "refresh_max")) ctx->refreshMax = std::max(0, (int)v);
  else if (!strcmp(name, "refresh_tolerance")) ctx->refreshTolerance = v;
  else if (!strcmp(name, "refresh_refine")) ctx->refreshRefine = v != 0.0;
  else if (!strcmp(name, "refresh_residual_max")) ctx->refreshResidualMax = v;
  else if (!strcmp(name, "flip_slot_cap")) { ctx->flipSlotCap = std::max(1, std::min((int)v, (int)FLIP_SLOTS)); ctx->dropGraph(); }
  else if (!strcmp(name, "flip_scatter")) { ctx->flipScatter = v >= 2.0 ? 2 : (v != 0.0 ? 1 : 0); ctx->dropGraph(); }
  else if (!strcmp(name, "row_price_frac")) { ctx->rowPriceFrac = v < 0.0 ? 0.0 : v; ctx->dropGraph(); }
  else if (!strcmp(name, "sell_windows")) {
    ctx->sellWindows = v != 0.0;
    if (ctx->n > 0 && ctx->D.colStart) {
      ctx->dropGraph();
      return ctx->buildSell();
    }
  }
  else if (!strcmp(name, "shard_grow"))
    ctx->shardGrow = v != 0.0;
  else if (!strcmp(name, "dc_wide")) {
    if ((int)v != ctx->dcWide)
      ctx->dropGraph();
    ctx->dcWide = (int)v;
  }
  else if (!strcmp(name, "price_lds_min_windows") || !strcmp(name, "price_lds_grid")) {
    if (ctx->n > 0 && ctx->D.colStart)
      return -2;  // read by buildSell: set before clpgpu_load_problem
    (name[10] == 'm' ? ctx->priceLdsMinWindows : ctx->priceLdsGridCap) = std::max(1, (int)v);
  }
  else if (!strcmp(name, "price_lds")) {
    // 0: never; 1 (default): the host picks the chain's pricing form per batch; 2: the LDS form on every pivot (tests)
    if (v != 0.0 && !ctx->priceLds && ctx->n > 0 && ctx->D.colStart)
      return -2;  // decides what buildSell lays out: switch it ON before clpgpu_load_problem
    if ((int)v != ctx->priceLds)
      ctx->dropGraph();
    ctx->priceLds = (int)v;
    if (!ctx->priceLds)
      ctx->priceMode = 0;
  }
  else if (!strcmp(name, "scaling")) {
    if (ctx->n > 0 && ctx->D.colStart)
      return -2;  // the matrix is already on the device in its current units: set before clpgpu_load_problem
    ctx->scalingMode = (int)v;
  }
  else if (!strcmp(name, "flip_list_cap")) { ctx->flipListCap = std::max(1, std::min((int)v, FLIP_LIST_CAP)); ctx->dropGraph(); }
  else return -1;
  return 0;
}

int clpgpu_set_status(clpgpu_context *ctx, const unsigned char *status)
{
  if (!ctx)
    return -99;
  ctx->userStatus.assign(status, status + ctx->N);
  ctx->haveStatus = true;
  return 0;
}

// ClpModel::chgRowLower / chgRowUpper / chgColumnLower / chgColumnUpper / chgObjCoefficients
// (src/ClpModel.cpp:2669-2770): whole-array replacement, the matrix stays resident; the next
// clpgpu_dual starts from these (and from the status given with clpgpu_set_status) -- the re-solve
// pattern of branch and bound (ClpSimplex::dual with a warm basis).
// (with option "scaling" the arrays arrive in the caller's units and are kept in scaled units)
// the caller's arrays are kept (origRow*/origCol*); with scaling on, the pair is rebuilt the way the
// load builds it (scaleBoundPair: infinities stay infinite, gaps below the primal tolerance close,
// ClpSimplex::createRim src/ClpSimplex.cpp:3920-3980)
static void rebuildRowBounds(clpgpu_context *ctx)
{
  for (int i = 0; i < ctx->m; i++) {
    if (ctx->scaled)
      scaleBoundPair(ctx->origRowLower[i], ctx->origRowUpper[i], ctx->rowScale[i], ctx->primalTolerance, ctx->rowLower[i], ctx->rowUpper[i]);
    else {
      ctx->rowLower[i] = ctx->origRowLower[i];
      ctx->rowUpper[i] = ctx->origRowUpper[i];
    }
  }
}
static void rebuildColumnBounds(clpgpu_context *ctx)
{
  for (int j = 0; j < ctx->n; j++) {
    if (ctx->scaled)
      scaleBoundPair(ctx->origColLower[j], ctx->origColUpper[j], 1.0 / ctx->colScale[j], ctx->primalTolerance, ctx->colLower[j], ctx->colUpper[j]);
    else {
      ctx->colLower[j] = ctx->origColLower[j];
      ctx->colUpper[j] = ctx->origColUpper[j];
    }
  }
}
int clpgpu_chg_row_lower(clpgpu_context *ctx, const double *rowLower)
{
  if (!ctx || !ctx->m)
    return -99;
  for (int i = 0; i < ctx->m; i++)
    ctx->origRowLower[i] = rowLower ? rowLower[i] : -1.0e30;
  rebuildRowBounds(ctx);
  return 0;
}
int clpgpu_chg_row_upper(clpgpu_context *ctx, const double *rowUpper)
{
  if (!ctx || !ctx->m)
    return -99;
  for (int i = 0; i < ctx->m; i++)
    ctx->origRowUpper[i] = rowUpper ? rowUpper[i] : 1.0e30;
  rebuildRowBounds(ctx);
  return 0;
}
int clpgpu_chg_column_lower(clpgpu_context *ctx, const double *columnLower)
{
  if (!ctx || !ctx->n)
    return -99;
  for (int j = 0; j < ctx->n; j++)
    ctx->origColLower[j] = columnLower ? columnLower[j] : 0.0;
  rebuildColumnBounds(ctx);
  return 0;
}
int clpgpu_chg_column_upper(clpgpu_context *ctx, const double *columnUpper)
{
  if (!ctx || !ctx->n)
    return -99;
  for (int j = 0; j < ctx->n; j++)
    ctx->origColUpper[j] = columnUpper ? columnUpper[j] : 1.0e30;
  rebuildColumnBounds(ctx);
  return 0;
}
int clpgpu_chg_obj_coefficients(clpgpu_context *ctx, const double *objIn)
{
  if (!ctx || !ctx->n)
    return -99;
  for (int j = 0; j < ctx->n; j++) {
    double v = objIn ? objIn[j] : 0.0;
    ctx->origObj[j] = v;
    ctx->obj[j] = ctx->scaled ? v * ctx->colScale[j] : v;
  }
  return 0;
}

// ClpPackedMatrix::scale (src/ClpPackedMatrix.cpp:4120) as a stand-alone host computation: the row
// and column factors the engine applies when the "scaling" option is set.  No context and no device
// needed.  Returns 1 if the matrix would be left unscaled (factors all 1), 0 otherwise, -1 on bad input.
int clpgpu_scale_factors(int numberRows, int numberColumns, const int *columnStart, const int *rowIndex, const double *element,
                         const double *columnLower, const double *columnUpper, const double *rowLower, const double *rowUpper,
                         int mode, double primalTolerance, double *rowScale, double *columnScale)
{
  if (numberRows <= 0 || numberColumns <= 0 || !columnStart || !rowIndex || !element || !rowScale || !columnScale || mode < 1 ||
      mode > 4 || !columnLower || !columnUpper || !rowLower || !rowUpper)
    return -1;
  double dualTolerance = 1.0e-7, zeroTolerance = 1.0e-13;
  std::fill(rowScale, rowScale + numberRows, 1.0);
  std::fill(columnScale, columnScale + numberColumns, 1.0);
  int rc = computeScaleFactors(numberRows, numberColumns, columnStart, rowIndex, element, columnLower, columnUpper, rowLower,
                               rowUpper, mode, primalTolerance, dualTolerance, zeroTolerance, rowScale, columnScale);
  if (rc) {
    std::fill(rowScale, rowScale + numberRows, 1.0);
    std::fill(columnScale, columnScale + numberColumns, 1.0);
  }
  return rc;
}

int clpgpu_dual(clpgpu_context *ctx)
{
  if (!ctx)
    return -99;
  (void)hipSetDevice(ctx->device);
  ctx->started = false;
  ctx->seconds = 0.0;
  return ctx->run(-1);
}

int clpgpu_dual_steps(clpgpu_context *ctx, int iterations)
{
  if (!ctx)
    return -99;
  (void)hipSetDevice(ctx->device);
  if (ctx->started && ctx->problemStatus >= 0)
    return ctx->problemStatus;
  return ctx->run(iterations);
}

// ClpSimplexDual::fastDual (src/ClpSimplexDual.cpp:7227-7480): the dual from the basis at hand with the
// iteration count restarted, as strong branching and the node loops of branch and bound call it.  Returns 0
// when the run came to a conclusion (optimal / infeasible: problem status as clpgpu_dual), 1 when it was
// stopped (iteration limit, or -- without alwaysFinish -- the first time the iteration loop asks for a
// refactorization); the problem status is then 3.
int clpgpu_context::fastDual(bool alwaysFinish)
{
  fastDualMode = alwaysFinish ? 1 : 2;
  started = false;
  seconds = 0.0;
  run(-1);
  fastDualMode = 0;
  if (problemStatus == 10)
    problemStatus = 3;
  return problemStatus == 3 ? 1 : 0;
}

int clpgpu_fast_dual(clpgpu_context *ctx, int alwaysFinish)
{
  if (!ctx || !ctx->m)
    return -99;
  (void)hipSetDevice(ctx->device);
  int rc = ctx->fastDual(alwaysFinish != 0);
  return ctx->problemStatus == 4 ? -99 : rc;
}

// ClpSimplexDual::strongBranching (src/ClpSimplexDual.cpp:6965-7226).  For every listed column: "down" (upper
// bound newUpper[i]) then "up" (lower bound newLower[i]), each a fastDual from the basis the context holds
// (the finished solve's), everything put back afterwards.  On return newUpper[i] / newLower[i] hold the
// change in objective of the down / up branch (1e100 = infeasible), outputStatus[2i], [2i+1] the branch
// status (0 finished, 1 infeasible, 2 unfinished), outputIterations the iteration counts and -- when
// outputSolution is given -- outputSolution[2i], [2i+1] the column solutions.  Return code 0 nothing
// interesting, 1 one branch of some column infeasible, -1 both branches of a column infeasible, -2 error.
// The reference keeps a copy of the factorization and puts it back per branch; here every branch starts
// with a re-inversion of the saved basis (same numbers up to rounding).
int clpgpu_strong_branching(clpgpu_context *ctx, int numberVariables, const int *variables, double *newLower, double *newUpper,
                            double **outputSolution, int *outputStatus, int *outputIterations, int stopOnFirstInfeasible,
                            int alwaysFinish)
{
  if (!ctx || !ctx->m || numberVariables < 0 || (numberVariables && (!variables || !newLower || !newUpper || !outputStatus || !outputIterations)))
    return -99;
  (void)hipSetDevice(ctx->device);
  if (!ctx->started) {
    ctx->setError("clpgpu_strong_branching: no solve to branch from (call clpgpu_dual first)");
    return -2;
  }
  const int n = ctx->n, N = ctx->N;
  for (int i = 0; i < numberVariables; i++)
    if (variables[i] < 0 || variables[i] >= n)
      return -99;
  const double saveObjectiveValue = ctx->objectiveValue;
  const int saveIterations = ctx->numberIterations, saveProblemStatus = ctx->problemStatus;
  std::vector<unsigned char> saveStatus(N);
  if (ctx->d2h(saveStatus.data(), ctx->D.status, N))
    return -2;
  for (int i = 0; i < N; i++)
    saveStatus[i] &= 7;
  const bool hadStatus = ctx->haveStatus;
  const std::vector<unsigned char> saveUserStatus = ctx->userStatus;
  int returnCode = 0;
  bool failed = false;
  auto branch = [&](int iColumn, bool down, double bound, double &objectiveChange, int slot) {
    double &external = down ? ctx->origColUpper[iColumn] : ctx->origColLower[iColumn];
    const double saveBound = external;
    external = bound;
    rebuildColumnBounds(ctx);
    ctx->userStatus = saveStatus;
    ctx->haveStatus = true;
    int status = ctx->fastDual(alwaysFinish != 0);
    if (ctx->problemStatus == 4)
      failed = true;
    // make sure plausible (:7060)
    const double obj = std::max(ctx->objectiveValue, saveObjectiveValue);
    if (status && ctx->problemStatus != 3) {
      // not finished - might be optimal (:7061-7070; no dual objective limit in the engine)
      if (!ctx->numberPrimalInfeasibilities)
        ctx->problemStatus = 0;
      status = ctx->problemStatus;
    }
    if (ctx->problemStatus == 3)
      status = 2;
    if (status || ctx->problemStatus == 0) {
      objectiveChange = obj - saveObjectiveValue;
    } else {
      objectiveChange = 1.0e100;
      status = 1;
    }
    if (outputSolution && outputSolution[slot]) {
      std::vector<double> all(N);
      if (clpgpu_get_solution(ctx, all.data()))
        failed = true;
      std::copy(all.begin(), all.begin() + n, outputSolution[slot]);
    }
    outputStatus[slot] = status;
    outputIterations[slot] = ctx->numberIterations;
    external = saveBound;
  };
  int iSolution = 0;
  for (int i = 0; i < numberVariables && !failed; i++) {
    const int iColumn = variables[i];
    double down, up;
    branch(iColumn, true, newUpper[i], down, iSolution++);
    if (failed)
      break;
    branch(iColumn, false, newLower[i], up, iSolution++);
    newUpper[i] = down;
    newLower[i] = up;
    // both sides feasible: nothing; one side: 1 (and stop if asked); neither: -1 and stop (:7177-7205)
    if (down < 1.0e100) {
      if (up >= 1.0e100) {
        returnCode = 1;
        if (stopOnFirstInfeasible)
          break;
      }
    } else {
      if (up < 1.0e100) {
        returnCode = 1;
        if (stopOnFirstInfeasible)
          break;
      } else {
        returnCode = -1;
        break;
      }
    }
  }
  // everything back: bounds, basis, the factorization and solution of the saved basis, the objective
  rebuildColumnBounds(ctx);
  ctx->userStatus = saveStatus;
  ctx->haveStatus = true;
  ctx->started = false;
  ctx->run(0);
  if (ctx->problemStatus == 4)
    failed = true;
  ctx->userStatus = hadStatus ? saveUserStatus : saveStatus;
  ctx->objectiveValue = saveObjectiveValue;
  ctx->numberIterations = saveIterations;
  if (!failed)
    ctx->problemStatus = saveProblemStatus;
  return failed ? -2 : returnCode;
}

int clpgpu_problem_status(const clpgpu_context *ctx) { return ctx ? ctx->problemStatus : -99; }
int clpgpu_number_iterations(const clpgpu_context *ctx) { return ctx ? ctx->numberIterations : 0; }
double clpgpu_objective_value(const clpgpu_context *ctx) { return ctx ? ctx->objectiveValue : 0.0; }

int clpgpu_get_solution(clpgpu_context *ctx, double *solution)
{
  if (!ctx)
    return -99;
  int rc = ctx->d2h(solution, ctx->D.sol, ctx->N);
  if (!rc && ctx->scaled) {  // unscale (ClpSimplex::deleteRim, src/ClpSimplex.cpp:3376-3412)
    for (int j = 0; j < ctx->n; j++)
      solution[j] *= ctx->colScale[j];
    for (int i = 0; i < ctx->m; i++)
      solution[ctx->n + i] /= ctx->rowScale[i];
  }
  return rc;
}
int clpgpu_get_reduced_costs(clpgpu_context *ctx, double *dj)
{
  if (!ctx)
    return -99;
  int rc = ctx->d2h(dj, ctx->D.dj, ctx->N);
  if (!rc && ctx->scaled) {
    for (int j = 0; j < ctx->n; j++)
      dj[j] /= ctx->colScale[j];
    for (int i = 0; i < ctx->m; i++)
      dj[ctx->n + i] *= ctx->rowScale[i];
  }
  return rc;
}
int clpgpu_get_status(clpgpu_context *ctx, unsigned char *status)
{
  if (!ctx)
    return -99;
  return ctx->d2h(status, ctx->D.status, ctx->N);
}
int clpgpu_get_pivot_variable(clpgpu_context *ctx, int *pivotVariable)
{
  if (!ctx)
    return -99;
  return ctx->d2h(pivotVariable, ctx->D.pivotVariable, ctx->m);
}
int clpgpu_get_pivot_log(clpgpu_context *ctx, clpgpu_pivot_record *out, int maxRecords)
{
  if (!ctx)
    return -99;
  if (ctx->pullCtrl())
    return -99;
  int total = ctx->hCtrl->logCount;
  int avail = total < ctx->logCapacity ? total : ctx->logCapacity;
  int count = avail < maxRecords ? avail : maxRecords;
  if (out && count > 0)
    if (ctx->d2h((PivotRecord *)out, ctx->D.log, count))
      return -99;
  return total;
}
int clpgpu_get_row_weights(clpgpu_context *ctx, double *weights, double *infeasibility)
{
  if (!ctx)
    return -99;
  int rc = 0;
  if (weights)
    rc |= ctx->d2h(weights, ctx->D.weights, ctx->m);
  if (infeasibility)
    rc |= ctx->d2h(infeasibility, ctx->D.infeas, ctx->m);
  return rc;
}
int clpgpu_get_kernel_times(clpgpu_context *ctx, int maxKernels, const char **names, double *milliseconds, long *launches)
{
  if (!ctx)
    return -99;
  int count = (int)ctx->ktNames.size();
  for (int i = 0; i < count && i < maxKernels; i++) {
    if (names)
      names[i] = ctx->ktNames[i];
    if (milliseconds)
      milliseconds[i] = ctx->ktMs[i];
    if (launches)
      launches[i] = ctx->ktCount[i];
  }
  return count;
}
int clpgpu_get_stats(clpgpu_context *ctx, clpgpu_stats *stats)
{
  if (!ctx || !stats || !ctx->hCtrl)
    return -99;
  if (ctx->pullCtrl())
    return -99;
  *stats = ctx->stats;
  if (getenv("CLPGPU_DEBUG_STATS")) {
    const long long *g = ctx->hCtrl->dbg;
    fprintf(stderr, "clpgpu dbg: dc small %lld big %lld passes %lld tries %lld sumNc %lld mapped %lld full %lld ticksSmall %lld ticksBig %lld | flip iters %lld flips %lld entries %lld sequential %lld | scattered %lld select rows %lld hot rows %lld\n",
            g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11], g[12], g[14], g[13], g[15]);
    const long long *q = ctx->hCtrl->dbgDc;
    fprintf(stderr, "clpgpu dbg: ratio test working-set path: calls %lld, ticks/call %.0f, before the passes %.0f, max %lld; candidates of breakpoint class "
                    "<= 0 / 1 / 2 summed over the pivots with a long list: %lld / %lld / %lld; fall-backs to the full list %lld (sum of their working-set classes %lld)\n",
            q[0], q[0] ? (double)q[1] / q[0] : 0.0, q[0] ? (double)q[2] / q[0] : 0.0, q[3], q[4], q[5], q[6], q[7] % 1000000LL, q[7] / 1000000LL);
    const long long *w = ctx->hCtrl->dbgCc;
    fprintf(stderr, "clpgpu dbg: ratio test, final batch: moved into one wave %lld times, too large %lld times, mean size %.1f; ticks per call in the trips %.0f, "
                    "in the coarse passes before them %.0f; calls of the wide kernel %lld\n",
            w[0], w[1], (w[0] + w[1]) ? (double)w[2] / (double)(w[0] + w[1]) : 0.0, (w[0] + w[1] + w[5]) ? (double)w[3] / (double)(w[0] + w[1] + w[5]) : 0.0,
            (w[0] + w[1] + w[5]) ? (double)w[4] / (double)(w[0] + w[1] + w[5]) : 0.0, w[5]);
    fprintf(stderr, "clpgpu dbg: ratio test, multi-wave working-set calls %lld, of which the test ended below 8 theta0 (class 0 would have sufficed) %lld\n", w[6], w[7]);
  }
  stats->price_bytes = ctx->hCtrl->statPriceBytes;
  stats->row_bytes = ctx->hCtrl->statRowBytes;
  if (!ctx->timing) {
    stats->row_launches = (long)ctx->hCtrl->statRowLaunches;
    stats->price_launches = (long)(ctx->hCtrl->statPriceLaunches - ctx->hCtrl->statRowLaunches);
  }
  stats->total_ms = ctx->seconds * 1.0e3;
  stats->iterations = ctx->numberIterations;
  stats->refactorizations = ctx->numberRefactorizations;
  stats->nucleus = ctx->luActive ? ctx->hLu.k : ctx->hCtrl->k;
  stats->lu_active = ctx->luActive ? 1 : 0;
  stats->lu_front = ctx->luActive ? ctx->hLu.nF : 0;
  stats->lu_tail = ctx->luActive ? ctx->hLu.k2 : 0;
  stats->lu_factorizations = ctx->luFactorizations;
  stats->lu_front_ms = ctx->luFrontSeconds * 1.0e3;
  stats->lu_invert_ms = ctx->luInvertSeconds * 1.0e3;
  stats->lu_build_ms = ctx->luBuildSeconds * 1.0e3;
  stats->eta_count = ctx->hCtrl->pivots;
  stats->perturbations = ctx->numberPerturbations;
  stats->backwards_restores = ctx->numberBackwards;
  stats->loop_flags = ctx->numberLoopFlags;
  stats->accuracy_restores = ctx->numberAccuracyRestores;
  stats->singular_restores = ctx->numberSingularRestores;
  stats->price_form = (ctx->priceMode == 1 && ctx->jdsReady) ? 1 : 0;
  stats->dense_pi_launches = (long)ctx->hCtrl->statDensePi;
  stats->price_form_switches = ctx->priceFormSwitches;
  stats->exits_scheduled = ctx->exitScheduled;
  stats->exits_alpha_check = ctx->exitAlphaCheck;
  stats->exits_backwards = ctx->exitBackwards;
  stats->exits_bad_update = ctx->exitBadUpdate;
  stats->comm_mode = ctx->commActive ? ctx->commMode : 0;
  stats->shard_cand_cap = ctx->shardCandCap;
  stats->free_first_rows = ctx->numberFreeFirstRows;
  stats->free_entered = ctx->numberFreeEntered;
  stats->try_primal_exits = ctx->numberTryPrimal;
  stats->dc_wide_timeouts = ctx->numberDcWideTimeouts;
  stats->chuzr_partial_scans = ctx->hCtrl->chuzrPartialScans;
  stats->chuzr_recalls = ctx->hCtrl->chuzrRecalls;
  stats->chuzr_ordered_walks = ctx->hCtrl->chuzrOrdered;
  stats->eta_compact_slots = (ctx->luActive && ctx->hCtrl->luCompactOn) ? ctx->hCtrl->luCompactCount : 0;
  stats->factor_elements = (long)ctx->hCtrl->factorElements;
  stats->nucleus_capacity = ctx->kcap;
  stats->refreshes = ctx->numberRefreshes;
  stats->refreshes_rejected = ctx->numberRefreshesRejected;
  return 0;
}

}  // extern "C"

/*
 * clp_dual_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see clp_dual_oracle.h).
 *
 * Restates the revised dual simplex of coin-or/Clp: by default for the fast branch "no free / superbasic
 * nonbasic variables" (moreSpecialOptions_&8, src/ClpSimplexDual.cpp:3685), with option "free_nonbasic" the general
 * one as well; scaling (option "scaling") and
 * cost perturbation (option "perturbation", ClpSimplexDual::perturb :6533) are off unless asked for.  Each
 * function cites the reference lines it follows.
 * Variable order is Clp's: sequences [0,n) structurals, [n,n+m) row slacks, slack column = -e_i
 * (src/ClpSimplex.cpp:3442-3474).
 *
 * Deliberate restrictions (documented in DESIGN.md):
 *  - factorization = slack singletons (src/CoinAbcBaseFactorization1.cpp:2589 pivotColumnSingleton)
 *    followed by the dense LU with partial pivoting of CoinAbcDenseFactorization::factor on the
 *    remaining nucleus, product-form eta updates (replaceColumnPart3).  Because a slack column -e_i
 *    has a single entry, doing the slacks first leaves the other columns untouched, so this is
 *    arithmetically the dense factorization of the whole basis with the zero work skipped.
 *  - of statusOfProblemInDual: the cost rescale after 4(m+n) iterations (:5009-5021), the pivot-tolerance changes
 *    (this factorization has none) and the Cbc-only branches are not restated; when the basis a singular
 *    refactorization falls back to is singular too the solve ends with status 4 (the reference factorizes "safely"
 *    with slacks put in, :5100-5117).
 *  - nonbasic free columns are given "bothFake" bounds at start by default, on this side and on the HIP engine's.  Option
 *    "free_nonbasic" 1 (the same option there) keeps them isFree as the reference does: dualRow's free-first entry (:3005-3055), the
 *    general branch of dualColumn0 (:4058-4179), the free branches of checkDualSolution / checkBothSolutions, firstFree_,
 *    "only free dual infeasibilities: use primal" (:5619-5622).  Checked against HiGHS (tests/test_oracle_free.py), there being
 *    no reference binary; the engine's half is held to this one in tests/test_gpu_free.py.
 *  - CoinThreadRandom lives in CoinUtils (absent); the 32-bit LCG form is used [unverifiable here].
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction, so the arithmetic order is the source order).
 */
#include "clp_dual_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ClpSimplex::Status, src/ClpSimplex.hpp:119-126 */
enum { ST_FREE = 0, ST_BASIC = 1, ST_UPPER = 2, ST_LOWER = 3, ST_SUPER = 4, ST_FIXED = 5 };
/* ClpSimplexDual::FakeBound, src/ClpSimplexDual.hpp (bits 3-4 of status_) */
enum { FAKE_NONE = 0, FAKE_LOWER = 1, FAKE_UPPER = 2, FAKE_BOTH = 3 };
#define FLAGGED_BIT 64
#define REALLY_TINY 1.0e-100 /* COIN_INDEXED_REALLY_TINY_ELEMENT [CoinUtils; from memory] */
#define DEVEX_TRY_NORM 1.0e-4 /* src/ClpSimplex.hpp:2056 */

typedef struct {
  int k;        /* nucleus size = number of basic structurals */
  int *kcol;    /* [k] structural sequence of nucleus column c */
  int *krow;    /* [k] original row pivoted at elimination step c (== basis position of kcol[c]) */
  int *rowToK;  /* [m] -1 when the row's slack is basic, else elimination step of that row */
  double *lu;   /* k*k column-major in pivoted row order: multipliers below the diagonal,
                   U on/above it, diagonal stored inverted (CoinAbcDenseFactorization.cpp:281) */
  long luCap;
  double *t;    /* [m] scratch */
  /* product-form eta file (replaceColumnPart3 :480); stored sparse, arithmetic as dense */
  int nEta, maxEta;
  int *etaPivot;
  double *etaPivotValue; /* 1/alpha */
  int *etaStart;         /* [maxEta+1] */
  int *etaIndex;
  double *etaValue;
  long etaCap;
} Factor;

#define ORC_CYCLE 12 /* CLP_CYCLE, src/ClpSolve.hpp:435 */
#define ORC_PROGRESS 5 /* CLP_PROGRESS, src/ClpSolve.hpp */
struct OrcModel {
  int m, n;
  int *colStart, *row;
  double *elem;
  int *rowStart, *rcol; /* row copy (ClpSimplex::createRim builds rowCopy_, src/ClpSimplex.cpp:3648) */
  double *relem;
  double *alphaDense; /* [n] scratch of the by-row path */
  unsigned char *touched;
  int *touchList;
  int priceByRow; /* 1: choose by row / by column like ClpPackedMatrix::transposeTimes :727-754 */
  long pricedByRow, pricedByColumn;
  double *colLower, *colUpper, *obj, *rowLower, *rowUpper;
  /* rim arrays [columns | rows], src/ClpSimplex.hpp:1864-1922 */
  double *lower, *upper, *cost, *dj, *sol;
  unsigned char *status;
  int *pivotVariable;
  int haveStatus;
  /* parameters (defaults src/ClpSimplex.cpp:49-80) */
  double primalTolerance, dualTolerance, dualToleranceBase, dualBound, zeroTolerance, acceptablePivot_,
      largeValue;
  int maximumIterations, pivotRule, maximumPivots, logLevel;
  /* state */
  int problemStatus, numberIterations, numberRefactorizations;
  int pivotRow, sequenceIn, sequenceOut, directionIn, directionOut;
  double alpha, theta, dualOut, dualIn, valueIn, valueOut, lowerIn, upperIn, lowerOut, upperOut;
  double objectiveValue;
  double largestPrimalError, largestDualError;
  double sumPrimalInfeasibilities, sumDualInfeasibilities, sumOfRelaxedPrimalInfeasibilities,
      sumOfRelaxedDualInfeasibilities;
  int numberPrimalInfeasibilities, numberDualInfeasibilities;
  int numberFake, numberChanged, numberTimesOptimal, forceFactorization, lastBadIteration;
  unsigned int seed;
  int perturbation, perturbationOption; /* ClpSimplex::perturbation_: 50 auto, 100 "only the kick", 101 perturbed, 102 never again */
  double *perturbationArray;            /* ClpSimplex::perturbationArray_ [2n], filled on the first perturb() */
  const double *objBeforeScaling;       /* objective() as the caller gave it (perturb() looks at that one, :6566) */
  int numberPerturbations;              /* how many times perturb() changed costs in the last solve (test hook) */
  /* ClpSimplexProgress (src/ClpSolve.cpp:4289-4725), the part the dual uses: the last ORC_PROGRESS status checks */
  double progObjective[ORC_PROGRESS], progInfeasibility[ORC_PROGRESS];
  int progNumberInfeasibilities[ORC_PROGRESS], progIteration[ORC_PROGRESS];
  int progTimes, progBadTimes, progReallyBadTimes, progTimesFlagged;
  int progressFlag;               /* ClpSimplex::progressFlag_: 1 a fixed variable left, 2 a free one came in, 4 costs copied, 8 has looked optimal */
  double *costCopy;               /* the second half of cost_ once progressFlag_ & 4 (:5381-5390) */
  double bestPossibleImprovement; /* ClpSimplex::checkDualSolution :3087 */
  unsigned char *saveStatus;      /* ClpSimplex::saveStatus_ / savedSolution_: the basis of the last good status check (:6160-6175) */
  double *savedSolution;
  int numberBackwards, numberLoopFlags; /* test hooks: times the "objective going backwards" restore ran, times looping() acted */
  int checkBoth;                  /* option "check_both": gutsOfSolution ends in checkBothSolutions (default 1) or in the older pair (0) */
  int rimInfeasible;              /* the start-up sanity check found crossing bounds: status 1 without a rim to look at */
  int freeNonbasic;               /* option "free_nonbasic": 1 = nonbasic free columns stay isFree as in the reference (dualRow's free-first entry
                                     :2962-3140, the general branch of dualColumn0 :4058-4179); 0 (default, what the HIP engine does) = they
                                     are given bothFake bounds at start */
  int noFreeOrSuper;              /* moreSpecialOptions_ & 8: "no free or super basic" as the last checkBothSolutions saw it (always 1 with
                                     free_nonbasic 0) */
  int firstFree;                  /* ClpSimplex::firstFree_ */
  int numberDualInfeasibilitiesWithoutFree;
  double badFree;                 /* dualColumn0's badFree of the last pivot row */
  int numberFreeFirstRows, numberFreeEntered; /* test hooks: pivot rows chosen by the free-first entry, pivots that brought a free variable in
                                                 through the general branch's freePivot */
  int debugSingularAt;            /* fault injection (option "debug_singular_at"): the refactorization of the first status check at or after
                                     this iteration is taken as singular; -1 off */
  int tryPrimal;                  /* option "try_primal" (0 default, as the engine's): 1 = gutsOfDual's "problems - try primal" exit is there */
  int steepestMode;               /* option "steepest_mode": ClpDualRowSteepest::mode_ (0 uninitialized, 1 full, 2 partial, 3 = the
                                     constructor's default: partial scan sized by the factorization's fill, :258-278); start-up weights
                                     are the unit weights of every mode but 1 */
  int steepestElements;           /* option "steepest_elements": what stands for factorization()->numberElements() in mode 3 (see
                                     factorElementsModel): 0 = entries of the basic structural columns, 1 = this LU's own nonzeros */
  long factorElements;            /* as of the last factorization */
  double debugToleranceFactor;    /* option "debug_tolerance_factor" (fault injection, see steepestPivotRow); 0 off */
  int chuzrFloor;                 /* option "debug_chuzr_floor": the 2000 of :260-276 (tests lower it so that small LPs scan partially) */
  int numberPartialScans, numberChuzrRecalls; /* test hooks: pivotRow calls that scanned part of the list; second calls (:338-346) */
  int debugPlainLu;               /* option "debug_plain_lu": 1 = the nucleus is eliminated by the unblocked loop (luPlain) */
  int numberTryPrimal;            /* test hook: times gutsOfDual's "problems - try primal" exit was taken (:540-547) */
  int numberSingularRestores;     /* test hook: times a singular refactorization sent the solve back to the saved basis (:5060-5125) */
  int debugBadAccuracyAt;         /* fault injection (option "debug_bad_accuracy_at"): the first status check at or after this iteration
                                     finds largestPrimalError_ = 1e16; -1 off */
  int numberAccuracyRestores;     /* test hook: times the "bad accuracy, treat as singular" restore ran (:5237-5318) */
  int debugBackwardsAt;           /* fault injection (option "debug_backwards_at"): pretend the objective dropped at the first status
                                     check at or after this iteration; -1 off */
  int cycIn[ORC_CYCLE], cycOut[ORC_CYCLE]; /* ClpSimplexProgress in_ / out_ / way_ */
  char cycWay[ORC_CYCLE];
  int scalingMode;            /* ClpModel::scaling(): 0 off (default here), 1 equilibrium, 2 geometric, 3/4 auto */
  int scalingApplied;         /* last orc_dual really scaled (scale() returned 0) */
  double *rowScale, *colScale; /* [m], [n] of the last scaled solve */
  Factor fac;
  /* dual row pivot */
  double *weights;      /* [m] by basis position */
  double *infeas;       /* [m] dense squared infeasibility */
  int *infIndex;        /* insertion-ordered list, as CoinIndexedVector infeasible_ */
  int numberInfeasible;
  double *savedWeights; /* saveWeights(1)/(2): weights keyed by sequence */
  int *savedWhich;
  int haveSavedWeights;
  double *altWeightValue; /* alternateWeights_ for unrollWeights */
  int *altWeightIndex;
  int numberAlt;
  /* work vectors */
  double *rowWork0, *rowWork1, *rowWork2, *rowWork3; /* dense length m */
  int *piIndex;
  double *piValue;
  int numberPi; /* packed BTRAN result / row part of tableau row */
  int *colIndex;
  double *colValue;
  int numberColNz; /* packed column part of tableau row */
  int *wIndex;
  double *wValue;
  int numberW; /* packed updated column */
  int *spareIndex[2];
  double *spareValue[2]; /* the two flip-flop candidate lists of dualColumn */
  int numberCandidates;
  double upperThetaFirst;
  int *rowFlip, numberRowFlip, *colFlip, numberColFlip;
  /* log */
  OrcPivotRecord *log;
  int logCount, logCap;
  double seconds, startupSeconds; /* wall clock of the last orc_dual; its part before the first status check (start-up factorization + resync) */
};

/* ------------------------------------------------------------------------------------------ */
static inline int getStatus(const OrcModel *M, int i) { return M->status[i] & 7; }
static inline void setStatus(OrcModel *M, int i, int s) { M->status[i] = (unsigned char)((M->status[i] & ~7) | s); }
static inline int getFake(const OrcModel *M, int i) { return (M->status[i] >> 3) & 3; }
static inline void setFake(OrcModel *M, int i, int f) { M->status[i] = (unsigned char)((M->status[i] & ~24) | (f << 3)); }
static inline int flagged(const OrcModel *M, int i) { return (M->status[i] & FLAGGED_BIT) != 0; }
static inline void setFlagged(OrcModel *M, int i) { M->status[i] |= FLAGGED_BIT; }
static inline void clearFlagged(OrcModel *M, int i) { M->status[i] &= (unsigned char)~FLAGGED_BIT; }
static void restoreCosts(OrcModel *M);
/* ClpSimplexProgress accessors (src/ClpSolve.cpp:4676-4712); slot ORC_PROGRESS-1 is the newest status check */
static inline double progressLastObjective(const OrcModel *M, int back) { return M->progObjective[ORC_PROGRESS - 1 - back]; }
static inline int progressLastIteration(const OrcModel *M, int back) { return M->progIteration[ORC_PROGRESS - 1 - back]; }
static inline void progressModifyObjective(OrcModel *M, double value) { M->progObjective[ORC_PROGRESS - 1] = value; }
static inline void progressClearBadTimes(OrcModel *M) { M->progBadTimes = 0; }
static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }

/* CoinThreadRandom::randomDouble, COIN_OWN_RANDOM_32 form [CoinUtils, from memory] */
static double randomDouble(OrcModel *M)
{
  M->seed = 1664525u * M->seed + 1013904223u;
  return ((double)M->seed) / 4294967296.0;
}

static double originalLower(const OrcModel *M, int i) { return i < M->n ? M->colLower[i] : M->rowLower[i - M->n]; }
static double originalUpper(const OrcModel *M, int i) { return i < M->n ? M->colUpper[i] : M->rowUpper[i - M->n]; }

/* ------------------------------------------------------------------------------------------ */
OrcModel *orc_create(int m, int n, const int *colStart, const int *row, const double *elem,
                     const double *colLower, const double *colUpper, const double *obj,
                     const double *rowLower, const double *rowUpper)
{
  OrcModel *M = (OrcModel *)calloc(1, sizeof(OrcModel));
  int nz = colStart[n], N = m + n;
  M->m = m;
  M->n = n;
  M->colStart = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  memcpy(M->colStart, colStart, sizeof(int) * (size_t)(n + 1));
  M->row = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));
  memcpy(M->row, row, sizeof(int) * (size_t)nz);
  M->elem = (double *)malloc(sizeof(double) * (size_t)(nz > 0 ? nz : 1));
  memcpy(M->elem, elem, sizeof(double) * (size_t)nz);
  M->rowStart = (int *)calloc((size_t)m + 2, sizeof(int));
  M->rcol = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));
  M->relem = (double *)malloc(sizeof(double) * (size_t)(nz > 0 ? nz : 1));
  for (int p = 0; p < nz; p++)
    M->rowStart[row[p] + 1]++;
  for (int i = 0; i < m; i++)
    M->rowStart[i + 1] += M->rowStart[i];
  {
    int *fill = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    memcpy(fill, M->rowStart, sizeof(int) * (size_t)m);
    for (int j = 0; j < n; j++)
      for (int p = colStart[j]; p < colStart[j + 1]; p++) {
        int q = fill[row[p]]++;
        M->rcol[q] = j;
        M->relem[q] = elem[p];
      }
    free(fill);
  }
  M->alphaDense = (double *)calloc((size_t)n + 1, sizeof(double));
  M->touched = (unsigned char *)calloc((size_t)n + 1, 1);
  M->touchList = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  M->priceByRow = 1;
#define DUP(dst, src, cnt)                                    \
  dst = (double *)malloc(sizeof(double) * (size_t)((cnt) + 1)); \
  memcpy(dst, src, sizeof(double) * (size_t)(cnt))
  DUP(M->colLower, colLower, n);
  DUP(M->colUpper, colUpper, n);
  DUP(M->obj, obj, n);
  DUP(M->rowLower, rowLower, m);
  DUP(M->rowUpper, rowUpper, m);
#undef DUP
#define DALLOC(cnt) (double *)calloc((size_t)(cnt) + 1, sizeof(double))
#define IALLOC(cnt) (int *)calloc((size_t)(cnt) + 1, sizeof(int))
  M->lower = DALLOC(N);
  M->upper = DALLOC(N);
  M->cost = DALLOC(N);
  M->dj = DALLOC(N);
  M->sol = DALLOC(N);
  M->status = (unsigned char *)calloc((size_t)N + 1, 1);
  M->pivotVariable = IALLOC(m);
  /* ClpSimplex constructor defaults, src/ClpSimplex.cpp:44-163 */
  M->primalTolerance = 1.0e-7;
  M->dualTolerance = M->dualToleranceBase = 1.0e-7;
  M->dualBound = 1.0e10;
  M->zeroTolerance = 1.0e-13;
  M->acceptablePivot_ = 1.0e-8;
  M->largeValue = 1.0e15;
  M->maximumIterations = 2147483647;
  M->pivotRule = 1;
  M->checkBoth = 1; /* gutsOfSolution ends in checkBothSolutions, as in this reference version (src/ClpSimplex.cpp:762) */
  M->freeNonbasic = 0;
  M->tryPrimal = 0;
  M->noFreeOrSuper = 1;
  M->firstFree = -1;
  M->maximumPivots = 200; /* CoinAbcBaseFactorization1.cpp:142 default */
  M->chuzrFloor = 2000;
  M->steepestMode = 3;    /* ClpDualRowSteepest(int mode = 3), src/ClpDualRowSteepest.hpp:118; ClpSimplex's constructor takes it, src/ClpSimplex.cpp:158 */
  M->seed = 1234567u;     /* src/ClpModel.cpp:149 */
  M->debugBackwardsAt = -1;
  M->debugBadAccuracyAt = -1;
  M->debugSingularAt = -1;
  M->saveStatus = (unsigned char *)calloc((size_t)N + 1, 1);
  M->savedSolution = DALLOC(N);
  M->costCopy = DALLOC(N);
  M->perturbationOption = 102; /* off; the reference's constructor default is 100 (src/ClpSimplex.cpp:114), the clp CLI's 50 */
  M->forceFactorization = -1;
  M->lastBadIteration = -999999;
  M->fac.rowToK = IALLOC(m);
  M->fac.kcol = IALLOC(m);
  M->fac.krow = IALLOC(m);
  M->fac.t = DALLOC(m);
  M->weights = DALLOC(m);
  M->infeas = DALLOC(m);
  M->infIndex = IALLOC(m);
  M->savedWeights = DALLOC(m);
  M->savedWhich = IALLOC(m);
  M->altWeightValue = DALLOC(m);
  M->altWeightIndex = IALLOC(m);
  M->rowWork0 = DALLOC(m);
  M->rowWork1 = DALLOC(m);
  M->rowWork2 = DALLOC(m);
  M->rowWork3 = DALLOC(m);
  M->piIndex = IALLOC(m);
  M->piValue = DALLOC(m);
  M->colIndex = IALLOC(n);
  M->colValue = DALLOC(n);
  M->wIndex = IALLOC(m);
  M->wValue = DALLOC(m);
  for (int i = 0; i < 2; i++) {
    M->spareIndex[i] = IALLOC(N);
    M->spareValue[i] = DALLOC(N);
  }
  M->rowFlip = IALLOC(m);
  M->colFlip = IALLOC(n);
  return M;
}

static void freeFactor(Factor *F)
{
  free(F->kcol);
  free(F->krow);
  free(F->rowToK);
  free(F->lu);
  free(F->t);
  free(F->etaPivot);
  free(F->etaPivotValue);
  free(F->etaStart);
  free(F->etaIndex);
  free(F->etaValue);
}

void orc_destroy(OrcModel *M)
{
  if (!M)
    return;
  free(M->colStart); free(M->row); free(M->elem);
  free(M->rowStart); free(M->rcol); free(M->relem); free(M->alphaDense); free(M->touched); free(M->touchList);
  free(M->colLower); free(M->colUpper); free(M->obj); free(M->rowLower); free(M->rowUpper);
  free(M->lower); free(M->upper); free(M->cost); free(M->dj); free(M->sol); free(M->status);
  free(M->pivotVariable);
  free(M->rowScale); free(M->colScale);
  freeFactor(&M->fac);
  free(M->weights); free(M->infeas); free(M->infIndex); free(M->savedWeights); free(M->savedWhich);
  free(M->altWeightValue); free(M->altWeightIndex);
  free(M->rowWork0); free(M->rowWork1); free(M->rowWork2); free(M->rowWork3);
  free(M->piIndex); free(M->piValue); free(M->colIndex); free(M->colValue); free(M->wIndex); free(M->wValue);
  for (int i = 0; i < 2; i++) { free(M->spareIndex[i]); free(M->spareValue[i]); }
  free(M->rowFlip); free(M->colFlip); free(M->log);
  free(M->perturbationArray);
  free(M->saveStatus); free(M->savedSolution); free(M->costCopy);
  free(M);
}

int orc_set_option(OrcModel *M, const char *name, double v)
{
  if (!strcmp(name, "pivot_rule")) M->pivotRule = (int)v;
  else if (!strcmp(name, "max_iterations")) M->maximumIterations = (int)v;
  else if (!strcmp(name, "max_pivots")) {
    if (v > 0) {
      M->maximumPivots = (int)v;
    } else { /* ClpSimplex::defaultFactorizationFrequency, src/ClpSimplex.cpp:11401-11429 */
      int mm = M->m, f;
      if (mm < 10000)
        f = 75 + mm / 50;
      else if (mm < 100000)
        f = 75 + 200 + (mm - 10000) / 200;
      else
        f = 1000;
      M->maximumPivots = f < 1000 ? f : 1000;
    }
  }
  else if (!strcmp(name, "dual_bound")) M->dualBound = v;
  else if (!strcmp(name, "primal_tolerance")) M->primalTolerance = v;
  else if (!strcmp(name, "dual_tolerance")) M->dualTolerance = M->dualToleranceBase = v;
  else if (!strcmp(name, "log_level")) M->logLevel = (int)v;
  else if (!strcmp(name, "random_seed")) M->seed = (unsigned int)v;
  else if (!strcmp(name, "price_by_row")) M->priceByRow = (int)v;
  else if (!strcmp(name, "scaling")) M->scalingMode = (int)v;
  else if (!strcmp(name, "perturbation")) M->perturbationOption = (int)v;
  else if (!strcmp(name, "debug_backwards_at")) M->debugBackwardsAt = (int)v;
  else if (!strcmp(name, "debug_bad_accuracy_at")) M->debugBadAccuracyAt = (int)v;
  else if (!strcmp(name, "debug_singular_at")) M->debugSingularAt = (int)v;
  else if (!strcmp(name, "check_both")) M->checkBoth = (int)v;
  else if (!strcmp(name, "free_nonbasic")) M->freeNonbasic = (int)v;
  else if (!strcmp(name, "try_primal")) M->tryPrimal = (int)v;
  else if (!strcmp(name, "debug_plain_lu")) M->debugPlainLu = (int)v;
  else if (!strcmp(name, "steepest_mode")) M->steepestMode = (int)v;
  else if (!strcmp(name, "steepest_elements")) M->steepestElements = (int)v;
  else if (!strcmp(name, "debug_last_bad_iteration")) M->lastBadIteration = (int)v; /* fault injection: lastBadIteration_ the solve starts with (-999999) */
  else if (!strcmp(name, "debug_tolerance_factor")) M->debugToleranceFactor = v;
  else if (!strcmp(name, "debug_chuzr_floor")) M->chuzrFloor = (int)v > 1 ? (int)v : 1;
  else return -1;
  return 0;
}

void orc_set_status(OrcModel *M, const unsigned char *status)
{
  memcpy(M->status, status, (size_t)(M->m + M->n));
  M->haveStatus = 1;
}

/* ------------------------------------------------------------------------------------------ */
/* matrix kernels                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* ClpPackedMatrix::times :296 -- y += scalar*A*x, column loop, skips x_j == 0 */
void orc_times(const OrcModel *M, double scalar, const double *x, double *y)
{
  for (int j = 0; j < M->n; j++) {
    double value = x[j];
    if (value) {
      value *= scalar;
      for (int p = M->colStart[j]; p < M->colStart[j + 1]; p++)
        y[M->row[p]] += value * M->elem[p];
    }
  }
}

/* ClpPackedMatrix::transposeTimes :362 -- y_j += scalar * sum_i x_i a_ij */
void orc_transpose_times(const OrcModel *M, double scalar, const double *x, double *y)
{
  for (int j = 0; j < M->n; j++) {
    double value = 0.0;
    for (int p = M->colStart[j]; p < M->colStart[j + 1]; p++)
      value += x[M->row[p]] * M->elem[p];
    y[j] += value * scalar;
  }
}

/* Fused row pricing + first ratio pass.  Row part: ClpPackedMatrix.cpp:1039-1072, column part:
 * gutsOfTransposeTimesUnscaled :1856-1988.  `pi` is negated while being expanded (:1009-1014)
 * because the caller passes scalar = -1 (ClpSimplexDual.cpp:1300). */
static int priceRowFused(const OrcModel *M, int numberPi, const int *piIndex, const double *piValue,
                         double *piDense /* zeroed length m, returned zeroed */,
                         const unsigned char *status, const double *dj, double zeroTolerance,
                         double dualTolerance, double acceptablePivot, int *outIndex, double *outValue,
                         int *numberCandidatesOut, int *candIndex, double *candValue, double *upperThetaOut)
{
  const int n = M->n;
  const double multiplier[2] = { -1.0, 1.0 };
  const double dualT = -dualTolerance;
  const double tentativeTheta = 1.0e15; /* :1857 (pass 0 proper uses 1e25, ClpSimplexDual.cpp:3679) */
  double upperTheta = 1.0e31;
  int numberRemaining = 0;
  for (int i = 0; i < numberPi; i++)
    piDense[piIndex[i]] = -piValue[i];
  /* row (slack) part of the tableau row is pi itself */
  const unsigned char *statusRow = status + n;
  const double *djRow = dj + n;
  for (int i = 0; i < numberPi; i++) {
    int iSequence = piIndex[i];
    int iStatus = (statusRow[iSequence] & 3) - 1;
    if (iStatus > 0) {
      double mult = multiplier[iStatus - 1];
      double alpha = piValue[i] * mult;
      if (alpha > 0.0) {
        double oldValue = djRow[iSequence] * mult;
        double value = oldValue - tentativeTheta * alpha;
        if (value < dualT) {
          value = oldValue - upperTheta * alpha;
          if (value < dualT && alpha >= acceptablePivot)
            upperTheta = (oldValue - dualT) / alpha;
          candValue[numberRemaining] = alpha * mult;
          candIndex[numberRemaining++] = iSequence + n;
        }
      }
    }
  }
  int numberNonZero = 0;
  /* by row or by column?  ClpPackedMatrix::transposeTimes :727-754 (packed pi, no column copy) */
  int byRow = 0;
  if (M->priceByRow) {
    double factor = 0.5;
    if ((double)n * sizeof(double) > 1000000.0) {
      if (M->m * 10 < n)
        factor *= 0.333333333;
      else if (M->m * 4 < n)
        factor *= 0.5;
      else if (M->m * 2 < n)
        factor *= 0.66666666667;
    }
    byRow = !(numberPi > factor * M->m);
  }
  OrcModel *MM = (OrcModel *)M; /* scratch arrays and counters only */
  if (byRow) {
    /* ClpPackedMatrix::transposeTimesByRow :1307 / gutsOfTransposeTimesByRowGE3 :5176: accumulate
       pi_i * row_i into a dense scratch, first-touch list.  pi is walked in ascending row order so
       every alpha_j receives its terms in the same order as the by-column loop (zero terms omitted):
       the values are bit-identical; the touched list is then sorted so the output order is too. */
    MM->pricedByRow++;
    int nTouched = 0;
    for (int i = 0; i < numberPi; i++) {
      int iRow = piIndex[i];
      double piv = -piValue[i];
      for (int q = M->rowStart[iRow]; q < M->rowStart[iRow + 1]; q++) {
        int j = M->rcol[q];
        if (!((status[j] & 3) - 1))
          continue;
        if (!MM->touched[j]) {
          MM->touched[j] = 1;
          MM->touchList[nTouched++] = j;
        }
        MM->alphaDense[j] += piv * M->relem[q];
      }
    }
    /* ascending column order (insertion sort on small lists, qsort otherwise) */
    if (nTouched > 1) {
      int *a = MM->touchList;
      if (nTouched < 64) {
        for (int x = 1; x < nTouched; x++) {
          int v = a[x], y = x - 1;
          while (y >= 0 && a[y] > v) {
            a[y + 1] = a[y];
            y--;
          }
          a[y + 1] = v;
        }
      } else {
        /* counting pass through the touched flags keeps it O(n) worst case */
        int w = 0;
        for (int j = 0; j < n && w < nTouched; j++)
          if (MM->touched[j])
            a[w++] = j;
      }
    }
    for (int t = 0; t < nTouched; t++) {
      int iColumn = MM->touchList[t];
      int wanted = (status[iColumn] & 3) - 1;
      double value = MM->alphaDense[iColumn];
      MM->alphaDense[iColumn] = 0.0;
      MM->touched[iColumn] = 0;
      if (fabs(value) > zeroTolerance) {
        outValue[numberNonZero] = value;
        outIndex[numberNonZero++] = iColumn;
        if (wanted > 0) {
          double mult = multiplier[wanted - 1];
          double alpha = value * mult;
          if (alpha > 0.0) {
            double oldValue = dj[iColumn] * mult;
            double value2 = oldValue - tentativeTheta * alpha;
            if (value2 < dualT) {
              value2 = oldValue - upperTheta * alpha;
              if (value2 < dualT && alpha >= acceptablePivot)
                upperTheta = (oldValue - dualT) / alpha;
              candValue[numberRemaining] = alpha * mult;
              candIndex[numberRemaining++] = iColumn;
            }
          }
        }
      }
    }
  } else {
  MM->pricedByColumn++;
  for (int iColumn = 0; iColumn < n; iColumn++) {
    int wanted = (status[iColumn] & 3) - 1;
    if (wanted) {
      double value = 0.0;
      for (int p = M->colStart[iColumn]; p < M->colStart[iColumn + 1]; p++)
        value += piDense[M->row[p]] * M->elem[p]; /* sequential, ascending p (:1872-1886) */
      if (fabs(value) > zeroTolerance) {
        outValue[numberNonZero] = value;
        outIndex[numberNonZero++] = iColumn;
        if (wanted > 0) {
          double mult = multiplier[wanted - 1];
          double alpha = value * mult;
          if (alpha > 0.0) {
            double oldValue = dj[iColumn] * mult;
            double value2 = oldValue - tentativeTheta * alpha;
            if (value2 < dualT) {
              value2 = oldValue - upperTheta * alpha;
              if (value2 < dualT && alpha >= acceptablePivot)
                upperTheta = (oldValue - dualT) / alpha;
              candValue[numberRemaining] = alpha * mult;
              candIndex[numberRemaining++] = iColumn;
            }
          }
        }
      }
    }
  }
  }
  for (int i = 0; i < numberPi; i++)
    piDense[piIndex[i]] = 0.0;
  *numberCandidatesOut = numberRemaining;
  *upperThetaOut = upperTheta;
  return numberNonZero;
}

int orc_price_row_fused(const OrcModel *M, int numberPi, const int *piIndex, const double *piValue,
                        const unsigned char *status, const double *dj, double zeroTolerance,
                        double dualTolerance, double acceptablePivot, int *outIndex, double *outValue,
                        int *numberCandidates, int *candIndex, double *candValue, double *upperTheta)
{
  double *dense = (double *)calloc((size_t)M->m + 1, sizeof(double));
  int r = priceRowFused(M, numberPi, piIndex, piValue, dense, status, dj, zeroTolerance, dualTolerance,
                        acceptablePivot, outIndex, outValue, numberCandidates, candIndex, candValue, upperTheta);
  free(dense);
  return r;
}

/* ClpPackedMatrix::add :4874 / ClpSimplex::add for slacks: array += multiplier * column(iSequence) */
static void addColumn(const OrcModel *M, double *array, int iSequence, double multiplier)
{
  if (iSequence < M->n) {
    for (int p = M->colStart[iSequence]; p < M->colStart[iSequence + 1]; p++)
      array[M->row[p]] += multiplier * M->elem[p];
  } else {
    array[iSequence - M->n] -= multiplier; /* slack column is -e_i */
  }
}

/* ------------------------------------------------------------------------------------------ */
/* factorization                                                                               */
/* ------------------------------------------------------------------------------------------ */
static void facReserveEta(Factor *F, int maxEta, long nnz)
{
  if (maxEta > F->maxEta || !F->etaPivot) {
    F->maxEta = maxEta;
    F->etaPivot = (int *)realloc(F->etaPivot, sizeof(int) * (size_t)(maxEta + 1));
    F->etaPivotValue = (double *)realloc(F->etaPivotValue, sizeof(double) * (size_t)(maxEta + 1));
    F->etaStart = (int *)realloc(F->etaStart, sizeof(int) * (size_t)(maxEta + 2));
    if (!F->nEta)
      F->etaStart[0] = 0;
  }
  if (nnz > F->etaCap) {
    F->etaCap = nnz + nnz / 2 + 1024;
    F->etaIndex = (int *)realloc(F->etaIndex, sizeof(int) * (size_t)F->etaCap);
    F->etaValue = (double *)realloc(F->etaValue, sizeof(double) * (size_t)F->etaCap);
  }
}

/* The elimination of CoinAbcDenseFactorization::factor :262-313 on the k x k nucleus (column-major, lu[r + c k]): first largest
 * pivot of column i among rows >= i, multipliers a_ji (1 / pivot), a_jc -= a_ic l_j.  luPlain is that loop as written (one pass over
 * the whole trailing matrix per pivot: 8 k^3 / 3 bytes through memory, twenty minutes at k = 10 500); luBlocked makes the same
 * floating-point operations on every entry in the same order -- each a_jc still receives its updates one at a time, by ascending
 * pivot -- but panel by panel, LAPACK style: a panel of LU_NB columns is eliminated on its own, its row interchanges are applied to
 * the other columns afterwards, and its LU_NB updates reach the columns to the right while a tile of the panel's multipliers sits in
 * cache.  Updates with a_ic == 0 are skipped (an exact no-op: active entries are never -0.0), which is most of them while the
 * nucleus is sparse.  Bit-identical results, tests/test_oracle_golden.py::test_blocked_lu_is_the_plain_loop; option
 * "debug_plain_lu" 1 runs luPlain. */
static int luPlain(double *lu, int k, int *perm, double zeroTolerance)
{
  double *elements = lu;
  for (int i = 0; i < k; i++) {
    int iRow = -1;
    double largest = zeroTolerance;
    for (int j = i; j < k; j++) {
      double value = fabs(elements[j]);
      if (value > largest) {
        largest = value;
        iRow = j;
      }
    }
    if (iRow < 0)
      return -1;
    if (iRow != i) {
      /* full row swap (the reference swaps columns <= i now and later columns lazily, :271-300) */
      for (int c = 0; c < k; c++) {
        double value = lu[i + (size_t)c * k];
        lu[i + (size_t)c * k] = lu[iRow + (size_t)c * k];
        lu[iRow + (size_t)c * k] = value;
      }
      int ip = perm[i];
      perm[i] = perm[iRow];
      perm[iRow] = ip;
    }
    double pivotValue = 1.0 / elements[i];
    elements[i] = pivotValue;
    for (int j = i + 1; j < k; j++)
      elements[j] *= pivotValue;
    double *elementsA = elements;
    for (int c = i + 1; c < k; c++) {
      elementsA += k;
      double value = elementsA[i];
      for (int j = i + 1; j < k; j++)
        elementsA[j] -= value * elements[j];
    }
    elements += k;
  }
  return 0;
}

#define LU_NB 64   /* panel width */
#define LU_RT 256  /* rows of the panel's multipliers kept in cache while they are applied (256 x 64 doubles = 128 KB) */
/* columns [c0, c1) to the right of the panel [p, p + nb): the panel's nb updates, ascending pivot order per entry */
__attribute__((target_clones("avx2", "default"))) static void luApplyPanel(double *lu, int k, int p, int nb, int c0, int c1)
{
  const int pe = p + nb;
  /* (the columns are independent of one another: with OpenMP they are dealt over the threads, which changes no operation on any entry)
     rows inside the panel: the column's entries of U come out one after the other */
#pragma omp parallel for schedule(static) if (c1 - c0 >= 512)
  for (int c = c0; c < c1; c++) {
    double *a = lu + (size_t)c * k;
    for (int i = p; i < pe; i++) {
      const double value = a[i];
      if (value) {
        const double *l = lu + (size_t)i * k;
        for (int j = i + 1; j < pe; j++)
          a[j] -= value * l[j];
      }
    }
  }
  /* rows below the panel, a tile of rows at a time, four columns per pass over the tile */
  const int groups = (c1 - c0) / 4;
#pragma omp parallel for schedule(dynamic, 16) if (c1 - c0 >= 512)
  for (int gq = 0; gq < groups; gq++)
  for (int r0 = pe; r0 < k; r0 += LU_RT) {
    const int r1 = r0 + LU_RT < k ? r0 + LU_RT : k;
    {
      const int c = c0 + 4 * gq;
      double *a0 = lu + (size_t)c * k, *a1 = a0 + k, *a2 = a1 + k, *a3 = a2 + k;
      for (int i = p; i < pe; i++) {
        const double u0 = a0[i], u1 = a1[i], u2 = a2[i], u3 = a3[i];
        if (u0 == 0.0 && u1 == 0.0 && u2 == 0.0 && u3 == 0.0)
          continue;
        const double *l = lu + (size_t)i * k;
        if (u0 != 0.0 && u1 != 0.0 && u2 != 0.0 && u3 != 0.0) {
          for (int j = r0; j < r1; j++) {
            const double lj = l[j];
            a0[j] -= u0 * lj;
            a1[j] -= u1 * lj;
            a2[j] -= u2 * lj;
            a3[j] -= u3 * lj;
          }
        } else {
          if (u0 != 0.0)
            for (int j = r0; j < r1; j++)
              a0[j] -= u0 * l[j];
          if (u1 != 0.0)
            for (int j = r0; j < r1; j++)
              a1[j] -= u1 * l[j];
          if (u2 != 0.0)
            for (int j = r0; j < r1; j++)
              a2[j] -= u2 * l[j];
          if (u3 != 0.0)
            for (int j = r0; j < r1; j++)
              a3[j] -= u3 * l[j];
        }
      }
    }
  }
  for (int r0 = pe; r0 < k; r0 += LU_RT) {
    const int r1 = r0 + LU_RT < k ? r0 + LU_RT : k;
    for (int c = c0 + 4 * groups; c < c1; c++) {
      double *a = lu + (size_t)c * k;
      for (int i = p; i < pe; i++) {
        const double value = a[i];
        if (value != 0.0) {
          const double *l = lu + (size_t)i * k;
          for (int j = r0; j < r1; j++)
            a[j] -= value * l[j];
        }
      }
    }
  }
}

static int luBlocked(double *lu, int k, int *perm, double zeroTolerance)
{
  int swapWith[LU_NB];
  for (int p = 0; p < k; p += LU_NB) {
    const int nb = p + LU_NB < k ? LU_NB : k - p, pe = p + nb;
    for (int i = p; i < pe; i++) {
      double *elements = lu + (size_t)i * k;
      int iRow = -1;
      double largest = zeroTolerance;
      for (int j = i; j < k; j++) {
        double value = fabs(elements[j]);
        if (value > largest) {
          largest = value;
          iRow = j;
        }
      }
      if (iRow < 0)
        return -1;
      swapWith[i - p] = iRow;
      if (iRow != i) {
        for (int c = p; c < pe; c++) { /* the panel's own columns now, the others after the panel */
          double value = lu[i + (size_t)c * k];
          lu[i + (size_t)c * k] = lu[iRow + (size_t)c * k];
          lu[iRow + (size_t)c * k] = value;
        }
        int ip = perm[i];
        perm[i] = perm[iRow];
        perm[iRow] = ip;
      }
      double pivotValue = 1.0 / elements[i];
      elements[i] = pivotValue;
      for (int j = i + 1; j < k; j++)
        elements[j] *= pivotValue;
      for (int c = i + 1; c < pe; c++) {
        double *elementsA = lu + (size_t)c * k;
        double value = elementsA[i];
        if (value)
          for (int j = i + 1; j < k; j++)
            elementsA[j] -= value * elements[j];
      }
    }
    for (int c = 0; c < k; c++) {
      if (c == p) {
        c = pe - 1;
        continue;
      }
      double *a = lu + (size_t)c * k;
      for (int i = p; i < pe; i++) {
        const int iRow = swapWith[i - p];
        if (iRow != i) {
          double value = a[i];
          a[i] = a[iRow];
          a[iRow] = value;
        }
      }
    }
    if (pe < k)
      luApplyPanel(lu, k, p, nb, pe, k);
  }
  return 0;
}

/* What ClpDualRowSteepest::pivotRow reads as model_->factorization()->numberElements() (src/ClpDualRowSteepest.cpp:262).  Behind
 * ClpFactorization that is CoinFactorization::numberElements() = totalElements_ (CoinUtils, not in this tree; the in-tree twin keeps
 * the same counter: entries of U less the slack pivots, src/CoinAbcBaseFactorization1.cpp:3800 / :3860, plus the dense block and L,
 * :3282 / :3323) -- a property of that LU code's ordering and fill, which no other factorization reproduces.  Model 0 (default, and
 * the one the HIP engine can compute bit for bit): the entries of the basic structural columns, i.e. that counter for an LU without
 * fill -- exact for the slack basis (0) and for every basis that triangularizes, a lower bound otherwise.  Model 1: the nonzeros this
 * oracle's own dense LU of the nucleus holds (what a plug-in factorization would answer, CoinOtherFactorization::numberElements). */
static void factorElementsModel(OrcModel *M)
{
  const Factor *F = &M->fac;
  long count = 0;
  if (M->steepestElements == 0) {
    for (int c = 0; c < F->k; c++)
      count += M->colStart[F->kcol[c] + 1] - M->colStart[F->kcol[c]];
  } else {
    const size_t kk = (size_t)F->k * (size_t)F->k;
    for (size_t q = 0; q < kk; q++)
      count += F->lu[q] != 0.0;
  }
  M->factorElements = count;
}

/* ClpFactorization::factorize :1649 (collect basic rows then columns, slack value -1) +
 * CoinAbcDenseFactorization::factor :216-331 on the nucleus.  Returns 0, or -1 if singular. */
static int factorize(OrcModel *M)
{
  Factor *F = &M->fac;
  const int m = M->m, n = M->n;
  int k = 0, numberBasic = 0;
  for (int i = 0; i < m; i++) {
    if (getStatus(M, n + i) == ST_BASIC) {
      F->rowToK[i] = -1;
      numberBasic++;
    } else {
      F->rowToK[i] = -2; /* nucleus row, step not yet known */
    }
  }
  for (int j = 0; j < n; j++)
    if (getStatus(M, j) == ST_BASIC) {
      if (k < m)
        F->kcol[k] = j;
      k++;
      numberBasic++;
    }
  if (numberBasic != m)
    return -2;
  F->k = k;
  F->nEta = 0;
  facReserveEta(F, M->maximumPivots, 0);
  F->etaStart[0] = 0;
  M->numberRefactorizations++;
  /* list of nucleus rows, ascending */
  int *rrows = (int *)malloc(sizeof(int) * (size_t)(k + 1));
  int nr = 0;
  for (int i = 0; i < m; i++)
    if (F->rowToK[i] == -2)
      rrows[nr++] = i;
  if ((long)k * k > F->luCap) {
    F->luCap = (long)k * k + 16;
    free(F->lu);
    F->lu = (double *)malloc(sizeof(double) * (size_t)F->luCap);
  }
  double *lu = F->lu;
  if (k)
    memset(lu, 0, sizeof(double) * (size_t)k * (size_t)k);
  int *where = F->krow; /* reuse as temp: row -> local index */
  int *local = (int *)malloc(sizeof(int) * (size_t)(m + 1));
  for (int i = 0; i < m; i++)
    local[i] = -1;
  for (int r = 0; r < k; r++)
    local[rrows[r]] = r;
  for (int c = 0; c < k; c++) {
    int j = F->kcol[c];
    for (int p = M->colStart[j]; p < M->colStart[j + 1]; p++) {
      int r = local[M->row[p]];
      if (r >= 0)
        lu[r + (size_t)c * k] = M->elem[p];
    }
  }
  (void)where;
  int *perm = (int *)malloc(sizeof(int) * (size_t)(k + 1));
  for (int i = 0; i < k; i++)
    perm[i] = i;
  int status = M->debugPlainLu ? luPlain(lu, k, perm, M->zeroTolerance) : luBlocked(lu, k, perm, M->zeroTolerance);
  if (!status) {
    for (int c = 0; c < k; c++) {
      F->krow[c] = rrows[perm[c]];
      F->rowToK[F->krow[c]] = c;
    }
    /* postProcess: pivotVariable (basis position == row) */
    for (int i = 0; i < m; i++)
      if (F->rowToK[i] == -1)
        M->pivotVariable[i] = n + i;
    for (int c = 0; c < k; c++)
      M->pivotVariable[F->krow[c]] = F->kcol[c];
    factorElementsModel(M);
  }
  free(perm);
  free(local);
  free(rrows);
  return status;
}

/* CoinAbcDenseFactorization::updateColumn :571-610: L, U, then the eta file.  `x` is indexed by
 * row on input and by basis position (== row) on output. */
static void ftran(const OrcModel *M, double *x)
{
  const Factor *F = &M->fac;
  const int k = F->k, m = M->m;
  double *t = F->t;
  const double *lu = F->lu;
  for (int c = 0; c < k; c++)
    t[c] = x[F->krow[c]];
  for (int i = 0; i < k; i++) {
    double value = t[i];
    if (value) {
      const double *col = lu + (size_t)i * k;
      for (int j = i + 1; j < k; j++)
        t[j] -= value * col[j];
    }
  }
  for (int i = k - 1; i >= 0; i--) {
    const double *col = lu + (size_t)i * k;
    double value = t[i] * col[i];
    t[i] = value;
    if (value) {
      for (int j = 0; j < i; j++)
        t[j] -= value * col[j];
      int jcol = F->kcol[i];
      for (int p = M->colStart[jcol]; p < M->colStart[jcol + 1]; p++) {
        int r = M->row[p];
        if (F->rowToK[r] < 0)
          x[r] -= value * M->elem[p];
      }
    }
  }
  for (int r = 0; r < m; r++)
    if (F->rowToK[r] < 0)
      x[r] = x[r] * -1.0; /* inverse pivot of the slack column -e_r */
  for (int c = 0; c < k; c++)
    x[F->krow[c]] = t[c];
  for (int e = 0; e < F->nEta; e++) {
    int iPivot = F->etaPivot[e];
    double value = x[iPivot] * F->etaPivotValue[e];
    if (value) {
      for (int p = F->etaStart[e]; p < F->etaStart[e + 1]; p++)
        x[F->etaIndex[p]] -= value * F->etaValue[p];
    }
    x[iPivot] = value;
  }
}

/* CoinAbcDenseFactorization::updateColumnTranspose :634-683: etas reversed, U^T, L^T. */
static void btran(const OrcModel *M, double *y)
{
  const Factor *F = &M->fac;
  const int k = F->k, m = M->m;
  double *t = F->t;
  const double *lu = F->lu;
  for (int e = F->nEta - 1; e >= 0; e--) {
    int iPivot = F->etaPivot[e];
    double value = y[iPivot];
    for (int p = F->etaStart[e]; p < F->etaStart[e + 1]; p++)
      value -= y[F->etaIndex[p]] * F->etaValue[p];
    y[iPivot] = value * F->etaPivotValue[e];
  }
  for (int r = 0; r < m; r++)
    if (F->rowToK[r] < 0)
      y[r] = y[r] * -1.0;
  for (int c = 0; c < k; c++) {
    double value = y[F->krow[c]];
    int jcol = F->kcol[c];
    for (int p = M->colStart[jcol]; p < M->colStart[jcol + 1]; p++) {
      int r = M->row[p];
      if (F->rowToK[r] < 0)
        value -= y[r] * M->elem[p];
    }
    const double *col = lu + (size_t)c * k;
    for (int j = 0; j < c; j++)
      value -= t[j] * col[j];
    t[c] = value * col[c];
  }
  for (int i = k - 1; i >= 0; i--) {
    const double *col = lu + (size_t)i * k;
    double value = t[i];
    for (int j = i + 1; j < k; j++)
      value -= t[j] * col[j];
    t[i] = value;
  }
  for (int c = 0; c < k; c++)
    y[F->krow[c]] = t[c];
}

/* CoinAbcDenseFactorization::checkReplacePart2 :470-478 + replaceColumnPart3 :480-544 */
static int replaceColumn(OrcModel *M, const int *wIndex, const double *wValue, int numberW, int pivotRow, double alpha)
{
  Factor *F = &M->fac;
  if (F->nEta >= M->maximumPivots)
    return 3;
  if (fabs(alpha) < M->zeroTolerance)
    return 2;
  int e = F->nEta;
  long need = F->etaStart[e] + numberW + 1;
  facReserveEta(F, M->maximumPivots, need);
  int put = F->etaStart[e];
  for (int i = 0; i < numberW; i++) {
    if (wIndex[i] != pivotRow) {
      F->etaIndex[put] = wIndex[i];
      F->etaValue[put++] = wValue[i];
    }
  }
  F->etaStart[e + 1] = put;
  F->etaPivot[e] = pivotRow;
  F->etaPivotValue[e] = 1.0 / alpha;
  F->nEta++;
  return 0;
}

int orc_factorize(OrcModel *M, const unsigned char *status, int *pivotVariable)
{
  memcpy(M->status, status, (size_t)(M->m + M->n));
  int rc = factorize(M);
  if (!rc && pivotVariable)
    memcpy(pivotVariable, M->pivotVariable, sizeof(int) * (size_t)M->m);
  return rc;
}
void orc_ftran(OrcModel *M, double *region) { ftran(M, region); }
void orc_btran(OrcModel *M, double *region) { btran(M, region); }
int orc_replace_column(OrcModel *M, const double *w, int pivotRow, double alpha)
{
  int nw = 0;
  for (int i = 0; i < M->m; i++)
    if (w[i]) {
      M->wIndex[nw] = i;
      M->wValue[nw++] = w[i];
    }
  int rc = replaceColumn(M, M->wIndex, M->wValue, nw, pivotRow, alpha);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* primal / dual solutions  (ClpSimplex::gutsOfSolution :574)                                  */
/* ------------------------------------------------------------------------------------------ */

/* ClpSimplex::computePrimals :914-1162 (numberRefinements_ = 0) */
static void computePrimals(OrcModel *M)
{
  const int m = M->m, n = M->n;
  double *array = M->rowWork1;
  for (int i = 0; i < m; i++)
    M->sol[M->pivotVariable[i]] = 0.0;
  memset(array, 0, sizeof(double) * (size_t)m);
  orc_times(M, -1.0, M->sol, array);
  for (int i = 0; i < m; i++)
    array[i] = array[i] + M->sol[n + i];
  double *rhs = M->rowWork2;
  memcpy(rhs, array, sizeof(double) * (size_t)m);
  ftran(M, array);
  /* error check: B*x_B - rhs */
  double *work = M->rowWork3;
  memset(work, 0, sizeof(double) * (size_t)m);
  for (int i = 0; i < m; i++) {
    double value = array[i];
    if (value)
      addColumn(M, work, M->pivotVariable[i], value);
  }
  double largest = 0.0;
  for (int i = 0; i < m; i++) {
    double d = fabs(work[i] - rhs[i]);
    if (d > largest)
      largest = d;
  }
  M->largestPrimalError = largest;
  for (int i = 0; i < m; i++)
    M->sol[M->pivotVariable[i]] = array[i];
  memset(array, 0, sizeof(double) * (size_t)m);
  memset(rhs, 0, sizeof(double) * (size_t)m);
  memset(work, 0, sizeof(double) * (size_t)m);
}

/* ClpSimplex::computeDuals :1164-1400 */
static void computeDuals(OrcModel *M)
{
  const int m = M->m, n = M->n;
  double *array = M->rowWork1;
  for (int i = 0; i < m; i++)
    array[i] = M->cost[M->pivotVariable[i]];
  btran(M, array);
  double largest = 0.0;
  for (int i = 0; i < m; i++) {
    int iPivot = M->pivotVariable[i];
    double value;
    if (iPivot >= n) {
      value = M->cost[iPivot] + array[iPivot - n];
    } else {
      double v = 0.0;
      for (int p = M->colStart[iPivot]; p < M->colStart[iPivot + 1]; p++)
        v += array[M->row[p]] * M->elem[p];
      value = M->cost[iPivot] - v;
    }
    if (fabs(value) > largest)
      largest = fabs(value);
  }
  M->largestDualError = largest;
  for (int i = 0; i < m; i++)
    M->dj[n + i] = array[i] + M->cost[n + i];
  for (int j = 0; j < n; j++)
    M->dj[j] = M->cost[j];
  orc_transpose_times(M, -1.0, array, M->dj);
  memset(array, 0, sizeof(double) * (size_t)m);
}

/* ClpSimplex::checkPrimalSolution :2989-3069 */
static void checkPrimalSolution(OrcModel *M)
{
  const int N = M->m + M->n, n = M->n;
  double primalTolerance = M->primalTolerance;
  double relaxedTolerance = primalTolerance + dmin(1.0e-2, M->largestPrimalError);
  M->objectiveValue = 0.0;
  M->sumPrimalInfeasibilities = 0.0;
  M->numberPrimalInfeasibilities = 0;
  M->sumOfRelaxedPrimalInfeasibilities = 0.0;
  for (int pass = 0; pass < 2; pass++) {
    int lo = pass ? 0 : n, hi = pass ? n : N; /* rows first, then columns */
    for (int i = lo; i < hi; i++) {
      double infeasibility = 0.0;
      M->objectiveValue += M->sol[i] * M->cost[i];
      if (M->sol[i] > M->upper[i])
        infeasibility = M->sol[i] - M->upper[i];
      else if (M->sol[i] < M->lower[i])
        infeasibility = M->lower[i] - M->sol[i];
      if (infeasibility > primalTolerance) {
        M->sumPrimalInfeasibilities += infeasibility - primalTolerance;
        if (infeasibility > relaxedTolerance)
          M->sumOfRelaxedPrimalInfeasibilities += infeasibility - relaxedTolerance;
        M->numberPrimalInfeasibilities++;
      }
    }
  }
}

/* ClpSimplex::checkDualSolution :3070-3225.  With option free_nonbasic: the isFree branches ("free so relax a lot", :3125-3136), the count
 * without free variables and firstFree_ (:3214-3219). */
static void checkDualSolution(OrcModel *M)
{
  const int N = M->m + M->n, n = M->n;
  double relaxedTolerance = M->dualTolerance + dmin(1.0e-2, M->largestDualError);
  const double possTolerance = 5.0 * relaxedTolerance; /* a bigger tolerance for the possible improvement (:3093) */
  const int withFree = M->freeNonbasic;
  int firstFreePrimal = -1, firstFreeDual = -1, numberSuperBasicWithDj = 0;
  M->bestPossibleImprovement = 0.0;
  M->sumDualInfeasibilities = 0.0;
  M->numberDualInfeasibilities = 0;
  M->sumOfRelaxedDualInfeasibilities = 0.0;
  M->numberDualInfeasibilitiesWithoutFree = 0;
  for (int pass = 0; pass < 2; pass++) {
    int lo = pass ? n : 0, hi = pass ? N : n; /* columns first, then rows */
    for (int i = lo; i < hi; i++) {
      if (getStatus(M, i) != ST_BASIC && !flagged(M, i)) {
        double distanceUp = M->upper[i] - M->sol[i];
        double distanceDown = M->sol[i] - M->lower[i];
        double value = M->dj[i];
        const int isFree = withFree && getStatus(M, i) == ST_FREE;
        if (distanceUp > M->primalTolerance) {
          if (withFree && distanceDown > M->primalTolerance) { /* check if "free" (:3110-3118) */
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
            if (v > M->dualTolerance) {
              if (!(isFree && i < n)) { /* (the row loop has no relaxed form, :3180-3190) */
                if (!isFree)
                  M->numberDualInfeasibilitiesWithoutFree++;
                M->sumDualInfeasibilities += v - M->dualTolerance;
                if (v > possTolerance)
                  M->bestPossibleImprovement += dmin(distanceUp, 1.0e10) * v;
                if (v > relaxedTolerance)
                  M->sumOfRelaxedDualInfeasibilities += v - relaxedTolerance;
                M->numberDualInfeasibilities++;
              } else {
                v *= 0.01; /* free so relax a lot */
                if (v > M->dualTolerance) {
                  M->sumDualInfeasibilities += v - M->dualTolerance;
                  if (v > possTolerance)
                    M->bestPossibleImprovement = 1.0e100;
                  if (v > relaxedTolerance)
                    M->sumOfRelaxedDualInfeasibilities += v - relaxedTolerance;
                  M->numberDualInfeasibilities++;
                }
              }
            }
          }
        }
        if (distanceDown > M->primalTolerance) {
          if (value > 0.0) {
            if (value > M->dualTolerance) {
              M->sumDualInfeasibilities += value - M->dualTolerance;
              if (value > possTolerance)
                M->bestPossibleImprovement += value * dmin(distanceDown, 1.0e10);
              if (value > relaxedTolerance)
                M->sumOfRelaxedDualInfeasibilities += value - relaxedTolerance;
              M->numberDualInfeasibilities++;
              if (!isFree)
                M->numberDualInfeasibilitiesWithoutFree++;
            }
          }
        }
      }
    }
  }
  if (withFree) {
    if (firstFreeDual >= 0)
      M->firstFree = firstFreeDual;
    else if (numberSuperBasicWithDj || progressLastIteration(M, 0) <= 0)
      M->firstFree = firstFreePrimal;
  }
}

/* ClpSimplex::checkBothSolutions :3226-3440.  What gutsOfSolution ends in in this reference version (:762); the default since round 4, on
 * both sides (option "check_both" 0 restores the checkPrimalSolution + checkDualSolution pair, which statusOfProblemInDual still calls directly
 * where the reference does).  With option free_nonbasic the bookkeeping for free and superbasic variables as well: moreSpecialOptions_ & 8
 * ("no free or super basic", which picks the branch of dualColumn0), the count without free variables and firstFree_. */
static void checkBothSolutions(OrcModel *M)
{
  const int N = M->m + M->n;
  const double primalTolerance = M->primalTolerance, dualTolerance = M->dualTolerance;
  M->objectiveValue = 0.0;
  M->sumPrimalInfeasibilities = 0.0;
  M->numberPrimalInfeasibilities = 0;
  /* we can't really trust infeasibilities if there is primal / dual error */
  const double relaxedToleranceP = primalTolerance + dmin(1.0e-2, dmax(M->largestPrimalError, 0.0 * primalTolerance));
  const double relaxedToleranceD = dualTolerance + dmin(1.0e-2, dmax(M->largestDualError, 5.0 * dualTolerance));
  const double possTolerance = 5.0 * relaxedToleranceD; /* allow bigger tolerance for possible improvement */
  M->sumOfRelaxedPrimalInfeasibilities = 0.0;
  M->sumDualInfeasibilities = 0.0;
  M->numberDualInfeasibilities = 0;
  M->sumOfRelaxedDualInfeasibilities = 0.0;
  M->bestPossibleImprovement = 0.0;
  int numberDualInfeasibilitiesFree = 0, firstFreePrimal = -1, firstFreeDual = -1, numberSuperBasicWithDj = 0;
  int noFreeOrSuper = 1; /* say no free or superbasic (:3273) */
  for (int i = 0; i < N; i++) {
    const double value = M->sol[i];
    M->objectiveValue += value * M->cost[i];
    const double distanceUp = M->upper[i] - value, distanceDown = value - M->lower[i];
    if (distanceUp < -primalTolerance) {
      const double infeasibility = -distanceUp;
      if (getStatus(M, i) != ST_BASIC)
        noFreeOrSuper = 0; /* say superbasic variables exist (:3294) */
      M->sumPrimalInfeasibilities += infeasibility - primalTolerance;
      if (infeasibility > relaxedToleranceP)
        M->sumOfRelaxedPrimalInfeasibilities += infeasibility - relaxedToleranceP;
      M->numberPrimalInfeasibilities++;
    } else if (distanceDown < -primalTolerance) {
      const double infeasibility = -distanceDown;
      if (getStatus(M, i) != ST_BASIC)
        noFreeOrSuper = 0;
      M->sumPrimalInfeasibilities += infeasibility - primalTolerance;
      if (infeasibility > relaxedToleranceP)
        M->sumOfRelaxedPrimalInfeasibilities += infeasibility - relaxedToleranceP;
      M->numberPrimalInfeasibilities++;
    } else if (getStatus(M, i) != ST_BASIC && !flagged(M, i)) {
      /* feasible (so could be free) and not basic */
      double djValue = M->dj[i];
      if (distanceDown < primalTolerance) {
        if (distanceUp > primalTolerance && djValue < -dualTolerance) {
          M->sumDualInfeasibilities -= djValue + dualTolerance;
          if (djValue < -possTolerance)
            M->bestPossibleImprovement -= distanceUp * djValue;
          if (djValue < -relaxedToleranceD)
            M->sumOfRelaxedDualInfeasibilities -= djValue + relaxedToleranceD;
          M->numberDualInfeasibilities++;
        }
      } else if (distanceUp < primalTolerance) {
        if (djValue > dualTolerance) {
          M->sumDualInfeasibilities += djValue - dualTolerance;
          if (djValue > possTolerance)
            M->bestPossibleImprovement += distanceDown * djValue;
          if (djValue > relaxedToleranceD)
            M->sumOfRelaxedDualInfeasibilities += djValue - relaxedToleranceD;
          M->numberDualInfeasibilities++;
        }
      } else {
        /* strictly between its bounds: may be free -- say free or superbasic (:3345) */
        noFreeOrSuper = 0;
        djValue *= 100.0;
        if (fabs(djValue) > dualTolerance) {
          if (getStatus(M, i) == ST_FREE)
            numberDualInfeasibilitiesFree++;
          M->sumDualInfeasibilities += fabs(djValue) - dualTolerance;
          M->bestPossibleImprovement = 1.0e100;
          M->numberDualInfeasibilities++;
          if (fabs(djValue) > relaxedToleranceD) {
            M->sumOfRelaxedDualInfeasibilities += value - relaxedToleranceD; /* sic: `value`, :3358 */
            numberSuperBasicWithDj++;
            if (firstFreeDual < 0)
              firstFreeDual = i;
            if (firstFreePrimal < 0)
              firstFreePrimal = i;
          }
        } else if (getStatus(M, i) == ST_SUPER && firstFreePrimal < 0) {
          firstFreePrimal = i;
        }
      }
    }
  }
  M->numberDualInfeasibilitiesWithoutFree = M->numberDualInfeasibilities;
  if (M->freeNonbasic) {
    M->noFreeOrSuper = noFreeOrSuper;
    M->numberDualInfeasibilitiesWithoutFree = M->numberDualInfeasibilities - numberDualInfeasibilitiesFree;
    if (firstFreeDual >= 0)
      M->firstFree = firstFreeDual; /* dual (:3418) */
    else if (numberSuperBasicWithDj || progressLastIteration(M, 0) <= 0)
      M->firstFree = firstFreePrimal;
  }
}

static void gutsOfSolution(OrcModel *M)
{
  computePrimals(M);
  computeDuals(M);
  if (M->checkBoth) {
    checkBothSolutions(M);
  } else {
    if (M->freeNonbasic)
      M->noFreeOrSuper = 0; /* "say may be free or superbasic", the old way of src/ClpSimplex.cpp:3228-3234 */
    checkPrimalSolution(M);
    checkDualSolution(M);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* fake bounds (ClpSimplexDual::changeBounds :3148, originalBound :6403, changeBound :6445)    */
/* ------------------------------------------------------------------------------------------ */
static void originalBound(OrcModel *M, int i)
{
  if (getFake(M, i) != FAKE_NONE) {
    M->numberFake--;
    setFake(M, i, FAKE_NONE);
    M->lower[i] = originalLower(M, i);
    M->upper[i] = originalUpper(M, i);
  }
}

static int changeBound(OrcModel *M, int i)
{
  double oldLower = M->lower[i], oldUpper = M->upper[i], value = M->sol[i];
  int modified = 0;
  originalBound(M, i);
  double lowerValue = M->lower[i], upperValue = M->upper[i];
  M->lower[i] = oldLower;
  M->upper[i] = oldUpper;
  if (value == oldLower) {
    if (upperValue > oldLower + M->dualBound) {
      M->upper[i] = oldLower + M->dualBound;
      setFake(M, i, FAKE_UPPER);
      modified = 1;
      M->numberFake++;
    }
  } else if (value == oldUpper) {
    if (lowerValue < oldUpper - M->dualBound) {
      M->lower[i] = oldUpper - M->dualBound;
      setFake(M, i, FAKE_LOWER);
      modified = 1;
      M->numberFake++;
    }
  }
  return modified;
}

/* initialize == 1 or 3 (:3267-3440), 0 (:3153-3266) */
static int changeBounds(OrcModel *M, int initialize, double *outputArray, double *changeCost)
{
  const int N = M->m + M->n, n = M->n;
  M->numberFake = 0;
  if (!initialize) {
    int numberInfeasibilities = 0;
    double newBound = 5.0 * M->dualBound;
    *changeCost = 0.0;
    /* createRim1(false): put back original bounds */
    for (int i = 0; i < N; i++) {
      M->lower[i] = originalLower(M, i);
      M->upper[i] = originalUpper(M, i);
    }
    for (int i = 0; i < N; i++) {
      double lowerValue = M->lower[i], upperValue = M->upper[i], value = M->sol[i];
      setFake(M, i, FAKE_NONE);
      int st = getStatus(M, i);
      if (st == ST_UPPER) {
        if (fabs(value - upperValue) > M->primalTolerance) {
          if (fabs(M->dj[i]) > 1.0e-9)
            numberInfeasibilities++;
          else {
            setStatus(M, i, ST_SUPER);
            if (M->freeNonbasic)
              M->noFreeOrSuper = 0; /* moreSpecialOptions_ &= ~8 (:3181, :3192) */
          }
        }
      } else if (st == ST_LOWER) {
        if (fabs(value - lowerValue) > M->primalTolerance) {
          if (fabs(M->dj[i]) > 1.0e-9)
            numberInfeasibilities++;
          else {
            setStatus(M, i, ST_SUPER);
            if (M->freeNonbasic)
              M->noFreeOrSuper = 0; /* moreSpecialOptions_ &= ~8 (:3181, :3192) */
          }
        }
      }
    }
    if (numberInfeasibilities) {
      for (int i = 0; i < N; i++) {
        double lowerValue = M->lower[i], upperValue = M->upper[i];
        double newLowerValue, newUpperValue;
        int st = getStatus(M, i);
        if (st == ST_UPPER || st == ST_LOWER) {
          double value = M->sol[i];
          if (value - lowerValue <= upperValue - value) {
            newLowerValue = dmax(lowerValue, value - 0.666667 * newBound);
            newUpperValue = dmin(upperValue, newLowerValue + newBound);
          } else {
            newUpperValue = dmin(upperValue, value + 0.666667 * newBound);
            newLowerValue = dmax(lowerValue, newUpperValue - newBound);
          }
          if (newLowerValue > lowerValue) {
            if (newUpperValue < upperValue) {
              setFake(M, i, FAKE_BOTH);
              if (st == ST_LOWER) {
                newLowerValue = value;
                newUpperValue = dmin(upperValue, newLowerValue + newBound);
              } else {
                newUpperValue = value;
                newLowerValue = dmax(lowerValue, newUpperValue - newBound);
              }
              M->numberFake++;
            } else {
              setFake(M, i, FAKE_LOWER);
              M->numberFake++;
            }
          } else if (newUpperValue < upperValue) {
            setFake(M, i, FAKE_UPPER);
            M->numberFake++;
          }
          M->lower[i] = newLowerValue;
          M->upper[i] = newUpperValue;
          M->sol[i] = (st == ST_UPPER) ? newUpperValue : newLowerValue;
          double movement = M->sol[i] - value;
          if (movement && outputArray) {
            if (i >= n)
              outputArray[i - n] -= movement;
            else
              addColumn(M, outputArray, i, movement);
            *changeCost += movement * M->cost[i];
          }
        }
      }
      M->dualBound = newBound;
    } else {
      numberInfeasibilities = -1;
    }
    return numberInfeasibilities;
  } else {
    if (initialize == 3) {
      for (int i = 0; i < N; i++) {
        if (getFake(M, i) != FAKE_NONE) {
          M->lower[i] = originalLower(M, i);
          M->upper[i] = originalUpper(M, i);
          setFake(M, i, FAKE_NONE);
        }
      }
    }
    double testBound = 0.999999 * M->dualBound;
    for (int i = 0; i < N; i++) {
      int st = getStatus(M, i);
      if (st == ST_UPPER || st == ST_LOWER) {
        double lowerValue = M->lower[i], upperValue = M->upper[i], value = M->sol[i];
        if (lowerValue > -M->largeValue || upperValue < M->largeValue) {
          if (fabs(lowerValue - value) <= fabs(upperValue - value)) {
            if (upperValue > lowerValue + testBound) {
              if (getFake(M, i) == FAKE_NONE)
                M->numberFake++;
              M->upper[i] = lowerValue + M->dualBound;
              setFake(M, i, FAKE_UPPER);
            }
          } else {
            if (lowerValue < upperValue - testBound) {
              if (getFake(M, i) == FAKE_NONE)
                M->numberFake++;
              M->lower[i] = upperValue - M->dualBound;
              setFake(M, i, FAKE_LOWER);
            }
          }
          M->sol[i] = (st == ST_UPPER) ? M->upper[i] : M->lower[i];
        } else {
          M->lower[i] = -0.5 * M->dualBound;
          M->upper[i] = 0.5 * M->dualBound;
          setFake(M, i, FAKE_BOTH);
          M->numberFake++;
          setStatus(M, i, ST_UPPER);
          M->sol[i] = 0.5 * M->dualBound;
        }
      } else if (st == ST_BASIC) {
        setFake(M, i, FAKE_NONE);
        double gap = M->upper[i] - M->lower[i];
        if (gap > 0.5 * M->dualBound && gap < 2.0 * M->dualBound) {
          M->lower[i] = originalLower(M, i);
          M->upper[i] = originalUpper(M, i);
        }
      }
    }
    return 1;
  }
}

static int numberAtFakeBound(const OrcModel *M)
{
  int count = 0;
  for (int i = 0; i < M->m + M->n; i++) {
    int f = getFake(M, i), st = getStatus(M, i);
    if (st == ST_UPPER && (f == FAKE_UPPER || f == FAKE_BOTH))
      count++;
    else if (st == ST_LOWER && (f == FAKE_LOWER || f == FAKE_BOTH))
      count++;
  }
  return count;
}

/* ------------------------------------------------------------------------------------------ */
/* dual row pivot                                                                              */
/* ------------------------------------------------------------------------------------------ */
static void infeasibleAdd(OrcModel *M, int iRow, double value)
{
  if (M->infeas[iRow]) {
    M->infeas[iRow] = value;
  } else {
    M->infeas[iRow] = value;
    M->infIndex[M->numberInfeasible++] = iRow;
  }
}

/* ClpDualRowSteepest::saveWeights :773-1014, modes 1,2,3,4 */
static void saveWeights(OrcModel *M, int mode)
{
  const int m = M->m;
  if (M->pivotRule == 0) {
    return; /* Dantzig keeps nothing */
  }
  if (mode == 1) {
    if (M->haveSavedWeights) {
      /* change from row numbers to sequence numbers (:786-793) */
      for (int i = 0; i < m; i++)
        M->altWeightIndex[i] = M->pivotVariable[i];
      M->numberAlt = 0;
    }
    return;
  }
  if (mode == 2 || mode == 4) {
    if (!M->haveSavedWeights) {
      /* initialize weights to 1.0 (:820-870, mode_ != 1) */
      for (int i = 0; i < m; i++)
        M->weights[i] = 1.0;
      for (int i = 0; i < m; i++) {
        M->savedWeights[i] = M->weights[i];
        M->savedWhich[i] = M->pivotVariable[i];
      }
      M->haveSavedWeights = 1;
    } else {
      const int *which;
      int *back = (int *)malloc(sizeof(int) * (size_t)(m + M->n + 1));
      for (int i = 0; i < m + M->n; i++)
        back[i] = -1;
      if (mode != 4) {
        memcpy(M->savedWhich, M->altWeightIndex, sizeof(int) * (size_t)m);
        memcpy(M->savedWeights, M->weights, sizeof(double) * (size_t)m);
        which = M->altWeightIndex;
      } else {
        which = M->savedWhich;
      }
      for (int i = 0; i < m; i++)
        back[which[i]] = i;
      for (int i = 0; i < m; i++) {
        int iPivot = back[M->pivotVariable[i]];
        if (iPivot >= 0) {
          M->weights[i] = M->savedWeights[iPivot];
          if (M->weights[i] < DEVEX_TRY_NORM)
            M->weights[i] = DEVEX_TRY_NORM;
        } else {
          M->weights[i] = 1.0;
        }
      }
      free(back);
    }
  }
  if (mode == 6) {
    /* scale back weights as primal errors (:930-950; the reference stores `allowed` in every weight) */
    double primalError = M->largestPrimalError;
    double allowed = primalError > 1.0e3 ? 10.0 : (primalError > 1.0e2 ? 50.0 : (primalError > 1.0e1 ? 100.0 : 1000.0));
    for (int i = 0; i < m; i++)
      M->weights[i] = allowed;
  }
  if (mode >= 2) {
    for (int i = 0; i < M->numberInfeasible; i++)
      M->infeas[M->infIndex[i]] = 0.0;
    M->numberInfeasible = 0;
    double tolerance = M->primalTolerance;
    for (int iRow = 0; iRow < m; iRow++) {
      int iPivot = M->pivotVariable[iRow];
      double value = M->sol[iPivot], lower = M->lower[iPivot], upper = M->upper[iPivot];
      if (value < lower - tolerance) {
        value -= lower;
        value *= value;
        infeasibleAdd(M, iRow, value);
      } else if (value > upper + tolerance) {
        value -= upper;
        value *= value;
        infeasibleAdd(M, iRow, value);
      }
    }
  }
}

/* ClpDualRowSteepest::unrollWeights :1022 */
static void unrollWeights(OrcModel *M)
{
  if (M->pivotRule == 0)
    return;
  for (int i = 0; i < M->numberAlt; i++)
    M->weights[M->altWeightIndex[i]] = M->altWeightValue[i];
  M->numberAlt = 0;
}

/* ClpDualRowDantzig::pivotRow :56-92 */
static int dantzigPivotRow(OrcModel *M)
{
  double tolerance = M->primalTolerance;
  if (M->largestPrimalError > 1.0e-8)
    tolerance *= M->largestPrimalError / 1.0e-8;
  double largest = 0.0;
  int chosenRow = -1;
  for (int iRow = 0; iRow < M->m; iRow++) {
    int iSequence = M->pivotVariable[iRow];
    double value = M->sol[iSequence];
    double infeas = dmax(value - M->upper[iSequence], M->lower[iSequence] - value);
    if (infeas > tolerance) {
      if (infeas > largest) {
        if (!flagged(M, iSequence)) {
          chosenRow = iRow;
          largest = infeas;
        }
      }
    }
  }
  return chosenRow;
}

/* ClpDualRowSteepest::pivotRow :179-364, all of it: the touch-up of the last pivot row (:210-250), the "can't trust
 * infeasibilities" tolerance (:251-257), numberWanted from mode_ (:258-278: the partial scan of modes 2 and 3 -- the constructor's
 * default is mode 3, src/ClpDualRowSteepest.hpp:118), the random start and the two passes with the early break (:279-335; a flagged
 * candidate hands its ticket back, the last pivot row that is put off by `continue` does not use one), and the second call with
 * largestDualError_ = 0 when nothing was chosen under the changed tolerance (:338-346: everything again, another random number
 * included).  `factorization()->numberElements()` (:262) is the one input that depends on the LU code behind ClpFactorization:
 * M->factorElements, see factorElementsModel(). */
static int steepestPivotRow(OrcModel *M)
{
  double largest = 0.0;
  int chosenRow = -1;
  int lastPivotRow = M->pivotRow;
  double tolerance = M->primalTolerance;
  double error = dmin(1.0e-2, M->largestPrimalError);
  tolerance = tolerance + error;
  tolerance = dmin(1000.0, tolerance);
  tolerance *= tolerance;
  int toleranceChanged = 0;
  if (lastPivotRow >= 0 && lastPivotRow < M->m) {
    int iPivot = M->pivotVariable[lastPivotRow];
    double value = M->sol[iPivot], lower = M->lower[iPivot], upper = M->upper[iPivot];
    if (value > upper + tolerance) {
      value -= upper;
      value *= value;
      infeasibleAdd(M, lastPivotRow, value);
    } else if (value < lower - tolerance) {
      value -= lower;
      value *= value;
      infeasibleAdd(M, lastPivotRow, value);
    } else {
      if (M->infeas[lastPivotRow])
        M->infeas[lastPivotRow] = REALLY_TINY;
    }
  }
  int number = M->numberInfeasible;
  if (M->numberIterations < M->lastBadIteration + 200) {
    if (M->largestDualError > M->largestPrimalError) {
      tolerance *= dmin(M->largestDualError / M->largestPrimalError, 1000.0);
      toleranceChanged = 1;
    } else if (M->debugToleranceFactor > 0.0 && M->largestDualError >= 0.0) {
      /* fault injection (option "debug_tolerance_factor"): the two errors are rounding noise on a healthy LP and which of them is larger
       * is not reproducible between two factorizations; tests arm the branch with a factor of their own (the second call sees
       * largestDualError_ < 0 below and leaves it alone, as it leaves the real one) */
      tolerance *= M->debugToleranceFactor;
      toleranceChanged = 1;
    }
  }
  int numberWanted;
  if (M->steepestMode < 2) {
    numberWanted = number + 1;
  } else if (M->steepestMode == 2) {
    numberWanted = number / 8 > M->chuzrFloor ? number / 8 : M->chuzrFloor;
  } else {
    double ratio = (double)M->factorElements / (double)M->m;
    numberWanted = number / 8 > M->chuzrFloor ? number / 8 : M->chuzrFloor;
    if (ratio < 1.0) {
      numberWanted = number / 20 > M->chuzrFloor ? number / 20 : M->chuzrFloor;
    } else if (ratio > 10.0) {
      ratio = number * (ratio / 80.0);
      if (ratio > number)
        numberWanted = number + 1;
      else
        numberWanted = (int)ratio > M->chuzrFloor ? (int)ratio : M->chuzrFloor;
    }
  }
  if (M->largestPrimalError > 1.0e-3)
    numberWanted = number + 1; /* be safe */
  if (numberWanted <= number)
    M->numberPartialScans++;
  int start[4];
  start[1] = number;
  start[2] = 0;
  double dstart = ((double)number) * randomDouble(M);
  start[0] = (int)dstart;
  start[3] = start[0];
  for (int iPass = 0; iPass < 2; iPass++) {
    int end = start[2 * iPass + 1];
    for (int i = start[2 * iPass]; i < end; i++) {
      int iRow = M->infIndex[i];
      double value = M->infeas[iRow];
      if (value > tolerance) {
        double weight = dmin(M->weights[iRow], 1.0e50);
        if (value > largest * weight) {
          if (iRow == lastPivotRow) {
            if (value * 1.0e-10 < largest * weight)
              continue;
            else
              value *= 1.0e-10;
          }
          int iSequence = M->pivotVariable[iRow];
          if (!flagged(M, iSequence)) {
            if (M->sol[iSequence] > M->upper[iSequence] + tolerance || M->sol[iSequence] < M->lower[iSequence] - tolerance) {
              chosenRow = iRow;
              largest = value / weight;
            }
          } else {
            numberWanted++; /* "just to make sure we don't exit before got something" */
          }
        }
        numberWanted--;
        if (!numberWanted)
          break;
      }
    }
    if (!numberWanted)
      break;
  }
  if (chosenRow < 0 && toleranceChanged) {
    /* "won't line up with checkPrimalSolution - do again" (:338-346); cannot loop: the second call sees no dual error */
    double saveError = M->largestDualError;
    M->largestDualError = M->debugToleranceFactor > 0.0 ? -1.0 : 0.0; /* (-1 only tells the injected branch above that this is the second call) */
    M->numberChuzrRecalls++;
    chosenRow = steepestPivotRow(M);
    number = M->numberInfeasible;
    M->largestDualError = saveError;
  }
  if (chosenRow < 0 && lastPivotRow < 0) {
    int nLeft = 0;
    for (int i = 0; i < number; i++) {
      int iRow = M->infIndex[i];
      if (fabs(M->infeas[iRow]) > 1.0e-50)
        M->infIndex[nLeft++] = iRow;
      else
        M->infeas[iRow] = 0.0;
    }
    M->numberInfeasible = nLeft;
    M->numberPrimalInfeasibilities = nLeft;
  }
  return chosenRow;
}

/* ClpDualRowSteepest::updatePrimalSolution :630-763 / ClpDualRowDantzig :131-170 */
static void updatePrimalSolution(OrcModel *M, const int *which, const double *work, int number, double primalRatio,
                                 double *objectiveChange)
{
  double changeObj = 0.0;
  double tolerance = M->primalTolerance;
  for (int i = 0; i < number; i++) {
    int iRow = which[i];
    int iPivot = M->pivotVariable[iRow];
    double value = M->sol[iPivot];
    double cost = M->cost[iPivot];
    double change = primalRatio * work[i];
    value -= change;
    changeObj -= change * cost;
    M->sol[iPivot] = value;
    if (M->pivotRule) {
      double lower = M->lower[iPivot], upper = M->upper[iPivot];
      if (value < lower - tolerance) {
        value -= lower;
        value *= value;
        infeasibleAdd(M, iRow, value);
      } else if (value > upper + tolerance) {
        value -= upper;
        value *= value;
        infeasibleAdd(M, iRow, value);
      } else {
        if (M->infeas[iRow])
          M->infeas[iRow] = REALLY_TINY;
      }
    }
  }
  if (M->pivotRule) {
    int iRow = M->pivotRow;
    if (M->infeas[iRow])
      M->infeas[iRow] = REALLY_TINY;
  }
  *objectiveChange += changeObj;
}

/* pack a dense position-space vector, ascending (CoinIndexedVector::scan) */
static int packDense(double *dense, int m, int *index, double *value)
{
  int number = 0;
  for (int i = 0; i < m; i++) {
    if (dense[i]) {
      index[number] = i;
      value[number++] = dense[i];
      dense[i] = 0.0;
    }
  }
  return number;
}

/* ClpDualRowSteepest::updateWeights :375-540 (includes the FTRAN of the entering column, which
 * ClpFactorization::updateTwoColumnsFT :2889 fuses with the FTRAN of the BTRAN result) */
static double updateWeights(OrcModel *M)
{
  const int m = M->m;
  double alpha = 0.0;
  /* FTRAN entering column: rowWork1 holds the unpacked column on entry */
  ftran(M, M->rowWork1);
  M->numberW = packDense(M->rowWork1, m, M->wIndex, M->wValue);
  if (M->pivotRule == 0) {
    for (int i = 0; i < M->numberW; i++)
      if (M->wIndex[i] == M->pivotRow) {
        alpha = M->wValue[i];
        break;
      }
    return alpha;
  }
  double norm = 0.0;
  double *work2 = M->rowWork2;
  for (int i = 0; i < M->numberPi; i++) {
    double value = M->piValue[i];
    norm += value * value;
    work2[M->piIndex[i]] = value;
  }
  ftran(M, work2); /* tau = B^-1 rho */
  int pivotRow = M->pivotRow;
  norm /= M->alpha * M->alpha;
  double multiplier = 2.0 / M->alpha;
  int nSave = 0;
  for (int i = 0; i < M->numberW; i++) {
    int iRow = M->wIndex[i];
    double theta = M->wValue[i];
    if (iRow == pivotRow)
      alpha = theta;
    double devex = M->weights[iRow];
    M->altWeightValue[nSave] = devex;
    M->altWeightIndex[nSave++] = iRow;
    double value = work2[iRow];
    devex += theta * (theta * norm + value * multiplier);
    if (devex < DEVEX_TRY_NORM)
      devex = DEVEX_TRY_NORM;
    M->weights[iRow] = devex;
  }
  M->numberAlt = nSave;
  if (norm < DEVEX_TRY_NORM)
    norm = DEVEX_TRY_NORM;
  M->weights[pivotRow] = norm;
  memset(work2, 0, sizeof(double) * (size_t)m);
  return alpha;
}

/* ------------------------------------------------------------------------------------------ */
/* the dual simplex proper                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* ClpSimplexDual::dualRow :2962-3140 (no free variables, no values pass) */
static void unpackColumn(OrcModel *M, double *dense, int iSequence);

/* ClpSimplexDual::nextSuperBasic :8285-8302 */
static int nextSuperBasic(OrcModel *M)
{
  if (M->firstFree >= 0) {
    const int N = M->m + M->n;
    int returnValue = M->firstFree;
    int iColumn = M->firstFree + 1;
    for (; iColumn < N; iColumn++) {
      if (getStatus(M, iColumn) == ST_FREE)
        if (fabs(M->dj[iColumn]) > 1.0e2 * M->dualTolerance)
          break;
    }
    M->firstFree = iColumn;
    if (M->firstFree == N)
      M->firstFree = -1;
    return returnValue;
  } else {
    return -1;
  }
}

/* ClpSimplexDual::dualRow :2962-3140 */
static void dualRow(OrcModel *M)
{
  int chosenRow = -1;
  if (M->freeNonbasic) {
    /* first see if any free variables and put them in basis (:3005-3055) */
    int nextFree = nextSuperBasic(M);
    if (nextFree >= 0) {
      /* unpack vector and find a good pivot */
      double *work = M->rowWork3;
      unpackColumn(M, work, nextFree);
      ftran(M, work);
      double bestFeasibleAlpha = 0.0, bestInfeasibleAlpha = 0.0;
      int bestFeasibleRow = -1, bestInfeasibleRow = -1;
      for (int iRow = 0; iRow < M->m; iRow++) { /* (the reference walks the packed list of the updated column; the choices are strict
                                                     maxima, so the order matters only between exact ties) */
        double alpha = fabs(work[iRow]);
        work[iRow] = 0.0;
        if (alpha > 1.0e-3) {
          int iSequence = M->pivotVariable[iRow];
          double value = M->sol[iSequence], lower = M->lower[iSequence], upper = M->upper[iSequence];
          double infeasibility = 0.0;
          if (value > upper)
            infeasibility = value - upper;
          else if (value < lower)
            infeasibility = lower - value;
          if (infeasibility * alpha > bestInfeasibleAlpha && alpha > 1.0e-1) {
            if (!flagged(M, iSequence)) {
              bestInfeasibleAlpha = infeasibility * alpha;
              bestInfeasibleRow = iRow;
            }
          }
          if (alpha > bestFeasibleAlpha && (lower > -1.0e20 || upper < 1.0e20)) {
            bestFeasibleAlpha = alpha;
            bestFeasibleRow = iRow;
          }
        }
      }
      if (bestInfeasibleRow >= 0)
        chosenRow = bestInfeasibleRow;
      else if (bestFeasibleAlpha > 1.0e-2)
        chosenRow = bestFeasibleRow;
      if (chosenRow >= 0)
        M->numberFreeFirstRows++;
    }
  }
  if (chosenRow >= 0)
    M->pivotRow = chosenRow;
  else
    M->pivotRow = M->pivotRule ? steepestPivotRow(M) : dantzigPivotRow(M);
  if (M->pivotRow >= 0) {
    M->sequenceOut = M->pivotVariable[M->pivotRow];
    M->valueOut = M->sol[M->sequenceOut];
    M->lowerOut = M->lower[M->sequenceOut];
    M->upperOut = M->upper[M->sequenceOut];
    if (M->valueOut > M->upperOut) {
      M->directionOut = -1;
      M->dualOut = M->valueOut - M->upperOut;
    } else if (M->valueOut < M->lowerOut) {
      M->directionOut = 1;
      M->dualOut = M->lowerOut - M->valueOut;
    } else {
      if (M->valueOut - M->lowerOut < M->upperOut - M->valueOut) {
        M->directionOut = 1;
        M->dualOut = M->lowerOut - M->valueOut;
      } else {
        M->directionOut = -1;
        M->dualOut = M->valueOut - M->upperOut;
      }
    }
  }
}

/* ClpSimplexDual::dualColumn0, the general branch "some free or super basic" (:4058-4179), run on the tableau row when the last
 * checkBothSolutions cleared moreSpecialOptions_ & 8 (the fused first pass of the pricing is then not taken, ClpSimplexDual.cpp:1296-1300).
 * Rows first, then columns.  A free or superbasic variable worth keeping becomes the incoming one (the largest |alpha| of them: freePivot)
 * and is given fake bounds on the way when its value allows.  Fills candidate list 0; sets sequenceIn / theta / alpha for a free choice. */
static void dualColumn0General(OrcModel *M, double acceptablePivot)
{
  const double tentativeTheta = 1.0e25;
  double upperTheta = 1.0e31;
  double freePivot = acceptablePivot;
  int numberRemaining = 0;
  int *index = M->spareIndex[0];
  double *spare = M->spareValue[0];
  double badFree = 0.0;
  for (int iSection = 0; iSection < 2; iSection++) {
    const int number = iSection ? M->numberColNz : M->numberPi;
    const int *which = iSection ? M->colIndex : M->piIndex;
    const double *work = iSection ? M->colValue : M->piValue;
    const int addSequence = iSection ? 0 : M->n;
    for (int i = 0; i < number; i++) {
      const int jSequence = which[i] + addSequence;
      if (jSequence == M->sequenceOut)
        continue; /* the leaving variable is not a candidate (:4083) */
      double alpha, oldValue, value;
      int keep;
      switch (getStatus(M, jSequence)) {
      case ST_BASIC:
      case ST_FIXED:
        break;
      case ST_FREE:
      case ST_SUPER:
        alpha = work[i];
        oldValue = M->dj[jSequence];
        if (oldValue > M->dualTolerance) {
          keep = 1;
        } else if (oldValue < -M->dualTolerance) {
          keep = 1;
        } else {
          if (fabs(alpha) > dmax(10.0 * acceptablePivot, 1.0e-5)) {
            keep = 1;
          } else {
            keep = 0;
            badFree = dmax(badFree, fabs(alpha));
          }
        }
        if (keep) {
          /* free - choose largest */
          if (fabs(alpha) > freePivot) {
            freePivot = fabs(alpha);
            M->sequenceIn = jSequence;
            M->theta = oldValue / alpha;
            M->alpha = alpha;
          }
          /* give fake bounds if possible */
          if (2.0 * fabs(M->sol[jSequence]) < M->dualBound) {
            setFake(M, jSequence, FAKE_BOTH);
            M->numberFake++;
            value = oldValue - tentativeTheta * alpha;
            if (value > M->dualTolerance) {
              /* pretend coming in from upper bound */
              M->upper[jSequence] = M->sol[jSequence];
              M->lower[jSequence] = M->upper[jSequence] - M->dualBound;
              setStatus(M, jSequence, ST_UPPER);
            } else {
              /* pretend coming in from lower bound */
              M->lower[jSequence] = M->sol[jSequence];
              M->upper[jSequence] = M->lower[jSequence] + M->dualBound;
              setStatus(M, jSequence, ST_LOWER);
            }
          }
        }
        break;
      case ST_UPPER:
        alpha = work[i];
        oldValue = M->dj[jSequence];
        value = oldValue - tentativeTheta * alpha;
        if (value > M->dualTolerance) {
          value = oldValue - upperTheta * alpha;
          if (value > M->dualTolerance && -alpha >= acceptablePivot)
            upperTheta = (oldValue - M->dualTolerance) / alpha;
          spare[numberRemaining] = alpha;
          index[numberRemaining++] = jSequence;
        }
        break;
      case ST_LOWER:
        alpha = work[i];
        oldValue = M->dj[jSequence];
        value = oldValue - tentativeTheta * alpha;
        if (value < -M->dualTolerance) {
          value = oldValue - upperTheta * alpha;
          if (value < -M->dualTolerance && alpha >= acceptablePivot)
            upperTheta = (oldValue + M->dualTolerance) / alpha;
          spare[numberRemaining] = alpha;
          index[numberRemaining++] = jSequence;
        }
        break;
      }
    }
  }
  M->numberCandidates = numberRemaining;
  M->upperThetaFirst = upperTheta;
  M->badFree = badFree;
}

/* ClpSimplexDual::dualColumn :4192-4927; the first pass was fused into pricing (spareIntArray_[0]
 * == -2 path, :4273-4281).  Candidate list is in spare list 0.  Returns bestPossible. */
static double dualColumn(OrcModel *M, double acceptablePivot)
{
  const int top = M->m + M->n; /* the reference uses numberColumns_ as the top of the two lists */
  int numberPossiblySwapped = 0;
  int numberRemaining = M->numberCandidates;
  double totalThru = 0.0;
  double bestEverPivot = acceptablePivot;
  int lastSequence = -1;
  double lastPivot = 0.0;
  double upperTheta = M->upperThetaFirst;
  double newTolerance = M->dualTolerance;
  int modifyCosts = 0;
  double increaseInObjective = 0.0;
  int iFlip = 0;
  int interesting[2], swapped[2];
  double *array[2], *spare, *spare2;
  int *indices[2], *index, *index2;
  array[0] = M->spareValue[0];
  indices[0] = M->spareIndex[0];
  array[1] = M->spareValue[1];
  indices[1] = M->spareIndex[1];
  spare = array[0];
  index = indices[0];
  for (int i = 0; i < 2; i++) {
    interesting[i] = 0;
    swapped[i] = top;
  }
  double bestPossible = 1.0;
  const int freeChosen = M->sequenceIn >= 0; /* the general branch of dualColumn0 chose a free variable: "always choose" (:4321) */
  if (!freeChosen) {
    M->alpha = 0.0;
    M->sequenceIn = -1;
  }
  double tentativeTheta = 1.0e25;
  interesting[0] = numberRemaining;
  if (!numberRemaining && M->sequenceIn < 0)
    return 0.0; /* looks infeasible */
  int badSumPivots = 0;
  if (!freeChosen) {
  M->theta = 1.0e50;
  tentativeTheta = dmax(10.0 * upperTheta, 1.0e-7);
  while (tentativeTheta < 1.0e22) {
    double thruThis = 0.0;
    double bestPivot = acceptablePivot;
    int bestSequence = -1;
    numberPossiblySwapped = top;
    numberRemaining = 0;
    upperTheta = 1.0e50;
    spare = array[iFlip];
    index = indices[iFlip];
    spare2 = array[1 - iFlip];
    index2 = indices[1 - iFlip];
    double increaseInThis = 0.0;
    for (int i = 0; i < interesting[iFlip]; i++) {
      int iSequence = index[i];
      double alpha = spare[i];
      double oldValue = M->dj[iSequence];
      double value = oldValue - tentativeTheta * alpha;
      if (alpha < 0.0) {
        if (value > newTolerance) {
          double range = M->upper[iSequence] - M->lower[iSequence];
          thruThis -= range * alpha;
          increaseInThis -= (oldValue + M->dualTolerance) * range;
          spare2[--numberPossiblySwapped] = alpha;
          index2[numberPossiblySwapped] = iSequence;
          if (fabs(alpha) > bestPivot) {
            bestPivot = fabs(alpha);
            bestSequence = numberPossiblySwapped;
          }
        } else {
          value = oldValue - upperTheta * alpha;
          if (value > newTolerance && -alpha >= acceptablePivot)
            upperTheta = (oldValue - newTolerance) / alpha;
          spare2[numberRemaining] = alpha;
          index2[numberRemaining++] = iSequence;
        }
      } else {
        if (value < -newTolerance) {
          double range = M->upper[iSequence] - M->lower[iSequence];
          thruThis += range * alpha;
          increaseInThis += (oldValue - M->dualTolerance) * range;
          spare2[--numberPossiblySwapped] = alpha;
          index2[numberPossiblySwapped] = iSequence;
          if (fabs(alpha) > bestPivot) {
            bestPivot = fabs(alpha);
            bestSequence = numberPossiblySwapped;
          }
        } else {
          value = oldValue - upperTheta * alpha;
          if (value < -newTolerance && alpha >= acceptablePivot)
            upperTheta = (oldValue + newTolerance) / alpha;
          spare2[numberRemaining] = alpha;
          index2[numberRemaining++] = iSequence;
        }
      }
    }
    swapped[1 - iFlip] = numberPossiblySwapped;
    interesting[1 - iFlip] = numberRemaining;
    double check = fabs(totalThru + thruThis);
    check += 1.0e-8 + 1.0e-10 * check;
    if (check >= fabs(M->dualOut) || increaseInObjective + increaseInThis < 0.0) {
      /* we should be pivoting in this batch: compress down to this lot */
      numberRemaining = 0;
      for (int i = top - 1; i >= swapped[1 - iFlip]; i--) {
        spare[numberRemaining] = spare2[i];
        index[numberRemaining++] = index2[i];
      }
      interesting[iFlip] = numberRemaining;
      int iTry;
      const int MAXTRY = 100;
      for (iTry = 0; iTry < MAXTRY; iTry++) {
        upperTheta = 1.0e50;
        numberPossiblySwapped = top;
        numberRemaining = 0;
        increaseInThis = 0.0;
        thruThis = 0.0;
        spare = array[iFlip];
        index = indices[iFlip];
        spare2 = array[1 - iFlip];
        index2 = indices[1 - iFlip];
        for (int i = 0; i < interesting[iFlip]; i++) {
          int iSequence = index[i];
          double alpha = spare[i];
          double oldValue = M->dj[iSequence];
          double value = oldValue - upperTheta * alpha;
          if (alpha < 0.0) {
            if (value > newTolerance) {
              if (-alpha >= acceptablePivot)
                upperTheta = (oldValue - newTolerance) / alpha;
            }
          } else {
            if (value < -newTolerance) {
              if (alpha >= acceptablePivot)
                upperTheta = (oldValue + newTolerance) / alpha;
            }
          }
        }
        bestPivot = acceptablePivot;
        M->sequenceIn = -1;
        double largestPivot = acceptablePivot;
        double sumBadPivots = 0.0;
        badSumPivots = 0;
        upperTheta *= 1.0000000001;
        for (int i = 0; i < interesting[iFlip]; i++) {
          int iSequence = index[i];
          double alpha = spare[i];
          double value = M->dj[iSequence] - upperTheta * alpha;
          double badDj = 0.0;
          int addToSwapped = 0;
          if (alpha < 0.0) {
            if (value >= 0.0) {
              addToSwapped = 1;
              badDj = -M->dj[iSequence] - M->dualTolerance;
            }
          } else {
            if (value <= 0.0) {
              addToSwapped = 1;
              badDj = M->dj[iSequence] - M->dualTolerance;
            }
          }
          if (!addToSwapped) {
            spare2[numberRemaining] = alpha;
            index2[numberRemaining++] = iSequence;
          } else {
            spare2[--numberPossiblySwapped] = alpha;
            index2[numberPossiblySwapped] = iSequence;
            int take = 0;
            double absAlpha = fabs(alpha);
            if (absAlpha > bestPivot)
              take = 1;
            if (absAlpha < acceptablePivot && upperTheta < 1.0e20) {
              if (alpha < 0.0) {
                if (value > M->dualTolerance) {
                  double gap = M->upper[iSequence] - M->lower[iSequence];
                  if (gap < 1.0e20)
                    sumBadPivots += value * gap;
                  else
                    sumBadPivots += 1.0e20;
                }
              } else {
                if (value < -M->dualTolerance) {
                  double gap = M->upper[iSequence] - M->lower[iSequence];
                  if (gap < 1.0e20)
                    sumBadPivots -= value * gap;
                  else
                    sumBadPivots += 1.0e20;
                }
              }
            }
            if (take) {
              M->sequenceIn = numberPossiblySwapped;
              bestPivot = absAlpha;
              M->theta = M->dj[iSequence] / alpha;
              largestPivot = dmax(largestPivot, 0.5 * bestPivot);
            }
            double range = M->upper[iSequence] - M->lower[iSequence];
            thruThis += range * fabs(alpha);
            increaseInThis += badDj * range;
          }
        }
        if (sumBadPivots > 1.0e4) {
          if (M->fac.nEta > 3) {
            badSumPivots = 1;
            break;
          }
        }
        swapped[1 - iFlip] = numberPossiblySwapped;
        interesting[1 - iFlip] = numberRemaining;
        double increase = (fabs(M->dualOut) - totalThru) * M->theta;
        increase += increaseInObjective;
        if (M->theta < 0.0)
          thruThis += fabs(M->dualOut);
        if (increaseInObjective < 0.0 && increase < 0.0 && lastSequence >= 0) {
          bestPivot = 0.0;
        } else {
          totalThru += thruThis;
          increaseInObjective += increaseInThis;
        }
        if (bestPivot < 0.1 * bestEverPivot && bestEverPivot > 1.0e-6 && (bestPivot < 1.0e-3 || totalThru * 2.0 > fabs(M->dualOut))) {
          M->sequenceIn = lastSequence;
          iFlip = 1 - iFlip;
          break;
        } else if (M->sequenceIn == -1 && upperTheta > M->largeValue) {
          if (lastPivot > acceptablePivot) {
            M->sequenceIn = lastSequence;
            iFlip = 1 - iFlip;
          }
          break;
        } else if (totalThru >= fabs(M->dualOut)) {
          modifyCosts = 1;
          break;
        } else {
          lastSequence = M->sequenceIn;
          if (bestPivot > bestEverPivot)
            bestEverPivot = bestPivot;
          iFlip = 1 - iFlip;
          modifyCosts = 1;
        }
      }
      if (iTry == MAXTRY)
        iFlip = 1 - iFlip;
      break;
    } else {
      /* skip this lot */
      if (bestPivot > 1.0e-3 || bestPivot > bestEverPivot) {
        bestEverPivot = bestPivot;
        lastSequence = bestSequence;
      } else {
        /* keep old swapped */
        memcpy(array[1 - iFlip] + swapped[iFlip], array[iFlip] + swapped[iFlip], sizeof(double) * (size_t)(top - swapped[iFlip]));
        memcpy(indices[1 - iFlip] + swapped[iFlip], indices[iFlip] + swapped[iFlip], sizeof(int) * (size_t)(top - swapped[iFlip]));
        swapped[1 - iFlip] = swapped[iFlip];
      }
      increaseInObjective += increaseInThis;
      iFlip = 1 - iFlip;
      tentativeTheta = 2.0 * upperTheta;
      totalThru += thruThis;
    }
  }
  if (M->sequenceIn < 0 && lastSequence >= 0) {
    M->sequenceIn = lastSequence;
    iFlip = 1 - iFlip;
  }
  double minimumTheta = (M->upperOut > M->lowerOut) ? 1.0e-18 : 0.0;
  if (M->sequenceIn >= 0) {
    iFlip = 1 - iFlip;
    spare = array[iFlip];
    index = indices[iFlip];
    M->alpha = spare[M->sequenceIn];
    M->sequenceIn = index[M->sequenceIn];
    double oldValue = M->dj[M->sequenceIn];
    M->theta = dmax(oldValue / M->alpha, 0.0);
    if (M->theta < minimumTheta && fabs(M->alpha) < 1.0e5)
      M->theta = minimumTheta;
    if (modifyCosts && !badSumPivots) {
      for (int i = top - 1; i >= swapped[iFlip]; i--) {
        int iSequence = index[i];
        double alpha = spare[i];
        double value = M->dj[iSequence] - M->theta * alpha;
        if (alpha < 0.0) {
          if (value > M->dualTolerance) {
            double modification = alpha * M->theta - M->dj[iSequence] + newTolerance;
            M->dj[iSequence] += modification;
            M->cost[iSequence] += modification;
            if (modification)
              M->numberChanged++;
          }
        } else {
          if (-value > M->dualTolerance) {
            double modification = alpha * M->theta - M->dj[iSequence] - newTolerance;
            M->dj[iSequence] += modification;
            M->cost[iSequence] += modification;
            if (modification)
              M->numberChanged++;
          }
        }
      }
    }
  }
  } /* !freeChosen */
  if ((badSumPivots || fabs(M->theta * M->badFree) > 10.0 * M->dualTolerance) && M->fac.nEta) {
    /* things look bad: force a refactorization (:4776-4784) */
    M->sequenceIn = -1;
    M->acceptablePivot_ = -M->acceptablePivot_;
  }
  if (M->sequenceIn >= 0) {
    M->lowerIn = M->lower[M->sequenceIn];
    M->upperIn = M->upper[M->sequenceIn];
    M->valueIn = M->sol[M->sequenceIn];
    M->dualIn = M->dj[M->sequenceIn];
    /* MODIFYCOST > 1: modify cost to hit zero exactly (:4796-4834) */
    double modification = M->theta * M->alpha - M->dualIn;
    double moveObjective = fabs(modification * M->sol[M->sequenceIn]);
    double smallMove = dmax(fabs(M->objectiveValue), 1.0e-3);
    if (moveObjective > smallMove)
      modification *= smallMove / moveObjective;
    if (badSumPivots)
      modification = 0.0;
    M->dualIn += modification;
    M->dj[M->sequenceIn] = M->dualIn;
    M->cost[M->sequenceIn] += modification;
    if (modification)
      M->numberChanged++;
    if (M->alpha < 0.0) {
      M->directionIn = -1;
      M->upperIn = M->valueIn;
    } else {
      M->directionIn = 1;
      M->lowerIn = M->valueIn;
    }
    if (fabs(M->alpha) < 1.0e-6) {
      /* need bestPossible (:4851-4908) */
      bestPossible = 0.0;
      const double tent = 1.0e25, dualT = -M->dualTolerance;
      for (int iSection = 0; iSection < 2; iSection++) {
        int number = iSection ? M->numberColNz : M->numberPi;
        const int *which = iSection ? M->colIndex : M->piIndex;
        const double *work = iSection ? M->colValue : M->piValue;
        int addSequence = iSection ? 0 : M->n;
        for (int i = 0; i < number; i++) {
          int iSequence = which[i] + addSequence;
          int st = getStatus(M, iSequence);
          double mult = 1.0;
          if (st == ST_UPPER)
            mult = -1.0;
          if ((st == ST_FREE || st == ST_SUPER) && !M->noFreeOrSuper)
            bestPossible = dmax(bestPossible, fabs(work[i])); /* :4889-4893 */
          if (st == ST_UPPER || st == ST_LOWER) {
            double alpha = work[i] * mult;
            if (alpha > 0.0) {
              double oldValue = M->dj[iSequence] * mult;
              double value = oldValue - tent * alpha;
              if (value < dualT)
                bestPossible = dmax(bestPossible, alpha);
            }
          }
        }
      }
    } else {
      bestPossible = fabs(M->alpha);
    }
  } else {
    bestPossible = 0.0;
    M->alpha = 0.0;
  }
  return bestPossible;
}

/* ClpSimplexDual::flipBounds :6345-6401 */
static void flipBounds(OrcModel *M)
{
  for (int iSection = 0; iSection < 2; iSection++) {
    int number = iSection ? M->numberColFlip : M->numberRowFlip;
    const int *which = iSection ? M->colFlip : M->rowFlip;
    int addSequence = iSection ? 0 : M->n;
    for (int i = 0; i < number; i++) {
      int iSequence = which[i] + addSequence;
      int st = getStatus(M, iSequence);
      if (st == ST_UPPER) {
        setStatus(M, iSequence, ST_LOWER);
        M->sol[iSequence] = M->lower[iSequence];
      } else if (st == ST_LOWER) {
        setStatus(M, iSequence, ST_UPPER);
        M->sol[iSequence] = M->upper[iSequence];
      }
    }
  }
  M->numberRowFlip = 0;
  M->numberColFlip = 0;
}

/* ClpSimplexDual::updateDualsInDual :2430-2883.  outputArray (dense, length m) gets the rhs
 * movement of the flips.  Returns number of flips. */
static int updateDualsInDual(OrcModel *M, double *outputArray, double theta, double *objectiveChange, int fullRecompute)
{
  const int n = M->n, m = M->m;
  int numberInfeasibilities = 0;
  double tolerance = M->dualTolerance + dmin(1.0e-2, M->largestDualError);
  double changeObj = 0.0;
  M->numberRowFlip = 0;
  M->numberColFlip = 0;
  if (!fullRecompute) {
    {
      double *reducedCost = M->dj + n;
      const double *lower = M->lower + n, *upper = M->upper + n, *cost = M->cost + n;
      const unsigned char *statusArray = M->status + n;
      const double multiplier[] = { 0.0, 0.0, -1.0, 1.0 };
      for (int i = 0; i < M->numberPi; i++) {
        int iSequence = M->piIndex[i];
        double alphaI = M->piValue[i];
        int iStatus = (statusArray[iSequence] & 3) - 1;
        if (iStatus) {
          double value = reducedCost[iSequence] - theta * alphaI;
          reducedCost[iSequence] = value;
          double mult = multiplier[iStatus + 1];
          value *= mult;
          if (value < -tolerance) {
            double movement = mult * (lower[iSequence] - upper[iSequence]);
            M->rowFlip[M->numberRowFlip++] = iSequence;
            changeObj -= movement * cost[iSequence];
            outputArray[iSequence] += movement;
          }
        }
      }
    }
    {
      double *reducedCost = M->dj;
      const double *lower = M->lower, *upper = M->upper, *cost = M->cost;
      const unsigned char *statusArray = M->status;
      const double multiplier[] = { -1.0, 1.0, -1.0, 1.0 }; /* [0],[1] overwritten as in :2500 */
      for (int i = 0; i < M->numberColNz; i++) {
        int iSequence = M->colIndex[i];
        double alphaI = M->colValue[i];
        int iStatus = (statusArray[iSequence] & 3) - 1;
        if (!M->noFreeOrSuper && (statusArray[iSequence] & 7) == ST_SUPER)
          continue; /* the general column loop (:2596-2651) has no case for a superbasic variable: its dj is left as it was */
        if (iStatus) {
          double value = reducedCost[iSequence] - theta * alphaI;
          reducedCost[iSequence] = value;
          double mult = multiplier[iStatus + 1];
          value *= mult;
          if (value < -tolerance && iStatus > 0) {
            double movement = mult * (upper[iSequence] - lower[iSequence]);
            M->colFlip[M->numberColFlip++] = iSequence;
            changeObj += movement * cost[iSequence];
            addColumn(M, outputArray, iSequence, movement);
          }
        }
      }
    }
    numberInfeasibilities = M->numberRowFlip + M->numberColFlip;
    M->numberPi = 0;
    M->numberColNz = 0;
  } else {
    /* :2648-2868, with TRY_SET_FAKE */
    for (int iSection = 0; iSection < 2; iSection++) {
      int lo = iSection ? 0 : n, hi = iSection ? n : n + m;
      for (int iSequence = lo; iSequence < hi; iSequence++) {
        double value = M->dj[iSequence];
        int st = getStatus(M, iSequence);
        double movement = 0.0;
        int flip = 0;
        if (st == ST_UPPER) {
          if (value > tolerance) {
            flip = 1;
            movement = M->lower[iSequence] - M->upper[iSequence];
            if (fabs(movement) > M->dualBound) {
              if (getFake(M, iSequence) == FAKE_NONE) {
                setFake(M, iSequence, FAKE_LOWER);
                M->lower[iSequence] = M->upper[iSequence] - M->dualBound;
                movement = M->lower[iSequence] - M->upper[iSequence];
                M->numberFake++;
              }
            }
          } else if (value > -tolerance) {
            if (getFake(M, iSequence) == FAKE_UPPER) {
              movement = M->lower[iSequence] - M->upper[iSequence];
              setStatus(M, iSequence, ST_LOWER);
              M->sol[iSequence] = M->lower[iSequence];
              changeObj += movement * M->cost[iSequence];
            }
          }
        } else if (st == ST_LOWER) {
          if (value < -tolerance) {
            flip = 1;
            movement = M->upper[iSequence] - M->lower[iSequence];
            if (fabs(movement) > M->dualBound) {
              if (getFake(M, iSequence) == FAKE_NONE) {
                setFake(M, iSequence, FAKE_UPPER);
                M->upper[iSequence] = M->lower[iSequence] + M->dualBound;
                movement = M->upper[iSequence] - M->lower[iSequence];
                M->numberFake++;
              }
            }
          } else if (value < tolerance) {
            if (getFake(M, iSequence) == FAKE_LOWER) {
              movement = M->upper[iSequence] - M->lower[iSequence];
              setStatus(M, iSequence, ST_UPPER);
              M->sol[iSequence] = M->upper[iSequence];
              changeObj += movement * M->cost[iSequence];
            }
          }
        }
        if (flip) {
          changeObj += movement * M->cost[iSequence];
          if (iSection) {
            M->colFlip[M->numberColFlip++] = iSequence;
            addColumn(M, outputArray, iSequence, movement);
          } else {
            M->rowFlip[M->numberRowFlip++] = iSequence - n;
            outputArray[iSequence - n] += -movement;
          }
        }
      }
    }
    numberInfeasibilities = M->numberRowFlip + M->numberColFlip;
    flipBounds(M);
  }
  *objectiveChange += changeObj;
  return numberInfeasibilities;
}

static void logPivot(OrcModel *M, int numberFlipped)
{
  if (M->logCount == M->logCap) {
    M->logCap = M->logCap ? 2 * M->logCap : 1024;
    M->log = (OrcPivotRecord *)realloc(M->log, sizeof(OrcPivotRecord) * (size_t)M->logCap);
  }
  OrcPivotRecord *r = &M->log[M->logCount++];
  r->iteration = M->numberIterations;
  r->sequenceIn = M->sequenceIn;
  r->sequenceOut = M->sequenceOut;
  r->pivotRow = M->pivotRow;
  r->numberFlipped = numberFlipped;
  r->reserved = 0;
  r->theta = M->theta;
  r->alpha = M->alpha;
  r->dualOut = M->dualOut;
  r->objective = M->objectiveValue;
  if (M->logLevel > 7) {
    printf("%d %.10g In: %c%d Out: %c%d theta %g alpha %g\n", M->numberIterations, M->objectiveValue,
           M->sequenceIn < M->n ? 'C' : 'R', M->sequenceIn < M->n ? M->sequenceIn : M->sequenceIn - M->n,
           M->sequenceOut < M->n ? 'C' : 'R', M->sequenceOut < M->n ? M->sequenceOut : M->sequenceOut - M->n, M->theta,
           M->alpha);
  }
}

/* ClpSimplexProgress::cycle (ClpSolve.cpp:4726-4825): the last ORC_CYCLE (in, out, way) triples; a repeat
 * of the oldest one with everything after it repeating too is a cycle of that length, two irregular
 * repeats count as 100 */
static int progressCycle(OrcModel *M, int in, int out, int wayIn, int wayOut)
{
  int matched = 0;
  for (int i = 1; i < ORC_CYCLE; i++)
    if (in == M->cycOut[i]) {
      matched = -1;
      break;
    }
  if (matched && M->cycIn[0] >= 0) {
    matched = 0;
    int nMatched = 0;
    const char way0 = M->cycWay[0];
    const int in0 = M->cycIn[0], out0 = M->cycOut[0];
    for (int k = 1; k < ORC_CYCLE - 4; k++) {
      if (in0 == M->cycIn[k] && out0 == M->cycOut[k] && way0 == M->cycWay[k]) {
        nMatched++;
        int end = ORC_CYCLE - k, j;
        for (j = 1; j < end; j++)
          if (M->cycIn[j + k] != M->cycIn[j] || M->cycOut[j + k] != M->cycOut[j] || M->cycWay[j + k] != M->cycWay[j])
            break;
        if (j == end) {
          matched = k;
          break;
        }
      }
    }
    if (matched <= 0 && nMatched > 1)
      matched = 100;
  }
  for (int i = 0; i < ORC_CYCLE - 1; i++) {
    M->cycIn[i] = M->cycIn[i + 1];
    M->cycOut[i] = M->cycOut[i + 1];
    M->cycWay[i] = M->cycWay[i + 1];
  }
  M->cycIn[ORC_CYCLE - 1] = in;
  M->cycOut[ORC_CYCLE - 1] = out;
  M->cycWay[ORC_CYCLE - 1] = (char)(1 - wayIn + 4 * (1 - wayOut));
  return matched;
}

/* parity hook: a sequence of pivots through progressCycle on a scratch model (tests compare it with the device's
 * ring-buffer form and with a Python restatement of ClpSolve.cpp:4726-4825) */
void orc_test_cycle(int n, const int *in, const int *out, const int *wayIn, const int *wayOut, int *matched)
{
  OrcModel *M = (OrcModel *)calloc(1, sizeof(OrcModel));
  for (int i = 0; i < ORC_CYCLE; i++) {
    M->cycIn[i] = M->cycOut[i] = -1;
    M->cycWay[i] = 0;
  }
  for (int i = 0; i < n; i++)
    matched[i] = progressCycle(M, in[i], out[i], wayIn[i], wayOut[i]);
  free(M);
}

/* ClpSimplex::housekeeping :2065-2489.  Returns 0 carry on, 1 refactorize,
 * 2 iteration limit. */
static int housekeeping(OrcModel *M, double objectiveChange, int numberFlipped)
{
  M->numberIterations++;
  if (M->pivotRow >= 0)
    M->pivotVariable[M->pivotRow] = M->sequenceIn;
  if (M->upper[M->sequenceIn] > 1.0e20 && M->lower[M->sequenceIn] < -1.0e20)
    M->progressFlag |= 2; /* making real progress (:2096-2100) */
  M->sol[M->sequenceIn] = M->valueIn;
  if (M->upper[M->sequenceOut] - M->lower[M->sequenceOut] < 1.0e-12)
    M->progressFlag |= 1;
  if (M->sequenceIn != M->sequenceOut) {
    setStatus(M, M->sequenceIn, ST_BASIC);
    if (M->upper[M->sequenceOut] - M->lower[M->sequenceOut] > 0) {
      if (fabs(M->valueOut - M->lower[M->sequenceOut]) < fabs(M->valueOut - M->upper[M->sequenceOut]))
        setStatus(M, M->sequenceOut, ST_LOWER);
      else
        setStatus(M, M->sequenceOut, ST_UPPER);
    } else {
      setStatus(M, M->sequenceOut, ST_FIXED);
    }
    M->sol[M->sequenceOut] = M->valueOut;
  } else {
    if (fabs(M->valueIn - M->lower[M->sequenceIn]) < fabs(M->valueIn - M->upper[M->sequenceIn]))
      setStatus(M, M->sequenceIn, ST_LOWER);
    else
      setStatus(M, M->sequenceIn, ST_UPPER);
  }
  M->objectiveValue += objectiveChange;
  logPivot(M, numberFlipped);
  if (M->numberIterations >= M->maximumIterations)
    return 2;
  /* small cycles (:2397-2431, ClpSimplexProgress::cycle ClpSolve.cpp:4726-4825) */
  {
    int cycle = progressCycle(M, M->sequenceIn, M->sequenceOut, M->directionIn, M->directionOut);
    if (cycle > 0) {
      static const int off[] = { 1, 1, 1, 1, 2, 2, 2, 3, 3, 4 };
      for (int i = 0; i < ORC_CYCLE; i++) {
        M->cycIn[i] = M->cycOut[i] = -1;
        M->cycWay[i] = 0;
      }
      double random = randomDouble(M);
      int extra = (int)(9.999 * random);
      if (M->fac.nEta > cycle) {
        M->forceFactorization = cycle - off[extra] > 1 ? cycle - off[extra] : 1;
      } else {
        M->status[M->sequenceOut] |= 64; /* setFlagged(sequenceOut_) */
      }
      return 1;
    }
  }
  int numberPivots = M->fac.nEta;
  if (numberPivots == M->maximumPivots || M->maximumPivots < 2) {
    return 1;
  } else if (M->forceFactorization > 0 && numberPivots == M->forceFactorization) {
    M->forceFactorization = (3 + 5 * M->forceFactorization) / 4;
    if (M->forceFactorization > M->maximumPivots)
      M->forceFactorization = -1;
    return 1;
  }
  /* the randomised early refactorization after 1000+10*(m+n/4) iterations (:2469-2484) */
  if (M->numberIterations > 1000 + 10 * (M->m + (M->n >> 2))) {
    double random = randomDouble(M);
    while (random < 0.45)
      random *= 2.0;
    int maxNumber = (M->forceFactorization < 0) ? M->maximumPivots : (M->forceFactorization < M->maximumPivots ? M->forceFactorization : M->maximumPivots);
    if (numberPivots >= random * maxNumber)
      return 1;
  }
  return 0;
}

/* unpackPacked: entering column into dense rowWork1 (ClpSimplex::unpackPacked :3439-3495) */
static void unpackColumn(OrcModel *M, double *dense, int iSequence)
{
  if (iSequence >= M->n) {
    dense[iSequence - M->n] = -1.0;
  } else {
    for (int p = M->colStart[iSequence]; p < M->colStart[iSequence + 1]; p++)
      dense[M->row[p]] = M->elem[p];
  }
}

/* ClpSimplexDual::whileIterating :973-2384.  Returns returnCode. */
static int whileIterating(OrcModel *M)
{
  const int m = M->m;
  int returnCode = -1;
  double saveSumDual = M->sumDualInfeasibilities;
  while (M->problemStatus == -1) {
    dualRow(M);
    if (M->pivotRow >= 0) {
      double acceptablePivot = 1.0e-1 * M->acceptablePivot_;
      if (M->numberIterations > 100)
        acceptablePivot = M->acceptablePivot_;
      int pivots = M->fac.nEta;
      if (pivots > 10 || (pivots && saveSumDual))
        acceptablePivot = 1.0e+3 * M->acceptablePivot_;
      else if (pivots > 5)
        acceptablePivot = 1.0e+2 * M->acceptablePivot_;
      else if (pivots)
        acceptablePivot = M->acceptablePivot_;
      double bestPossiblePivot = 1.0;
      /* BTRAN: rho = B^-T (directionOut * e_r) (:1286-1288) */
      double *work = M->rowWork0;
      work[M->pivotRow] = (double)M->directionOut;
      btran(M, work);
      M->numberPi = 0;
      for (int i = 0; i < m; i++) {
        double value = work[i];
        work[i] = 0.0;
        if (fabs(value) > M->zeroTolerance) { /* packed output drops tiny (CoinFactorization BTRAN) */
          M->piIndex[M->numberPi] = i;
          M->piValue[M->numberPi++] = value;
        }
      }
      M->sequenceIn = -1;
      /* row of tableau + first ratio pass (:1300) */
      M->numberColNz = priceRowFused(M, M->numberPi, M->piIndex, M->piValue, M->rowWork1, M->status, M->dj, M->zeroTolerance,
                                     M->dualTolerance, acceptablePivot, M->colIndex, M->colValue, &M->numberCandidates,
                                     M->spareIndex[0], M->spareValue[0], &M->upperThetaFirst);
      M->badFree = 0.0;
      if (!M->noFreeOrSuper)
        dualColumn0General(M, acceptablePivot); /* (the tableau row is the same; only the first pass differs) */
      if (M->sequenceIn >= 0)
        M->numberFreeEntered++;
      bestPossiblePivot = dualColumn(M, acceptablePivot);
      if (M->sequenceIn < 0 && acceptablePivot <= M->acceptablePivot_) {
        if (!M->fac.nEta)
          M->problemStatus = 1;
      }
      if (M->sequenceIn >= 0) {
        double btranAlpha = -M->alpha * M->directionOut;
        unpackColumn(M, M->rowWork1, M->sequenceIn);
        M->alpha = updateWeights(M);
        double checkValue = 1.0e-7;
        if (M->largestPrimalError > 10.0)
          checkValue = dmin(1.0e-4, 1.0e-8 * M->largestPrimalError);
        if (fabs(btranAlpha) < 1.0e-12 || fabs(M->alpha) < 1.0e-12 || fabs(btranAlpha - M->alpha) > checkValue * (1.0 + fabs(M->alpha))) {
          if (M->fac.nEta) {
            unrollWeights(M);
            M->problemStatus = -2;
            M->numberPi = M->numberColNz = M->numberW = 0;
            returnCode = -2;
            break;
          } else {
            double test;
            if (fabs(btranAlpha) < 1.0e-8 || fabs(M->alpha) < 1.0e-8)
              test = 1.0e-1 * fabs(M->alpha);
            else
              test = 1.0e-4 * (1.0 + fabs(M->alpha));
            if (fabs(btranAlpha) < 1.0e-12 || fabs(M->alpha) < 1.0e-12 || fabs(btranAlpha - M->alpha) > test) {
              unrollWeights(M);
              setFlagged(M, M->sequenceOut);
            progressClearBadTimes(M);
              progressClearBadTimes(M);
              M->lastBadIteration = M->numberIterations;
              M->numberPi = M->numberColNz = M->numberW = 0;
              if (fabs(M->alpha) < 1.0e-10 && fabs(btranAlpha) < 1.0e-8 && M->numberIterations > 100) {
                M->problemStatus = 1;
                returnCode = 1;
                break;
              }
              continue;
            }
          }
        }
        double objectiveChange = 0.0;
        int saveStatus = getStatus(M, M->sequenceIn);
        setStatus(M, M->sequenceIn, ST_BASIC);
        double *flipRhs = M->rowWork2;
        int nswapped = updateDualsInDual(M, flipRhs, M->theta, &objectiveChange, 0);
        setStatus(M, M->sequenceIn, saveStatus);
        double oldDualOut = M->dualOut;
        if (nswapped) {
          ftran(M, flipRhs);
          int nf = packDense(flipRhs, m, M->piIndex, M->piValue);
          updatePrimalSolution(M, M->piIndex, M->piValue, nf, 1.0, &objectiveChange);
          M->valueOut = M->sol[M->sequenceOut];
          if (M->directionOut < 0)
            M->dualOut = M->valueOut - M->upperOut;
          else
            M->dualOut = M->lowerOut - M->valueOut;
        }
        double movement = -M->dualOut * M->directionOut / M->alpha;
        double movementOld = oldDualOut * M->directionOut / M->alpha;
        if (objectiveChange + fabs(movementOld * M->dualIn) < -dmax(1.0e-5, 1.0e-12 * fabs(M->objectiveValue))) {
          if (M->fac.nEta) {
            unrollWeights(M);
            M->problemStatus = -2;
            M->numberW = 0;
            M->numberRowFlip = M->numberColFlip = 0;
            returnCode = -2;
            break;
          }
        }
        int updateStatus = replaceColumn(M, M->wIndex, M->wValue, M->numberW, M->pivotRow, M->alpha);
        if (fabs(M->dualOut) > 1.0e50)
          updateStatus = 2;
        if (updateStatus == 2 && !M->fac.nEta && fabs(M->alpha) > 1.0e-5)
          updateStatus = 4;
        if (updateStatus == 1 || updateStatus == 4) {
          if (M->fac.nEta > 5 || updateStatus == 4) {
            M->problemStatus = -2;
            returnCode = -3;
          }
        } else if (updateStatus == 2) {
          unrollWeights(M);
          if (M->fac.nEta) {
            M->problemStatus = -2;
            returnCode = -2;
            M->numberW = 0;
            M->numberRowFlip = M->numberColFlip = 0;
            break;
          } else {
            setFlagged(M, M->sequenceOut);
            M->lastBadIteration = M->numberIterations;
            M->numberW = 0;
            double oc = 0.0;
            memset(M->rowWork2, 0, sizeof(double) * (size_t)m);
            updateDualsInDual(M, M->rowWork2, 0.0, &oc, 1);
            memset(M->rowWork2, 0, sizeof(double) * (size_t)m);
            continue;
          }
        } else if (updateStatus == 3 || updateStatus == 5) {
          M->problemStatus = -2;
        }
        if (M->theta < 0.0)
          M->theta = 0.0;
        int numberFlipped = M->numberRowFlip + M->numberColFlip;
        flipBounds(M);
        updatePrimalSolution(M, M->wIndex, M->wValue, M->numberW, movement, &objectiveChange);
        M->numberW = 0;
        M->dualOut /= M->alpha;
        M->dualOut *= -M->directionOut;
        M->dj[M->sequenceIn] = 0.0;
        double oldValue = M->valueIn;
        if (M->directionIn == -1)
          M->valueIn = M->upperIn + M->dualOut;
        else
          M->valueIn = M->lowerIn + M->dualOut;
        objectiveChange += M->cost[M->sequenceIn] * (M->valueIn - oldValue);
        if (M->directionOut > 0) {
          M->valueOut = M->lowerOut;
          M->dj[M->sequenceOut] = M->theta;
        } else {
          M->valueOut = M->upperOut;
          M->dj[M->sequenceOut] = -M->theta;
        }
        M->sol[M->sequenceOut] = M->valueOut;
        int whatNext = housekeeping(M, objectiveChange, numberFlipped);
        originalBound(M, M->sequenceIn);
        changeBound(M, M->sequenceOut);
        if (whatNext == 1) {
          M->problemStatus = -2;
        } else if (whatNext == 2) {
          M->problemStatus = 3;
          returnCode = 3;
          break;
        }
      } else {
        /* no incoming column is valid (:1869-2079) */
        M->pivotRow = -1;
        M->numberPi = M->numberColNz = 0;
        if (M->fac.nEta < 2 && M->acceptablePivot_ <= 1.0e-8 && M->acceptablePivot_ > 0.0) {
          double dualTest = 1.0e13;
          if (!numberAtFakeBound(M))
            dualTest = 0.0;
          if (bestPossiblePivot < 1.0e-11 && M->dualBound > dualTest) {
            /* "say infeasible ... unless primal feasible!!!!" (:1982-2027, specialOptions_ 0): the sums are those of the last
               status check; with fewer than two pivots since the factorization the -4 alternative (:1999: sumPrimal > 50 and
               more than two pivots) cannot be taken, so a nearly primal feasible or a dual infeasible point ends as 10, "use
               primal".  (The reference also drops the objective there, :2024; a caller of this port sees only the status.) */
            M->problemStatus = 1;
            if (M->sumPrimalInfeasibilities < 1.0e-3 || M->sumDualInfeasibilities > 1.0e-5)
              M->problemStatus = 10;
            returnCode = 1;
            break;
          }
          if (M->fac.nEta == 0)
            M->problemStatus = -4;
        }
        M->acceptablePivot_ = fabs(M->acceptablePivot_);
        if (M->fac.nEta < 5 && M->acceptablePivot_ > 1.0e-8)
          M->acceptablePivot_ = 1.0e-8;
        returnCode = 1;
        break;
      }
    } else {
      /* no pivot row (:2080-2331) */
      int numberPivots = M->fac.nEta;
      returnCode = 0;
      if (!numberPivots) {
        if (M->numberPrimalInfeasibilities && M->problemStatus == -1)
          M->problemStatus = -4;
        int iRow;
        for (iRow = 0; iRow < m; iRow++)
          if (flagged(M, M->pivotVariable[iRow]))
            break;
        if (M->numberFake || M->numberDualInfeasibilities) {
          M->problemStatus = -5;
        } else {
          if (iRow < m) {
            M->problemStatus = -5;
          } else {
            M->problemStatus = 0;
            M->numberPrimalInfeasibilities = 0;
            M->sumPrimalInfeasibilities = 0.0;
            M->numberDualInfeasibilities = 0;
            M->sumDualInfeasibilities = 0.0;
            if (M->perturbation == 101 || M->numberChanged) {
              /* costs were perturbed or modified: restore (createRim4) and recheck (:2222-2236) */
              M->numberChanged = 0;
              M->perturbation = 102; /* stop any perturbations */
              restoreCosts(M);
              computeDuals(M);
              progressModifyObjective(M, -DBL_MAX);
              checkDualSolution(M);
              if (M->numberDualInfeasibilities)
                M->problemStatus = 10;
              else
                checkPrimalSolution(M); /* computeObjectiveValue */
            }
          }
        }
      } else {
        M->problemStatus = -3;
        returnCode = -2;
        int half = (numberPivots + 1) >> 1;
        if (M->forceFactorization < 0 || half < M->forceFactorization)
          M->forceFactorization = half;
      }
      break;
    }
  }
  return returnCode;
}

/* ------------------------------------------------------------------------------------------ */
/* ClpSimplexProgress for the dual (src/ClpSolve.cpp:4289-4725): what happened at the last        */
/* ORC_PROGRESS status checks.  Slot ORC_PROGRESS-1 is the newest.                                 */
/* ------------------------------------------------------------------------------------------ */
static void progressReset(OrcModel *M) /* ClpSimplexProgress::reset :4613, algorithm_ < 0 */
{
  for (int i = 0; i < ORC_PROGRESS; i++) {
    M->progObjective[i] = -DBL_MAX * 1.0e-50;
    M->progInfeasibility[i] = -1.0; /* an impossible value */
    M->progNumberInfeasibilities[i] = -1;
    M->progIteration[i] = -1;
  }
  M->progTimes = M->progBadTimes = M->progReallyBadTimes = M->progTimesFlagged = 0;
}
static void progressStartCheck(OrcModel *M) /* :4715 */
{
  for (int i = 0; i < ORC_CYCLE; i++) {
    M->cycIn[i] = M->cycOut[i] = -1;
    M->cycWay[i] = 0;
  }
}
static int sameBits(double a, double b) /* equalDouble :4424 */
{
  return memcmp(&a, &b, sizeof(double)) == 0;
}
static void resetFakeBounds0(OrcModel *M);

/* ClpSimplexProgress::looping :4438-4611 for algorithm_ < 0.  Returns -1 carry on, -2 something was changed
 * (tolerance / dual bound / a variable flagged), 0 "declare victory", 3 / 4 give up. */
static int progressLooping(OrcModel *M)
{
  const double objective = M->objectiveValue - M->bestPossibleImprovement;
  const double infeasibility = M->sumPrimalInfeasibilities;
  const int numberInfeasibilities = M->numberPrimalInfeasibilities;
  const int iterationNumber = M->numberIterations;
  int numberMatched = 0, matched = 0, nsame = 0;
  for (int i = 0; i < ORC_PROGRESS; i++) {
    if (sameBits(objective, M->progObjective[i]) && sameBits(infeasibility, M->progInfeasibility[i])
        && numberInfeasibilities == M->progNumberInfeasibilities[i]) {
      matched |= (1 << i);
      if (iterationNumber != M->progIteration[i])
        numberMatched++; /* not the same iteration */
      else
        nsame++; /* stuck but code should notice */
    }
    if (i) {
      M->progObjective[i - 1] = M->progObjective[i];
      M->progInfeasibility[i - 1] = M->progInfeasibility[i];
      M->progNumberInfeasibilities[i - 1] = M->progNumberInfeasibilities[i];
      M->progIteration[i - 1] = M->progIteration[i];
    }
  }
  M->progObjective[ORC_PROGRESS - 1] = objective;
  M->progInfeasibility[ORC_PROGRESS - 1] = infeasibility;
  M->progNumberInfeasibilities[ORC_PROGRESS - 1] = numberInfeasibilities;
  M->progIteration[ORC_PROGRESS - 1] = iterationNumber;
  if (nsame == ORC_PROGRESS)
    numberMatched = ORC_PROGRESS; /* really stuck */
  if (M->progressFlag & 3)
    numberMatched = 0;
  M->progTimes++;
  if (M->progTimes < 10)
    numberMatched = 0;
  if (matched == (1 << (ORC_PROGRESS - 1)))
    numberMatched = 0; /* just last time: may be checking something */
  if (!numberMatched)
    return -1;
  M->numberLoopFlags++;
  M->progBadTimes++;
  if (M->progBadTimes < 10) {
    M->forceFactorization = 1; /* factorize every iteration */
    if (M->progBadTimes < 2) {
      progressStartCheck(M); /* clear other loop check */
      M->dualTolerance *= 1.05;
      if (M->dualBound < 1.0e17) { /* if infeasible increase dual bound */
        M->dualBound *= 1.1;
        resetFakeBounds0(M);
      }
    } else {
      if (M->dualBound > 1.0e14)
        M->dualBound = 1.0e14;
      int iSequence = M->cycIn[ORC_CYCLE - 1];
      if (iSequence >= 0) {
        setFlagged(M, iSequence);
        progressStartCheck(M);
      } else {
        return 4; /* all flagged? give up */
      }
      M->progBadTimes = 2;
    }
    return -2;
  }
  /* look at solution and maybe declare victory */
  return infeasibility < 1.0e-4 ? 0 : 3;
}

static int compareDoubles(const void *a, const void *b)
{
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

/* ClpSimplexDual::perturb :6533-6957 -- cost perturbation of the nonbasic, non-fixed structurals.
 * Restated for the values dual() can be entered with here: 50 (the clp default: perturb at start-up when at
 * most a quarter of the |costs| are distinct), 51-69 (fixed maximum fractions), 100 (no start-up
 * perturbation, the "kick" after 2(m+n) iterations), and below 50 "user is in charge" (10^value) without the
 * <= -10 experiments.  Row costs are never modified (modifyRowCosts is forced false at :6745), the Cbc
 * branches are out.
 * Returns 1 when the reference would rather use primal (all costs zero, :6588). */
static int perturb(OrcModel *M)
{
  const int m = M->m, n = M->n;
  if (M->perturbation > 100)
    return 0; /* perturbed already */
  if (M->perturbation == 100)
    M->perturbation = 50; /* treat as normal */
  const int savePerturbation = M->perturbation;
  double perturbation = 1.0e-20;
  double maximumFraction = 1.0e-5; /* maximum fraction of cost to perturb */
  const double constantPerturbation = 100.0 * M->dualTolerance;
  int maxLength = 0, minLength = m;
  double averageCost = 0.0;
  int numberNonZero = 0;
  if (!M->numberIterations && M->perturbation >= 50) {
    /* see if we need to perturb (:6562-6606) */
    double *sort = DALLOC(n > 0 ? n : 1);
    for (int i = 0; i < n; i++) {
      double value = fabs(M->objBeforeScaling[i]);
      sort[i] = value;
      averageCost += value;
      if (value)
        numberNonZero++;
    }
    if (numberNonZero)
      averageCost /= (double)numberNonZero;
    else
      averageCost = 1.0;
    qsort(sort, (size_t)n, sizeof(double), compareDoubles);
    int number = 1;
    double last = n ? sort[0] : 0.0;
    for (int i = 1; i < n; i++) {
      if (last != sort[i])
        number++;
      last = sort[i];
    }
    free(sort);
    if (!numberNonZero && M->perturbation < 55)
      return 1; /* safer to use primal */
    if (number * 4 > n) {
      M->perturbation = 100;
      return 0; /* good enough */
    }
  }
  for (int j = 0; j < n; j++) {
    if (M->lower[j] < M->upper[j]) {
      int length = M->colStart[j + 1] - M->colStart[j];
      if (length > 2) {
        if (length > maxLength)
          maxLength = length;
        if (length < minLength)
          minLength = length;
      }
    }
  }
  if (M->perturbation >= 70)
    M->perturbation -= 20; /* "do rows" -- but row costs are left alone, :6745 */
  if (M->perturbation > 50) {
    static const double fractions[] = { 1.0e-10, 1.0e-9, 1.0e-8, 1.0e-7, 1.0e-6, 1.0e-5, 1.0e-4, 1.0e-3, 1.0e-2, 1.0e-1, 1.0 };
    int whichOne = M->perturbation - 51;
    maximumFraction = fractions[whichOne < 10 ? whichOne : 10];
  }
  double smallestNonZero = 1.0e100;
  if (M->perturbation >= 50) {
    perturbation = 1.0e-8;
    if (M->perturbation > 50 && M->perturbation < 60)
      perturbation = dmax(1.0e-8, maximumFraction);
    int allSame = 1;
    double lastValue = 0.0;
    for (int i = 0; i < m; i++) {
      double lo = M->lower[n + i], up = M->upper[n + i];
      if (lo < up) {
        double value = fabs(M->cost[n + i]);
        perturbation = dmax(perturbation, value);
        if (value)
          smallestNonZero = dmin(smallestNonZero, value);
      }
      if (lo && lo > -1.0e10) {
        lo = fabs(lo);
        if (!lastValue)
          lastValue = lo;
        else if (fabs(lo - lastValue) > 1.0e-7)
          allSame = 0;
      }
      if (up && up < 1.0e10) {
        up = fabs(up);
        if (!lastValue)
          lastValue = up;
        else if (fabs(up - lastValue) > 1.0e-7)
          allSame = 0;
      }
    }
    double lastValue2 = 0.0;
    for (int j = 0; j < n; j++) {
      double lo = M->lower[j], up = M->upper[j];
      if (lo < up) {
        double value = fabs(M->cost[j]);
        perturbation = dmax(perturbation, value);
        if (value)
          smallestNonZero = dmin(smallestNonZero, value);
      }
      if (lo && lo > -1.0e10) {
        lo = fabs(lo);
        if (!lastValue2)
          lastValue2 = lo;
        else if (fabs(lo - lastValue2) > 1.0e-7)
          allSame = 0;
      }
      if (up && up < 1.0e10) {
        up = fabs(up);
        if (!lastValue2)
          lastValue2 = up;
        else if (fabs(up - lastValue2) > 1.0e-7)
          allSame = 0;
      }
    }
    if (allSame) {
      /* ClpPackedMatrix::rangeOfElements, src/ClpPackedMatrix.cpp:5229 */
      double smallestNegative = -DBL_MAX, largestNegative = 0.0, smallestPositive = DBL_MAX, largestPositive = 0.0;
      for (int p = 0; p < M->colStart[n]; p++) {
        double value = M->elem[p];
        if (value > 0.0) {
          smallestPositive = dmin(smallestPositive, value);
          largestPositive = dmax(largestPositive, value);
        } else if (value < 0.0) {
          smallestNegative = dmax(smallestNegative, value);
          largestNegative = dmin(largestNegative, value);
        }
      }
      if (smallestNegative == largestNegative && smallestPositive == largestPositive) {
        /* really hit perturbation */
        double adjust = dmin(100.0 * maximumFraction, 1.0e-3 * dmax(lastValue, lastValue2));
        maximumFraction = dmax(adjust, maximumFraction);
      }
    }
    perturbation = dmin(perturbation, smallestNonZero / maximumFraction);
  } else {
    /* user is in charge */
    maximumFraction = 1.0e-1;
    perturbation = pow(10.0, (double)M->perturbation);
  }
  double largestZero = 0.0, largest = 0.0;
  static const double weight[] = { 1.0e-4, 1.0e-2, 5.0e-1, 1.0, 2.0, 5.0, 10.0, 20.0, 30.0, 40.0, 100.0 };
  /* constantPerturbation is 100 x dualTolerance here, so the "scale back" of :6784 never applies */
  double factor = 1.0;
  if (maxLength)
    factor = 3.0 / (double)minLength;
  const double m1 = 0.5;
  const double smallestAllowed = dmin(1.0e-2 * M->dualTolerance, maximumFraction);
  double largestAllowed = dmax(1.0e3 * M->dualTolerance, maximumFraction * averageCost);
  if (M->perturbation == 51)
    largestAllowed = dmax(M->dualTolerance, maximumFraction);
  if (!M->perturbationArray) {
    M->perturbationArray = DALLOC(2 * n + 1);
    for (int j = 0; j < 2 * n; j++)
      M->perturbationArray[j] = randomDouble(M);
  }
  for (int j = 0; j < n; j++) {
    if (M->lower[j] < M->upper[j] && getStatus(M, j) != ST_BASIC) {
      double value = perturbation;
      const double currentValue = M->cost[j];
      value = dmin(value, constantPerturbation + maximumFraction * (fabs(currentValue) + 1.0e-1 * perturbation + 1.0e-8));
      double value2 = constantPerturbation + 1.0e-1 * smallestNonZero;
      if (M->lower[j] > -M->largeValue) {
        if (fabs(M->lower[j]) < fabs(M->upper[j])) {
          value *= (1.0 - m1 + m1 * M->perturbationArray[2 * j]);
          value2 *= (1.0 - m1 + m1 * M->perturbationArray[2 * j + 1]);
        } else {
          value = 0.0;
        }
      } else if (M->upper[j] < M->largeValue) {
        value *= -(1.0 - m1 + m1 * M->perturbationArray[2 * j]);
        value2 *= -(1.0 - m1 + m1 * M->perturbationArray[2 * j + 1]);
      } else {
        value = 0.0;
      }
      if (value) {
        int length = M->colStart[j + 1] - M->colStart[j];
        if (length > 3) {
          length = (int)((double)length * factor);
          if (length < 3)
            length = 3;
        }
        value *= (length < 10) ? weight[length] : weight[10];
        value = dmin(value, value2);
        if (savePerturbation < 50 || savePerturbation > 60) {
          if (fabs(value) <= M->dualTolerance)
            value = 0.0;
        } else if (value) {
          /* get in range */
          if (fabs(value) <= smallestAllowed) {
            value *= 10.0;
            while (fabs(value) <= smallestAllowed)
              value *= 10.0;
          } else if (fabs(value) > largestAllowed) {
            value *= 0.1;
            while (fabs(value) > largestAllowed)
              value *= 0.1;
          }
        }
        if (currentValue)
          largest = dmax(largest, fabs(value));
        else
          largestZero = dmax(largestZero, fabs(value));
        /* but negative if at ub */
        if (getStatus(M, j) == ST_UPPER)
          value = -value;
        M->cost[j] += value;
      }
    }
  }
  if (largestZero > 1.0 * largest && largest) {
    /* the perturbation of a zero cost must not dwarf those of the others (:6902-6917) */
    double test = dmax(1.0e-8, largest);
    for (int j = 0; j < n; j++) {
      if (!M->objBeforeScaling[j]) {
        double cost = M->cost[j];
        while (fabs(cost) > test)
          cost *= 0.5;
        M->cost[j] = cost;
      }
    }
  }
  M->perturbation = 101; /* say perturbed */
  M->numberPerturbations++;
  return 0;
}

/* createRim4(false): the original costs back (src/ClpSimplex.cpp:4645) */
static void restoreCosts(OrcModel *M)
{
  for (int j = 0; j < M->n; j++)
    M->cost[j] = M->obj[j];
  for (int i = 0; i < M->m; i++)
    M->cost[M->n + i] = 0.0;
}

/* ClpSimplexDual::resetFakeBounds(0) :8303-8309: original bounds back (createRim1), then the fake ones again */
static void resetFakeBounds0(OrcModel *M)
{
  const int N = M->m + M->n;
  for (int i = 0; i < N; i++) {
    M->lower[i] = originalLower(M, i);
    M->upper[i] = originalUpper(M, i);
  }
  double dummy = 0.0;
  changeBounds(M, 3, NULL, &dummy);
}

/* ClpSimplexDual::resetFakeBounds(type > 0) :8487-8600: original bounds back, then every variable that carries a fake
 * status gets the bound that status stands for again (and the value that goes with its nonbasic status) */
static void resetFakeBounds1(OrcModel *M)
{
  const int N = M->m + M->n;
  for (int i = 0; i < N; i++) {
    M->lower[i] = originalLower(M, i);
    M->upper[i] = originalUpper(M, i);
  }
  M->numberFake = 0;
  for (int i = 0; i < N; i++) {
    const int fakeStatus = getFake(M, i);
    if (fakeStatus == FAKE_NONE)
      continue;
    const int st = getStatus(M, i);
    if (st == ST_BASIC || st == ST_FIXED) {
      setFake(M, i, FAKE_NONE);
      continue;
    }
    const double lowerValue = M->lower[i], upperValue = M->upper[i], value = M->sol[i];
    M->numberFake++;
    if (fakeStatus == FAKE_UPPER) {
      M->upper[i] = lowerValue + M->dualBound;
      M->sol[i] = (st == ST_LOWER) ? lowerValue : M->upper[i];
    } else if (fakeStatus == FAKE_LOWER) {
      M->lower[i] = upperValue - M->dualBound;
      M->sol[i] = (st == ST_LOWER) ? M->lower[i] : upperValue;
    } else if (st == ST_LOWER) {
      M->lower[i] = value;
      M->upper[i] = value + M->dualBound;
    } else if (st == ST_UPPER) {
      M->upper[i] = value;
      M->lower[i] = value - M->dualBound;
    } else { /* isFree / superBasic */
      M->lower[i] = value - 0.5 * M->dualBound;
      M->upper[i] = value + 0.5 * M->dualBound;
    }
  }
}

/* ClpDualRowSteepest::looksOptimal (src/ClpDualRowSteepest.cpp:1070); the base class (Dantzig) says no */
static int looksOptimal(const OrcModel *M)
{
  if (M->pivotRule == 0)
    return 0;
  double tolerance = M->primalTolerance + dmin(1.0e-2, M->largestPrimalError);
  tolerance = dmin(1000.0, tolerance);
  int numberInfeasible = 0;
  for (int iRow = 0; iRow < M->m; iRow++) {
    int iPivot = M->pivotVariable[iRow];
    double value = M->sol[iPivot];
    if (value < M->lower[iPivot] - tolerance)
      numberInfeasible++;
    else if (value > M->upper[iPivot] + tolerance)
      numberInfeasible++;
  }
  return numberInfeasible == 0;
}

/* ClpSimplexDual::statusOfProblemInDual :4996-6343, the parts that matter without values pass, Cbc
 * options or primal fallback (status 10 is returned to the caller). */
static void statusOfProblemInDual(OrcModel *M, int *lastCleaned, int type)
{
  const int m = M->m;
  int numberPivots = M->fac.nEta;
  int tentativeStatus = M->problemStatus;
  int weightsSaved = 0;
  int unflagVariables = 1, reallyBadProblems = 0;
  double changeCost = 0.0;
  if (M->problemStatus > -3 || numberPivots > 0) {
    saveWeights(M, 1);
    weightsSaved = 1;
    if (type) {
      int rc = factorize(M);
      if (!rc && M->debugSingularAt >= 0 && M->numberIterations >= M->debugSingularAt && M->numberIterations > 0) {
        rc = 1; /* fault injection */
        M->debugSingularAt = -1;
      }
      if (rc) {
        /* is factorization okay?  no - restore previous basis (:5061-5125) */
        M->numberSingularRestores++;
        unflagVariables = 0;
        for (int i = 0; i < m + M->n; i++)
          if (flagged(M, i))
            M->saveStatus[i] |= FLAGGED_BIT; /* keep any flagged variables */
        memcpy(M->status, M->saveStatus, (size_t)(m + M->n));
        memcpy(M->sol, M->savedSolution, sizeof(double) * (size_t)(m + M->n));
        resetFakeBounds1(M); /* get correct bounds on all variables */
        setFlagged(M, M->sequenceOut); /* need to reject something */
        progressClearBadTimes(M);
        M->forceFactorization = 1; /* a bit drastic but .. */
        type = 2;
        if (factorize(M)) {
          M->problemStatus = 4; /* the saved basis is singular too: the reference goes on to a safe factorization with slacks */
          return;
        }
      }
    }
    if (M->problemStatus != -4 || numberPivots > 10)
      M->problemStatus = -3;
  }
  if (M->progInfeasibility[0] < 1.0e-1 && M->primalTolerance == 1.0e-7 && M->progIteration[0] > 0
      && M->progIteration[ORC_PROGRESS - 1] - M->progIteration[0] > 25) {
    /* the default primal tolerance (so the user did not set it) is loosened when the last checks all show tiny
       infeasibilities (:5136-5160) */
    int iP;
    double minAverage = DBL_MAX, maxAverage = 0.0;
    for (iP = 0; iP < ORC_PROGRESS; iP++) {
      int count = M->progNumberInfeasibilities[iP];
      if (!count)
        break;
      double average = M->progInfeasibility[iP];
      if (average > 0.1)
        break;
      average /= (double)count;
      minAverage = dmin(minAverage, average);
      maxAverage = dmax(maxAverage, average);
    }
    if (iP == ORC_PROGRESS && minAverage < 1.0e-5 && maxAverage < 1.0e-3)
      M->primalTolerance = 1.0e-6;
  }
  if (type)
    gutsOfSolution(M);
  if (M->debugBadAccuracyAt >= 0 && M->numberIterations >= M->debugBadAccuracyAt && M->numberIterations > 0) {
    M->largestPrimalError = 1.0e16; /* fault injection */
    M->debugBadAccuracyAt = -1;
  }
  if ((M->largestPrimalError > 1.0e15 || M->largestDualError > 1.0e15) && M->numberIterations) {
    /* bad accuracy: treat as singular -- back to the previous basis, reject a variable (:5237-5318) */
    M->numberAccuracyRestores++;
    unflagVariables = 0;
    for (int i = 0; i < m + M->n; i++)
      if (flagged(M, i))
        M->saveStatus[i] |= FLAGGED_BIT; /* keep any flagged variables */
    memcpy(M->status, M->saveStatus, (size_t)(m + M->n));
    memcpy(M->sol, M->savedSolution, sizeof(double) * (size_t)(m + M->n));
    resetFakeBounds1(M); /* get correct bounds on all variables */
    int rejectedVariable = M->sequenceOut;
    if (rejectedVariable < 0 || flagged(M, rejectedVariable)) {
      rejectedVariable = -1;
      for (int i = 0; i < m; i++) {
        int iSequence = M->pivotVariable[i];
        if (!flagged(M, iSequence)) {
          rejectedVariable = iSequence;
          break;
        }
      }
      if (rejectedVariable < 0) {
        M->problemStatus = 10; /* real trouble */
        return;
      }
    }
    setFlagged(M, rejectedVariable);
    progressClearBadTimes(M);
    M->forceFactorization = 1; /* a bit drastic but .. */
    type = 2;
    if (factorize(M)) {
      M->problemStatus = 4;
      return;
    }
    gutsOfSolution(M);
  }
  if (progressLastIteration(M, 0) == M->numberIterations) {
    /* double check infeasibility if no action (:5326-5330) */
    if (looksOptimal(M)) {
      M->numberPrimalInfeasibilities = 0;
      M->sumPrimalInfeasibilities = 0.0;
    }
  } else {
    /* has the objective gone backwards since the last check? (:5332-5488) */
    const double thisObj = M->objectiveValue - M->bestPossibleImprovement;
    double lastObj = progressLastObjective(M, 0);
    double testTol = 5.0e-3;
    if (M->progTimesFlagged > 10)
      testTol *= pow(2.0, M->progTimesFlagged - 8);
    else if (M->progTimesFlagged > 5)
      testTol *= 5.0;
    if (M->debugBackwardsAt >= 0 && M->numberIterations >= M->debugBackwardsAt && M->numberIterations > 0) {
      /* fault injection: two checks in a row see a drop, the first small (costs saved), the second large (restore) */
      lastObj = thisObj + ((M->progressFlag & 4) ? 2.0e4 : 1.0) * (testTol * 4.0 * (fabs(thisObj) + 1.0) + 1.0);
      if (M->progressFlag & 4)
        M->debugBackwardsAt = -1;
    }
    if (M->firstFree < 0 /* :5360 */ && lastObj > thisObj + testTol * (fabs(thisObj) + fabs(lastObj)) + testTol) {
      if (M->progTimesFlagged > 10)
        M->progReallyBadTimes++;
      if (M->maximumPivots > 1) {
        if ((M->progressFlag & 4) == 0 && lastObj < thisObj + 1.0e4 && M->largestPrimalError < 1.0e2) {
          /* just save costs */
          memcpy(M->costCopy, M->cost, sizeof(double) * (size_t)(m + M->n));
          M->progressFlag |= 4;
        } else {
          /* back to the basis of the last good check, refactorize every iteration */
          M->numberBackwards++;
          M->forceFactorization = 1;
          unflagVariables = 0;
          memcpy(M->status, M->saveStatus, (size_t)(m + M->n));
          memcpy(M->sol, M->savedSolution, sizeof(double) * (size_t)(m + M->n));
          if ((M->progressFlag & 4) == 0) {
            memcpy(M->costCopy, M->cost, sizeof(double) * (size_t)(m + M->n));
            M->progressFlag |= 4;
          } else {
            memcpy(M->cost, M->costCopy, sizeof(double) * (size_t)(m + M->n));
          }
          if (factorize(M)) { /* internalFactorize(1); the saved basis factorized before */
            M->problemStatus = 4;
            return;
          }
          resetFakeBounds0(M);
          type = 2; /* so will restore weights */
          gutsOfSolution(M);
          if (numberPivots < 2) {
            /* need to reject something */
            setFlagged(M, M->sequenceOut);
            progressClearBadTimes(M);
            M->progTimesFlagged++;
          }
          if (numberPivots < 10)
            reallyBadProblems = 1;
          progressModifyObjective(M, M->objectiveValue - M->bestPossibleImprovement);
        }
      }
    } else if (lastObj < thisObj - 1.0e-5 * dmax(fabs(thisObj), fabs(lastObj)) - 1.0e-3) {
      M->numberTimesOptimal = 0;
    }
  }
  /* check if looping (:5506-5536) */
  int loop = (type != 2) ? progressLooping(M) : -1;
  if (M->progReallyBadTimes > 10)
    M->problemStatus = 10; /* instead - try other algorithm */
  int situationChanged = 0;
  if (loop >= 0) {
    M->problemStatus = loop; /* exit if in loop */
    if (!M->problemStatus) {
      /* declaring victory */
      M->numberPrimalInfeasibilities = 0;
      M->sumPrimalInfeasibilities = 0.0;
    } else if (M->problemStatus != 3) {
      M->problemStatus = 10;
    }
    return;
  } else if (loop < -1) {
    /* something may have changed */
    gutsOfSolution(M);
    situationChanged = 1;
  }
  if (M->progressFlag & 2)
    situationChanged = 2; /* really for free variables in */
  M->progressFlag &= ~3;
  if (M->progressFlag & 4)
    memcpy(M->costCopy, M->cost, sizeof(double) * (size_t)(m + M->n)); /* save copy of cost_ (:5543-5547) */
  if (!M->numberPrimalInfeasibilities && !M->numberDualInfeasibilities)
    M->progressFlag |= 8; /* mark as having gone optimal if looks like it */
  /* if we are primal feasible and any dual infeasibilities are on free variables then it is better to go to primal (:5619-5622) */
  if (M->freeNonbasic && !M->numberPrimalInfeasibilities && !M->numberDualInfeasibilitiesWithoutFree && M->numberDualInfeasibilities) {
    M->problemStatus = 10;
    if (M->logLevel > 3)
      fprintf(stderr, "orc: iteration %d primal feasible, the %d dual infeasibilities are all on free variables: 10\n", M->numberIterations,
              M->numberDualInfeasibilities);
  }
  int needCleanFake = 0;
  double saveDualBound = M->dualBound;
  while (M->problemStatus <= -3 && saveDualBound == M->dualBound) {
    int cleanDuals = 0;
    if (situationChanged != 0)
      cleanDuals = 1;
    int numberChangedBounds = 0;
    int doOriginalTolerance = 0;
    if (*lastCleaned == M->numberIterations)
      doOriginalTolerance = 1;
    if (M->sumOfRelaxedDualInfeasibilities == 0.0 && M->sumOfRelaxedPrimalInfeasibilities == 0.0) {
      M->numberDualInfeasibilities = 0;
      M->sumDualInfeasibilities = 0.0;
      M->numberPrimalInfeasibilities = 0;
      M->sumPrimalInfeasibilities = 0.0;
    }
    if (M->numberDualInfeasibilities == 0 || M->problemStatus == -4) {
      progressModifyObjective(M, M->objectiveValue - M->bestPossibleImprovement); /* :5645 */
      if (M->numberPrimalInfeasibilities == 0) {
        /* may be optimal - or may be bounds are wrong (:5689-5764) */
        memset(M->rowWork3, 0, sizeof(double) * (size_t)m);
        numberChangedBounds = (M->dualBound < 1.0e20) ? changeBounds(M, 0, M->rowWork3, &changeCost) : 0;
        memset(M->rowWork3, 0, sizeof(double) * (size_t)m);
        if (numberChangedBounds <= 0 && !M->numberDualInfeasibilities) {
          if (M->perturbation == 101) {
            /* looks optimal with perturbed costs: the true costs back and look again (:5708-5733) */
            M->perturbation = 102; /* stop any perturbations */
            cleanDuals = 1;
            changeBounds(M, 1, NULL, &changeCost); /* make sure fake bounds are back */
            restoreCosts(M);
            computeDuals(M); /* make sure duals are current */
            checkDualSolution(M);
            progressModifyObjective(M, -DBL_MAX);
            if (M->numberDualInfeasibilities) {
              M->numberChanged = 1; /* force something to happen */
              *lastCleaned = M->numberIterations - 1;
            } else {
              checkPrimalSolution(M); /* computeObjectiveValue(true) */
            }
          }
          if (*lastCleaned < M->numberIterations && M->numberTimesOptimal < 4) {
            doOriginalTolerance = 2;
            M->numberTimesOptimal++;
            if (M->numberTimesOptimal == 1) {
              M->dualTolerance = M->dualToleranceBase;
            } else {
              M->dualTolerance = M->dualToleranceBase * pow(2.0, M->numberTimesOptimal - 1);
            }
            cleanDuals = 2; /* if nothing changed optimal else primal */
          } else {
            M->problemStatus = 0; /* optimal */
          }
        } else {
          cleanDuals = 1;
          if (doOriginalTolerance == 1) {
            /* checkUnbounded path (:5766-5826): without free variables we only get here with
               genuinely active fake bounds -> dual infeasible if the bound is already huge */
            if (M->dualBound > 1.0e17)
              M->problemStatus = 2;
            else
              M->problemStatus = -3;
            if (M->problemStatus == 2 && M->perturbation == 101) {
              /* unbounded only for the perturbed costs? (:5814-5820) */
              M->perturbation = 102;
              cleanDuals = 1;
              restoreCosts(M);
              progressModifyObjective(M, -DBL_MAX);
              M->problemStatus = -1;
            }
          } else {
            doOriginalTolerance = 2;
          }
        }
      }
      if (M->problemStatus == -4 || M->problemStatus == -5) {
        numberChangedBounds = changeBounds(M, 0, NULL, &changeCost);
        needCleanFake = 1;
        if ((numberChangedBounds <= 0 || M->dualBound > 1.0e20 || (M->largestPrimalError > 1.0 && M->dualBound > 1.0e17))
            && (numberPivots < 4 || M->sumPrimalInfeasibilities > 1.0e-6)) {
          M->problemStatus = 1; /* infeasible */
          if (M->perturbation == 101)
            M->perturbation = 102; /* stop any perturbations (:5860) */
          if (!M->numberPrimalInfeasibilities) {
            M->problemStatus = -1;
            doOriginalTolerance = 2;
          }
        } else {
          M->problemStatus = -1;
          cleanDuals = 1;
          if (numberChangedBounds <= 0)
            doOriginalTolerance = 2;
        }
      }
    } else {
      cleanDuals = 1;
    }
    if (M->problemStatus < 0) {
      if (doOriginalTolerance == 2) {
        *lastCleaned = M->numberIterations;
        M->numberChanged = 0;
        M->perturbation = 102; /* stop any perturbations (:5891) */
        restoreCosts(M);
        progressModifyObjective(M, -DBL_MAX);
        computeDuals(M);
        checkDualSolution(M);
        if (cleanDuals != 2) {
          changeBounds(M, 3, NULL, &changeCost);
          needCleanFake = 1;
          cleanDuals = 2;
        }
      }
      if (cleanDuals == 1 || (cleanDuals == 2 && !M->numberDualInfeasibilities)) {
        double objectiveChange = 0.0;
        memset(M->rowWork2, 0, sizeof(double) * (size_t)m);
        updateDualsInDual(M, M->rowWork2, 0.0, &objectiveChange, 1);
        memset(M->rowWork2, 0, sizeof(double) * (size_t)m);
        gutsOfSolution(M);
        updateDualsInDual(M, M->rowWork2, 0.0, &objectiveChange, 1);
        memset(M->rowWork2, 0, sizeof(double) * (size_t)m);
        if (M->numberDualInfeasibilities) {
          if ((M->numberPrimalInfeasibilities || numberPivots) && M->problemStatus != 10)
            M->problemStatus = -1;
          else
            M->problemStatus = 10;
        } else if (situationChanged == 2) {
          M->problemStatus = -1;
          changeBounds(M, 3, NULL, &changeCost);
        }
        situationChanged = 0;
      } else {
        if (cleanDuals != 2)
          M->problemStatus = -1;
        else
          M->problemStatus = 10; /* try primal */
      }
    }
  }
  if (tentativeStatus != -2 && tentativeStatus != -1 && unflagVariables) {
    /* unflag (:6079-6120) */
    int numberFlagged = 0;
    for (int iRow = 0; iRow < m; iRow++) {
      int iPivot = M->pivotVariable[iRow];
      if (flagged(M, iPivot)) {
        numberFlagged++;
        clearFlagged(M, iPivot);
      }
    }
    if (numberFlagged && !numberPivots) {
      if (M->numberTimesOptimal < 3) {
        M->numberTimesOptimal++;
        M->problemStatus = -1;
      } else {
        M->problemStatus = 10;
      }
    }
  }
  if (M->problemStatus < 0) {
    if (needCleanFake) {
      double dummy = 0.0;
      changeBounds(M, 3, NULL, &dummy);
    }
    if (type == 0 || type == 1) {
      /* the basis just checked is the one to come back to (saveStatus_ / savedSolution_, :6160-6175) */
      memcpy(M->saveStatus, M->status, (size_t)(m + M->n));
      memcpy(M->savedSolution, M->sol, sizeof(double) * (size_t)(m + M->n));
    }
    if (weightsSaved) {
      if (!reallyBadProblems && (M->largestPrimalError < 100.0 || numberPivots > 10)) {
        if (tentativeStatus > -3)
          saveWeights(M, (type < 2) ? 2 : 4);
        else
          saveWeights(M, 3);
      } else {
        saveWeights(M, 6); /* reset weights or scale back */
      }
    }
  }
  {
    /* refactorize more often when the recorded objective fell between the last two checks (:6316-6328) */
    const double thisObj = progressLastObjective(M, 0), lastObj = progressLastObjective(M, 1);
    if (lastObj > thisObj + 1.0e-4 * dmax(fabs(thisObj), fabs(lastObj)) + 1.0e-4 && M->firstFree < 0 /* :6319 */) {
      if (M->maximumPivots > 10) {
        if (M->forceFactorization < 0)
          M->forceFactorization = M->maximumPivots;
        M->forceFactorization = (M->forceFactorization >> 1) > 1 ? (M->forceFactorization >> 1) : 1;
      }
    }
  }
  if (M->problemStatus == 1 && (M->progressFlag & 8) != 0 && fabs(M->objectiveValue) > 1.0e10)
    M->problemStatus = 10; /* infeasible - but has looked feasible (:6338) */
}

/* ClpSimplexDual::dual :637 -> startupSolve :230 -> gutsOfDual :432 */
static int dualOnRim(OrcModel *M)
{
  const int m = M->m, n = M->n, N = m + n;
  struct timespec t0, t1;
  /* createRim: bounds, costs, solution; slack basis when no status given (allSlackBasis :7831) */
  for (int j = 0; j < n; j++) {
    M->lower[j] = M->colLower[j];
    M->upper[j] = M->colUpper[j];
    M->cost[j] = M->obj[j];
  }
  for (int i = 0; i < m; i++) {
    M->lower[n + i] = M->rowLower[i];
    M->upper[n + i] = M->rowUpper[i];
    M->cost[n + i] = 0.0;
  }
  {
    /* ClpSimplex::sanityCheck (src/ClpSimplex.cpp:7645-7790), the bound part, as createRim(63) runs it at start-up (:4270):
       bounds that cross by more than the primal tolerance make the problem infeasible before anything is solved (status 1,
       :7773-7780); bounds closer than the tolerance are made equal (:7695-7700) */
    double fixTolerance = M->primalTolerance;
    if (fixTolerance < 2.0e-8)
      fixTolerance *= 1.1;
    int numberBad = 0;
    for (int i = 0; i < N; i++) {
      double value = M->upper[i] - M->lower[i];
      if (value < -M->primalTolerance)
        numberBad++;
      else if (value <= fixTolerance && value)
        M->upper[i] = M->lower[i];
    }
    M->rimInfeasible = numberBad > 0;
    if (numberBad) {
      M->problemStatus = 1;
      M->numberIterations = 0;
      M->numberRefactorizations = 0;
      M->logCount = 0;
      M->objectiveValue = 0.0;
      M->seconds = 0.0;
      return 1;
    }
  }
  if (!M->haveStatus) {
    for (int i = 0; i < m; i++)
      M->status[n + i] = ST_BASIC;
    for (int j = 0; j < n; j++) {
      if (M->colLower[j] >= 0.0) {
        M->status[j] = ST_LOWER;
      } else if (M->colUpper[j] <= 0.0) {
        M->status[j] = ST_UPPER;
      } else if (M->colLower[j] < -1.0e20 && M->colUpper[j] > 1.0e20) {
        M->status[j] = M->freeNonbasic ? ST_FREE : ST_UPPER; /* free: the reference's isFree (allSlackBasis :7846-7849) with option
                                                                free_nonbasic, else bothFake bounds, see header */
      } else if (fabs(M->colLower[j]) < fabs(M->colUpper[j])) {
        M->status[j] = ST_LOWER;
      } else {
        M->status[j] = ST_UPPER;
      }
    }
  }
  for (int i = 0; i < N; i++) {
    int st = getStatus(M, i);
    M->status[i] = (unsigned char)st; /* clear fake/flag bits */
    if (st == ST_LOWER || st == ST_FIXED)
      M->sol[i] = M->lower[i];
    else if (st == ST_UPPER)
      M->sol[i] = M->upper[i];
    else if (st != ST_BASIC)
      M->sol[i] = 0.0;
    if (st != ST_BASIC && M->lower[i] == M->upper[i])
      setStatus(M, i, ST_FIXED);
    if (st == ST_LOWER && M->lower[i] < -1.0e20 && M->upper[i] < 1.0e20) {
      setStatus(M, i, ST_UPPER);
      M->sol[i] = M->upper[i];
    } else if (st == ST_UPPER && M->upper[i] > 1.0e20 && M->lower[i] > -1.0e20) {
      setStatus(M, i, ST_LOWER);
      M->sol[i] = M->lower[i];
    }
    if (M->freeNonbasic && (st == ST_LOWER || st == ST_UPPER) && M->lower[i] < -1.0e20 && M->upper[i] > 1.0e20) {
      setStatus(M, i, ST_FREE); /* createRim's clean-up of a caller's basis, src/ClpSimplex.cpp:4317-4338 */
      M->sol[i] = 0.0;
    }
  }
  M->problemStatus = -1;
  M->numberIterations = 0;
  M->numberRefactorizations = 0;
  M->logCount = 0;
  M->numberFake = 0;
  M->numberChanged = 0;
  M->numberTimesOptimal = 0;
  M->perturbation = M->perturbationOption; /* ClpDataSave: every dual() starts from the caller's value */
  M->numberPerturbations = 0;
  progressReset(M); /* ClpSimplex::saveData -> progress_.fillFromModel, src/ClpSimplex.cpp:9732 */
  M->progressFlag = 0; /* :461 */
  M->bestPossibleImprovement = 0.0;
  M->numberBackwards = M->numberLoopFlags = M->numberAccuracyRestores = M->numberSingularRestores = 0;
  M->numberTryPrimal = 0;
  M->numberPartialScans = M->numberChuzrRecalls = 0;
  M->noFreeOrSuper = 1;
  M->firstFree = -1;
  M->badFree = 0.0;
  M->numberDualInfeasibilitiesWithoutFree = 0;
  M->numberFreeFirstRows = M->numberFreeEntered = 0;
  for (int i = 0; i < ORC_CYCLE; i++) { /* progress_.startCheck(), ClpSimplexDual.cpp:452 */
    M->cycIn[i] = M->cycOut[i] = -1;
    M->cycWay[i] = 0;
  }
  M->pivotRow = -1;
  M->numberInfeasible = 0;
  M->haveSavedWeights = 0;
  M->objectiveValue = 0.0;
  M->largestPrimalError = M->largestDualError = 0.0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  if (factorize(M)) {
    M->problemStatus = 4;
    return 4;
  }
  {
    double dummy = 0.0;
    changeBounds(M, 1, NULL, &dummy);
  }
  gutsOfSolution(M);
  /* startupSolve :330-336: an optimal starting basis sets problemStatus_ = 0 FIRST, and the costs are perturbed only
     `if (problemStatus_ < 0 && perturbation_ < 100)` */
  const int optimalAtStart = !M->numberDualInfeasibilities && !M->numberPrimalInfeasibilities;
  if (M->perturbation < 100 && !optimalAtStart) {
    /* startupSolve :335-341.  perturb() returning 1 ("safer to use primal", all costs zero) is only a hint
       there (usePrimal, read by callers that hold a primal); dual carries on unperturbed. */
    perturb(M);
    gutsOfSolution(M);
  }
  if (!M->numberDualInfeasibilities && !M->numberPrimalInfeasibilities && M->perturbation < 101)
    M->problemStatus = 0; /* ClpSimplexDual::dual :664-666: nothing to do */
  int lastCleaned = 0;
  int factorType = 0;
  double smallestPrimalInfeasibility = 1.7976931348623157e308; /* COIN_DBL_MAX, gutsOfDual :442 */
  double lastObjectiveValue = -1.0e100;                         /* :460 */
  {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    M->startupSeconds = (double)(ts.tv_sec - t0.tv_sec) + 1.0e-9 * (double)(ts.tv_nsec - t0.tv_nsec);
  }
  while (M->problemStatus < 0) {
    for (int i = 0; i < m; i++)
      M->rowWork0[i] = M->rowWork1[i] = M->rowWork2[i] = M->rowWork3[i] = 0.0;
    M->numberPi = M->numberColNz = M->numberW = 0;
    if (M->perturbation < 101 && M->numberIterations > 2 * (m + n)) {
      /* if getting nowhere - why not give it a kick (gutsOfDual :488-492) */
      perturb(M);
      gutsOfSolution(M);
    }
    statusOfProblemInDual(M, &lastCleaned, factorType);
    factorType = 1;
    /* "problems - try primal" (gutsOfDual :533-547): the primal infeasibilities have grown 1e5-fold since the smallest sum seen while the
       objective stood still -- and either the last two recorded objectives say the solve fell off a cliff or the growth is 1e10-fold */
    if (M->objectiveValue > 1.0e-4 + 1.0e-9 * fabs(lastObjectiveValue) + lastObjectiveValue)
      smallestPrimalInfeasibility = 1.7976931348623157e308; /* reset smallest */
    smallestPrimalInfeasibility = dmin(smallestPrimalInfeasibility, M->sumPrimalInfeasibilities);
    lastObjectiveValue = M->objectiveValue;
    if (M->sumPrimalInfeasibilities > 1.0e5 && M->sumPrimalInfeasibilities > 1.0e5 * smallestPrimalInfeasibility
        && ((progressLastObjective(M, 0) < -1.0e10 && -progressLastObjective(M, 1) > -1.0e5)
            || M->sumPrimalInfeasibilities > 1.0e10 * smallestPrimalInfeasibility)
        && M->problemStatus < 0 && M->tryPrimal) {
      M->problemStatus = 10;
      M->sumPrimalInfeasibilities = -123456789.0; /* mark as large infeasibility cost wanted */
      M->numberTryPrimal++;
    }
    if (M->problemStatus < 0) {
      M->problemStatus = -1;
      whileIterating(M);
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  M->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1.0e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  if (M->problemStatus == 0 || M->problemStatus == 3 || M->problemStatus == 10) {
    /* finish(): true objective from original costs (10 = "clean up with primal": the point primal would start from) */
    double objective = 0.0;
    for (int j = 0; j < n; j++)
      objective += M->obj[j] * M->sol[j];
    M->objectiveValue = objective;
  }
  return M->problemStatus;
}

/* ------------------------------------------------------------------------------------------ */
/* Scaling -- ClpPackedMatrix::scale (src/ClpPackedMatrix.cpp:4120-4760) as ClpSimplex::createRim  */
/* calls it (matrix_->scale(this, this), src/ClpSimplex.cpp:3701), its application to bounds and     */
/* costs (createRim :3880-3980) and the unscaling of the results.  Modes: 1 equilibrium, 2           */
/* geometric, 3 and 4 "auto" (equilibrium first, geometric kept if its spread is more than twice     */
/* better).  automaticScale_ (objectiveScale_/rhsScale_) is off, as by default in the reference.     */
/* Returns 1 when the matrix is left unscaled (all |a_ij| within [0.5, 2], :4282), else 0.           */
/* Groundwork for SURVEY 8(f)3: the HIP engine does not scale yet; not part of any parity claim.     */
/* ------------------------------------------------------------------------------------------ */
static int computeScaling(OrcModel *M, double *rowScale, double *columnScale)
{
  const int m = M->m, n = M->n;
  char *usefulColumn = (char *)malloc((size_t)n + 1);
  char *usedRow = (char *)calloc((size_t)m + 1, 1);
  double largest = 0.0, smallest = 1.0e50;
  for (int j = 0; j < n; j++) {
    char useful = 0;
    if (M->colUpper[j] > M->colLower[j] + 1.0e-12 || (M->haveStatus && getStatus(M, j) == ST_BASIC)) {
      for (int p = M->colStart[j]; p < M->colStart[j + 1]; p++) {
        double value = fabs(M->elem[p]);
        if (value > 1.0e-20) {
          useful = 1;
          if (value > largest)
            largest = value;
          if (value < smallest)
            smallest = value;
        }
      }
    }
    usefulColumn[j] = useful;
  }
  if (smallest * 1.0e12 < largest) { /* :4268 increase tolerances */
    if (M->dualTolerance < 5.0e-7)
      M->dualTolerance = M->dualToleranceBase = 5.0e-7;
    if (M->primalTolerance < 5.0e-7)
      M->primalTolerance = 5.0e-7;
  }
  if (smallest >= 0.5 && largest <= 2.0) { /* :4273 don't bother scaling */
    free(usefulColumn);
    free(usedRow);
    return 1;
  }
  if (largest > 1.0e13 * smallest) { /* :4303 safer to have a smaller zero tolerance */
    double newTolerance = smallest / largest * 0.5;
    if (newTolerance < 1.0e-18)
      newTolerance = 1.0e-18;
    if (M->zeroTolerance > newTolerance)
      M->zeroTolerance = newTolerance;
  }
  int scalingMethod = M->scalingMode;
  if (scalingMethod == 4)
    scalingMethod = 3;
  double savedOverallRatio = 0.0;
  const double tolerance = 5.0 * M->primalTolerance;
  double overallLargest, overallSmallest = 1.0e20;
  int finished = 0;
  while (!finished) {
    int numberPass = 3;
    for (int i = 0; i < m; i++)
      rowScale[i] = 1.0;
    for (int j = 0; j < n; j++)
      columnScale[j] = 1.0;
    if (scalingMethod == 1 || scalingMethod == 3) {
      /* maximum in each row (:4340) */
      for (int i = 0; i < m; i++) {
        largest = 1.0e-10;
        for (int q = M->rowStart[i]; q < M->rowStart[i + 1]; q++)
          if (usefulColumn[M->rcol[q]]) {
            double value = fabs(M->relem[q]);
            if (value > largest)
              largest = value;
          }
        rowScale[i] = 1.0 / largest;
      }
    } else {
      /* geometric mean: rows, columns, rows (:4365-4445; the last column round is skipped) */
      while (numberPass) {
        numberPass--;
        for (int i = 0; i < m; i++) {
          largest = 1.0e-50;
          smallest = 1.0e50;
          for (int q = M->rowStart[i]; q < M->rowStart[i + 1]; q++) {
            int j = M->rcol[q];
            if (usefulColumn[j]) {
              double value = fabs(M->relem[q]) * columnScale[j];
              if (value > largest)
                largest = value;
              if (value < smallest)
                smallest = value;
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
            for (int p = M->colStart[j]; p < M->colStart[j + 1]; p++) {
              double value = fabs(M->elem[p]) * rowScale[M->row[p]];
              if (value > largest)
                largest = value;
              if (value < smallest)
                smallest = value;
            }
            columnScale[j] = 1.0 / sqrt(smallest * largest);
          }
      }
    }
    /* if ranges will make horrid then scale (:4451) */
    for (int i = 0; i < m; i++) {
      double difference = M->rowUpper[i] - M->rowLower[i];
      double scaledDifference = difference * rowScale[i];
      if (scaledDifference > tolerance && scaledDifference < 1.0e-4) {
        rowScale[i] *= 1.0e-4 / scaledDifference;
        if (rowScale[i] > 1.0e10)
          rowScale[i] = 1.0e10;
        if (rowScale[i] < 1.0e-10)
          rowScale[i] = 1.0e-10;
      }
    }
    /* what the smallest would be if every column's largest were 1.0 (:4465) */
    overallSmallest = 1.0e50;
    for (int j = 0; j < n; j++)
      if (usefulColumn[j]) {
        largest = 1.0e-20;
        smallest = 1.0e50;
        for (int p = M->colStart[j]; p < M->colStart[j + 1]; p++) {
          double value = fabs(M->elem[p] * rowScale[M->row[p]]);
          if (value > largest)
            largest = value;
          if (value < smallest)
            smallest = value;
        }
        if (overallSmallest * largest > smallest)
          overallSmallest = smallest / largest;
      }
    if (scalingMethod == 1 || scalingMethod == 2) {
      finished = 1;
    } else if (savedOverallRatio == 0.0 && scalingMethod != 4) {
      savedOverallRatio = overallSmallest;
      scalingMethod = 4;
    } else {
      if (overallSmallest > 2.0 * savedOverallRatio)
        finished = 1; /* geometric was better */
      else
        scalingMethod = 1; /* redo equilibrium */
    }
  }
  /* final pass: columns scaled so that their largest is reasonable (:4528-4590) */
  overallLargest = 1.0;
  if (overallSmallest < 1.0e-1)
    overallLargest = 1.0 / sqrt(overallSmallest);
  if (overallLargest > 100.0)
    overallLargest = 100.0;
  overallSmallest = 1.0e50;
  for (int j = 0; j < n; j++) {
    if (M->colUpper[j] > M->colLower[j] + 1.0e-12 && M->colStart[j + 1] > M->colStart[j]) {
      largest = 1.0e-20;
      smallest = 1.0e50;
      for (int p = M->colStart[j]; p < M->colStart[j + 1]; p++) {
        int i = M->row[p];
        usedRow[i] = 1;
        double value = fabs(M->elem[p] * rowScale[i]);
        if (value > largest)
          largest = value;
        if (value < smallest)
          smallest = value;
      }
      columnScale[j] = overallLargest / largest;
      double difference = M->colUpper[j] - M->colLower[j];
      if (difference < 1.0e-5 * columnScale[j])
        columnScale[j] = difference / 1.0e-5; /* make gap larger */
      double value = smallest * columnScale[j];
      if (overallSmallest > value)
        overallSmallest = value;
    } else {
      columnScale[j] = 1.0;
    }
  }
  for (int i = 0; i < m; i++)
    if (!usedRow[i])
      rowScale[i] = 1.0;
  if (overallSmallest < 1.0e-13) { /* :4601 */
    double newTolerance = overallSmallest * 0.5;
    if (newTolerance < 1.0e-18)
      newTolerance = 1.0e-18;
    M->zeroTolerance = newTolerance;
  }
  free(usefulColumn);
  free(usedRow);
  return 0;
}

/* scaled bound as createRim builds it (src/ClpSimplex.cpp:3920-3980): infinities stay infinite */
static void scaleBounds(double lowerValue, double upperValue, double multiplier, double primalTolerance, double *lo, double *up)
{
  const double INF = 1.0e30;
  if (lowerValue > -1.0e20) {
    *lo = lowerValue * multiplier;
    if (upperValue >= 1.0e20) {
      *up = INF;
    } else {
      *up = upperValue * multiplier;
      if (fabs(*up - *lo) <= primalTolerance) { /* fix variables with tiny gaps */
        if (*lo >= 0.0)
          *up = *lo;
        else if (*up <= 0.0)
          *lo = *up;
        else
          *lo = *up = 0.0;
      }
    }
  } else if (upperValue < 1.0e20) {
    *lo = -INF;
    *up = upperValue * multiplier;
  } else {
    *lo = -INF;
    *up = INF;
  }
}

static int dualScaledOrNot(OrcModel *M);

/* ClpSimplex::dual (src/ClpSimplex.cpp:5631): ClpSimplexDual::dual, then its own second thought about an "infeasible" that
 * was reached with fake bounds active -- "clean up in primal as fake bounds" (:5800-5803): status 10, not 1 */
int orc_dual(OrcModel *M)
{
  int status = dualScaledOrNot(M);
  if (status == 1 && !M->rimInfeasible && numberAtFakeBound(M) > 0)
    status = M->problemStatus = 10;
  return status;
}

static int dualScaledOrNot(OrcModel *M)
{
  const int m = M->m, n = M->n;
  M->scalingApplied = 0;
  M->objBeforeScaling = M->obj;
  if (M->scalingMode <= 0)
    return dualOnRim(M);
  free(M->rowScale);
  free(M->colScale);
  M->rowScale = (double *)malloc(sizeof(double) * (size_t)(m + 1));
  M->colScale = (double *)malloc(sizeof(double) * (size_t)(n + 1));
  if (computeScaling(M, M->rowScale, M->colScale))
    return dualOnRim(M); /* not scaled after all (scalingFlag_ = -scalingFlag_, src/ClpSimplex.cpp:3702) */
  const double *rs = M->rowScale, *cs = M->colScale;
  const int nnz = M->colStart[n];
  /* scaled model: A_s = R A C, x_s = x / c_j, row_s = r_i * row, cost_s = c_j * cost */
  double *elemS = (double *)malloc(sizeof(double) * (size_t)(nnz + 1));
  double *relemS = (double *)malloc(sizeof(double) * (size_t)(nnz + 1));
  double *clS = (double *)malloc(sizeof(double) * (size_t)(n + 1)), *cuS = (double *)malloc(sizeof(double) * (size_t)(n + 1));
  double *objS = (double *)malloc(sizeof(double) * (size_t)(n + 1));
  double *rlS = (double *)malloc(sizeof(double) * (size_t)(m + 1)), *ruS = (double *)malloc(sizeof(double) * (size_t)(m + 1));
  for (int j = 0; j < n; j++) {
    for (int p = M->colStart[j]; p < M->colStart[j + 1]; p++)
      elemS[p] = M->elem[p] * (cs[j] * rs[M->row[p]]); /* element *= scale * rowScale[iRow], src/ClpPackedMatrix.cpp:4789-4794 */
    objS[j] = M->obj[j] * cs[j];
    scaleBounds(M->colLower[j], M->colUpper[j], 1.0 / cs[j], M->primalTolerance, &clS[j], &cuS[j]);
  }
  for (int i = 0; i < m; i++) {
    for (int q = M->rowStart[i]; q < M->rowStart[i + 1]; q++)
      relemS[q] = M->relem[q] * (rs[i] * cs[M->rcol[q]]); /* the row copy: element *= scale * columnScale[iColumn], :5577-5584 -- the same
                                                             product, so the two copies hold the same numbers */
    scaleBounds(M->rowLower[i], M->rowUpper[i], rs[i], M->primalTolerance, &rlS[i], &ruS[i]);
  }
  double *saveElem = M->elem, *saveRelem = M->relem, *saveCl = M->colLower, *saveCu = M->colUpper, *saveObj = M->obj,
         *saveRl = M->rowLower, *saveRu = M->rowUpper;
  M->elem = elemS;
  M->relem = relemS;
  M->colLower = clS;
  M->colUpper = cuS;
  M->obj = objS;
  M->rowLower = rlS;
  M->rowUpper = ruS;
  int status = dualOnRim(M);
  M->elem = saveElem;
  M->relem = saveRelem;
  M->colLower = saveCl;
  M->colUpper = saveCu;
  M->obj = saveObj;
  M->rowLower = saveRl;
  M->rowUpper = saveRu;
  /* unscale: activities, reduced costs, duals (deleteRim, src/ClpSimplex.cpp:3376-3412) */
  for (int j = 0; j < n; j++) {
    M->sol[j] *= cs[j];
    M->dj[j] /= cs[j];
  }
  for (int i = 0; i < m; i++) {
    M->sol[n + i] /= rs[i];
    M->dj[n + i] *= rs[i];
  }
  if (status == 0 || status == 3 || status == 10) {
    double objective = 0.0;
    for (int j = 0; j < n; j++)
      objective += M->obj[j] * M->sol[j];
    M->objectiveValue = objective;
  }
  M->scalingApplied = 1;
  free(elemS);
  free(relemS);
  free(clS);
  free(cuS);
  free(objS);
  free(rlS);
  free(ruS);
  return status;
}

/* scale factors of the last scaled solve (1.0 everywhere if it was not scaled); returns scalingApplied */
int orc_get_scale_factors(const OrcModel *M, double *rowScale, double *columnScale)
{
  for (int i = 0; i < M->m; i++)
    rowScale[i] = (M->scalingApplied && M->rowScale) ? M->rowScale[i] : 1.0;
  for (int j = 0; j < M->n; j++)
    columnScale[j] = (M->scalingApplied && M->colScale) ? M->colScale[j] : 1.0;
  return M->scalingApplied;
}

/* test hook: a sequence of status checks through progressLooping (ClpSimplexProgress::looping for the dual) on a scratch
 * model without rows or columns.  Check i sees objective[i] (the possible improvement taken as zero), infeasibility[i],
 * numberInfeasibilities[i] at iteration[i], with progressFlag_ & 3 = flagBits[i] and newestIncoming[i] as the last entry of
 * the cycle detector's in_ list.  Out: code[i] = what looping() returned, and the model's dual tolerance, dual bound,
 * forceFactorization_ afterwards; flagged[i] = the sequence it flagged or -1.  Sequences must be below 64. */
void orc_test_looping(int count, const double *objective, const double *infeasibility, const int *numberInfeasibilities,
                      const int *iteration, const int *flagBits, const int *newestIncoming, int *code, double *dualTolerance,
                      double *dualBound, int *forceFactorization, int *flaggedOut)
{
  OrcModel *M = (OrcModel *)calloc(1, sizeof(OrcModel));
  M->status = (unsigned char *)calloc(64, 1);
  M->dualTolerance = M->dualToleranceBase = 1.0e-7;
  M->dualBound = 1.0e10;
  M->forceFactorization = -1;
  progressReset(M);
  for (int i = 0; i < count; i++) {
    M->objectiveValue = objective[i];
    M->bestPossibleImprovement = 0.0;
    M->sumPrimalInfeasibilities = infeasibility[i];
    M->numberPrimalInfeasibilities = numberInfeasibilities[i];
    M->numberIterations = iteration[i];
    M->progressFlag = flagBits[i];
    progressStartCheck(M);
    M->cycIn[ORC_CYCLE - 1] = newestIncoming[i];
    memset(M->status, 0, 64);
    code[i] = progressLooping(M);
    dualTolerance[i] = M->dualTolerance;
    dualBound[i] = M->dualBound;
    forceFactorization[i] = M->forceFactorization;
    flaggedOut[i] = -1;
    for (int j = 0; j < 64; j++)
      if (flagged(M, j))
        flaggedOut[i] = j;
  }
  free(M->status);
  free(M);
}

/* test hook: ClpSimplexDual::perturb on a fresh rim (createRim: the model's bounds and costs, unscaled) with the given
 * statuses, as if numberIterations pivots had been made.  cost[n+m] receives the perturbed costs; returns
 * 1000 * (perturb's return code) + perturbation_ afterwards.  tests/test_perturb_host.py holds the engine's host
 * arithmetic (clp_amd/csrc/perturb_host.h) against this, bit for bit. */
int orc_test_perturb(OrcModel *M, int perturbation, int numberIterations, const unsigned char *status, double *cost)
{
  const int m = M->m, n = M->n;
  for (int j = 0; j < n; j++) {
    M->lower[j] = M->colLower[j];
    M->upper[j] = M->colUpper[j];
    M->cost[j] = M->obj[j];
  }
  for (int i = 0; i < m; i++) {
    M->lower[n + i] = M->rowLower[i];
    M->upper[n + i] = M->rowUpper[i];
    M->cost[n + i] = 0.0;
  }
  memcpy(M->status, status, (size_t)(m + n));
  M->perturbation = perturbation;
  M->numberIterations = numberIterations;
  M->objBeforeScaling = M->obj;
  M->dualTolerance = M->dualToleranceBase;
  int rc = perturb(M);
  memcpy(cost, M->cost, sizeof(double) * (size_t)(m + n));
  return 1000 * rc + M->perturbation;
}

int orc_number_iterations(const OrcModel *M) { return M->numberIterations; }
double orc_objective_value(const OrcModel *M) { return M->objectiveValue; }
int orc_number_refactorizations(const OrcModel *M) { return M->numberRefactorizations; }
int orc_number_perturbations(const OrcModel *M) { return M->numberPerturbations; }
int orc_number_backwards(const OrcModel *M) { return M->numberBackwards; }
int orc_number_loop_flags(const OrcModel *M) { return M->numberLoopFlags; }
int orc_number_accuracy_restores(const OrcModel *M) { return M->numberAccuracyRestores; }
int orc_number_singular_restores(const OrcModel *M) { return M->numberSingularRestores; }
int orc_number_try_primal(const OrcModel *M) { return M->numberTryPrimal; }
int orc_number_partial_scans(const OrcModel *M) { return M->numberPartialScans; }
int orc_number_chuzr_recalls(const OrcModel *M) { return M->numberChuzrRecalls; }
long orc_factor_elements(const OrcModel *M) { return M->factorElements; }
int orc_number_free_first_rows(const OrcModel *M) { return M->numberFreeFirstRows; }
int orc_number_free_entered(const OrcModel *M) { return M->numberFreeEntered; }
double orc_iteration_seconds(const OrcModel *M) { return M->seconds; }
double orc_startup_seconds(const OrcModel *M) { return M->startupSeconds; }
void orc_get_solution(const OrcModel *M, double *s) { memcpy(s, M->sol, sizeof(double) * (size_t)(M->m + M->n)); }
void orc_get_reduced_costs(const OrcModel *M, double *d) { memcpy(d, M->dj, sizeof(double) * (size_t)(M->m + M->n)); }
void orc_get_status(const OrcModel *M, unsigned char *s) { memcpy(s, M->status, (size_t)(M->m + M->n)); }
void orc_get_pivot_variable(const OrcModel *M, int *p) { memcpy(p, M->pivotVariable, sizeof(int) * (size_t)M->m); }
void orc_get_row_duals(const OrcModel *M, double *d) { memcpy(d, M->dj + M->n, sizeof(double) * (size_t)M->m); }
void orc_get_row_weights(const OrcModel *M, double *w, double *inf)
{
  memcpy(w, M->weights, sizeof(double) * (size_t)M->m);
  memcpy(inf, M->infeas, sizeof(double) * (size_t)M->m);
}
int orc_get_pivot_log(const OrcModel *M, OrcPivotRecord *out, int maxRecords)
{
  int count = M->logCount < maxRecords ? M->logCount : maxRecords;
  if (out)
    memcpy(out, M->log, sizeof(OrcPivotRecord) * (size_t)count);
  return M->logCount;
}

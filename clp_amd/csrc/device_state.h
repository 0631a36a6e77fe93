// device_state.h -- layout of the device-resident simplex state of libclpgpu (gfx950 only).
//
// Everything the per-iteration loop touches lives in HBM; the host sees only the small control
// block `Ctrl` (pinned mirror) and syncs the rim arrays at refactorization boundaries
// ("host mirrors to sync at refactor boundaries").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clpgpu {

// ClpSimplex::Status (reference src/ClpSimplex.hpp:119-126)
enum : int { ST_FREE = 0, ST_BASIC = 1, ST_UPPER = 2, ST_LOWER = 3, ST_SUPER = 4, ST_FIXED = 5 };
// ClpSimplexDual::FakeBound, bits 3-4 of the status byte
enum : int { FAKE_NONE = 0, FAKE_LOWER = 1, FAKE_UPPER = 2, FAKE_BOTH = 3 };
constexpr unsigned char FLAGGED_BIT = 64;

// Ctrl::state while the device loop is running / why it stopped
enum : int {
  RUN = -1,
  EXIT_NO_PIVOT_ROW = 100,  // dualRow found nothing (ClpSimplexDual.cpp:2080)
  EXIT_NO_INCOMING = 101,   // ratio test found nothing (:1869)
  EXIT_ALPHA_CHECK = 102,   // btran/ftran alpha disagree (:1451)
  EXIT_REFACTOR = 103,      // housekeeping asked for a refactorization (:1849)
  EXIT_MAX_ITERATIONS = 104,
  EXIT_BACKWARDS = 105,     // objective going backwards (:1574)
  EXIT_BAD_UPDATE = 106,    // replaceColumn says singular (:1618)
  EXIT_STEP_LIMIT = 107,    // bench stepping: requested number of pivots done
  EXIT_SHARD_OVERFLOW = 108 // column-sharded run: a rank's candidate / flip records exceed the exchange buffer
};

#define FLIP_HOT_CAP 1024
struct Ctrl {
  int state;
  int numberIterations;
  int pivots;  // basis updates since the last refactorization
  int k;       // size of the nucleus (number of basic structurals)
  int pivotRow, sequenceIn, sequenceOut, directionIn, directionOut;
  int lastPivotRow;
  int numberInfeasible;
  int numberCandidates;
  int numberFlips;
  int numberAppend;
  int updateCase;  // 0 struct->struct, 1 slack out/struct in, 2 struct out/slack in, 3 slack->slack
  int slotColOut;  // col-slot of a leaving structural
  int slotRowIn;   // row-slot of the row whose slack enters
  int rowOfSlackOut;
  int maximumPivots, maximumIterations, forceFactorization, stepLimit;
  int numberChanged;
  int progressFlag;  // ClpSimplex::progressFlag_ bits 1 (a fixed variable left) and 2 (a free one came in), set per pivot (ClpSimplex.cpp:2096-2100)
  int logCount, logCapacity;
  int pivotRule;
  int lastBadIteration;
  int badSumPivots;
  int modifyCosts;
  unsigned int seed;
  int kcap;
  int scratchCount;  // generic compaction total
  double alpha, theta, dualOut, dualIn, valueIn, valueOut, lowerIn, upperIn, lowerOut, upperOut;
  double btranAlpha, movement;
  double objectiveValue, objectiveChange;
  double upperTheta, acceptablePivot, acceptablePivotBase;
  double primalTolerance, dualTolerance, zeroTolerance, dualBound, largeValue;
  double largestPrimalError, largestDualError;
  double saveSumDual;
  double norm;
  double bestPossible;
  double scratchSum;
  double statPriceBytes, statPriceLaunches;  // by-column bytes; launches of either form
  double statRowBytes, statRowLaunches;      // the by-row form's share
  double statDensePi;                        // pricing launches whose pi was dense (12 nnz >= m): what the host picks the chain's pricing kernel by
  // CHUZR hand-over between its three kernels
  double chuzrTolerance;
  int chuzrNumber, chuzrStart, chuzrLast, chuzrOrdered;  // chuzrOrdered: partial scans that had to walk the whole list in order (a flagged candidate / the last pivot row in the scanned part)
  int classCount[4];
  int tCount, preDone;  // ratio-test candidates by breakpoint class (k_cand_scatter)
  long long dbg[16];    // development counters (CLPGPU_DEBUG_STATS)
  long long dbg2[8];    // phase clocks of k_flip_apply2
  int ticket[8];        // "last workgroup done" counters (always 0 between launches)
  int ticketGroup[8][64];  // first level of the same: one counter per 32 workgroups (<= 2048 workgroups)
  int flipAppend, numberAppend1;
  int flipHotCount;  // rows with more flip contributions than slots this pivot (k_dj_flags appends, CHUZR resets)
  int lastPriceByRow, shardRowCands;  // row candidates ahead of the column candidates in the local list (sharded runs)
  int cycIn[12], cycOut[12], cycWay[12], cycHead;  // ClpSimplexProgress in_ / out_ / way_ (CLP_CYCLE = 12, src/ClpSolve.hpp:435)  // form the last pricing launch took (k_price_row_finish)
  int appendGo, flipDense;  // flipDense: this pivot's flip rhs is left to k_flip_dense
  int updGo[2], updK, updPad;  // basis-update branch: go flag per pivot parity, k at the time of the fork
  long long dbgDc[8];
  int pendingState, pendingPad;  // stepped run: what housekeeping decided on the pivot the step limit stopped at (RUN or EXIT_REFACTOR); acted on when the run resumes
  int wsJ, wsCount;  // ratio test: breakpoint class prefix of the working set k_dc_working_set compacted, and its size (wsJ < 0: none)  // ratio test, working-set path: calls, ticks of the whole kernel, ticks before the passes start, max ticks of one call
  long long dbgCc[8];  // development counters of the ratio test's final batch: compacted calls, too large to compact, sum of batch sizes, ticks of the trips, ticks of the coarse passes, wide calls
  int dcArrive, dcWide;  // k_dual_column_wide: grid-barrier arrivals of this pivot; 1 = this pivot's ratio test was left to it, -1 = a barrier timed out
  // option free_nonbasic (isFree / superbasic nonbasics, src/ClpSimplexDual.cpp:3005-3055, :4058-4179); all zero without it.
  // presetRowPlus1: the host's free-first row for this pivot (dualRow :3005-3055), taken by CHUZR's final selection instead of its own;
  // freeHold: the tail of the pivot does not run the head of the next CHUZR (the host decides first whether that pivot is free-first);
  // freeCount: entries of Dev::freeList (0: the fast branch of dualColumn0, no k_free_scan work); freeChosen: k_free_scan brought a
  // free variable in, the ratio test is skipped (:4321 "always choose"); badFree: dualColumn0's badFree_ of this pivot row
  int presetRowPlus1, freeHold, freeCount, freeChosen;
  double badFree;
  int freeEntered, freePad;
  // ClpDualRowSteepest::pivotRow's partial scan and second call (src/ClpDualRowSteepest.cpp:258-278, :329-346): steepestMode = mode_
  // (option steepest_mode, 3 = the constructor's default); factorElements = what stands for factorization()->numberElements() as of the
  // last factorization (host, option steepest_elements); chuzrWanted = numberWanted of this call, chuzrTolChanged = toleranceChanged;
  // counters: calls that stopped on numberWanted / second calls
  int steepestMode, chuzrWanted, chuzrTolChanged, chuzrRecalls;
  long long factorElements;
  int chuzrPartialScans, chuzrFloor;  // chuzrFloor: the 2000 of :260-276 (option debug_chuzr_floor)
  int luCompactCount, luCompactOn;  // compact eta file: slots in use; 1 = the chain's FTRAN reads it (option lu_compact_eta)
  int debugDcTimeoutAt, debugPad;  // option debug_dc_wide_timeout_at: k_dual_column_wide reports a barrier timeout at this iteration
  double debugToleranceFactor;  // option debug_tolerance_factor (fault injection for the changed tolerance of CHUZR); 0 off
  double tailAlpha, tailValueOut;  // w[pivotRow] / sol[sequenceOut] handed to the serial tail of k_ftran_scatter3  // this pivot's primal update completed: its list appends may be scattered  // unordered append count of k_dj_flags (ordered later by k_flip_apply2)
};

struct PivotRecord {  // == clpgpu_pivot_record
  int iteration, sequenceIn, sequenceOut, pivotRow, numberFlipped, reserved;
  double theta, alpha, dualOut, objective;
};


// ---- LU factorization mode (lu_kernels.hip): frozen B0 = [slack singletons | Markowitz front | dense tail]
// plus a product-form eta file.  One gather-form level schedule per triangular sweep.
struct LuTri {
  int nLevels, nItems;
  const int *levelStart;  // [nLevels+1] into the items
  const int *tgt, *src;   // [nItems] out[tgt] = (srcv[src] - sum val * vec[idx]) / div
  const int *entStart;    // [nItems+1]
  const int *entIdx;
  const double *entVal;
  const double *div;      // [nItems]
};
struct LuDev {
  int k, nF, k2, ns;   // nucleus order, front pivots, tail order, rows whose slack was basic at the refactorization
  int kpad, tcap;      // stride of the per-right-hand-side work vectors; capacity of the eta file
  LuTri Lf, Ub, Utf, Ltb;  // L^-1 rows; U11^-1 [I | -U12] rows; its transpose (front pivots, then tail columns); L^-T
  const int *rowOfLocal;  // [k] local nucleus row -> row
  const int *posOfCol;    // [k] local nucleus column -> basis position
  const int *tailRow, *tailCol;  // [k2] tail slot -> local nucleus row / column
  // U rows of the slack singletons: row i, entries (local nucleus column, value); and the same by column
  const int *sRowIndex, *sRowStart, *sRowCol;
  const double *sRowVal;
  const int *sColStart, *sColRow;  // sColRow holds ROW INDICES (positions of the slacks)
  const double *sColVal;
  const double *MinvT;  // [k2 x ld] transpose of the tail inverse (the BTRAN's contiguous rows)
  // work
  double *wr, *xc;     // [3 * kpad] by local nucleus row / column
  double *tcv;         // [kpad]
  double *x0;          // [3 * m] B0 solves by position
  double *cp;          // [m] c' of the BTRAN (zero between solves)
  double *y;           // [m] BTRAN result by row (chain)
  // eta file
  double *H;           // [tcap * m] eta j at H + j*m, by position
  double *G;           // [tcap * tcap] (I + N)^-1, row-major, lower triangular
  double *GT;          // [tcap * tcap] its transpose (column sweeps read it row-wise)
  int *P;              // [tcap] position of eta j
  int *prevSame, *nextSame;  // [tcap] etas on the same position
  int *lastOfPos;      // [m]
  double *s, *g, *d;   // [3 * tcap], [tcap], [tcap]
  // compact copy of the eta file for the chain's FTRAN (option lu_compact_eta): only the positions that hold -- or held at some pivot
  // since the refactorization -- a structural.  Slot q < Ctrl::luCompactCount stands for position posOfCslot[q]; the structurals of the
  // refactorization come first, in position order, then one slot per position whose slack left since (its column of H is copied in
  // when that happens).  Every basic structural sits at a position with a slot; a position without one still holds the slack of its
  // own row, and the FTRAN gets its value from the row itself: x_i = A[i, K] x_K - v_i over the row copy's basic part.
  double *Hc;          // [ldc * tcap] slot-major: slot q holds its entries of etas 0 .. t - 1 at Hc + q * tcap
  int ldc;             // slots allocated
  int *cslotOfPos;     // [m] slot of a position, -1 none
  int *posOfCslot;     // [ldc]
  const int *sRowOf;   // [m] index of a row among the frozen slack rows (sRow*), -1 none
  int ncs0;            // slots at the refactorization (the later ones are positions whose slack left since)
  double *xK;          // [ldc * 4] the three FTRAN results at the slots, packed (x, tau, flip part, 0): what the slack rows gather
  int *posOfBasicCol;  // [n] basis position of a basic structural (kept per pivot by the housekeeping kernel)
};

// All device pointers of one context.  Passed by value to kernels.
struct Dev {
  int m, n, N;
  int firstColumn, lastColumn;  // column keys handled by the N-wide kernels (candidates, dj update)
  int priceFirst, priceLast;    // columns this GPU prices (its SELL copy); == key range on one GPU
  // A by column (ClpPackedMatrix / CoinPackedMatrix layout)
  const int *colStart;
  const int *row;
  const double *elem;
  // A by row, each row partitioned [basic structurals | nonbasic]; cross indices keep the
  // partition maintainable in O(column length) per pivot (cf. ClpPackedMatrix3::swapOne)
  const int *rowStart;
  int *ccol;
  int *cslot;  // [nnz] col-slot of the entry's column while it is in the basic part of its row
  double *relem;
  int *csrToCsc;
  int *cscToCsr;
  int *basicCount;
  // rim arrays [columns | rows]
  double *lower, *upper, *cost, *dj, *sol;
  const double *origLower, *origUpper;
  unsigned char *status;
  // basis bookkeeping
  int *pivotVariable;  // [m] sequence at basis position p
  int *posOfSlack;     // [m] position of the basic slack of row i, or -1
  int *slotOfRow;      // [m] row-slot of a nucleus row, or -1
  int *slotOfCol;      // [n] col-slot of a basic structural, or -1
  int *slotRow;        // [kcap]
  int *slotCol;        // [kcap]
  int *slotPos;        // [kcap] basis position of col-slot
  double *Minv;        // [kcap*ld] row-major: x_K[sc] = sum_sr Minv[sc*ld+sr] * v_R[sr]
  int ld;
  // work vectors
  double *vecC;      // [m] BTRAN input by position
  double *rho;       // [m] BTRAN result by row, |.|<=zeroTolerance flushed (the packed pi)
  double *piNeg;     // [m] -rho (what the pricing kernel gathers)
  unsigned long long *piBits;  // [(m+63)/64 + 4] bitmap of the nonzero rows of pi
  double *alphaCol;  // [n] tableau row, column part (0 where skipped)
  double *vecV1, *vecV2;  // [m] FTRAN inputs by row
  double *w, *tau, *x3;   // [m] FTRAN results by position
  double *flipRhs;        // [m]
  double *slotA, *slotB, *slotC, *slotD, *slotE, *slotF;  // [kcap] nucleus-sized scratch
  int *tIndex;       // [m] nonzero col-slots of the BTRAN t-vector
  double *tValue;
  double *rhoSlot;   // [kcap] unpruned rho on nucleus rows (for the rank-1 update)
  double *slotV1, *rhoSlotF, *flipSlot;  // [kcap] the three FTRAN right-hand sides by nucleus row-slot (entering column, pruned rho, flip rhs)
  double *partial;   // gemvT partials [(kcap/64+1) * kcap]
  // dual row pivot
  double *weights, *altWeights, *infeas, *weightBySeq;
  double *savedWeightBySeq;  // ClpDualRowSteepest::savedWeights_: the by-sequence weights of the last saveWeights(2), what mode 4 restores
  int *infIndex;
  // ratio test
  unsigned char *candFlag;  // [N] by key (rows first, then columns)
  int *candSeq;
  double *candAlpha;
  int *candTag;
  unsigned char *candLive;
  double *candDj, *candRange;  // [N] dj and upper - lower of every candidate (snapshot taken by k_cand_scatter)
  int *wsIdxG;                 // [DC_WS_CAP] working set of the ratio test (candidate indices, list order)
  int *freeList;               // [N] option free_nonbasic: the sequences whose status was isFree / superBasic at the last status check, rows first
                               // (the order dualColumn0's general branch meets them in); Ctrl::freeCount of them
  double *dcPart;              // [2 * DCW_BLOCKS * DCW_PART] per-workgroup partials of k_dual_column_wide, two sets
  int *candBlk, *candRk;       // [N] compaction block of the candidate; its rank among classes <= 0 / 1 / 2 inside that block (10 bits each)
  double *flipRecMv, *flipRecObj;   // [FLIP_LIST_CAP] per appended flip: movement, objective term
  int *flipRecStart, *flipRecLen;   // [FLIP_LIST_CAP] its column extent (a row flip: length 1)
  int *blockCount, *blockOffset;
  int *classBlock;  // [3 * blocks] ratio-test breakpoint classes per compaction block
  double *blockMin, *blockSum;
  int *flipSeq;
  double *flipMv;  // [FLIP_MAX_FLIPS] movement of each flip, list order (dense-column mode)
  double *rowDot;  // [3m] wide-row mode: slack-row parts of the three FTRANs (k_slack_dots)
  int *flipKey;  // [FLIP_LIST_CAP] flagged bound flips in arrival order, as compaction keys
  int *appendFlag;  // [m]
  int *appendFlag1, *blockOffset1;  // the same for the flip part of the primal update (scattered together later)
  int *flipHot;     // [FLIP_HOT_CAP] those rows
  int *flipTouch;   // [m] contributors per row while the flip rhs is assembled (zero otherwise)
  int *flipRowKey;  // [m * 8] their flip keys ...
  double *flipRowVal;  // [m * 8] ... and movement * element, in ticket order
  // sliced-ELL copy of the priced column range: slice = 64 columns (one wave), entry t of the
  // slice's lane l at sellStart[slice] + t*64 + l; columns sorted by length so padding is ~1%
  const int *sellStart;  // [numSlices+1]
  const int *sellCol;    // [numSlices*64] original column, -1 padding
  const int *sellLen;    // [numSlices*64]
  const int *sellRow;
  const double *sellElem;
  int numSlices;
  int sellWindowed, sellWinBase;  // windowed SELL copy: workgroup b of the pricing kernel holds the columns of compaction block sellWinBase + b (buildSell)
  // the same windows stored for the dense-pi form (k_price_lds, round 5): the rows cut into jdsTiles tiles of jdsTileRows (a pi
  // tile lives in LDS), every slice's entries as ONE jagged stream, tile after tile: inside a (slice, tile) segment the 64
  // columns are ordered by their entry count in that tile, so the lanes that still hold an entry at step t are a prefix and
  // the segment is stored without padding -- per PAIR of steps one record per such lane: two 16-bit tile-local row indices
  // (jdsRowPair) and two elements (jdsElemPair; the second 0.0 when the column's count in the tile is odd)
  int jdsWindows, jdsTiles, jdsTileRows;
  const int *jdsSegStart;         // [jdsWindows * 4] first record of the slice's stream
  const unsigned char *jdsCnt;    // [(slice * jdsTiles + tile) * 64 + p] entries in the tile of the column at position p of THAT tile's order
  const unsigned char *jdsSrc;    // [(slice * jdsTiles + tile) * 64 + p] position of the same column in the previous tile's order
  const unsigned char *jdsHome;   // [slice * 64 + l] position, in the last tile's order, of the column at home position l
  const int *jdsCol;              // [slice * 64 + l] column key at home position l (sorted by length inside the window), -1 none
  const unsigned *jdsRowPair;
  const double2 *jdsElemPair;
  int *touchCol;  // [n] by-row pricing: contributors per column while a tableau row is assembled (zero otherwise)
  int *touchRow;  // [n * 8] their rows ...
  double *touchVal;  // [n * 8] ... and products, in ticket order
  const int *longCol;  // [numLong] columns too long for a SELL lane (a wave strides each)
  int numLong;
  double *sellMin, *sellBytes;  // per pricing workgroup
  double *chzBest;
  int *chzKey, *chzRow;
  int *chzCnt;  // partial scan: entries above the tolerance in the workgroup's span of the list (bit 30: a flagged one or the last pivot row among them)
  double *normPartial;
  // refactorization scratch
  double *workW, *workX;  // [kcap*ld]
  int *perm;
  double *gjL;  // [kcap * 32] multipliers of the current block (blocked re-inversion)
  double *gjU;  // [32 * 2*ld]  pivot-row values of the current block
  int *gjPiv;   // [64]
  double *gjL2; // [kcap * 64]  multipliers of the current outer block (two-level re-inversion)
  double *gjU2; // [64 * ld]    its pivot-row values
  int luMode;       // 1: the factorization on the device is the LU form (lu_kernels.hip)
  const LuDev *lu;  // its descriptor, in device memory
  Ctrl *ctrl;
  PivotRecord *log;
};

}  // namespace clpgpu

// kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for one pivot of the revised
// dual simplex.  Each kernel cites the reference loop it replaces (paths relative to the Clp tree).
//
// Rules followed throughout:
//  * every kernel of the iteration chain starts with `if (ctrl->state != RUN) return;` -- control
//    flow lives on the device, the host only polls the control block;
//  * no floating-point atomics and fixed reduction trees: results are deterministic run to run;
//  * compiled with -ffp-contract=off so per-column dot products are the same sequence of IEEE
//    operations as the reference's scalar loops (bit-identical tableau rows).
#include "device_state.h"

namespace clpgpu {

#define WAVE 64
constexpr double REALLY_TINY = 1.0e-100;  // COIN_INDEXED_REALLY_TINY_ELEMENT
constexpr double DEVEX_TRY_NORM = 1.0e-4; // src/ClpSimplex.hpp:2056

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ inline double waveSum(double v)
{
  for (int o = 32; o > 0; o >>= 1)
    v += __shfl_down(v, o);
  return v;  // valid in lane 0; fixed tree => deterministic
}
__device__ inline double waveMin(double v)
{
  for (int o = 32; o > 0; o >>= 1)
    v = fmin(v, __shfl_down(v, o));
  return v;
}
// block-wide deterministic sum; result valid in every thread. blockDim.x multiple of 64, <= 1024
__device__ inline double blockSum(double v, double *sh /*[16]*/)
{
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = waveSum(v);
  __syncthreads();
  if (lane == 0)
    sh[wv] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; i++)
    t += sh[i];
  return t;
}
__device__ inline double blockMin(double v, double *sh)
{
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = waveMin(v);
  __syncthreads();
  if (lane == 0)
    sh[wv] = v;
  __syncthreads();
  double t = sh[0];
  for (int i = 1; i < nw; i++)
    t = fmin(t, sh[i]);
  return t;
}
// block argmax of (value, smallest key wins ties); value <= floorValue => none (key stays -1)
__device__ inline void blockArgMax(double &value, int &key, double *shv, int *shk)
{
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_down(value, o);
    int ok = __shfl_down(key, o);
    if (ok >= 0 && (key < 0 || ov > value || (ov == value && ok < key))) {
      value = ov;
      key = ok;
    }
  }
  __syncthreads();
  if (lane == 0) {
    shv[wv] = value;
    shk[wv] = key;
  }
  __syncthreads();
  value = shv[0];
  key = shk[0];
  for (int i = 1; i < nw; i++) {
    if (shk[i] >= 0 && (key < 0 || shv[i] > value || (shv[i] == value && shk[i] < key))) {
      value = shv[i];
      key = shk[i];
    }
  }
}
// exclusive rank of `flag` inside the block (threads in index order) + block total
__device__ inline int blockRank(int flag, int &total, int *sh /*[17]*/)
{
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  unsigned long long mask = __ballot(flag);
  int rank = __popcll(mask & ((1ull << lane) - 1ull));
  __syncthreads();
  if (lane == 0)
    sh[wv] = __popcll(mask);
  __syncthreads();
  int base = 0;
  total = 0;
  for (int i = 0; i < nw; i++) {
    if (i < wv)
      base += sh[i];
    total += sh[i];
  }
  return base + rank;
}

__device__ inline double randomDouble(Ctrl *c)
{
  // CoinThreadRandom::randomDouble, 32-bit LCG form [CoinUtils, not in the reference tree]
  c->seed = 1664525u * c->seed + 1013904223u;
  return ((double)c->seed) / 4294967296.0;
}

// =============================================================================================
// CHUZR -- ClpDualRowSteepest::pivotRow (src/ClpDualRowSteepest.cpp:179-364, full scan) or
// ClpDualRowDantzig::pivotRow (src/ClpDualRowDantzig.cpp:56-92), then the scalar part of
// ClpSimplexDual::dualRow (src/ClpSimplexDual.cpp:3079-3102) and the acceptablePivot choice of
// whileIterating (:1270-1278).  One workgroup: the list is at most m long and is read once.
// =============================================================================================
__global__ void __launch_bounds__(1024) k_chuzr(Dev D)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double shv[16];
  __shared__ int shk[16];
  __shared__ double s_tolerance;
  __shared__ int s_number, s_start, s_last;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_number = 0;
    s_start = 0;
    s_tolerance = 0.0;
    s_last = -1;
  }
  if (tid == 0 && c->stepLimit >= 0 && c->numberIterations >= c->stepLimit) {
    c->state = EXIT_STEP_LIMIT;
  } else if (tid == 0) {
    int last = c->pivotRow;  // model_->pivotRow(): persists across refactorizations
    double tolerance = c->primalTolerance;
    if (c->pivotRule) {
      tolerance = tolerance + fmin(1.0e-2, c->largestPrimalError);
      tolerance = fmin(1000.0, tolerance);
      tolerance *= tolerance;
      if (last >= 0 && last < D.m) {
        int iPivot = D.pivotVariable[last];
        double value = D.sol[iPivot], lower = D.lower[iPivot], upper = D.upper[iPivot];
        if (value > upper + tolerance) {
          value -= upper;
          value *= value;
          if (D.infeas[last] == 0.0)
            D.infIndex[c->numberInfeasible++] = last;
          D.infeas[last] = value;
        } else if (value < lower - tolerance) {
          value -= lower;
          value *= value;
          if (D.infeas[last] == 0.0)
            D.infIndex[c->numberInfeasible++] = last;
          D.infeas[last] = value;
        } else if (D.infeas[last] != 0.0) {
          D.infeas[last] = REALLY_TINY;
        }
      }
      if (c->numberIterations < c->lastBadIteration + 200) {
        if (c->largestDualError > c->largestPrimalError)
          tolerance *= fmin(c->largestDualError / c->largestPrimalError, 1000.0);
      }
      int number = c->numberInfeasible;
      double dstart = ((double)number) * randomDouble(c);
      s_number = number;
      s_start = (int)dstart;
    } else {
      if (c->largestPrimalError > 1.0e-8)
        tolerance *= c->largestPrimalError / 1.0e-8;
      s_number = D.m;
      s_start = 0;
    }
    s_tolerance = tolerance;
    s_last = last;
  }
  __syncthreads();
  if (c->state != RUN)
    return;
  const double tolerance = s_tolerance;
  const int number = s_number, start = s_start, last = s_last;
  double best = 0.0;
  int bestKey = -1;  // key = rank in scan order (smaller = scanned earlier)
  int bestRow = -1;
  for (int i = tid; i < number; i += blockDim.x) {
    if (c->pivotRule) {
      int iRow = D.infIndex[i];
      double value = D.infeas[iRow];
      if (value > tolerance) {
        double weight = fmin(D.weights[iRow], 1.0e50);
        if (iRow == last)
          value *= 1.0e-10;  // last pivot row is the last resort (:302-307)
        int iSequence = D.pivotVariable[iRow];
        if (!(D.status[iSequence] & FLAGGED_BIT)) {
          double s = D.sol[iSequence];
          if (s > D.upper[iSequence] + tolerance || s < D.lower[iSequence] - tolerance) {
            double ratio = value / weight;
            int rank = i - start;
            if (rank < 0)
              rank += number;
            if (ratio > best || (ratio == best && bestKey >= 0 && rank < bestKey)) {
              best = ratio;
              bestKey = rank;
              bestRow = iRow;
            }
          }
        }
      }
    } else {
      int iSequence = D.pivotVariable[i];
      double value = D.sol[iSequence];
      double infeas = fmax(value - D.upper[iSequence], D.lower[iSequence] - value);
      if (infeas > tolerance && !(D.status[iSequence] & FLAGGED_BIT)) {
        if (infeas > best || (infeas == best && bestKey >= 0 && i < bestKey)) {
          best = infeas;
          bestKey = i;
          bestRow = i;
        }
      }
    }
  }
  // block argmax carries the scan rank as key; recover the row through a second shared slot
  __shared__ int shRow[16];
  {
    int lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    for (int o = 32; o > 0; o >>= 1) {
      double ov = __shfl_down(best, o);
      int ok = __shfl_down(bestKey, o);
      int orow = __shfl_down(bestRow, o);
      if (ok >= 0 && (bestKey < 0 || ov > best || (ov == best && ok < bestKey))) {
        best = ov;
        bestKey = ok;
        bestRow = orow;
      }
    }
    if (lane == 0) {
      shv[wv] = best;
      shk[wv] = bestKey;
      shRow[wv] = bestRow;
    }
    __syncthreads();
    if (tid == 0) {
      for (int i = 1; i < nw; i++) {
        if (shk[i] >= 0 && (bestKey < 0 || shv[i] > best || (shv[i] == best && shk[i] < bestKey))) {
          best = shv[i];
          bestKey = shk[i];
          bestRow = shRow[i];
        }
      }
    }
  }
  if (tid == 0) {
    int chosen = bestRow;
    c->pivotRow = chosen;
    if (chosen < 0) {
      c->state = EXIT_NO_PIVOT_ROW;
    } else {
      int seqOut = D.pivotVariable[chosen];
      c->sequenceOut = seqOut;
      double valueOut = D.sol[seqOut], lowerOut = D.lower[seqOut], upperOut = D.upper[seqOut];
      c->valueOut = valueOut;
      c->lowerOut = lowerOut;
      c->upperOut = upperOut;
      if (valueOut > upperOut) {
        c->directionOut = -1;
        c->dualOut = valueOut - upperOut;
      } else if (valueOut < lowerOut) {
        c->directionOut = 1;
        c->dualOut = lowerOut - valueOut;
      } else if (valueOut - lowerOut < upperOut - valueOut) {
        c->directionOut = 1;
        c->dualOut = lowerOut - valueOut;
      } else {
        c->directionOut = -1;
        c->dualOut = valueOut - upperOut;
      }
      // acceptablePivot (:1270-1278)
      double acceptablePivot = 1.0e-1 * c->acceptablePivotBase;
      if (c->numberIterations > 100)
        acceptablePivot = c->acceptablePivotBase;
      if (c->pivots > 10 || (c->pivots && c->saveSumDual != 0.0))
        acceptablePivot = 1.0e+3 * c->acceptablePivotBase;
      else if (c->pivots > 5)
        acceptablePivot = 1.0e+2 * c->acceptablePivotBase;
      else if (c->pivots)
        acceptablePivot = c->acceptablePivotBase;
      c->acceptablePivot = acceptablePivot;
      D.vecC[chosen] = (double)c->directionOut;  // BTRAN input: directionOut * e_r (:1286)
      c->sequenceIn = -1;
      c->numberFlips = 0;
      c->objectiveChange = 0.0;
    }
  }
}

// =============================================================================================
// BTRAN  y = B^-T c  for the nucleus representation  B^-1 = [slack part | Minv] :
//   y_i   = -c[pos(slack i)]                         rows whose slack is basic  (slack column -e_i)
//   t_sc  = c[pos(col sc)] - sum_{i in S} a_{i,col} y_i
//   y_R   = Minv^T t
// Stands in for ClpFactorization::updateColumnTranspose (src/ClpFactorization.cpp:2993) ->
// CoinAbcDenseFactorization::updateColumnTranspose (src/CoinAbcDenseFactorization.cpp:634).
// =============================================================================================
__global__ void k_btran_slack(Dev D, const double *cvec, double *y, int iter)
{
  if (iter && D.ctrl->state != RUN)
    return;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D.m) {
    int p = D.posOfSlack[i];
    y[i] = (p >= 0) ? cvec[p] * -1.0 : 0.0;
  }
}

__global__ void k_btran_t(Dev D, const double *cvec, const double *y, double *t, int iter)
{
  if (iter && D.ctrl->state != RUN)
    return;
  int sc = blockIdx.x * blockDim.x + threadIdx.x;
  if (sc < D.ctrl->k) {
    int col = D.slotCol[sc];
    double value = cvec[D.slotPos[sc]];
    for (int p = D.colStart[col]; p < D.colStart[col + 1]; p++) {
      int r = D.row[p];
      if (D.slotOfRow[r] < 0)
        value -= y[r] * D.elem[p];
    }
    t[sc] = value;
  }
}

// partial[chunk][sr] = sum_{sc in chunk (64 rows), ascending} Minv[sc][sr] * t[sc]
__global__ void k_gemvT_partial(Dev D, const double *t, int iter)
{
  if (iter && D.ctrl->state != RUN)
    return;
  const int k = D.ctrl->k;
  int sr = blockIdx.x * blockDim.x + threadIdx.x;
  int chunk = blockIdx.y;
  int sc0 = chunk * 64;
  if (sc0 >= k || sr >= k)
    return;
  int sc1 = min(sc0 + 64, k);
  double acc = 0.0;
  const double *Mp = D.Minv + (size_t)sc0 * D.ld + sr;
  for (int sc = sc0; sc < sc1; sc++) {
    acc += *Mp * t[sc];
    Mp += D.ld;
  }
  D.partial[(size_t)chunk * D.ld + sr] = acc;
}

// y[slotRow[sr]] = sum_chunks partial; mode 1 = iteration BTRAN: flush tiny, fill rho/piNeg/rhoSlot
__global__ void k_gemvT_final(Dev D, double *y, int mode, int iter)
{
  if (iter && D.ctrl->state != RUN)
    return;
  const int k = D.ctrl->k;
  int sr = blockIdx.x * blockDim.x + threadIdx.x;
  if (sr >= k)
    return;
  int nchunk = (k + 63) >> 6;
  double acc = 0.0;
  for (int ch = 0; ch < nchunk; ch++)
    acc += D.partial[(size_t)ch * D.ld + sr];
  if (mode == 1)
    D.rhoSlot[sr] = acc;
  y[D.slotRow[sr]] = acc;
}

// flush |rho| <= zeroTolerance (the packed BTRAN result drops them) and build piNeg = -rho
__global__ void k_rho_finish(Dev D)
{
  if (D.ctrl->state != RUN)
    return;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D.m) {
    double v = D.rho[i];
    if (fabs(v) <= D.ctrl->zeroTolerance)
      v = 0.0;
    D.rho[i] = v;
    D.piNeg[i] = -v;
  }
}

// =============================================================================================
// Row pricing by column, fused with the first ratio-test pass.
//   ClpPackedMatrix::transposeTimesByColumn  src/ClpPackedMatrix.cpp:1007-1090 (row part, pi negate)
//   ClpPackedMatrix::gutsOfTransposeTimesUnscaled (fused variant) :1799-1993 (column part)
// key space: [0,m) rows (slacks), [m,m+n) columns; one thread per key, 256 keys per workgroup.
// v1 kernel: one lane walks one column sequentially (bit-identical summation order).
// Writes alphaCol[j], candFlag[key], per-block candidate count and per-block min ratio.
// =============================================================================================
#define PRICE_BLOCK 256
__global__ void __launch_bounds__(PRICE_BLOCK) k_price(Dev D, int nbRows)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double shd[16];
  __shared__ int shi[17];
  const double dualT = -c->dualTolerance;
  const double acceptablePivot = c->acceptablePivot;
  const double zeroTolerance = c->zeroTolerance;
  const double tentativeTheta = 1.0e15;  // ClpPackedMatrix.cpp:1857
  int flag = 0;
  double ratio = 1.0e31;
  double bytes = 0.0;
  if ((int)blockIdx.x < nbRows) {
    int i = blockIdx.x * PRICE_BLOCK + threadIdx.x;
    if (i < D.m) {
      double value = D.rho[i];
      if (value != 0.0) {
        int iStatus = (D.status[D.n + i] & 3) - 1;
        if (iStatus > 0) {
          double mult = (iStatus == 1) ? -1.0 : 1.0;
          double alpha = value * mult;
          if (alpha > 0.0) {
            double oldValue = D.dj[D.n + i] * mult;
            double v2 = oldValue - tentativeTheta * alpha;
            if (v2 < dualT) {
              flag = 1;
              if (alpha >= acceptablePivot)
                ratio = (oldValue - dualT) / alpha;
            }
          }
        }
      }
      D.candFlag[i] = (unsigned char)flag;
    }
  } else {
    int j = D.firstColumn + ((int)blockIdx.x - nbRows) * PRICE_BLOCK + threadIdx.x;
    if (j < D.lastColumn) {
      int wanted = (D.status[j] & 3) - 1;
      double value = 0.0;
      if (wanted) {
        const int start = D.colStart[j], end = D.colStart[j + 1];
        for (int p = start; p < end; p++)
          value += D.piNeg[D.row[p]] * D.elem[p];
        bytes = 12.0 * (end - start) + 4.0;
        if (fabs(value) > zeroTolerance) {
          bytes += 20.0;
          if (wanted > 0) {
            double mult = (wanted == 1) ? -1.0 : 1.0;
            double alpha = value * mult;
            if (alpha > 0.0) {
              double oldValue = D.dj[j] * mult;
              double v2 = oldValue - tentativeTheta * alpha;
              if (v2 < dualT) {
                flag = 1;
                if (alpha >= acceptablePivot)
                  ratio = (oldValue - dualT) / alpha;
              }
            }
          }
        } else {
          value = 0.0;
        }
      }
      D.alphaCol[j] = value;
      D.candFlag[D.m + j] = (unsigned char)flag;
    }
  }
  int total;
  blockRank(flag, total, shi);
  double bmin = blockMin(ratio, shd);
  double bsum = blockSum(bytes, shd);
  if (threadIdx.x == 0) {
    D.blockCount[blockIdx.x] = total;
    D.blockMin[blockIdx.x] = bmin;
    D.blockSum[blockIdx.x] = bsum;
  }
}

// exclusive scan over per-block counts (<= 1M/256 blocks), min over per-block ratios
// what: 0 candidates (-> numberCandidates, upperTheta), 1 flips, 2 infeasibility-list appends
__global__ void __launch_bounds__(1024) k_scan_blocks(Dev D, int nb, int what, int iter, int nSell = 0)
{
  Ctrl *c = D.ctrl;
  if (iter && c->state != RUN)
    return;
  __shared__ int shi[17];
  __shared__ double shd[16];
  __shared__ int s_base;
  if (threadIdx.x == 0)
    s_base = 0;
  __syncthreads();
  double vmin = 1.0e31;
  double bytes = 0.0;
  for (int b0 = 0; b0 < nb; b0 += blockDim.x) {
    int b = b0 + threadIdx.x;
    int cnt = (b < nb) ? D.blockCount[b] : 0;
    if (what == 0 && b < nb) {
      vmin = fmin(vmin, D.blockMin[b]);
      bytes += D.blockSum[b];
    }
    // inclusive scan inside the block via wave ballots is for flags only; counts need a real scan
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int v = cnt;
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(v, o);
      if (lane >= o)
        v += t;
    }
    __syncthreads();
    if (lane == 63)
      shi[wv] = v;
    __syncthreads();
    int base = s_base;
    for (int i = 0; i < wv; i++)
      base += shi[i];
    if (b < nb)
      D.blockOffset[b] = base + v - cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int i = 0; i < nw; i++)
        tot += shi[i];
      s_base += tot;
    }
    __syncthreads();
  }
  if (what == 0) {
    for (int b = threadIdx.x; b < nSell; b += blockDim.x) {
      vmin = fmin(vmin, D.sellMin[b]);
      bytes += D.sellBytes[b];
    }
    vmin = blockMin(vmin, shd);
    bytes = blockSum(bytes, shd);
  }
  if (threadIdx.x == 0) {
    if (what == 0) {
      c->numberCandidates = s_base;
      c->upperTheta = vmin;
      c->classCount[0] = c->classCount[1] = c->classCount[2] = 0;
      // algorithmic bytes of this pricing launch: per scanned column 12*len+4 (+20 per
      // emitted nonzero), plus status 1*n, pi 8*m, one extra colStart
      c->statPriceBytes += bytes + (double)(D.lastColumn - D.firstColumn) + 8.0 * D.m + 4.0;
      c->statPriceLaunches += 1.0;
    } else if (what == 1) {
      c->numberFlips = s_base;
    } else {
      c->numberAppend = s_base;
    }
  }
}

__global__ void __launch_bounds__(PRICE_BLOCK) k_cand_scatter(Dev D, int nbRows)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ int shi[17];
  int flag = 0, seq = -1;
  double alpha = 0.0;
  if ((int)blockIdx.x < nbRows) {
    int i = blockIdx.x * PRICE_BLOCK + threadIdx.x;
    if (i < D.m && D.candFlag[i]) {
      flag = 1;
      seq = D.n + i;
      alpha = D.rho[i];
    }
  } else {
    int j = D.firstColumn + ((int)blockIdx.x - nbRows) * PRICE_BLOCK + threadIdx.x;
    if (j < D.lastColumn && D.candFlag[D.m + j]) {
      flag = 1;
      seq = j;
      alpha = D.alphaCol[j];
    }
  }
  int total;
  int rank = blockRank(flag, total, shi);
  int cls = 3;
  if (flag) {
    int o = D.blockOffset[blockIdx.x] + rank;
    D.candSeq[o] = seq;
    D.candAlpha[o] = alpha;
    // breakpoint of the coarse ratio passes (ClpSimplexDual.cpp:4384 / :4412) against theta0
    const double tol = c->dualTolerance;
    const double djv = D.dj[seq];
    const double x = (alpha < 0.0) ? (djv - tol) / alpha : (djv + tol) / alpha;
    const double theta0 = fmax(10.0 * c->upperTheta, 1.0e-7);
    cls = (x <= theta0 * 8.0) ? 0 : ((x <= theta0 * 256.0) ? 1 : ((x <= theta0 * 16384.0) ? 2 : 3));
    D.candLive[o] = (unsigned char)cls;
  }
  // per-class totals (integer atomics: order independent)
  for (int j = 0; j < 3; j++) {
    unsigned long long mk = __ballot(cls == j);
    if ((threadIdx.x & 63) == 0 && mk)
      atomicAdd(&D.ctrl->classCount[j], (int)__popcll(mk));
  }
}

// =============================================================================================
// Dual ratio test -- ClpSimplexDual::dualColumn  src/ClpSimplexDual.cpp:4192-4927 (bound-flipping
// long-step test, pass 0 already fused into pricing: the spareIntArray_[0]==-2 path :4273-4281).
// One workgroup walks the candidate list once per pass.  The reference's two ping-pong lists are
// represented by per-candidate state: live[i] (still in the "remaining" list) and tag[i] (id of the
// pass that moved it to a "swapped" list); sid[a] is the id of the swapped set held by list a.
// List order (needed only for "first largest |alpha| wins", :4533) is candidate index order.
// =============================================================================================
struct DcAcc {
  double thru, incr, ut, sumBad, bestPivot;
  int bestIdx;
};
// one combined block reduction (sum, sum, min, sum, argmax-first) with a single LDS exchange
__device__ inline void dcReduce(DcAcc &a, double (*shd)[16], int *shk)
{
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    a.thru += __shfl_down(a.thru, o);
    a.incr += __shfl_down(a.incr, o);
    a.sumBad += __shfl_down(a.sumBad, o);
    a.ut = fmin(a.ut, __shfl_down(a.ut, o));
    double ov = __shfl_down(a.bestPivot, o);
    int ok = __shfl_down(a.bestIdx, o);
    if (ok >= 0 && (a.bestIdx < 0 || ov > a.bestPivot || (ov == a.bestPivot && ok < a.bestIdx))) {
      a.bestPivot = ov;
      a.bestIdx = ok;
    }
  }
  __syncthreads();
  if (lane == 0) {
    shd[0][wv] = a.thru;
    shd[1][wv] = a.incr;
    shd[2][wv] = a.ut;
    shd[3][wv] = a.sumBad;
    shd[4][wv] = a.bestPivot;
    shk[wv] = a.bestIdx;
  }
  __syncthreads();
  a.thru = a.incr = a.sumBad = 0.0;
  a.ut = shd[2][0];
  a.bestPivot = shd[4][0];
  a.bestIdx = shk[0];
  for (int i = 0; i < nw; i++) {
    a.thru += shd[0][i];
    a.incr += shd[1][i];
    a.sumBad += shd[3][i];
    a.ut = fmin(a.ut, shd[2][i]);
    if (i && shk[i] >= 0 && (a.bestIdx < 0 || shd[4][i] > a.bestPivot || (shd[4][i] == a.bestPivot && shk[i] < a.bestIdx))) {
      a.bestPivot = shd[4][i];
      a.bestIdx = shk[i];
    }
  }
}

// CPT > 0: every thread keeps CPT candidates (alpha, dj, range, state) in registers, a pass is
// pure ALU + one block reduction; CPT == 0: candidates stay in global memory (very long rows).
// all-lanes result of the combined reduction inside one wave (no LDS, no barrier)
__device__ inline void dcReduceWave(DcAcc &a)
{
  for (int o = 32; o > 0; o >>= 1) {
    a.thru += __shfl_down(a.thru, o);
    a.incr += __shfl_down(a.incr, o);
    a.sumBad += __shfl_down(a.sumBad, o);
    a.ut = fmin(a.ut, __shfl_down(a.ut, o));
    double ov = __shfl_down(a.bestPivot, o);
    int ok = __shfl_down(a.bestIdx, o);
    if (ok >= 0 && (a.bestIdx < 0 || ov > a.bestPivot || (ov == a.bestPivot && ok < a.bestIdx))) {
      a.bestPivot = ov;
      a.bestIdx = ok;
    }
  }
  a.thru = __shfl(a.thru, 0);
  a.incr = __shfl(a.incr, 0);
  a.sumBad = __shfl(a.sumBad, 0);
  a.ut = __shfl(a.ut, 0);
  a.bestPivot = __shfl(a.bestPivot, 0);
  a.bestIdx = __shfl(a.bestIdx, 0);
}

template <int CPT, bool ONEWAVE, bool MAPPED = false>
__device__ bool dualColumnImpl(Dev D, const int *map = nullptr, int count = -1, double tauGuard = 0.0)
{
  Ctrl *c = D.ctrl;
  __shared__ double shd[5][16];
  __shared__ int shk[16];
  const int tid = threadIdx.x, nthr = ONEWAVE ? 64 : blockDim.x;
  const int nc = MAPPED ? count : c->numberCandidates;  // MAPPED: a prefiltered working set (see k_dual_column)
  const double acceptablePivot = c->acceptablePivot;
  const double dualTolerance = c->dualTolerance;
  const double newTolerance = dualTolerance;
  const double absDualOut = fabs(c->dualOut);
  auto reduce = [&](DcAcc &a) {
    if constexpr (ONEWAVE)
      dcReduceWave(a);
    else
      dcReduce(a, shd, shk);
  };
  constexpr int R = CPT > 0 ? CPT : 1;
  double ra[R], rd[R], rr[R];
  int rt[R], ri[R];
  bool rl[R];
  if constexpr (CPT > 0) {
#pragma unroll
    for (int q = 0; q < R; q++) {
      int pos = tid + q * nthr;
      rl[q] = false;
      rt[q] = -1;
      ri[q] = -1;
      ra[q] = rd[q] = rr[q] = 0.0;
      if (pos < nc) {
        int i = MAPPED ? map[pos] : pos;  // original candidate index: the list order for ties
        ri[q] = i;
        int seq = D.candSeq[i];
        ra[q] = D.candAlpha[i];
        rd[q] = D.dj[seq];
        rr[q] = D.upper[seq] - D.lower[seq];
        rl[q] = true;
      }
    }
  } else {
    for (int i = tid; i < nc; i += nthr) {
      D.candLive[i] = 1;
      D.candTag[i] = -1;
    }
    __syncthreads();
  }
  auto forEach = [&](auto body) {
    if constexpr (CPT > 0) {
#pragma unroll
      for (int q = 0; q < R; q++) {
        if (ri[q] >= 0)
          body(ri[q], ra[q], rd[q], rr[q], rl[q], rt[q]);
      }
    } else {
      for (int i = tid; i < nc; i += nthr) {
        int seq = D.candSeq[i];
        bool live = D.candLive[i] != 0;
        int tag = D.candTag[i];
        body(i, D.candAlpha[i], D.dj[seq], D.upper[seq] - D.lower[seq], live, tag);
        D.candLive[i] = live ? 1 : 0;
        D.candTag[i] = tag;
      }
    }
  };
  double totalThru = 0.0, bestEverPivot = acceptablePivot, increaseInObjective = 0.0;
  int lastIdx = -1;
  double upperTheta = c->upperTheta;
  int modifyCosts = 0, badSumPivots = 0;
  int iFlip = 0;
  int sid[2] = { -1, -1 };
  int passId = 0;
  int seqIdx = -1;
  double theta = 1.0e50;
  double tentativeTheta = fmax(10.0 * upperTheta, 1.0e-7);
  const double lastPivot = 0.0;  // never updated in the reference either (:4207)
  while (tentativeTheta < 1.0e22) {
    // a prefiltered run is only valid while theta stays below the threshold every excluded
    // candidate is known to exceed; otherwise the caller redoes the test on the full list
    if (MAPPED && tentativeTheta >= tauGuard)
      return false;
    // ---- coarse pass (:4355-4417)
    DcAcc acc = { 0.0, 0.0, 1.0e50, 0.0, acceptablePivot, -1 };
    forEach([&](int i, double alpha, double oldValue, double range, bool &live, int &tag) {
      if (!live)
        return;
      double value = oldValue - tentativeTheta * alpha;
      if (alpha < 0.0) {
        if (value > newTolerance) {
          acc.thru -= range * alpha;
          acc.incr -= (oldValue + dualTolerance) * range;
          live = false;
          tag = passId;
          if (fabs(alpha) > acc.bestPivot) {
            acc.bestPivot = fabs(alpha);
            acc.bestIdx = i;
          }
        } else if (-alpha >= acceptablePivot) {
          acc.ut = fmin(acc.ut, (oldValue - newTolerance) / alpha);
        }
      } else {
        if (value < -newTolerance) {
          acc.thru += range * alpha;
          acc.incr += (oldValue - dualTolerance) * range;
          live = false;
          tag = passId;
          if (fabs(alpha) > acc.bestPivot) {
            acc.bestPivot = fabs(alpha);
            acc.bestIdx = i;
          }
        } else if (alpha >= acceptablePivot) {
          acc.ut = fmin(acc.ut, (oldValue + newTolerance) / alpha);
        }
      }
    });
    reduce(acc);
    double thruThis = acc.thru, increaseInThis = acc.incr, bestPivot = acc.bestPivot;
    int bestIdx = acc.bestIdx;
    upperTheta = acc.ut;
    if (bestIdx < 0)
      bestPivot = acceptablePivot;
    sid[1 - iFlip] = passId;
    double check = fabs(totalThru + thruThis);
    check += 1.0e-8 + 1.0e-10 * check;
    if (check >= absDualOut || increaseInObjective + increaseInThis < 0.0) {
      // ---- pivot in this batch: the list becomes the swapped set of this pass (:4427-4434)
      forEach([&](int, double, double, double, bool &live, int &tag) { live = (tag == passId); });
      if constexpr (CPT == 0)
        __syncthreads();
      int iTry;
      const int MAXTRY = 100;
      for (iTry = 0; iTry < MAXTRY; iTry++) {
        passId++;
        DcAcc a1 = { 0.0, 0.0, 1.0e50, 0.0, acceptablePivot, -1 };
        forEach([&](int, double alpha, double oldValue, double, bool &live, int &) {
          if (!live)
            return;
          if (alpha < 0.0) {
            if (-alpha >= acceptablePivot)
              a1.ut = fmin(a1.ut, (oldValue - newTolerance) / alpha);
          } else {
            if (alpha >= acceptablePivot)
              a1.ut = fmin(a1.ut, (oldValue + newTolerance) / alpha);
          }
        });
        reduce(a1);
        upperTheta = a1.ut;
        badSumPivots = 0;
        upperTheta *= 1.0000000001;
        DcAcc a2 = { 0.0, 0.0, 1.0e50, 0.0, acceptablePivot, -1 };
        forEach([&](int i, double alpha, double djv, double range, bool &live, int &tag) {
          if (!live)
            return;
          double value = djv - upperTheta * alpha;
          double badDj = 0.0;
          int addToSwapped = 0;
          if (alpha < 0.0) {
            if (value >= 0.0) {
              addToSwapped = 1;
              badDj = -djv - dualTolerance;
            }
          } else {
            if (value <= 0.0) {
              addToSwapped = 1;
              badDj = djv - dualTolerance;
            }
          }
          if (addToSwapped) {
            live = false;
            tag = passId;
            double absAlpha = fabs(alpha);
            if (absAlpha > a2.bestPivot) {
              a2.bestPivot = absAlpha;
              a2.bestIdx = i;
            }
            if (absAlpha < acceptablePivot && upperTheta < 1.0e20) {
              if (alpha < 0.0) {
                if (value > dualTolerance)
                  a2.sumBad += (range < 1.0e20) ? value * range : 1.0e20;
              } else {
                if (value < -dualTolerance)
                  a2.sumBad += (range < 1.0e20) ? -(value * range) : 1.0e20;
              }
            }
            a2.thru += range * fabs(alpha);
            a2.incr += badDj * range;
          }
        });
        reduce(a2);
        thruThis = a2.thru;
        increaseInThis = a2.incr;
        bestPivot = a2.bestPivot;
        bestIdx = a2.bestIdx;
        double sumBadPivots = a2.sumBad;
        if (bestIdx < 0)
          bestPivot = acceptablePivot;
        seqIdx = bestIdx;
        if (bestIdx >= 0)
          theta = D.dj[D.candSeq[bestIdx]] / D.candAlpha[bestIdx];
        if (sumBadPivots > 1.0e4) {
          if (c->pivots > 3) {
            badSumPivots = 1;
            break;
          }
        }
        sid[1 - iFlip] = passId;
        double increase = (absDualOut - totalThru) * theta;
        increase += increaseInObjective;
        if (theta < 0.0)
          thruThis += absDualOut;  // force using this one
        if (increaseInObjective < 0.0 && increase < 0.0 && lastIdx >= 0) {
          bestPivot = 0.0;
        } else {
          totalThru += thruThis;
          increaseInObjective += increaseInThis;
        }
        if (bestPivot < 0.1 * bestEverPivot && bestEverPivot > 1.0e-6 && (bestPivot < 1.0e-3 || totalThru * 2.0 > absDualOut)) {
          seqIdx = lastIdx;
          iFlip = 1 - iFlip;
          break;
        } else if (seqIdx == -1 && upperTheta > c->largeValue) {
          if (lastPivot > acceptablePivot) {
            seqIdx = lastIdx;
            iFlip = 1 - iFlip;
          }
          break;
        } else if (totalThru >= absDualOut) {
          modifyCosts = 1;
          break;
        } else {
          lastIdx = seqIdx;
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
      // ---- skip this lot (:4640-4657)
      if (bestPivot > 1.0e-3 || bestPivot > bestEverPivot) {
        bestEverPivot = bestPivot;
        lastIdx = bestIdx;
      } else {
        sid[1 - iFlip] = sid[iFlip];  // keep old swapped
      }
      increaseInObjective += increaseInThis;
      iFlip = 1 - iFlip;
      tentativeTheta = 2.0 * upperTheta;
      totalThru += thruThis;
      passId++;
    }
  }
  if (seqIdx < 0 && lastIdx >= 0) {
    seqIdx = lastIdx;
    iFlip = 1 - iFlip;
  }
  double minimumTheta = (c->upperOut > c->lowerOut) ? 1.0e-18 : 0.0;
  int sequenceIn = -1;
  double alphaIn = 0.0;
  if (seqIdx >= 0) {
    iFlip = 1 - iFlip;
    alphaIn = D.candAlpha[seqIdx];
    sequenceIn = D.candSeq[seqIdx];
    double oldValue = D.dj[sequenceIn];
    theta = fmax(oldValue / alphaIn, 0.0);
    if (theta < minimumTheta && fabs(alphaIn) < 1.0e5)
      theta = minimumTheta;
    if (modifyCosts && !badSumPivots) {
      // cost shifting so everything that went through stays dual feasible (:4705-4772)
      const int sidFinal = sid[iFlip];
      int changed = 0;
      forEach([&](int i, double alpha, double djv, double, bool &, int &tag) {
        if (tag != sidFinal)
          return;
        int iSequence = D.candSeq[i];
        double value = djv - theta * alpha;
        if (alpha < 0.0) {
          if (value > dualTolerance) {
            double modification = alpha * theta - djv + newTolerance;
            D.dj[iSequence] = djv + modification;
            D.cost[iSequence] += modification;
            if (modification != 0.0)
              changed++;
          }
        } else {
          if (-value > dualTolerance) {
            double modification = alpha * theta - djv - newTolerance;
            D.dj[iSequence] = djv + modification;
            D.cost[iSequence] += modification;
            if (modification != 0.0)
              changed++;
          }
        }
      });
      DcAcc a3 = { (double)changed, 0.0, 1.0e50, 0.0, 0.0, -1 };
      reduce(a3);
      if (tid == 0)
        c->numberChanged += (int)a3.thru;
    }
  }
  if (badSumPivots && c->pivots) {
    sequenceIn = -1;
    if (tid == 0)
      c->acceptablePivotBase = -c->acceptablePivotBase;
  }
  if (sequenceIn >= D.n) {
    if (tid == 0)
      D.vecV1[sequenceIn - D.n] = -1.0;
  } else if (sequenceIn >= 0) {
    for (int p = D.colStart[sequenceIn] + tid; p < D.colStart[sequenceIn + 1]; p += nthr)
      D.vecV1[D.row[p]] = D.elem[p];
  }
  if (tid == 0) {
    c->badSumPivots = badSumPivots;
    c->modifyCosts = modifyCosts;
    if (sequenceIn >= 0) {
      c->sequenceIn = sequenceIn;
      c->alpha = alphaIn;
      c->theta = theta;
      double lowerIn = D.lower[sequenceIn], upperIn = D.upper[sequenceIn], valueIn = D.sol[sequenceIn];
      double dualIn = D.dj[sequenceIn];
      // modify cost so the incoming dj is exactly theta*alpha (:4796-4834)
      double modification = theta * alphaIn - dualIn;
      double moveObjective = fabs(modification * valueIn);
      double smallMove = fmax(fabs(c->objectiveValue), 1.0e-3);
      if (moveObjective > smallMove)
        modification *= smallMove / moveObjective;
      if (badSumPivots)
        modification = 0.0;
      dualIn += modification;
      D.dj[sequenceIn] = dualIn;
      D.cost[sequenceIn] += modification;
      if (modification != 0.0)
        c->numberChanged++;
      c->dualIn = dualIn;
      c->valueIn = valueIn;
      if (alphaIn < 0.0) {
        c->directionIn = -1;
        upperIn = valueIn;
      } else {
        c->directionIn = 1;
        lowerIn = valueIn;
      }
      c->lowerIn = lowerIn;
      c->upperIn = upperIn;
      c->bestPossible = fabs(alphaIn);
      c->btranAlpha = -alphaIn * c->directionOut;
    } else {
      c->sequenceIn = -1;
      c->alpha = 0.0;
      c->bestPossible = 0.0;
      c->state = EXIT_NO_INCOMING;
    }
  }
  return true;
}

#define DC_CPT 4
#define DC_SMALL (8 * 64)
// typical sparse tableau row: one wave, candidates in registers, shuffle-only reductions
__global__ void __launch_bounds__(64) k_dual_column_small(Dev D)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int nc = c->numberCandidates;
  if (!nc) {
    if (threadIdx.x == 0) {
      c->sequenceIn = -1;
      c->alpha = 0.0;
      c->bestPossible = 0.0;
      c->state = EXIT_NO_INCOMING;
    }
    return;
  }
  if (nc <= 4 * 64)
    dualColumnImpl<4, true>(D);
  else if (nc <= DC_SMALL)
    dualColumnImpl<8, true>(D);
}
// Long candidate lists (dense tableau rows, up to ~n/2 entries).  Only the few dozen candidates
// with the smallest breakpoints ever take part in the passes; the rest only bound theta from above.
// k_cand_scatter classified every candidate by its breakpoint against theta0 * {2^3, 2^8, 2^14};
// the largest class prefix that fits in registers becomes the working set, the test runs on it, and
// is repeated on the full list only if theta ever reaches the class threshold (exact either way).
#define DC_WS_CAP (DC_CPT * 1024)
__global__ void __launch_bounds__(1024) k_dual_column(Dev D)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int nc = c->numberCandidates;
  if (nc <= DC_SMALL)
    return;  // handled (or rejected) by k_dual_column_small
  __shared__ int wsIdx[DC_WS_CAP];
  __shared__ int shw[17];
  __shared__ int s_done;
  const int tid = threadIdx.x;
  int cum[3];
  cum[0] = c->classCount[0];
  cum[1] = cum[0] + c->classCount[1];
  cum[2] = cum[1] + c->classCount[2];
  int J = -1;
  for (int j = 0; j < 3; j++)
    if (cum[j] <= DC_WS_CAP && cum[j] < nc)
      J = j;
  if (J >= 0 && cum[J] > 0) {
    const int ws = cum[J];
    const double theta0 = fmax(10.0 * c->upperTheta, 1.0e-7);
    const double tau = theta0 * (J == 0 ? 8.0 : (J == 1 ? 256.0 : 16384.0));
    // ordered compaction of the working set (thread t owns a contiguous slice of the list)
    const int per = (nc + (int)blockDim.x - 1) / (int)blockDim.x;
    const int lo = min(nc, tid * per), hi = min(nc, lo + per);
    int cnt = 0;
    for (int i = lo; i < hi; i++)
      cnt += (D.candLive[i] <= J);
    const int lane = tid & 63, wv = tid >> 6;
    int v = cnt;
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(v, o);
      if (lane >= o)
        v += t;
    }
    if (lane == 63)
      shw[wv] = v;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wv; i++)
      base += shw[i];
    int o = base + v - cnt;
    for (int i = lo; i < hi; i++)
      if (D.candLive[i] <= J)
        wsIdx[o++] = i;
    if (tid == 0)
      s_done = 0;
    __syncthreads();
    bool ok;
    if (ws <= DC_SMALL) {
      ok = true;
      if (tid < 64) {
        ok = dualColumnImpl<8, true, true>(D, wsIdx, ws, tau);
        if (tid == 0)
          s_done = ok ? 1 : 0;
      }
      __syncthreads();
      ok = s_done != 0;
    } else {
      ok = dualColumnImpl<DC_CPT, false, true>(D, wsIdx, ws, tau);
    }
    if (ok)
      return;
    __syncthreads();
  }
  if (nc <= DC_CPT * (int)blockDim.x) {
    dualColumnImpl<DC_CPT, false>(D);
  } else {
    dualColumnImpl<0, false>(D);
  }
}

// =============================================================================================
// FTRAN  x = B^-1 v :  x_K = Minv v_R ;  x[pos(slack i)] = sum_{j in K} a_ij x_j - v_i
// Stands in for ClpFactorization::updateColumn / updateTwoColumnsFT (src/ClpFactorization.cpp:2803,
// :2889) -> CoinAbcDenseFactorization::updateColumn (src/CoinAbcDenseFactorization.cpp:571).
// Two right-hand sides share one sweep over Minv (the entering column and the DSE vector).
// =============================================================================================
__global__ void k_unpack_in(Dev D)
{
  // ClpSimplex::unpackPacked (src/ClpSimplex.cpp:3439-3495): a_q, or -e_i for a slack
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  int q = c->sequenceIn;
  if (q >= D.n) {
    if (threadIdx.x == 0)
      D.vecV1[q - D.n] = -1.0;
  } else {
    for (int p = D.colStart[q] + threadIdx.x; p < D.colStart[q + 1]; p += blockDim.x)
      D.vecV1[D.row[p]] = D.elem[p];
  }
}

__global__ void k_ftran_gather(Dev D, const double *v1, const double *v2, double *g1, double *g2, int iter)
{
  if (iter && D.ctrl->state != RUN)
    return;
  if (iter == 2 && D.ctrl->numberFlips == 0)
    return;
  int sr = blockIdx.x * blockDim.x + threadIdx.x;
  if (sr < D.ctrl->k) {
    int r = D.slotRow[sr];
    g1[sr] = v1[r];
    if (v2)
      g2[sr] = v2[r];
  }
}

// one wave per nucleus row: x[sc] = sum_sr Minv[sc][sr] * g[sr]; lanes stride the row (coalesced)
__global__ void __launch_bounds__(256) k_gemv2(Dev D, const double *g1, const double *g2, double *x1, double *x2, int iter)
{
  if (iter && D.ctrl->state != RUN)
    return;
  if (iter == 2 && D.ctrl->numberFlips == 0)
    return;
  const int k = D.ctrl->k;
  const int lane = threadIdx.x & 63;
  const int wavesPerBlock = blockDim.x >> 6;
  for (int sc = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6); sc < k; sc += gridDim.x * wavesPerBlock) {
    const double *Mrow = D.Minv + (size_t)sc * D.ld;
    double a1 = 0.0, a2 = 0.0;
    if (g2) {
      for (int sr = lane; sr < k; sr += 64) {
        double mv = Mrow[sr];
        a1 += mv * g1[sr];
        a2 += mv * g2[sr];
      }
      a2 = waveSum(a2);
    } else {
      for (int sr = lane; sr < k; sr += 64)
        a1 += Mrow[sr] * g1[sr];
    }
    a1 = waveSum(a1);
    if (lane == 0) {
      x1[sc] = a1;
      if (g2)
        x2[sc] = a2;
    }
  }
}

// scatter nucleus results to basis positions and do the slack rows through the partitioned row copy
__global__ void k_ftran_scatter(Dev D, const double *v1, const double *v2, const double *xk1, const double *xk2, double *x1,
                                double *x2, int iter)
{
  if (iter && D.ctrl->state != RUN)
    return;
  if (iter == 2 && D.ctrl->numberFlips == 0)
    return;
  const int k = D.ctrl->k;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < D.m) {
    int p = D.posOfSlack[t];
    if (p >= 0) {
      double a1 = 0.0, a2 = 0.0;
      int s = D.rowStart[t], e = s + D.basicCount[t];
      for (int q = s; q < e; q++) {
        int sc = D.slotOfCol[D.ccol[q]];
        double a = D.relem[q];
        a1 += a * xk1[sc];
        if (v2)
          a2 += a * xk2[sc];
      }
      x1[p] = a1 - v1[t];
      if (v2)
        x2[p] = a2 - v2[t];
    }
  } else if (t < D.m + k) {
    int sc = t - D.m;
    int p = D.slotPos[sc];
    x1[p] = xk1[sc];
    if (v2)
      x2[p] = xk2[sc];
  }
}

// =============================================================================================
// DSE: norm, alpha check and weight update -- ClpDualRowSteepest::updateWeights
// (src/ClpDualRowSteepest.cpp:375-540) and the accuracy test of whileIterating (:1447-1501).
// =============================================================================================
__global__ void __launch_bounds__(1024) k_norm_alpha(Dev D)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double sh[16];
  double acc = 0.0;
  if (c->pivotRule) {
    for (int i = threadIdx.x; i < D.m; i += blockDim.x) {
      double v = D.rho[i];
      acc += v * v;
    }
    acc = blockSum(acc, sh);
  }
  if (threadIdx.x == 0) {
    double alphaOld = c->alpha;  // from the ratio test (btran side)
    double norm = acc / (alphaOld * alphaOld);
    c->norm = norm;
    double alpha = D.w[c->pivotRow];
    double btranAlpha = c->btranAlpha;
    double checkValue = 1.0e-7;
    if (c->largestPrimalError > 10.0)
      checkValue = fmin(1.0e-4, 1.0e-8 * c->largestPrimalError);
    // multiplier uses the old alpha (model_->alpha() inside updateWeights)
    c->scratchSum = 2.0 / alphaOld;
    if (fabs(btranAlpha) < 1.0e-12 || fabs(alpha) < 1.0e-12 || fabs(btranAlpha - alpha) > checkValue * (1.0 + fabs(alpha))) {
      int bad = 1;
      if (!c->pivots) {
        double test;
        if (fabs(btranAlpha) < 1.0e-8 || fabs(alpha) < 1.0e-8)
          test = 1.0e-1 * fabs(alpha);
        else
          test = 1.0e-4 * (1.0 + fabs(alpha));
        if (!(fabs(btranAlpha) < 1.0e-12 || fabs(alpha) < 1.0e-12 || fabs(btranAlpha - alpha) > test))
          bad = 0;  // accepted under the relaxed criterion (:1466-1471)
      }
      if (bad) {
        c->alpha = alpha;
        c->state = EXIT_ALPHA_CHECK;
        return;
      }
    }
    c->alpha = alpha;
  }
}

__global__ void k_weights(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN || !c->pivotRule)
    return;
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= D.m)
    return;
  double theta = D.w[p];
  if (theta != 0.0) {
    double devex = D.weights[p];
    D.altWeights[p] = devex;
    double norm = c->norm, multiplier = c->scratchSum;
    if (p == c->pivotRow) {
      devex = (norm < DEVEX_TRY_NORM) ? DEVEX_TRY_NORM : norm;
    } else {
      double value = D.tau[p];
      devex += theta * (theta * norm + value * multiplier);
      if (devex < DEVEX_TRY_NORM)
        devex = DEVEX_TRY_NORM;
    }
    D.weights[p] = devex;
  }
}

// ClpDualRowSteepest::unrollWeights (:1022): w is still intact when the host asks for this
__global__ void k_unroll_weights(Dev D)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m && D.w[p] != 0.0)
    D.weights[p] = D.altWeights[p];
}

// =============================================================================================
// Dual update + flip detection -- ClpSimplexDual::updateDualsInDual fast path
// (src/ClpSimplexDual.cpp:2454-2592).  Key space as in pricing.  candFlag doubles as flip flag.
// =============================================================================================
__global__ void __launch_bounds__(PRICE_BLOCK) k_dj_update(Dev D, int nbRows)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ int shi[17];
  const double theta = c->theta;
  const double tolerance = c->dualTolerance + fmin(1.0e-2, c->largestDualError);
  const int seqIn = c->sequenceIn;
  int flag = 0;
  if ((int)blockIdx.x < nbRows) {
    int i = blockIdx.x * PRICE_BLOCK + threadIdx.x;
    if (i < D.m) {
      double alphaI = D.rho[i];
      int seq = D.n + i;
      if (alphaI != 0.0 && seq != seqIn) {
        int iStatus = (D.status[seq] & 3) - 1;
        if (iStatus) {
          double value = D.dj[seq] - theta * alphaI;
          D.dj[seq] = value;
          double mult = (iStatus == 1) ? -1.0 : ((iStatus == 2) ? 1.0 : 0.0);
          value *= mult;
          if (value < -tolerance)
            flag = 1;
        }
      }
      D.candFlag[i] = (unsigned char)flag;
    }
  } else {
    int j = D.firstColumn + ((int)blockIdx.x - nbRows) * PRICE_BLOCK + threadIdx.x;
    if (j < D.lastColumn) {
      double alphaI = D.alphaCol[j];
      if (alphaI != 0.0 && j != seqIn) {
        int iStatus = (D.status[j] & 3) - 1;
        if (iStatus) {
          double value = D.dj[j] - theta * alphaI;
          D.dj[j] = value;
          double mult = (iStatus == 1) ? -1.0 : ((iStatus == 2) ? 1.0 : -1.0);
          value *= mult;
          if (value < -tolerance && iStatus > 0)
            flag = 1;
        }
      }
      D.candFlag[D.m + j] = (unsigned char)flag;
    }
  }
  int total;
  blockRank(flag, total, shi);
  if (threadIdx.x == 0)
    D.blockCount[blockIdx.x] = total;
}

__global__ void __launch_bounds__(PRICE_BLOCK) k_flip_scatter(Dev D, int nbRows)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN || c->numberFlips == 0)
    return;
  __shared__ int shi[17];
  int flag = 0, seq = -1;
  if ((int)blockIdx.x < nbRows) {
    int i = blockIdx.x * PRICE_BLOCK + threadIdx.x;
    if (i < D.m && D.candFlag[i]) {
      flag = 1;
      seq = D.n + i;
    }
  } else {
    int j = D.firstColumn + ((int)blockIdx.x - nbRows) * PRICE_BLOCK + threadIdx.x;
    if (j < D.lastColumn && D.candFlag[D.m + j]) {
      flag = 1;
      seq = j;
    }
  }
  int total;
  int rank = blockRank(flag, total, shi);
  if (flag)
    D.flipSeq[D.blockOffset[blockIdx.x] + rank] = seq;
}

// movement of each flip into the dense rhs (matrix_->add, src/ClpPackedMatrix.cpp:4874), in list
// order, entries of one column in parallel (distinct rows) => deterministic
__global__ void __launch_bounds__(256) k_flip_apply(Dev D, int nbPos)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN || c->numberFlips == 0)
    return;
  double changeObj = 0.0;
  for (int f = 0; f < c->numberFlips; f++) {
    int seq = D.flipSeq[f];
    int iStatus = (D.status[seq] & 3) - 1;
    if (seq >= D.n) {
      double mult = (iStatus == 1) ? -1.0 : 1.0;
      double movement = mult * (D.lower[seq] - D.upper[seq]);
      if (threadIdx.x == 0) {
        changeObj -= movement * D.cost[seq];
        D.flipRhs[seq - D.n] += movement;
      }
    } else {
      double mult = (iStatus == 1) ? -1.0 : 1.0;
      double movement = mult * (D.upper[seq] - D.lower[seq]);
      if (threadIdx.x == 0)
        changeObj += movement * D.cost[seq];
      for (int p = D.colStart[seq] + threadIdx.x; p < D.colStart[seq + 1]; p += blockDim.x)
        D.flipRhs[D.row[p]] += movement * D.elem[p];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    c->objectiveChange += changeObj;
  for (int b = threadIdx.x; b < nbPos; b += blockDim.x)
    D.blockCount[b] = 0;
}

// ClpSimplexDual::flipBounds (:6345-6401)
__global__ void k_flip_bounds(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < c->numberFlips; f += gridDim.x * blockDim.x) {
    int seq = D.flipSeq[f];
    int st = D.status[seq] & 7;
    if (st == ST_UPPER) {
      D.status[seq] = (unsigned char)((D.status[seq] & ~7) | ST_LOWER);
      D.sol[seq] = D.lower[seq];
    } else if (st == ST_LOWER) {
      D.status[seq] = (unsigned char)((D.status[seq] & ~7) | ST_UPPER);
      D.sol[seq] = D.upper[seq];
    }
  }
}

// =============================================================================================
// Primal update -- ClpDualRowSteepest::updatePrimalSolution (src/ClpDualRowSteepest.cpp:630-763)
// x_B -= ratio * vec ; refresh squared infeasibilities ; new entries are appended to the list in
// ascending position order (count / scan / scatter keeps CoinIndexedVector's insertion order).
// which: 0 -> vec = w, ratio = ctrl.movement ; 1 -> vec = x3 (flip FTRAN), ratio = 1
// =============================================================================================
__global__ void __launch_bounds__(256) k_primal_update(Dev D, int which)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  if (which == 1 && c->numberFlips == 0)
    return;
  __shared__ double shd[16];
  __shared__ int shi[17];
  const double *vec = which ? D.x3 : D.w;
  const double ratio = which ? 1.0 : c->movement;
  const double tolerance = c->primalTolerance;
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (which == 0) {
    // ClpSimplexDual::flipBounds (:6345-6401), after the backwards check of the scalar block
    for (int f = p; f < c->numberFlips; f += gridDim.x * blockDim.x) {
      int seq = D.flipSeq[f];
      int st = D.status[seq] & 7;
      if (st == ST_UPPER) {
        D.status[seq] = (unsigned char)((D.status[seq] & ~7) | ST_LOWER);
        D.sol[seq] = D.lower[seq];
      } else if (st == ST_LOWER) {
        D.status[seq] = (unsigned char)((D.status[seq] & ~7) | ST_UPPER);
        D.sol[seq] = D.upper[seq];
      }
    }
  } else if (p < D.m) {
    D.flipRhs[p] = 0.0;  // consumed by the flip FTRAN
  }
  double changeObj = 0.0;
  int append = 0;
  if (p < D.m) {
    double v = vec[p];
    if (v != 0.0) {
      int iPivot = D.pivotVariable[p];
      double value = D.sol[iPivot];
      double change = ratio * v;
      value -= change;
      changeObj -= change * D.cost[iPivot];
      D.sol[iPivot] = value;
      if (c->pivotRule) {
        double lower = D.lower[iPivot], upper = D.upper[iPivot];
        double old = D.infeas[p];
        if (value < lower - tolerance) {
          value -= lower;
          value *= value;
          if (old == 0.0)
            append = 1;
          D.infeas[p] = value;
        } else if (value > upper + tolerance) {
          value -= upper;
          value *= value;
          if (old == 0.0)
            append = 1;
          D.infeas[p] = value;
        } else if (old != 0.0) {
          D.infeas[p] = REALLY_TINY;
        }
      }
    }
    D.appendFlag[p] = append;
  }
  int total;
  blockRank(append, total, shi);
  double s = blockSum(changeObj, shd);
  if (threadIdx.x == 0) {
    D.blockCount[blockIdx.x] = total;
    D.blockSum[blockIdx.x] = s;
  }
}

__global__ void __launch_bounds__(256) k_append_scatter(Dev D, int which, int iter)
{
  const Ctrl *c = D.ctrl;
  if ((iter && c->state != RUN) || c->numberAppend == 0)
    return;
  if (iter && which == 1 && c->numberFlips == 0)
    return;
  __shared__ int shi[17];
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int flag = (p < D.m) ? D.appendFlag[p] : 0;
  int total;
  int rank = blockRank(flag, total, shi);
  if (flag)
    D.infIndex[c->numberInfeasible + D.blockOffset[blockIdx.x] + rank] = p;
}

// scalar tail of updatePrimalSolution + the scalar block of whileIterating between the two
// primal updates (:1531-1588): objective change, dualOut recompute, movement, backwards check,
// the pivot-size gate of replaceColumn (CoinAbcDenseFactorization::checkReplacePart2 :470).
__global__ void k_after_primal(Dev D, int nb, int which)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  if (which == 1 && c->numberFlips == 0) {
    // no flips: just the scalar block
  } else {
    double s = 0.0;
    for (int b = 0; b < nb; b++)
      s += D.blockSum[b];
    c->objectiveChange += s;
    c->numberInfeasible += c->numberAppend;
    c->numberAppend = 0;
    if (c->pivotRule) {
      int iRow = c->pivotRow;
      if (D.infeas[iRow] != 0.0)
        D.infeas[iRow] = REALLY_TINY;
    }
  }
  if (which == 1) {
    double oldDualOut = c->dualOut;
    if (c->numberFlips) {
      c->valueOut = D.sol[c->sequenceOut];
      if (c->directionOut < 0)
        c->dualOut = c->valueOut - c->upperOut;
      else
        c->dualOut = c->lowerOut - c->valueOut;
    }
    double alpha = c->alpha;
    c->movement = -c->dualOut * c->directionOut / alpha;
    double movementOld = oldDualOut * c->directionOut / alpha;
    if (c->objectiveChange + fabs(movementOld * c->dualIn) < -fmax(1.0e-5, 1.0e-12 * fabs(c->objectiveValue))) {
      if (c->pivots) {
        c->state = EXIT_BACKWARDS;
        return;
      }
    }
    if (fabs(alpha) < c->zeroTolerance || fabs(c->dualOut) > 1.0e50) {
      c->state = EXIT_BAD_UPDATE;
      return;
    }
    if (c->theta < 0.0)
      c->theta = 0.0;
    // classify the basis change for the nucleus update
    int seqIn = c->sequenceIn, seqOut = c->sequenceOut;
    int inStruct = seqIn < D.n, outStruct = seqOut < D.n;
    c->updateCase = outStruct ? (inStruct ? 0 : 2) : (inStruct ? 1 : 3);
    c->slotColOut = outStruct ? D.slotOfCol[seqOut] : -1;
    c->rowOfSlackOut = outStruct ? -1 : (seqOut - D.n);
    c->slotRowIn = inStruct ? -1 : D.slotOfRow[seqIn - D.n];
  }
}

// =============================================================================================
// Basis update on the nucleus inverse (the Forrest-Tomlin stand-in,
// ClpFactorization::replaceColumn src/ClpFactorization.cpp:2584).  With w = B^-1 a_q (by col-slot),
// rho = B^-T(dir e_p) (by row-slot, unpruned) and g = dir*rho/alpha, all four pivot types are
//     Minv[i][j] -= w_i * g_j          (one rank-1 sweep, k^2 reads + writes)
// followed by a row/column fix-up:   struct->struct: row a := g ;  slack out/struct in: append row g,
// column w/alpha, corner -1/alpha ;  struct out/slack in: delete row a, column b ;  slack->slack:
// column b := w/alpha.
// =============================================================================================
__global__ void k_update_vectors(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < c->k) {
    D.slotE[s] = D.w[D.slotPos[s]];                             // w by col-slot
    D.slotF[s] = ((double)c->directionOut) * D.rhoSlot[s] / c->alpha;  // g by row-slot
  }
}

__global__ void __launch_bounds__(256) k_rank1(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int k = c->k;
  const double dir = (double)c->directionOut, alpha = c->alpha;
  // blockIdx.x * 256 + thread = column j (coalesced along the row), blockIdx.y strides rows
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < k; j += gridDim.x * blockDim.x) {
    const double gj = dir * D.rhoSlot[j] / alpha;
    for (int i = blockIdx.y; i < k; i += gridDim.y) {
      double wi = D.w[D.slotPos[i]];
      if (wi != 0.0)
        D.Minv[(size_t)i * D.ld + j] -= wi * gj;
    }
  }
}

__global__ void k_rank1_fix(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int k = c->k;
  const int ucase = c->updateCase;
  const double alpha = c->alpha;
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  double slotFs = 0.0, slotEs = 0.0;
  if (s < k) {
    slotFs = ((double)c->directionOut) * D.rhoSlot[s] / alpha;  // g by row-slot
    slotEs = D.w[D.slotPos[s]];                                  // w by col-slot
  }
  if (ucase == 0) {
    int a = c->slotColOut;
    if (s < k)
      D.Minv[(size_t)a * D.ld + s] = slotFs;
  } else if (ucase == 1) {
    if (s < k) {
      D.Minv[(size_t)k * D.ld + s] = slotFs;
      D.Minv[(size_t)s * D.ld + k] = slotEs / alpha;
    } else if (s == k) {
      D.Minv[(size_t)k * D.ld + k] = -1.0 / alpha;
    }
  } else if (ucase == 2) {
    // delete col-slot a (a matrix row) and row-slot b (a matrix column): move the last ones in
    int a = c->slotColOut, b = c->slotRowIn, last = k - 1;
    if (s < k) {
      // first the column move (within every row), then the row move; rows a/last handled once
      double vlast = D.Minv[(size_t)s * D.ld + last];
      if (b != last)
        D.Minv[(size_t)s * D.ld + b] = vlast;
    }
  } else {
    int b = c->slotRowIn;
    if (s < k)
      D.Minv[(size_t)s * D.ld + b] = slotEs / alpha;
  }
}
// second half of the delete: copy matrix row `last` over row a (after the column move)
__global__ void k_rank1_fix2(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN || c->updateCase != 2)
    return;
  const int k = c->k;
  int a = c->slotColOut, last = k - 1;
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (a != last && s < k)
    D.Minv[(size_t)a * D.ld + s] = D.Minv[(size_t)last * D.ld + s];
}

// =============================================================================================
// Housekeeping -- tail of whileIterating (:1679-1713, :1828-1829), ClpSimplex::housekeeping
// (src/ClpSimplex.cpp:2065-2489), basis bookkeeping of the nucleus, row-copy partition
// maintenance and the pivot log record.  One workgroup; entry loops are thread-parallel.
// =============================================================================================
__device__ inline void rowCopySwap(const Dev &D, int e, int b)
{
  if (e == b)
    return;
  int ce = D.ccol[e], cb = D.ccol[b];
  double ve = D.relem[e], vb = D.relem[b];
  int pe = D.csrToCsc[e], pb = D.csrToCsc[b];
  D.ccol[e] = cb;
  D.relem[e] = vb;
  D.csrToCsc[e] = pb;
  D.cscToCsr[pb] = e;
  D.ccol[b] = ce;
  D.relem[b] = ve;
  D.csrToCsc[b] = pe;
  D.cscToCsr[pe] = b;
}

__device__ void houseBody(Dev D)
{
  Ctrl *c = D.ctrl;
  const int tid = threadIdx.x;
  const int seqIn = c->sequenceIn, seqOut = c->sequenceOut, pivotRow = c->pivotRow;
  const int n = D.n;
  // ---- row copy partition: leaving structural goes to the nonbasic part, entering to the basic
  if (seqOut < n) {
    for (int p = D.colStart[seqOut] + tid; p < D.colStart[seqOut + 1]; p += blockDim.x) {
      int r = D.row[p];
      int e = D.cscToCsr[p];
      int b = D.rowStart[r] + D.basicCount[r] - 1;
      rowCopySwap(D, e, b);
      D.basicCount[r] -= 1;
    }
  }
  __syncthreads();
  if (seqIn < n) {
    for (int p = D.colStart[seqIn] + tid; p < D.colStart[seqIn + 1]; p += blockDim.x) {
      int r = D.row[p];
      int e = D.cscToCsr[p];
      int b = D.rowStart[r] + D.basicCount[r];
      rowCopySwap(D, e, b);
      D.basicCount[r] += 1;
    }
  }
  // ---- clear the sparse work vectors of this iteration
  if (seqIn < n) {
    for (int p = D.colStart[seqIn] + tid; p < D.colStart[seqIn + 1]; p += blockDim.x)
      D.vecV1[D.row[p]] = 0.0;
  } else if (tid == 0) {
    D.vecV1[seqIn - n] = 0.0;
  }
  __syncthreads();
  if (tid != 0)
    return;
  D.vecC[pivotRow] = 0.0;
  // ---- nucleus bookkeeping
  int k = c->k;
  const int ucase = c->updateCase;
  if (ucase == 0) {
    int a = c->slotColOut;
    D.slotOfCol[seqOut] = -1;
    D.slotOfCol[seqIn] = a;
    D.slotCol[a] = seqIn;  // position unchanged (== pivotRow)
  } else if (ucase == 1) {
    int r = c->rowOfSlackOut;  // its slack leaves the basis: row joins the nucleus
    D.posOfSlack[r] = -1;
    D.slotOfRow[r] = k;
    D.slotRow[k] = r;
    D.slotOfCol[seqIn] = k;
    D.slotCol[k] = seqIn;
    D.slotPos[k] = pivotRow;
    k++;
  } else if (ucase == 2) {
    int a = c->slotColOut, b = c->slotRowIn, last = k - 1;
    int rIn = seqIn - n;
    D.slotOfCol[seqOut] = -1;
    D.slotOfRow[rIn] = -1;
    D.posOfSlack[rIn] = pivotRow;
    if (a != last) {
      int colLast = D.slotCol[last];
      D.slotCol[a] = colLast;
      D.slotPos[a] = D.slotPos[last];
      D.slotOfCol[colLast] = a;
    }
    if (b != last) {
      int rowLast = D.slotRow[last];
      D.slotRow[b] = rowLast;
      D.slotOfRow[rowLast] = b;
    }
    k--;
  } else {
    int b = c->slotRowIn;
    int rIn = seqIn - n, rOut = c->rowOfSlackOut;
    D.posOfSlack[rOut] = -1;
    D.posOfSlack[rIn] = pivotRow;
    D.slotOfRow[rIn] = -1;
    D.slotOfRow[rOut] = b;
    D.slotRow[b] = rOut;
  }
  c->k = k;
  // ---- whileIterating :1687-1712
  double dualOut = c->dualOut / c->alpha;
  dualOut *= -c->directionOut;
  c->dualOut = dualOut;
  D.dj[seqIn] = 0.0;
  double oldValue = c->valueIn;
  double valueIn = (c->directionIn == -1) ? c->upperIn + dualOut : c->lowerIn + dualOut;
  c->valueIn = valueIn;
  double objectiveChange = c->objectiveChange + D.cost[seqIn] * (valueIn - oldValue);
  double valueOut;
  if (c->directionOut > 0) {
    valueOut = c->lowerOut;
    D.dj[seqOut] = c->theta;
  } else {
    valueOut = c->upperOut;
    D.dj[seqOut] = -c->theta;
  }
  c->valueOut = valueOut;
  D.sol[seqOut] = valueOut;
  // ---- housekeeping (ClpSimplex.cpp:2065-2140)
  c->numberIterations++;
  D.pivotVariable[pivotRow] = seqIn;
  D.sol[seqIn] = valueIn;
  unsigned char stIn = D.status[seqIn], stOut = D.status[seqOut];
  if (seqIn != seqOut) {
    stIn = (unsigned char)((stIn & ~7) | ST_BASIC);
    if (D.upper[seqOut] - D.lower[seqOut] > 0) {
      if (fabs(valueOut - D.lower[seqOut]) < fabs(valueOut - D.upper[seqOut]))
        stOut = (unsigned char)((stOut & ~7) | ST_LOWER);
      else
        stOut = (unsigned char)((stOut & ~7) | ST_UPPER);
    } else {
      stOut = (unsigned char)((stOut & ~7) | ST_FIXED);
    }
    D.sol[seqOut] = valueOut;
  }
  // originalBound(sequenceIn) / changeBound(sequenceOut) (ClpSimplexDual.cpp:6403, :6445)
  if ((stIn >> 3) & 3) {
    stIn = (unsigned char)(stIn & ~24);
    D.lower[seqIn] = D.origLower[seqIn];
    D.upper[seqIn] = D.origUpper[seqIn];
  }
  {
    double oldLower = D.lower[seqOut], oldUpper = D.upper[seqOut], value = D.sol[seqOut];
    stOut = (unsigned char)(stOut & ~24);
    double lowerValue = D.origLower[seqOut], upperValue = D.origUpper[seqOut];
    if (value == oldLower) {
      if (upperValue > oldLower + c->dualBound) {
        D.upper[seqOut] = oldLower + c->dualBound;
        stOut = (unsigned char)(stOut | (FAKE_UPPER << 3));
      }
    } else if (value == oldUpper) {
      if (lowerValue < oldUpper - c->dualBound) {
        D.lower[seqOut] = oldUpper - c->dualBound;
        stOut = (unsigned char)(stOut | (FAKE_LOWER << 3));
      }
    }
  }
  D.status[seqIn] = stIn;
  D.status[seqOut] = stOut;
  c->objectiveValue += objectiveChange;
  c->pivots++;
  // ---- pivot log (CLP_SIMPLEX_HOUSE2, src/ClpMessage.cpp:48)
  if (c->logCount < c->logCapacity) {
    PivotRecord *r = &D.log[c->logCount];
    r->iteration = c->numberIterations;
    r->sequenceIn = seqIn;
    r->sequenceOut = seqOut;
    r->pivotRow = pivotRow;
    r->numberFlipped = c->numberFlips;
    r->reserved = c->numberCandidates;  // length of the ratio-test candidate list (diagnostics)
    r->theta = c->theta;
    r->alpha = c->alpha;
    r->dualOut = dualOut;
    r->objective = c->objectiveValue;
  }
  c->logCount++;
  // ---- refactorization decision (ClpSimplex.cpp:2435-2488)
  if (c->numberIterations >= c->maximumIterations) {
    c->state = EXIT_MAX_ITERATIONS;
    return;
  }
  int numberPivots = c->pivots;
  if (numberPivots == c->maximumPivots || c->maximumPivots < 2) {
    c->state = EXIT_REFACTOR;
  } else if (c->forceFactorization > 0 && numberPivots == c->forceFactorization) {
    c->forceFactorization = (3 + 5 * c->forceFactorization) / 4;
    if (c->forceFactorization > c->maximumPivots)
      c->forceFactorization = -1;
    c->state = EXIT_REFACTOR;
  } else if (c->numberIterations > 1000 + 10 * (D.m + (D.n >> 2))) {
    double random = randomDouble(c);
    while (random < 0.45)
      random *= 2.0;
    int maxNumber = (c->forceFactorization < 0) ? c->maximumPivots : min(c->forceFactorization, c->maximumPivots);
    if (numberPivots >= random * maxNumber)
      c->state = EXIT_REFACTOR;
  }
  if (c->state == RUN && c->k + 2 >= c->kcap)
    c->state = EXIT_REFACTOR;  // nucleus storage nearly full: host regrows it at the refactorization
}


// =============================================================================================
// v2 kernels (round 1, after the first rocprof pass: profiles/r01_bench_v1_kernel_stats.txt)
// =============================================================================================

// ---- CHUZR split in three so the list scan uses the whole chip ---------------------------------
__device__ void chuzrPreBody(Dev D)
{
  Ctrl *c = D.ctrl;
  if (c->stepLimit >= 0 && c->numberIterations >= c->stepLimit) {
    c->state = EXIT_STEP_LIMIT;
    return;
  }
  int last = c->pivotRow;  // model_->pivotRow(): persists across refactorizations
  double tolerance = c->primalTolerance;
  if (c->pivotRule) {
    tolerance = tolerance + fmin(1.0e-2, c->largestPrimalError);
    tolerance = fmin(1000.0, tolerance);
    tolerance *= tolerance;
    if (last >= 0 && last < D.m) {
      int iPivot = D.pivotVariable[last];
      double value = D.sol[iPivot], lower = D.lower[iPivot], upper = D.upper[iPivot];
      if (value > upper + tolerance) {
        value -= upper;
        value *= value;
        if (D.infeas[last] == 0.0)
          D.infIndex[c->numberInfeasible++] = last;
        D.infeas[last] = value;
      } else if (value < lower - tolerance) {
        value -= lower;
        value *= value;
        if (D.infeas[last] == 0.0)
          D.infIndex[c->numberInfeasible++] = last;
        D.infeas[last] = value;
      } else if (D.infeas[last] != 0.0) {
        D.infeas[last] = REALLY_TINY;
      }
    }
    if (c->numberIterations < c->lastBadIteration + 200) {
      if (c->largestDualError > c->largestPrimalError)
        tolerance *= fmin(c->largestDualError / c->largestPrimalError, 1000.0);
    }
    int number = c->numberInfeasible;
    double dstart = ((double)number) * randomDouble(c);
    c->chuzrNumber = number;
    c->chuzrStart = (int)dstart;
  } else {
    if (c->largestPrimalError > 1.0e-8)
      tolerance *= c->largestPrimalError / 1.0e-8;
    c->chuzrNumber = D.m;
    c->chuzrStart = 0;
  }
  c->chuzrTolerance = tolerance;
  c->chuzrLast = last;
  c->preDone = 1;
}

// Start-of-pivot scalars of CHUZR.  Runs standalone at the head of a batch; inside a batch the
// previous pivot's k_fix_house has already done it (preDone) -- unless a host-side refactorization
// intervened, in which case that tail never ran.
__global__ void k_chuzr_pre(Dev D)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  if (threadIdx.x != 0)
    return;
  if (c->preDone)
    return;
  chuzrPreBody(D);
}

#define CHZ_ITEMS 4
__global__ void __launch_bounds__(256) k_chuzr_scan(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double shv[4];
  __shared__ int shk[4], shr[4];
  const double tolerance = c->chuzrTolerance;
  const int number = c->chuzrNumber, start = c->chuzrStart, last = c->chuzrLast;
  double best = 0.0;
  int bestKey = -1, bestRow = -1;
  const int base = blockIdx.x * (256 * CHZ_ITEMS);
#pragma unroll
  for (int q = 0; q < CHZ_ITEMS; q++) {
    int i = base + q * 256 + threadIdx.x;
    if (i >= number)
      continue;
    if (c->pivotRule) {
      int iRow = D.infIndex[i];
      double value = D.infeas[iRow];
      if (value > tolerance) {
        double weight = fmin(D.weights[iRow], 1.0e50);
        if (iRow == last)
          value *= 1.0e-10;
        int iSequence = D.pivotVariable[iRow];
        if (!(D.status[iSequence] & FLAGGED_BIT)) {
          double s = D.sol[iSequence];
          if (s > D.upper[iSequence] + tolerance || s < D.lower[iSequence] - tolerance) {
            double ratio = value / weight;
            int rank = i - start;
            if (rank < 0)
              rank += number;
            if (ratio > best || (ratio == best && bestKey >= 0 && rank < bestKey)) {
              best = ratio;
              bestKey = rank;
              bestRow = iRow;
            }
          }
        }
      }
    } else {
      int iSequence = D.pivotVariable[i];
      double value = D.sol[iSequence];
      double infeas = fmax(value - D.upper[iSequence], D.lower[iSequence] - value);
      if (infeas > tolerance && !(D.status[iSequence] & FLAGGED_BIT)) {
        if (infeas > best || (infeas == best && bestKey >= 0 && i < bestKey)) {
          best = infeas;
          bestKey = i;
          bestRow = i;
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_down(best, o);
    int ok = __shfl_down(bestKey, o);
    int orow = __shfl_down(bestRow, o);
    if (ok >= 0 && (bestKey < 0 || ov > best || (ov == best && ok < bestKey))) {
      best = ov;
      bestKey = ok;
      bestRow = orow;
    }
  }
  if (lane == 0) {
    shv[wv] = best;
    shk[wv] = bestKey;
    shr[wv] = bestRow;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; i++)
      if (shk[i] >= 0 && (bestKey < 0 || shv[i] > best || (shv[i] == best && shk[i] < bestKey))) {
        best = shv[i];
        bestKey = shk[i];
        bestRow = shr[i];
      }
    D.chzBest[blockIdx.x] = best;
    D.chzKey[blockIdx.x] = bestKey;
    D.chzRow[blockIdx.x] = bestRow;
  }
}

__global__ void __launch_bounds__(256) k_chuzr_final(Dev D, int nblocks)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double shv[4];
  __shared__ int shk[4], shr[4];
  double best = 0.0;
  int bestKey = -1, bestRow = -1;
  int used = (c->chuzrNumber + 256 * CHZ_ITEMS - 1) / (256 * CHZ_ITEMS);
  if (used > nblocks)
    used = nblocks;
  for (int b = threadIdx.x; b < used; b += blockDim.x) {
    double ov = D.chzBest[b];
    int ok = D.chzKey[b];
    if (ok >= 0 && (bestKey < 0 || ov > best || (ov == best && ok < bestKey))) {
      best = ov;
      bestKey = ok;
      bestRow = D.chzRow[b];
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_down(best, o);
    int ok = __shfl_down(bestKey, o);
    int orow = __shfl_down(bestRow, o);
    if (ok >= 0 && (bestKey < 0 || ov > best || (ov == best && ok < bestKey))) {
      best = ov;
      bestKey = ok;
      bestRow = orow;
    }
  }
  if (lane == 0) {
    shv[wv] = best;
    shk[wv] = bestKey;
    shr[wv] = bestRow;
  }
  __syncthreads();
  if (threadIdx.x != 0)
    return;
  for (int i = 1; i < 4; i++)
    if (shk[i] >= 0 && (bestKey < 0 || shv[i] > best || (shv[i] == best && shk[i] < bestKey))) {
      best = shv[i];
      bestKey = shk[i];
      bestRow = shr[i];
    }
  int chosen = bestRow;
  c->pivotRow = chosen;
  if (chosen < 0) {
    c->state = EXIT_NO_PIVOT_ROW;
    return;
  }
  int seqOut = D.pivotVariable[chosen];
  c->sequenceOut = seqOut;
  double valueOut = D.sol[seqOut], lowerOut = D.lower[seqOut], upperOut = D.upper[seqOut];
  c->valueOut = valueOut;
  c->lowerOut = lowerOut;
  c->upperOut = upperOut;
  if (valueOut > upperOut) {
    c->directionOut = -1;
    c->dualOut = valueOut - upperOut;
  } else if (valueOut < lowerOut) {
    c->directionOut = 1;
    c->dualOut = lowerOut - valueOut;
  } else if (valueOut - lowerOut < upperOut - valueOut) {
    c->directionOut = 1;
    c->dualOut = lowerOut - valueOut;
  } else {
    c->directionOut = -1;
    c->dualOut = valueOut - upperOut;
  }
  double acceptablePivot = 1.0e-1 * c->acceptablePivotBase;
  if (c->numberIterations > 100)
    acceptablePivot = c->acceptablePivotBase;
  if (c->pivots > 10 || (c->pivots && c->saveSumDual != 0.0))
    acceptablePivot = 1.0e+3 * c->acceptablePivotBase;
  else if (c->pivots > 5)
    acceptablePivot = 1.0e+2 * c->acceptablePivotBase;
  else if (c->pivots)
    acceptablePivot = c->acceptablePivotBase;
  c->acceptablePivot = acceptablePivot;
  D.vecC[chosen] = (double)c->directionOut;
  c->sequenceIn = -1;
  c->numberFlips = 0;
  c->objectiveChange = 0.0;
}

// ---- BTRAN t-vector: one wave per nucleus column (the slack part of y has a single nonzero for
// the unit-vector BTRAN of the iteration, so the lane-parallel sum is exact there) ----------------
__global__ void __launch_bounds__(256) k_btran_t2(Dev D, const double *cvec, const double *y, double *t, int iter)
{
  if (iter && D.ctrl->state != RUN)
    return;
  const int k = D.ctrl->k;
  const int lane = threadIdx.x & 63;
  int sc = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sc >= k)
    return;
  int col = D.slotCol[sc];
  double acc = 0.0;
  for (int p = D.colStart[col] + lane; p < D.colStart[col + 1]; p += 64) {
    int r = D.row[p];
    if (D.slotOfRow[r] < 0)
      acc += y[r] * D.elem[p];
  }
  acc = waveSum(acc);
  if (lane == 0)
    t[sc] = cvec[D.slotPos[sc]] - acc;
}

// iteration BTRAN, t-vector: the input is dir*e_p, so y_S has at most one nonzero (row of the
// leaving slack) and t follows analytically -- no pass over the slack rows needed.
__global__ void k_btran_t3(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  int sc = blockIdx.x * blockDim.x + threadIdx.x;
  if (sc >= c->k)
    return;
  const double dir = (double)c->directionOut;
  const int seqOut = c->sequenceOut;
  double value = 0.0;
  if (seqOut < D.n) {
    if (D.slotOfCol[seqOut] == sc)
      value = dir;
  } else {
    const int rOut = seqOut - D.n;
    const double y = dir * -1.0;  // y_rOut = -c[pos]
    const int col = D.slotCol[sc];
    const int s = D.rowStart[rOut], e = s + D.basicCount[rOut];
    for (int q = s; q < e; q++)
      if (D.ccol[q] == col)
        value -= y * D.relem[q];
  }
  D.slotA[sc] = value;
}

// iteration BTRAN, back end: rho[i] = slack part or sum of the gemvT partials, flush tiny, piNeg,
// rhoSlot (unpruned, for the nucleus update) and the per-block partial of sum rho^2 (DSE norm)
__global__ void __launch_bounds__(256) k_rho_finish3(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double sh[16];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double sq = 0.0;
  if (i < D.m) {
    double v;
    int sr = D.slotOfRow[i];
    if (sr >= 0) {
      // y_R = Minv^T t with t given as a short list: read only those rows of Minv
      const int tc = c->tCount;
      v = 0.0;
      for (int q = 0; q < tc; q++)
        v += D.Minv[(size_t)D.tIndex[q] * D.ld + sr] * D.tValue[q];
      D.rhoSlot[sr] = v;
    } else {
      int p = D.posOfSlack[i];
      v = (p >= 0) ? D.vecC[p] * -1.0 : 0.0;
    }
    if (fabs(v) <= c->zeroTolerance)
      v = 0.0;
    D.rho[i] = v;
    D.piNeg[i] = -v;
    sq = v * v;
  }
  // bitmap of the nonzero rows of pi, one 64-bit word per wave (the pricing kernel keeps it in LDS)
  unsigned long long mask = __ballot(sq != 0.0);
  if ((threadIdx.x & 63) == 0 && i < D.m + 63)
    D.piBits[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = mask;
  double s = blockSum(sq, sh);
  if (threadIdx.x == 0)
    D.normPartial[blockIdx.x] = s;
}

// gemvT partial with 8 independent loads in flight per lane
__global__ void k_gemvT_partial2(Dev D, const double *t, int iter)
{
  if (iter && D.ctrl->state != RUN)
    return;
  const int k = D.ctrl->k;
  int sr = blockIdx.x * blockDim.x + threadIdx.x;
  int chunk = blockIdx.y;
  int sc0 = chunk * 64;
  if (sc0 >= k || sr >= k)
    return;
  int sc1 = min(sc0 + 64, k);
  double acc = 0.0;
  const double *Mp = D.Minv + (size_t)sc0 * D.ld + sr;
  int sc = sc0;
  for (; sc + 8 <= sc1; sc += 8) {
    double v0 = Mp[0], v1 = Mp[D.ld], v2 = Mp[2 * (size_t)D.ld], v3 = Mp[3 * (size_t)D.ld];
    double v4 = Mp[4 * (size_t)D.ld], v5 = Mp[5 * (size_t)D.ld], v6 = Mp[6 * (size_t)D.ld], v7 = Mp[7 * (size_t)D.ld];
    acc += v0 * t[sc];
    acc += v1 * t[sc + 1];
    acc += v2 * t[sc + 2];
    acc += v3 * t[sc + 3];
    acc += v4 * t[sc + 4];
    acc += v5 * t[sc + 5];
    acc += v6 * t[sc + 6];
    acc += v7 * t[sc + 7];
    Mp += 8 * (size_t)D.ld;
  }
  for (; sc < sc1; sc++) {
    acc += *Mp * t[sc];
    Mp += D.ld;
  }
  D.partial[(size_t)chunk * D.ld + sr] = acc;
}

// rho finish + per-block partial of sum rho^2 (DSE norm)
__global__ void __launch_bounds__(256) k_rho_finish2(Dev D)
{
  if (D.ctrl->state != RUN)
    return;
  __shared__ double sh[16];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double sq = 0.0;
  if (i < D.m) {
    double v = D.rho[i];
    if (fabs(v) <= D.ctrl->zeroTolerance)
      v = 0.0;
    D.rho[i] = v;
    D.piNeg[i] = -v;
    sq = v * v;
  }
  double s = blockSum(sq, sh);
  if (threadIdx.x == 0)
    D.normPartial[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_norm_alpha2(Dev D, int nb)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double sh[16];
  double acc = 0.0;
  if (c->pivotRule)
    for (int b = threadIdx.x; b < nb; b += blockDim.x)
      acc += D.normPartial[b];
  acc = blockSum(acc, sh);
  if (threadIdx.x == 0) {
    double alphaOld = c->alpha;
    double norm = acc / (alphaOld * alphaOld);
    c->norm = norm;
    double alpha = D.w[c->pivotRow];
    double btranAlpha = c->btranAlpha;
    double checkValue = 1.0e-7;
    if (c->largestPrimalError > 10.0)
      checkValue = fmin(1.0e-4, 1.0e-8 * c->largestPrimalError);
    c->scratchSum = 2.0 / alphaOld;
    c->alpha = alpha;
    if (fabs(btranAlpha) < 1.0e-12 || fabs(alpha) < 1.0e-12 || fabs(btranAlpha - alpha) > checkValue * (1.0 + fabs(alpha))) {
      int bad = 1;
      if (!c->pivots) {
        double test;
        if (fabs(btranAlpha) < 1.0e-8 || fabs(alpha) < 1.0e-8)
          test = 1.0e-1 * fabs(alpha);
        else
          test = 1.0e-4 * (1.0 + fabs(alpha));
        if (!(fabs(btranAlpha) < 1.0e-12 || fabs(alpha) < 1.0e-12 || fabs(btranAlpha - alpha) > test))
          bad = 0;
      }
      if (bad)
        c->state = EXIT_ALPHA_CHECK;
    }
  }
}

// parallel version of the scalar tail after a primal update
__global__ void __launch_bounds__(256) k_after_primal2(Dev D, int nb, int which)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double sh[16];
  const bool active = !(which == 1 && c->numberFlips == 0);
  double s = 0.0;
  if (active)
    for (int b = threadIdx.x; b < nb; b += blockDim.x)
      s += D.blockSum[b];
  s = blockSum(s, sh);
  if (threadIdx.x != 0)
    return;
  if (active) {
    c->objectiveChange += s;
    c->numberInfeasible += c->numberAppend;
    c->numberAppend = 0;
    if (c->pivotRule) {
      int iRow = c->pivotRow;
      if (D.infeas[iRow] != 0.0)
        D.infeas[iRow] = REALLY_TINY;
    }
  }
  if (which == 1) {
    double oldDualOut = c->dualOut;
    if (c->numberFlips) {
      c->valueOut = D.sol[c->sequenceOut];
      if (c->directionOut < 0)
        c->dualOut = c->valueOut - c->upperOut;
      else
        c->dualOut = c->lowerOut - c->valueOut;
    }
    double alpha = c->alpha;
    c->movement = -c->dualOut * c->directionOut / alpha;
    double movementOld = oldDualOut * c->directionOut / alpha;
    if (c->objectiveChange + fabs(movementOld * c->dualIn) < -fmax(1.0e-5, 1.0e-12 * fabs(c->objectiveValue))) {
      if (c->pivots) {
        c->state = EXIT_BACKWARDS;
        return;
      }
    }
    if (fabs(alpha) < c->zeroTolerance || fabs(c->dualOut) > 1.0e50) {
      c->state = EXIT_BAD_UPDATE;
      return;
    }
    if (c->theta < 0.0)
      c->theta = 0.0;
    int seqIn = c->sequenceIn, seqOut = c->sequenceOut;
    int inStruct = seqIn < D.n, outStruct = seqOut < D.n;
    c->updateCase = outStruct ? (inStruct ? 0 : 2) : (inStruct ? 1 : 3);
    c->slotColOut = outStruct ? D.slotOfCol[seqOut] : -1;
    c->rowOfSlackOut = outStruct ? -1 : (seqOut - D.n);
    c->slotRowIn = inStruct ? -1 : D.slotOfRow[seqIn - D.n];
  }
}

// =============================================================================================
// Row pricing, v2: sliced-ELL (SELL-64) sweep.  One wave owns a slice of 64 columns; entry t of
// lane l sits at sellStart[slice] + 64*t + l, so every step is one fully coalesced 512 B (elements)
// + 256 B (row indices) wave transaction, 8 steps in flight per lane.  Each lane still adds its own
// column's products in ascending entry order: the result is bit-identical to the reference's scalar
// loop (ClpPackedMatrix.cpp:1872-1886) and to the v1 kernel.  Fused first ratio pass as in v1.
// =============================================================================================
#define SELL_U 8
#define SELL_BITS_MAX 8192  // 64-bit words of the pi bitmap kept in LDS (rows <= 524288)
// PIPE: software-pipelined loads; NT: non-temporal matrix loads; BITS: gather pi only where the
// row's bit is set (pi is sparse for most pivots: the gather traffic scales with nnz(pi)/m)
template <bool PIPE, bool NT, bool BITS> __device__ inline void priceSellBody(Dev D, unsigned long long *bits)
{
  const Ctrl *c = D.ctrl;
  __shared__ double shd[16];
  const double dualT = -c->dualTolerance;
  const double acceptablePivot = c->acceptablePivot;
  const double zeroTolerance = c->zeroTolerance;
  const double tentativeTheta = 1.0e15;
  const int lane = threadIdx.x & 63;
  const int slice = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwords = (D.m + 63) >> 6;
  if (BITS) {
    for (int w = threadIdx.x; w < nwords; w += blockDim.x)
      bits[w] = D.piBits[w];
    __syncthreads();
  }
  auto piAt = [&](int r) -> double {
    if (BITS) {
      return ((bits[r >> 6] >> (r & 63)) & 1ull) ? D.piNeg[r] : 0.0;
    } else {
      return D.piNeg[r];
    }
  };
  double ratio = 1.0e31, bytes = 0.0;
  if (slice < D.numSlices) {
    const int idx = slice * 64 + lane;
    const int j = D.sellCol[idx];
    int len = 0, wanted = 0;
    if (j >= 0) {
      wanted = (D.status[j] & 3) - 1;
      if (wanted)
        len = D.sellLen[idx];
    }
    int maxLen = len;
    for (int o = 32; o > 0; o >>= 1)
      maxLen = max(maxLen, __shfl_xor(maxLen, o));
    double value = 0.0;
    if (maxLen > 0) {
      const int start = D.sellStart[slice];
      const int *rp = D.sellRow + start + lane;
      const double *ep = D.sellElem + start + lane;
      if (PIPE) {
        int r0[SELL_U];
        double e0[SELL_U];
#pragma unroll
        for (int u = 0; u < SELL_U; u++) {
          r0[u] = NT ? __builtin_nontemporal_load(&rp[u * 64]) : rp[u * 64];
          e0[u] = NT ? __builtin_nontemporal_load(&ep[u * 64]) : ep[u * 64];
        }
        for (int t = 0; t < maxLen; t += SELL_U) {
          int r1[SELL_U];
          double e1[SELL_U], pv[SELL_U];
          const bool more = t + SELL_U < maxLen;
          if (more) {
#pragma unroll
            for (int u = 0; u < SELL_U; u++) {
              r1[u] = NT ? __builtin_nontemporal_load(&rp[(t + SELL_U + u) * 64]) : rp[(t + SELL_U + u) * 64];
              e1[u] = NT ? __builtin_nontemporal_load(&ep[(t + SELL_U + u) * 64]) : ep[(t + SELL_U + u) * 64];
            }
          }
#pragma unroll
          for (int u = 0; u < SELL_U; u++)
            pv[u] = piAt(r0[u]);
#pragma unroll
          for (int u = 0; u < SELL_U; u++)
            if (t + u < len)
              value += pv[u] * e0[u];
          if (more) {
#pragma unroll
            for (int u = 0; u < SELL_U; u++) {
              r0[u] = r1[u];
              e0[u] = e1[u];
            }
          }
        }
      } else {
        for (int t = 0; t < maxLen; t += SELL_U) {
          int r[SELL_U];
          double e[SELL_U], pv[SELL_U];
#pragma unroll
          for (int u = 0; u < SELL_U; u++) {
            r[u] = NT ? __builtin_nontemporal_load(&rp[(t + u) * 64]) : rp[(t + u) * 64];
            e[u] = NT ? __builtin_nontemporal_load(&ep[(t + u) * 64]) : ep[(t + u) * 64];
          }
#pragma unroll
          for (int u = 0; u < SELL_U; u++)
            pv[u] = piAt(r[u]);
#pragma unroll
          for (int u = 0; u < SELL_U; u++)
            if (t + u < len)
              value += pv[u] * e[u];
        }
      }
    }
    if (j >= 0) {
      int flag = 0;
      if (wanted) {
        bytes = 12.0 * len + 4.0;
        if (fabs(value) > zeroTolerance) {
          bytes += 20.0;
          if (wanted > 0) {
            double mult = (wanted == 1) ? -1.0 : 1.0;
            double alpha = value * mult;
            if (alpha > 0.0) {
              double oldValue = D.dj[j] * mult;
              double v2 = oldValue - tentativeTheta * alpha;
              if (v2 < dualT) {
                flag = 1;
                if (alpha >= acceptablePivot)
                  ratio = (oldValue - dualT) / alpha;
              }
            }
          }
        } else {
          value = 0.0;
        }
      }
      D.alphaCol[j] = value;
      D.candFlag[D.m + j] = (unsigned char)flag;
    }
  }
  double bmin = blockMin(ratio, shd);
  double bsum = blockSum(bytes, shd);
  if (threadIdx.x == 0) {
    D.sellMin[blockIdx.x] = bmin;
    D.sellBytes[blockIdx.x] = bsum;
  }
}

// variant: 1 plain, 2 bitmap, 3 pipelined+bitmap, 4 pipelined+nt+bitmap, 5 nt+bitmap
__global__ void __launch_bounds__(256) k_price_sell(Dev D, int variant)
{
  extern __shared__ __attribute__((aligned(16))) unsigned long long sellBits[];  // (m+63)/64 words for variants >= 2
  if (D.ctrl->state != RUN)
    return;
  switch (variant) {
  case 2:
    priceSellBody<false, false, true>(D, sellBits);
    break;
  case 3:
    priceSellBody<true, false, true>(D, sellBits);
    break;
  case 4:
    priceSellBody<true, true, true>(D, sellBits);
    break;
  case 5:
    priceSellBody<false, true, true>(D, sellBits);
    break;
  default:
    priceSellBody<false, false, false>(D, sellBits);
    break;
  }
}

// Row pricing for long columns (dense or few-column LPs): one lane per column cannot fill the chip
// when n < ~10^5, so here a whole wave strides one CSC column (coalesced) and reduces.  The per-column
// sum is then a fixed 64-way tree instead of the reference's sequential order: deterministic, but
// only equal to the sequential sum to rounding (used when the mean column length is >= 256).
#define WIDE_BLOCKS 4096
__global__ void __launch_bounds__(256) k_price_wide(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double shd[16];
  const double dualT = -c->dualTolerance;
  const double acceptablePivot = c->acceptablePivot;
  const double zeroTolerance = c->zeroTolerance;
  const int lane = threadIdx.x & 63;
  double ratio = 1.0e31, bytes = 0.0;
  for (int j = D.priceFirst + blockIdx.x * 4 + (threadIdx.x >> 6); j < D.priceLast; j += gridDim.x * 4) {
    int wanted = (D.status[j] & 3) - 1;
    double value = 0.0;
    int flag = 0;
    if (wanted) {
      const int start = D.colStart[j], end = D.colStart[j + 1];
      double acc = 0.0;
      for (int p = start + lane; p < end; p += 64)
        acc += D.piNeg[D.row[p]] * D.elem[p];
      value = waveSum(acc);
      value = __shfl(value, 0);
      if (lane == 0)
        bytes += 12.0 * (end - start) + 4.0;
      if (fabs(value) > zeroTolerance) {
        if (lane == 0)
          bytes += 20.0;
        if (wanted > 0) {
          double mult = (wanted == 1) ? -1.0 : 1.0;
          double alpha = value * mult;
          if (alpha > 0.0) {
            double oldValue = D.dj[j] * mult;
            double v2 = oldValue - 1.0e15 * alpha;
            if (v2 < dualT) {
              flag = 1;
              if (alpha >= acceptablePivot)
                ratio = fmin(ratio, (oldValue - dualT) / alpha);
            }
          }
        }
      } else {
        value = 0.0;
      }
    }
    if (lane == 0) {
      D.alphaCol[j] = value;
      D.candFlag[D.m + j] = (unsigned char)flag;
    }
  }
  double bmin = blockMin(ratio, shd);
  double bsum = blockSum(bytes, shd);
  if (threadIdx.x == 0) {
    D.sellMin[blockIdx.x] = bmin;
    D.sellBytes[blockIdx.x] = bsum;
  }
}

// row (slack) part of the first ratio pass + per-key-block candidate counts for the ordered
// compaction (columns were flagged by k_price_sell in slice order; counts must be in key order)
__global__ void __launch_bounds__(PRICE_BLOCK) k_cand_count(Dev D, int nbRows, int recomputeRatio)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double shd[16];
  __shared__ int shi[17];
  int flag = 0;
  double ratio = 1.0e31;
  if ((int)blockIdx.x < nbRows) {
    const double dualT = -c->dualTolerance;
    int i = blockIdx.x * PRICE_BLOCK + threadIdx.x;
    if (i < D.m) {
      double value = D.rho[i];
      if (value != 0.0) {
        int iStatus = (D.status[D.n + i] & 3) - 1;
        if (iStatus > 0) {
          double mult = (iStatus == 1) ? -1.0 : 1.0;
          double alpha = value * mult;
          if (alpha > 0.0) {
            double oldValue = D.dj[D.n + i] * mult;
            double v2 = oldValue - 1.0e15 * alpha;
            if (v2 < dualT) {
              flag = 1;
              if (alpha >= c->acceptablePivot)
                ratio = (oldValue - dualT) / alpha;
            }
          }
        }
      }
      D.candFlag[i] = (unsigned char)flag;
    }
  } else {
    int j = D.firstColumn + ((int)blockIdx.x - nbRows) * PRICE_BLOCK + threadIdx.x;
    if (j < D.lastColumn) {
      flag = D.candFlag[D.m + j];
      if (flag && recomputeRatio) {
        // same expressions as the pricing kernel, so the min over all ranks' columns is bit-identical
        int wanted = (D.status[j] & 3) - 1;
        double mult = (wanted == 1) ? -1.0 : 1.0;
        double alpha = D.alphaCol[j] * mult;
        if (alpha >= c->acceptablePivot) {
          double oldValue = D.dj[j] * mult;
          ratio = (oldValue - (-c->dualTolerance)) / alpha;
        }
      }
    }
  }
  int total;
  blockRank(flag, total, shi);
  double bmin = blockMin(ratio, shd);
  if (threadIdx.x == 0) {
    D.blockCount[blockIdx.x] = total;
    D.blockMin[blockIdx.x] = bmin;
    D.blockSum[blockIdx.x] = 0.0;
  }
}


// =============================================================================================
// v4 fusions: fewer grid-wide dependencies per pivot (each launch costs ~4 us on 256 CUs)
// =============================================================================================

// k_chuzr_final + k_btran_t3 in one workgroup
__global__ void __launch_bounds__(256) k_chuzr_final_btran(Dev D, int nblocks)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double shv[4];
  __shared__ int shk[4], shr[4];
  __shared__ int s_ok;
  double best = 0.0;
  int bestKey = -1, bestRow = -1;
  int used = (c->chuzrNumber + 256 * CHZ_ITEMS - 1) / (256 * CHZ_ITEMS);
  if (used > nblocks)
    used = nblocks;
  for (int b = threadIdx.x; b < used; b += blockDim.x) {
    double ov = D.chzBest[b];
    int ok = D.chzKey[b];
    if (ok >= 0 && (bestKey < 0 || ov > best || (ov == best && ok < bestKey))) {
      best = ov;
      bestKey = ok;
      bestRow = D.chzRow[b];
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_down(best, o);
    int ok = __shfl_down(bestKey, o);
    int orow = __shfl_down(bestRow, o);
    if (ok >= 0 && (bestKey < 0 || ov > best || (ov == best && ok < bestKey))) {
      best = ov;
      bestKey = ok;
      bestRow = orow;
    }
  }
  if (lane == 0) {
    shv[wv] = best;
    shk[wv] = bestKey;
    shr[wv] = bestRow;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; i++)
      if (shk[i] >= 0 && (bestKey < 0 || shv[i] > best || (shv[i] == best && shk[i] < bestKey))) {
        best = shv[i];
        bestKey = shk[i];
        bestRow = shr[i];
      }
    int chosen = bestRow;
    c->pivotRow = chosen;
    c->preDone = 0;
    s_ok = chosen >= 0;
    if (chosen < 0) {
      c->state = EXIT_NO_PIVOT_ROW;
    } else {
      int seqOut = D.pivotVariable[chosen];
      c->sequenceOut = seqOut;
      double valueOut = D.sol[seqOut], lowerOut = D.lower[seqOut], upperOut = D.upper[seqOut];
      c->valueOut = valueOut;
      c->lowerOut = lowerOut;
      c->upperOut = upperOut;
      if (valueOut > upperOut) {
        c->directionOut = -1;
        c->dualOut = valueOut - upperOut;
      } else if (valueOut < lowerOut) {
        c->directionOut = 1;
        c->dualOut = lowerOut - valueOut;
      } else if (valueOut - lowerOut < upperOut - valueOut) {
        c->directionOut = 1;
        c->dualOut = lowerOut - valueOut;
      } else {
        c->directionOut = -1;
        c->dualOut = valueOut - upperOut;
      }
      double acceptablePivot = 1.0e-1 * c->acceptablePivotBase;
      if (c->numberIterations > 100)
        acceptablePivot = c->acceptablePivotBase;
      if (c->pivots > 10 || (c->pivots && c->saveSumDual != 0.0))
        acceptablePivot = 1.0e+3 * c->acceptablePivotBase;
      else if (c->pivots > 5)
        acceptablePivot = 1.0e+2 * c->acceptablePivotBase;
      else if (c->pivots)
        acceptablePivot = c->acceptablePivotBase;
      c->acceptablePivot = acceptablePivot;
      D.vecC[chosen] = (double)c->directionOut;
      c->sequenceIn = -1;
      c->numberFlips = 0;
      c->objectiveChange = 0.0;
    }
  }
  __syncthreads();
  if (!s_ok)
    return;
  // BTRAN t-vector for dir*e_p (see k_btran_t3), kept as a short list: t has one nonzero when a
  // structural leaves, and one per basic entry of the leaving slack's row otherwise
  const double dir = (double)c->directionOut;
  const int seqOut = c->sequenceOut;
  if (seqOut < D.n) {
    if (threadIdx.x == 0) {
      D.tIndex[0] = D.slotOfCol[seqOut];
      D.tValue[0] = dir;
      c->tCount = 1;
    }
  } else {
    const int rOut = seqOut - D.n;
    const double y = dir * -1.0;
    const int s = D.rowStart[rOut], cnt = D.basicCount[rOut];
    // ascending col-slot order so the later sum matches the dense form; rows are short
    for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
      int sc = D.slotOfCol[D.ccol[s + q]];
      int rank = 0;
      for (int q2 = 0; q2 < cnt; q2++) {
        int sc2 = D.slotOfCol[D.ccol[s + q2]];
        rank += (sc2 < sc);
      }
      D.tIndex[rank] = sc;
      D.tValue[rank] = 0.0 - y * D.relem[s + q];
    }
    if (threadIdx.x == 0)
      c->tCount = cnt;
  }
}

// one kernel for every candidate count: <= 1024 one wave (16 per lane), <= 4096 four waves
__global__ void __launch_bounds__(256) k_dual_column_fused(Dev D)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int nc = c->numberCandidates;
  if (!nc) {
    if (threadIdx.x == 0) {
      c->sequenceIn = -1;
      c->alpha = 0.0;
      c->bestPossible = 0.0;
      c->state = EXIT_NO_INCOMING;
    }
    return;
  }
  if (nc <= 16 * 64) {
    if (threadIdx.x < 64)
      dualColumnImpl<16, true>(D);
  } else if (nc <= 16 * 256) {
    dualColumnImpl<16, false>(D);
  } else {
    dualColumnImpl<0, false>(D);
  }
}

// FTRAN nucleus GEMV with the gather of the right-hand side folded in
__global__ void __launch_bounds__(256) k_gemv2g(Dev D, const double *v1, const double *v2, double *x1, double *x2, int iter)
{
  if (iter && D.ctrl->state != RUN)
    return;
  if (iter == 2 && D.ctrl->numberFlips == 0)
    return;
  const int k = D.ctrl->k;
  const int lane = threadIdx.x & 63;
  const int wavesPerBlock = blockDim.x >> 6;
  for (int sc = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6); sc < k; sc += gridDim.x * wavesPerBlock) {
    const double *Mrow = D.Minv + (size_t)sc * D.ld;
    double a1 = 0.0, a2 = 0.0;
    if (v2) {
      for (int sr = lane; sr < k; sr += 64) {
        double mv = Mrow[sr];
        int r = D.slotRow[sr];
        a1 += mv * v1[r];
        a2 += mv * v2[r];
      }
      a2 = waveSum(a2);
    } else {
      for (int sr = lane; sr < k; sr += 64)
        a1 += Mrow[sr] * v1[D.slotRow[sr]];
    }
    a1 = waveSum(a1);
    if (lane == 0) {
      x1[sc] = a1;
      if (v2)
        x2[sc] = a2;
    }
  }
}

// DSE weight update (positions) and dual update + flip detection (keys) in one N-wide pass.
// The alpha check runs afterwards (k_scan_flips_alpha); on failure the host unrolls the weights and
// refactorizes, which recomputes every dj, exactly as the reference does after its `break` (:1456).
__global__ void __launch_bounds__(PRICE_BLOCK) k_weights_dj(Dev D, int nbRows, int nbNorm)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ int shi[17];
  __shared__ double shd[16];
  const double theta = c->theta;
  const double tolerance = c->dualTolerance + fmin(1.0e-2, c->largestDualError);
  const int seqIn = c->sequenceIn;
  int flag = 0;
  if ((int)blockIdx.x < nbRows) {
    // every row block needs the DSE norm: sum of the per-block partials of sum rho^2
    double norm = 0.0, multiplier = 0.0;
    if (c->pivotRule) {
      double acc = 0.0;
      for (int b = threadIdx.x; b < nbNorm; b += blockDim.x)
        acc += D.normPartial[b];
      acc = blockSum(acc, shd);
      double alphaOld = c->alpha;
      norm = acc / (alphaOld * alphaOld);
      multiplier = 2.0 / alphaOld;
    }
    int i = blockIdx.x * PRICE_BLOCK + threadIdx.x;
    if (i < D.m) {
      if (c->pivotRule) {
        double thetaW = D.w[i];
        if (thetaW != 0.0) {
          double devex = D.weights[i];
          D.altWeights[i] = devex;
          if (i == c->pivotRow) {
            devex = (norm < DEVEX_TRY_NORM) ? DEVEX_TRY_NORM : norm;
          } else {
            devex += thetaW * (thetaW * norm + D.tau[i] * multiplier);
            if (devex < DEVEX_TRY_NORM)
              devex = DEVEX_TRY_NORM;
          }
          D.weights[i] = devex;
        }
      }
      double alphaI = D.rho[i];
      int seq = D.n + i;
      if (alphaI != 0.0 && seq != seqIn) {
        int iStatus = (D.status[seq] & 3) - 1;
        if (iStatus) {
          double value = D.dj[seq] - theta * alphaI;
          D.dj[seq] = value;
          double mult = (iStatus == 1) ? -1.0 : ((iStatus == 2) ? 1.0 : 0.0);
          value *= mult;
          if (value < -tolerance)
            flag = 1;
        }
      }
      D.candFlag[i] = (unsigned char)flag;
    }
  } else {
    int j = D.firstColumn + ((int)blockIdx.x - nbRows) * PRICE_BLOCK + threadIdx.x;
    if (j < D.lastColumn) {
      double alphaI = D.alphaCol[j];
      if (alphaI != 0.0 && j != seqIn) {
        int iStatus = (D.status[j] & 3) - 1;
        if (iStatus) {
          double value = D.dj[j] - theta * alphaI;
          D.dj[j] = value;
          double mult = (iStatus == 1) ? -1.0 : ((iStatus == 2) ? 1.0 : -1.0);
          value *= mult;
          if (value < -tolerance && iStatus > 0)
            flag = 1;
        }
      }
      D.candFlag[D.m + j] = (unsigned char)flag;
    }
  }
  int total;
  blockRank(flag, total, shi);
  if (threadIdx.x == 0)
    D.blockCount[blockIdx.x] = total;
}

// flip-count scan + the btran/ftran alpha accuracy test (whileIterating :1447-1501)
__global__ void __launch_bounds__(1024) k_scan_flips_alpha(Dev D, int nb)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ int shi[17];
  __shared__ int s_base;
  if (threadIdx.x == 0)
    s_base = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += blockDim.x) {
    int b = b0 + threadIdx.x;
    int cnt = (b < nb) ? D.blockCount[b] : 0;
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int v = cnt;
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(v, o);
      if (lane >= o)
        v += t;
    }
    __syncthreads();
    if (lane == 63)
      shi[wv] = v;
    __syncthreads();
    int base = s_base;
    for (int i = 0; i < wv; i++)
      base += shi[i];
    if (b < nb)
      D.blockOffset[b] = base + v - cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int i = 0; i < nw; i++)
        tot += shi[i];
      s_base += tot;
    }
    __syncthreads();
  }
  if (threadIdx.x != 0)
    return;
  c->numberFlips = s_base;
  double alpha = D.w[c->pivotRow];
  double btranAlpha = c->btranAlpha;
  double checkValue = 1.0e-7;
  if (c->largestPrimalError > 10.0)
    checkValue = fmin(1.0e-4, 1.0e-8 * c->largestPrimalError);
  c->alpha = alpha;
  if (fabs(btranAlpha) < 1.0e-12 || fabs(alpha) < 1.0e-12 || fabs(btranAlpha - alpha) > checkValue * (1.0 + fabs(alpha))) {
    int bad = 1;
    if (!c->pivots) {
      double test;
      if (fabs(btranAlpha) < 1.0e-8 || fabs(alpha) < 1.0e-8)
        test = 1.0e-1 * fabs(alpha);
      else
        test = 1.0e-4 * (1.0 + fabs(alpha));
      if (!(fabs(btranAlpha) < 1.0e-12 || fabs(alpha) < 1.0e-12 || fabs(btranAlpha - alpha) > test))
        bad = 0;
    }
    if (bad)
      c->state = EXIT_ALPHA_CHECK;
  }
}

// flip FTRAN back end fused with the primal update by the flip movement (ratio 1.0): the thread
// that produces x[p] applies it.  Appends are counted per position block with integer atomics
// (blockCount zeroed by k_flip_apply); objective partials are per launch block (fixed mapping).
__global__ void __launch_bounds__(256) k_ftran_scatter_flip(Dev D, const double *xk)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN || c->numberFlips == 0)
    return;
  __shared__ double shd[16];
  const int k = c->k;
  const double tolerance = c->primalTolerance;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int p = -1;
  double v = 0.0;
  if (t < D.m) {
    p = D.posOfSlack[t];
    if (p >= 0) {
      double a1 = 0.0;
      int s = D.rowStart[t], e = s + D.basicCount[t];
      for (int q = s; q < e; q++)
        a1 += D.relem[q] * xk[D.slotOfCol[D.ccol[q]]];
      v = a1 - D.flipRhs[t];
    }
    D.flipRhs[t] = 0.0;  // consumed (the nucleus rows were read by k_gemv2g)
  } else if (t < D.m + k) {
    int sc = t - D.m;
    p = D.slotPos[sc];
    v = xk[sc];
  }
  double changeObj = 0.0;
  if (p >= 0) {
    int append = 0;
    if (v != 0.0) {
      int iPivot = D.pivotVariable[p];
      double value = D.sol[iPivot];
      value -= v;
      changeObj -= v * D.cost[iPivot];
      D.sol[iPivot] = value;
      if (c->pivotRule) {
        double lower = D.lower[iPivot], upper = D.upper[iPivot];
        double old = D.infeas[p];
        if (value < lower - tolerance) {
          value -= lower;
          value *= value;
          if (old == 0.0)
            append = 1;
          D.infeas[p] = value;
        } else if (value > upper + tolerance) {
          value -= upper;
          value *= value;
          if (old == 0.0)
            append = 1;
          D.infeas[p] = value;
        } else if (old != 0.0) {
          D.infeas[p] = REALLY_TINY;
        }
      }
    }
    D.appendFlag[p] = append;
    if (append)
      atomicAdd(&D.blockCount[p >> 8], 1);
  }
  double s = blockSum(changeObj, shd);
  if (threadIdx.x == 0)
    D.blockSum[blockIdx.x] = s;
}

// append scan with absolute offsets + the scalar tail that used to be k_after_primal2
__global__ void __launch_bounds__(1024) k_scan_tail(Dev D, int nbCount, int nbSum, int which, int alphaTest = 0)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ int shi[17];
  __shared__ double shd[16];
  __shared__ int s_base;
  const bool active = !(which == 1 && c->numberFlips == 0);
  if (threadIdx.x == 0)
    s_base = c->numberInfeasible;
  __syncthreads();
  if (active) {
    for (int b0 = 0; b0 < nbCount; b0 += blockDim.x) {
      int b = b0 + threadIdx.x;
      int cnt = (b < nbCount) ? D.blockCount[b] : 0;
      int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
      int v = cnt;
      for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane >= o)
          v += t;
      }
      __syncthreads();
      if (lane == 63)
        shi[wv] = v;
      __syncthreads();
      int base = s_base;
      for (int i = 0; i < wv; i++)
        base += shi[i];
      if (b < nbCount)
        D.blockOffset[b] = base + v - cnt;
      __syncthreads();
      if (threadIdx.x == 0) {
        int tot = 0;
        for (int i = 0; i < nw; i++)
          tot += shi[i];
        s_base += tot;
      }
      __syncthreads();
    }
  }
  double s = 0.0;
  if (active)
    for (int b = threadIdx.x; b < nbSum; b += blockDim.x)
      s += D.blockSum[b];
  s = blockSum(s, shd);
  if (threadIdx.x != 0)
    return;
  if (active) {
    c->objectiveChange += s;
    c->numberAppend = s_base - c->numberInfeasible;
    c->numberInfeasible = s_base;
    if (c->pivotRule) {
      int iRow = c->pivotRow;
      if (D.infeas[iRow] != 0.0)
        D.infeas[iRow] = REALLY_TINY;
    }
  } else {
    c->numberAppend = 0;
  }
  if (which == 1) {
    if (alphaTest) {
      // btran/ftran alpha accuracy test (whileIterating :1447-1501)
      double alphaNew = D.w[c->pivotRow];
      double btranAlpha = c->btranAlpha;
      double checkValue = 1.0e-7;
      if (c->largestPrimalError > 10.0)
        checkValue = fmin(1.0e-4, 1.0e-8 * c->largestPrimalError);
      c->alpha = alphaNew;
      if (fabs(btranAlpha) < 1.0e-12 || fabs(alphaNew) < 1.0e-12 || fabs(btranAlpha - alphaNew) > checkValue * (1.0 + fabs(alphaNew))) {
        int bad = 1;
        if (!c->pivots) {
          double test;
          if (fabs(btranAlpha) < 1.0e-8 || fabs(alphaNew) < 1.0e-8)
            test = 1.0e-1 * fabs(alphaNew);
          else
            test = 1.0e-4 * (1.0 + fabs(alphaNew));
          if (!(fabs(btranAlpha) < 1.0e-12 || fabs(alphaNew) < 1.0e-12 || fabs(btranAlpha - alphaNew) > test))
            bad = 0;
        }
        if (bad) {
          c->state = EXIT_ALPHA_CHECK;
          return;
        }
      }
    }
    double oldDualOut = c->dualOut;
    if (c->numberFlips) {
      c->valueOut = D.sol[c->sequenceOut];
      if (c->directionOut < 0)
        c->dualOut = c->valueOut - c->upperOut;
      else
        c->dualOut = c->lowerOut - c->valueOut;
    }
    double alpha = c->alpha;
    c->movement = -c->dualOut * c->directionOut / alpha;
    double movementOld = oldDualOut * c->directionOut / alpha;
    if (c->objectiveChange + fabs(movementOld * c->dualIn) < -fmax(1.0e-5, 1.0e-12 * fabs(c->objectiveValue))) {
      if (c->pivots) {
        c->state = EXIT_BACKWARDS;
        return;
      }
    }
    if (fabs(alpha) < c->zeroTolerance || fabs(c->dualOut) > 1.0e50) {
      c->state = EXIT_BAD_UPDATE;
      return;
    }
    if (c->theta < 0.0)
      c->theta = 0.0;
    int seqIn = c->sequenceIn, seqOut = c->sequenceOut;
    int inStruct = seqIn < D.n, outStruct = seqOut < D.n;
    c->updateCase = outStruct ? (inStruct ? 0 : 2) : (inStruct ? 1 : 3);
    c->slotColOut = outStruct ? D.slotOfCol[seqOut] : -1;
    c->rowOfSlackOut = outStruct ? -1 : (seqOut - D.n);
    c->slotRowIn = inStruct ? -1 : D.slotOfRow[seqIn - D.n];
  }
}

__global__ void __launch_bounds__(256) k_append_scatter_abs(Dev D, int which)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN || c->numberAppend == 0)
    return;
  if (which == 1 && c->numberFlips == 0)
    return;
  __shared__ int shi[17];
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int flag = (p < D.m) ? D.appendFlag[p] : 0;
  int total;
  int rank = blockRank(flag, total, shi);
  if (flag)
    D.infIndex[D.blockOffset[blockIdx.x] + rank] = p;
}

__global__ void __launch_bounds__(256) k_house(Dev D)
{
  if (D.ctrl->state != RUN)
    return;
  houseBody(D);
}

// row/column fix-up of the nucleus update (k_rank1_fix + k_rank1_fix2) and housekeeping, one workgroup
__global__ void __launch_bounds__(256) k_fix_house(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int k = c->k;
  const int ucase = c->updateCase;
  const double alpha = c->alpha;
  const double dir = (double)c->directionOut;
  const int a = c->slotColOut, b = c->slotRowIn, last = k - 1;
  for (int s = threadIdx.x; s < k; s += blockDim.x) {
    double slotFs = dir * D.rhoSlot[s] / alpha;  // g by row-slot
    double slotEs = D.w[D.slotPos[s]];           // w by col-slot
    if (ucase == 0) {
      D.Minv[(size_t)a * D.ld + s] = slotFs;
    } else if (ucase == 1) {
      D.Minv[(size_t)k * D.ld + s] = slotFs;
      D.Minv[(size_t)s * D.ld + k] = slotEs / alpha;
    } else if (ucase == 2) {
      if (b != last)
        D.Minv[(size_t)s * D.ld + b] = D.Minv[(size_t)s * D.ld + last];
    } else {
      D.Minv[(size_t)s * D.ld + b] = slotEs / alpha;
    }
  }
  if (ucase == 1 && threadIdx.x == 0)
    D.Minv[(size_t)k * D.ld + k] = -1.0 / alpha;
  __syncthreads();
  if (ucase == 2 && a != last) {
    for (int s = threadIdx.x; s < k; s += blockDim.x)
      D.Minv[(size_t)a * D.ld + s] = D.Minv[(size_t)last * D.ld + s];
  }
  __syncthreads();
  houseBody(D);
  // head of the next pivot (only if this one ended normally and no exit was raised)
  if (threadIdx.x == 0 && D.ctrl->state == RUN)
    chuzrPreBody(D);
}


// =============================================================================================
// v5: flips are known as soon as theta is (they do not depend on the FTRAN), so the flip right-hand
// side joins the entering column and the DSE vector in ONE three-vector FTRAN sweep over Minv.
// =============================================================================================

// dual update + flip detection only (the weights need the FTRAN and come later)
__global__ void __launch_bounds__(PRICE_BLOCK) k_dj_flags(Dev D, int nbRows)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ int shi[17];
  const double theta = c->theta;
  const double tolerance = c->dualTolerance + fmin(1.0e-2, c->largestDualError);
  const int seqIn = c->sequenceIn;
  int flag = 0;
  if ((int)blockIdx.x < nbRows) {
    int i = blockIdx.x * PRICE_BLOCK + threadIdx.x;
    if (i < D.m) {
      double alphaI = D.rho[i];
      int seq = D.n + i;
      if (alphaI != 0.0 && seq != seqIn) {
        int iStatus = (D.status[seq] & 3) - 1;
        if (iStatus) {
          double value = D.dj[seq] - theta * alphaI;
          D.dj[seq] = value;
          double mult = (iStatus == 1) ? -1.0 : ((iStatus == 2) ? 1.0 : 0.0);
          value *= mult;
          if (value < -tolerance)
            flag = 1;
        }
      }
      D.candFlag[i] = (unsigned char)flag;
    }
  } else {
    int j = D.firstColumn + ((int)blockIdx.x - nbRows) * PRICE_BLOCK + threadIdx.x;
    if (j < D.lastColumn) {
      double alphaI = D.alphaCol[j];
      if (alphaI != 0.0 && j != seqIn) {
        int iStatus = (D.status[j] & 3) - 1;
        if (iStatus) {
          double value = D.dj[j] - theta * alphaI;
          D.dj[j] = value;
          double mult = (iStatus == 1) ? -1.0 : ((iStatus == 2) ? 1.0 : -1.0);
          value *= mult;
          if (value < -tolerance && iStatus > 0)
            flag = 1;
        }
      }
      D.candFlag[D.m + j] = (unsigned char)flag;
    }
  }
  int total;
  blockRank(flag, total, shi);
  if (threadIdx.x == 0)
    D.blockCount[blockIdx.x] = total;
}

// Flip right-hand side, all flips at once (matrix_->add per flipped column, src/ClpPackedMatrix.cpp
// :4874).  One workgroup; thread e owns one (flip, entry) pair.  Rows hit by a single flip are stored
// directly; rows hit by several flips are collected, ordered by (row, flip) and summed in flip order,
// so the result is bit-identical to the sequential loop of the reference whatever the schedule.
// sequential form (very many flips or dense columns): flips in order, entries of one column in parallel
__device__ void flipSequential(Dev D)
{
  Ctrl *c = D.ctrl;
  const int nf = c->numberFlips;
  const int tid = threadIdx.x;
  double changeObj = 0.0;
  for (int f = 0; f < nf; f++) {
    int seq = D.flipSeq[f];
    int iStatus = (D.status[seq] & 3) - 1;
    double mult = (iStatus == 1) ? -1.0 : 1.0;
    if (seq >= D.n) {
      double movement = mult * (D.lower[seq] - D.upper[seq]);
      if (tid == 0) {
        changeObj -= movement * D.cost[seq];
        D.flipRhs[seq - D.n] += movement;
      }
    } else {
      double movement = mult * (D.upper[seq] - D.lower[seq]);
      if (tid == 0)
        changeObj += movement * D.cost[seq];
      for (int p = D.colStart[seq] + tid; p < D.colStart[seq + 1]; p += blockDim.x)
        D.flipRhs[D.row[p]] += movement * D.elem[p];
    }
    __syncthreads();
  }
  if (tid == 0)
    c->objectiveChange += changeObj;
}
#define FLIP_MAX_FLIPS 1024
#define FLIP_MAX_ENTRIES 8192
#define FLIP_MAX_COLLIDE 1024
__global__ void __launch_bounds__(1024) k_flip_apply2(Dev D, int nbPos)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int nf = c->numberFlips;
  const int tid = threadIdx.x;
  // counters of k_ftran_scatter3's appends (position blocks) are reset here, flips or not
  for (int b = tid; b < nbPos; b += blockDim.x)
    D.blockCount[b] = 0;
  if (nf == 0)
    return;
  __shared__ double s_mv[FLIP_MAX_FLIPS];
  __shared__ int s_start[FLIP_MAX_FLIPS + 1];
  __shared__ int s_cRow[FLIP_MAX_COLLIDE], s_cFlip[FLIP_MAX_COLLIDE], s_cSorted[FLIP_MAX_COLLIDE];
  __shared__ double s_cVal[FLIP_MAX_COLLIDE];
  __shared__ int s_nCollide, s_total;
  __shared__ double shd[16];
  bool fallback = nf > FLIP_MAX_FLIPS;
  double changeObj = 0.0;
  if (!fallback) {
    // per-flip scalars
    for (int f = tid; f < nf; f += blockDim.x) {
      int seq = D.flipSeq[f];
      int iStatus = (D.status[seq] & 3) - 1;
      double mult = (iStatus == 1) ? -1.0 : 1.0;
      double mv;
      int len;
      if (seq >= D.n) {
        mv = mult * (D.lower[seq] - D.upper[seq]);
        changeObj -= mv * D.cost[seq];
        len = 1;
      } else {
        mv = mult * (D.upper[seq] - D.lower[seq]);
        changeObj += mv * D.cost[seq];
        len = D.colStart[seq + 1] - D.colStart[seq];
      }
      s_mv[f] = mv;
      s_start[f + 1] = len;
    }
    if (tid == 0) {
      s_start[0] = 0;
      s_nCollide = 0;
    }
    __syncthreads();
    if (tid == 0) {
      int acc = 0;
      for (int f = 0; f < nf; f++) {
        acc += s_start[f + 1];
        s_start[f + 1] = acc;
      }
      s_total = acc;
    }
    __syncthreads();
    fallback = s_total > FLIP_MAX_ENTRIES;
  }
  if (fallback) {
    flipSequential(D);
    return;
  }
  const int total = s_total;
  // phase 1: count the contributors of every touched row (touchCount is all zero between calls)
  int myRow[FLIP_MAX_ENTRIES / 1024], myFlip[FLIP_MAX_ENTRIES / 1024];
  double myVal[FLIP_MAX_ENTRIES / 1024];
#pragma unroll
  for (int q = 0; q < FLIP_MAX_ENTRIES / 1024; q++) {
    int e = tid + q * 1024;
    myRow[q] = -1;
    if (e < total) {
      int lo = 0, hi = nf;  // largest f with s_start[f] <= e
      while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (s_start[mid] <= e)
          lo = mid;
        else
          hi = mid;
      }
      int f = lo;
      int seq = D.flipSeq[f];
      int r;
      double v;
      if (seq >= D.n) {
        r = seq - D.n;
        v = s_mv[f];
      } else {
        int p = D.colStart[seq] + (e - s_start[f]);
        r = D.row[p];
        v = s_mv[f] * D.elem[p];
      }
      myRow[q] = r;
      myFlip[q] = f;
      myVal[q] = v;
      atomicAdd(&D.touchCount[r], 1);
    }
  }
  __syncthreads();
  {
    // how many entries share their row with another flip?  dense columns collide everywhere:
    // clean up and take the sequential form instead
    double nColl = 0.0;
#pragma unroll
    for (int q = 0; q < FLIP_MAX_ENTRIES / 1024; q++)
      if (myRow[q] >= 0 && atomicAdd(&D.touchCount[myRow[q]], 0) > 1)
        nColl += 1.0;
    nColl = blockSum(nColl, shd);
    if (nColl > (double)FLIP_MAX_COLLIDE) {
#pragma unroll
      for (int q = 0; q < FLIP_MAX_ENTRIES / 1024; q++)
        if (myRow[q] >= 0)
          D.touchCount[myRow[q]] = 0;
      __syncthreads();
      flipSequential(D);
      return;
    }
  }
  // phase 2: single contributors store, the others queue up
  bool overflow = false;
#pragma unroll
  for (int q = 0; q < FLIP_MAX_ENTRIES / 1024; q++) {
    if (myRow[q] >= 0) {
      int cnt = atomicAdd(&D.touchCount[myRow[q]], 0);
      if (cnt == 1) {
        D.flipRhs[myRow[q]] += myVal[q];
      } else {
        int o = atomicAdd(&s_nCollide, 1);
        if (o < FLIP_MAX_COLLIDE) {
          s_cRow[o] = myRow[q];
          s_cFlip[o] = myFlip[q];
          s_cVal[o] = myVal[q];
        } else {
          overflow = true;
        }
      }
    }
  }
  __syncthreads();
  const int ncol = min(s_nCollide, FLIP_MAX_COLLIDE);
  // phase 3: order the collisions by (row, flip) with a rank sort, then one thread per row segment
  for (int i = tid; i < ncol; i += blockDim.x) {
    int r = s_cRow[i], f = s_cFlip[i], rank = 0;
    for (int j = 0; j < ncol; j++) {
      int rj = s_cRow[j], fj = s_cFlip[j];
      rank += (rj < r) || (rj == r && fj < f);
    }
    s_cSorted[rank] = i;
  }
  __syncthreads();
  for (int i = tid; i < ncol; i += blockDim.x) {
    int e = s_cSorted[i];
    int r = s_cRow[e];
    if (i == 0 || s_cRow[s_cSorted[i - 1]] != r) {
      double acc = D.flipRhs[r];
      for (int j = i; j < ncol && s_cRow[s_cSorted[j]] == r; j++)
        acc += s_cVal[s_cSorted[j]];
      D.flipRhs[r] = acc;
    }
  }
  // phase 4: clean the counters
#pragma unroll
  for (int q = 0; q < FLIP_MAX_ENTRIES / 1024; q++)
    if (myRow[q] >= 0)
      D.touchCount[myRow[q]] = 0;
  double s = blockSum(changeObj, shd);
  if (tid == 0) {
    c->objectiveChange += s;
    if (overflow || s_nCollide > FLIP_MAX_COLLIDE)
      c->state = EXIT_BAD_UPDATE;  // cannot happen with the caps above unless thousands of flips collide
  }
}

// three right-hand sides in one sweep over Minv: entering column, DSE vector (rho), flip rhs.
// The gathered right-hand sides (v[slotRow[sr]]) are staged once per workgroup in LDS (chunks of
// GEMV_TILE slots), each wave then streams GEMV_ROWS rows of Minv against them.
#define GEMV_TILE 2048
#define GEMV_ROWS 4
__global__ void __launch_bounds__(256) k_gemv3g(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double s1[GEMV_TILE], s2[GEMV_TILE], s3[GEMV_TILE];
  const int k = c->k;
  const bool doTau = c->pivotRule != 0, doFlip = c->numberFlips != 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rowsPerBlock = 4 * GEMV_ROWS;
  for (int base = blockIdx.x * rowsPerBlock; base < k; base += gridDim.x * rowsPerBlock) {
    double a1[GEMV_ROWS], a2[GEMV_ROWS], a3[GEMV_ROWS];
#pragma unroll
    for (int q = 0; q < GEMV_ROWS; q++)
      a1[q] = a2[q] = a3[q] = 0.0;
    for (int t0 = 0; t0 < k; t0 += GEMV_TILE) {
      const int tn = min(GEMV_TILE, k - t0);
      __syncthreads();
      for (int i = threadIdx.x; i < tn; i += blockDim.x) {
        int r = D.slotRow[t0 + i];
        s1[i] = D.vecV1[r];
        s2[i] = doTau ? D.rho[r] : 0.0;
        s3[i] = doFlip ? D.flipRhs[r] : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < GEMV_ROWS; q++) {
        int sc = base + wv * GEMV_ROWS + q;
        if (sc < k) {
          const double *Mrow = D.Minv + (size_t)sc * D.ld + t0;
          for (int i = lane; i < tn; i += 64) {
            double mv = Mrow[i];
            a1[q] += mv * s1[i];
            a2[q] += mv * s2[i];
            a3[q] += mv * s3[i];
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < GEMV_ROWS; q++) {
      int sc = base + wv * GEMV_ROWS + q;
      double r1 = waveSum(a1[q]), r2 = waveSum(a2[q]), r3 = waveSum(a3[q]);
      if (lane == 0 && sc < k) {
        D.slotC[sc] = r1;
        D.slotD[sc] = r2;
        D.slotE[sc] = r3;
      }
    }
  }
}

// back end of the three FTRANs: w, tau and -- when there are flips -- x3 together with the primal
// update it drives (ratio 1.0, ClpSimplexDual.cpp:1535-1536)
__global__ void __launch_bounds__(256) k_ftran_scatter3(Dev D, int nbNorm)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double shd[16];
  const int k = c->k;
  const bool doFlip = c->numberFlips != 0;
  const double tolerance = c->primalTolerance;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int p = -1;
  double x1 = 0.0, x2 = 0.0, x3 = 0.0;
  if (t < D.m) {
    p = D.posOfSlack[t];
    if (p >= 0) {
      double a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int s = D.rowStart[t], e = s + D.basicCount[t];
      for (int q = s; q < e; q++) {
        int sc = D.slotOfCol[D.ccol[q]];
        double a = D.relem[q];
        a1 += a * D.slotC[sc];
        a2 += a * D.slotD[sc];
        if (doFlip)
          a3 += a * D.slotE[sc];
      }
      x1 = a1 - D.vecV1[t];
      x2 = a2 - D.rho[t];
      if (doFlip)
        x3 = a3 - D.flipRhs[t];
    }
    if (doFlip)
      D.flipRhs[t] = 0.0;  // consumed (nucleus rows were read by k_gemv3g)
  } else if (t < D.m + k) {
    int sc = t - D.m;
    p = D.slotPos[sc];
    x1 = D.slotC[sc];
    x2 = D.slotD[sc];
    x3 = D.slotE[sc];
  }
  // DSE norm for the weight update (ClpDualRowSteepest::updateWeights :516-538): sum of the
  // per-block partials of sum rho^2; alpha is still the ratio-test alpha here
  double norm = 0.0, multiplier = 0.0;
  if (c->pivotRule) {
    double acc = 0.0;
    for (int b = threadIdx.x; b < nbNorm; b += blockDim.x)
      acc += D.normPartial[b];
    acc = blockSum(acc, shd);
    const double alphaOld = c->alpha;
    norm = acc / (alphaOld * alphaOld);
    multiplier = 2.0 / alphaOld;
  }
  double changeObj = 0.0;
  if (p >= 0) {
    D.w[p] = x1;
    D.tau[p] = x2;
    if (c->pivotRule && x1 != 0.0) {
      double devex = D.weights[p];
      D.altWeights[p] = devex;
      if (p == c->pivotRow) {
        devex = (norm < DEVEX_TRY_NORM) ? DEVEX_TRY_NORM : norm;
      } else {
        devex += x1 * (x1 * norm + x2 * multiplier);
        if (devex < DEVEX_TRY_NORM)
          devex = DEVEX_TRY_NORM;
      }
      D.weights[p] = devex;
    }
    if (doFlip) {
      int append = 0;
      if (x3 != 0.0) {
        int iPivot = D.pivotVariable[p];
        double value = D.sol[iPivot];
        value -= x3;
        changeObj -= x3 * D.cost[iPivot];
        D.sol[iPivot] = value;
        if (c->pivotRule) {
          double lower = D.lower[iPivot], upper = D.upper[iPivot];
          double old = D.infeas[p];
          if (value < lower - tolerance) {
            value -= lower;
            value *= value;
            if (old == 0.0)
              append = 1;
            D.infeas[p] = value;
          } else if (value > upper + tolerance) {
            value -= upper;
            value *= value;
            if (old == 0.0)
              append = 1;
            D.infeas[p] = value;
          } else if (old != 0.0) {
            D.infeas[p] = REALLY_TINY;
          }
        }
      }
      D.appendFlag[p] = append;
      if (append)
        atomicAdd(&D.blockCount[p >> 8], 1);
    }
  }
  double s = blockSum(changeObj, shd);
  if (threadIdx.x == 0)
    D.blockSum[blockIdx.x] = s;
}

// DSE weight update (needs w, tau) -- positions only
__global__ void __launch_bounds__(256) k_weights2(Dev D, int nbNorm)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN || !c->pivotRule)
    return;
  __shared__ double shd[16];
  double acc = 0.0;
  for (int b = threadIdx.x; b < nbNorm; b += blockDim.x)
    acc += D.normPartial[b];
  acc = blockSum(acc, shd);
  const double alphaOld = c->alpha;  // still the ratio-test alpha: the accuracy test comes after
  const double norm = acc / (alphaOld * alphaOld);
  const double multiplier = 2.0 / alphaOld;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D.m) {
    double thetaW = D.w[i];
    if (thetaW != 0.0) {
      double devex = D.weights[i];
      D.altWeights[i] = devex;
      if (i == c->pivotRow) {
        devex = (norm < DEVEX_TRY_NORM) ? DEVEX_TRY_NORM : norm;
      } else {
        devex += thetaW * (thetaW * norm + D.tau[i] * multiplier);
        if (devex < DEVEX_TRY_NORM)
          devex = DEVEX_TRY_NORM;
      }
      D.weights[i] = devex;
    }
  }
}

// flip-count scan only (the accuracy test moved to k_scan_tail(which = 1), after the FTRAN)
__global__ void __launch_bounds__(1024) k_scan_flips(Dev D, int nb)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ int shi[17];
  __shared__ int s_base;
  if (threadIdx.x == 0)
    s_base = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += blockDim.x) {
    int b = b0 + threadIdx.x;
    int cnt = (b < nb) ? D.blockCount[b] : 0;
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int v = cnt;
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(v, o);
      if (lane >= o)
        v += t;
    }
    __syncthreads();
    if (lane == 63)
      shi[wv] = v;
    __syncthreads();
    int base = s_base;
    for (int i = 0; i < wv; i++)
      base += shi[i];
    if (b < nb)
      D.blockOffset[b] = base + v - cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int i = 0; i < nw; i++)
        tot += shi[i];
      s_base += tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    c->numberFlips = s_base;
}

__global__ void k_zero(double *p, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    p[i] = 0.0;
}
__global__ void k_zero_if_flips(Dev D, double *p, int n)
{
  if (D.ctrl->numberFlips == 0)
    return;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    p[i] = 0.0;
}

// =============================================================================================
// Refactorization of the nucleus: gather C = A[R,K], Gauss-Jordan with partial pivoting whose
// arithmetic on the not-yet-pivoted rows is exactly the right-looking LU of
// CoinAbcDenseFactorization::factor (src/CoinAbcDenseFactorization.cpp:262-313): multiplier
// l_j = a_ji * (1/pivot), a_jc -= a_ic * l_j, first-largest pivot in physical row order.
// The same row operations applied to the identity give X with X*C = D, so Minv = D^-1 X.
// =============================================================================================
__global__ void k_gather_nucleus(Dev D, const int *kcol, const int *localOfRow, int k)
{
  // one block per nucleus column
  int cidx = blockIdx.x;
  if (cidx >= k)
    return;
  int j = kcol[cidx];
  for (int p = D.colStart[j] + threadIdx.x; p < D.colStart[j + 1]; p += blockDim.x) {
    int r = localOfRow[D.row[p]];
    if (r >= 0)
      D.workW[(size_t)r * D.ld + cidx] = D.elem[p];
  }
}
__global__ void k_identity(Dev D, int k)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k)
    D.workX[(size_t)i * D.ld + i] = 1.0;
}

// pivot search in column i among physical rows >= i (first largest, > zeroTolerance)
__global__ void __launch_bounds__(1024) k_gj_pivot(Dev D, int i, int k, int *info /*[0]=singular flag, [1]=pivot row*/)
{
  __shared__ double shv[16];
  __shared__ int shk[16];
  if (info[0])
    return;
  double best = D.ctrl->zeroTolerance;
  int key = -1;
  for (int j = i + threadIdx.x; j < k; j += blockDim.x) {
    double v = fabs(D.workW[(size_t)j * D.ld + i]);
    if (v > best) {
      best = v;
      key = j;
    }
  }
  blockArgMax(best, key, shv, shk);
  if (threadIdx.x == 0) {
    if (key < 0)
      info[0] = 1 + i;
    info[1] = key;
  }
}
// swap physical rows i and pivot row in W and X, record permutation, store multipliers
__global__ void k_gj_swap(Dev D, int i, int k, int *info)
{
  if (info[0])
    return;
  int iRow = info[1];
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (iRow != i && j < k) {
    size_t a = (size_t)i * D.ld + j, b = (size_t)iRow * D.ld + j;
    double t = D.workW[a];
    D.workW[a] = D.workW[b];
    D.workW[b] = t;
    t = D.workX[a];
    D.workX[a] = D.workX[b];
    D.workX[b] = t;
  }
  if (j == 0 && iRow != i) {
    int t = D.perm[i];
    D.perm[i] = D.perm[iRow];
    D.perm[iRow] = t;
  }
}
// multipliers l_r = W[r][i] * (1/pivot) for every row r != i (kept in slotA), pivot inverse in slotB[i]
__global__ void k_gj_mult(Dev D, int i, int k, int *info)
{
  if (info[0])
    return;
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < k) {
    double pivotValue = 1.0 / D.workW[(size_t)i * D.ld + i];
    if (r == i) {
      D.slotB[i] = pivotValue;
      D.slotA[r] = 0.0;
    } else {
      D.slotA[r] = D.workW[(size_t)r * D.ld + i] * pivotValue;
    }
  }
}
// pivot search + row swap + multipliers of one elimination step, one workgroup
__global__ void __launch_bounds__(1024) k_gj_step(Dev D, int i, int k, int *info)
{
  __shared__ double shv[16];
  __shared__ int shk[16];
  __shared__ int s_row;
  if (info[0])
    return;
  double best = D.ctrl->zeroTolerance;
  int key = -1;
  for (int j = i + threadIdx.x; j < k; j += blockDim.x) {
    double v = fabs(D.workW[(size_t)j * D.ld + i]);
    if (v > best) {
      best = v;
      key = j;
    }
  }
  blockArgMax(best, key, shv, shk);
  if (threadIdx.x == 0) {
    if (key < 0)
      info[0] = 1 + i;
    info[1] = key;
    s_row = key;
  }
  __syncthreads();
  const int iRow = s_row;
  if (iRow < 0)
    return;
  if (iRow != i) {
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
      size_t a = (size_t)i * D.ld + j, b = (size_t)iRow * D.ld + j;
      double t = D.workW[a];
      D.workW[a] = D.workW[b];
      D.workW[b] = t;
      t = D.workX[a];
      D.workX[a] = D.workX[b];
      D.workX[b] = t;
    }
    if (threadIdx.x == 0) {
      int t = D.perm[i];
      D.perm[i] = D.perm[iRow];
      D.perm[iRow] = t;
    }
  }
  __syncthreads();
  const double pivotValue = 1.0 / D.workW[(size_t)i * D.ld + i];
  for (int r = threadIdx.x; r < k; r += blockDim.x) {
    if (r == i) {
      D.slotB[i] = pivotValue;
      D.slotA[r] = 0.0;
    } else {
      D.slotA[r] = D.workW[(size_t)r * D.ld + i] * pivotValue;
    }
  }
}
// row_r -= l_r * row_i  for all r != i, on W (columns > i) and X (all columns)
__global__ void __launch_bounds__(256) k_gj_elim(Dev D, int i, int k, int *info)
{
  if (info[0])
    return;
  const double *Wi = D.workW + (size_t)i * D.ld;
  const double *Xi = D.workX + (size_t)i * D.ld;
  for (int r = blockIdx.y; r < k; r += gridDim.y) {
    if (r == i)
      continue;
    double l = D.slotA[r];
    if (l == 0.0)
      continue;
    double *Wr = D.workW + (size_t)r * D.ld;
    double *Xr = D.workX + (size_t)r * D.ld;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < k; j += gridDim.x * blockDim.x) {
      if (j > i)
        Wr[j] -= Wi[j] * l;
      Xr[j] -= Xi[j] * l;
    }
  }
}
// Minv = D^-1 X
__global__ void __launch_bounds__(256) k_gj_finish(Dev D, int k)
{
  for (int r = blockIdx.y; r < k; r += gridDim.y) {
    double inv = D.slotB[r];
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < k; j += gridDim.x * blockDim.x)
      D.Minv[(size_t)r * D.ld + j] = D.workX[(size_t)r * D.ld + j] * inv;
  }
}

// =============================================================================================
// full-length matrix products for the resync after a refactorization
//   ClpPackedMatrix::times :296 (by the row copy: deterministic, no atomics)
//   ClpPackedMatrix::transposeTimes :362
// =============================================================================================
__global__ void k_times(Dev D, double scalar, const double *x, double *y)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D.m) {
    double acc = 0.0;
    for (int q = D.rowStart[i]; q < D.rowStart[i + 1]; q++)
      acc += D.relem[q] * x[D.ccol[q]];
    y[i] += scalar * acc;
  }
}
__global__ void k_transpose_times(Dev D, double scalar, const double *x, double *y)
{
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < D.n) {
    double value = 0.0;
    for (int p = D.colStart[j]; p < D.colStart[j + 1]; p++)
      value += x[D.row[p]] * D.elem[p];
    y[j] += value * scalar;
  }
}

// computePrimals helpers (src/ClpSimplex.cpp:914): zero basics, rhs = rowActivity - A x_N
__global__ void k_zero_basic(Dev D)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m)
    D.sol[D.pivotVariable[p]] = 0.0;
}
__global__ void k_primal_rhs(Dev D, double *rhs)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D.m) {
    double acc = 0.0;
    for (int q = D.rowStart[i]; q < D.rowStart[i + 1]; q++)
      acc += D.relem[q] * D.sol[D.ccol[q]];
    rhs[i] = -acc + D.sol[D.n + i];
  }
}
// max |(A x)_i - s_i| per block (largestPrimalError of computePrimals)
__global__ void __launch_bounds__(256) k_primal_residual(Dev D)
{
  __shared__ double sh[16];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double r = 0.0;
  if (i < D.m) {
    double acc = 0.0;
    for (int q = D.rowStart[i]; q < D.rowStart[i + 1]; q++)
      acc += D.relem[q] * D.sol[D.ccol[q]];
    r = fabs(acc - D.sol[D.n + i]);
  }
  // max via min of negatives
  double mx = -blockMin(-r, sh);
  if (threadIdx.x == 0)
    D.normPartial[blockIdx.x] = mx;
}
__global__ void k_store_basic(Dev D, const double *x)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m)
    D.sol[D.pivotVariable[p]] = x[p];
}
// computeDuals helpers (src/ClpSimplex.cpp:1164)
__global__ void k_basic_costs(Dev D, double *cB)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m)
    cB[p] = D.cost[D.pivotVariable[p]];
}
__global__ void k_djs(Dev D, const double *y)
{
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < D.n) {
    double value = 0.0;
    for (int p = D.colStart[t]; p < D.colStart[t + 1]; p++)
      value += y[D.row[p]] * D.elem[p];
    D.dj[t] = D.cost[t] + value * -1.0;
  } else if (t < D.N) {
    D.dj[t] = y[t - D.n] + D.cost[t];
  }
}

// saveWeights (src/ClpDualRowSteepest.cpp:773): weights follow their sequence across a
// refactorization; mode >= 2 rebuilds the infeasibility list in ascending position order
__global__ void k_weights_to_seq(Dev D)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m)
    D.weightBySeq[D.pivotVariable[p]] = D.weights[p];
}
__global__ void k_weights_from_seq(Dev D, int initialize)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m) {
    double wgt = 1.0;
    if (!initialize) {
      wgt = D.weightBySeq[D.pivotVariable[p]];
      if (wgt < 0.0)
        wgt = 1.0;  // "odd": was not basic at save time
      else if (wgt < DEVEX_TRY_NORM)
        wgt = DEVEX_TRY_NORM;
    }
    D.weights[p] = wgt;
  }
}
__global__ void k_fill(double *p, double v, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    p[i] = v;
}
__global__ void __launch_bounds__(256) k_infeas_flags(Dev D)
{
  __shared__ int shi[17];
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int flag = 0;
  if (p < D.m) {
    const double tolerance = D.ctrl->primalTolerance;
    int iPivot = D.pivotVariable[p];
    double value = D.sol[iPivot], lower = D.lower[iPivot], upper = D.upper[iPivot];
    double inf = 0.0;
    if (value < lower - tolerance) {
      value -= lower;
      inf = value * value;
      flag = 1;
    } else if (value > upper + tolerance) {
      value -= upper;
      inf = value * value;
      flag = 1;
    }
    D.infeas[p] = inf;
    D.appendFlag[p] = flag;
  }
  int total;
  blockRank(flag, total, shi);
  if (threadIdx.x == 0)
    D.blockCount[blockIdx.x] = total;
}
__global__ void k_infeas_finish(Dev D)
{
  D.ctrl->numberInfeasible = D.ctrl->numberAppend;
  D.ctrl->numberAppend = 0;
}
__global__ void k_set_state(Dev D, int state)
{
  D.ctrl->state = state;
}

}  // namespace clpgpu

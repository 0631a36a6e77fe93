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

constexpr double REALLY_TINY = 1.0e-100;  // COIN_INDEXED_REALLY_TINY_ELEMENT
constexpr double DEVEX_TRY_NORM = 1.0e-4; // src/ClpSimplex.hpp:2056

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
// maximum that keeps a NaN (fmax drops it): residual checks must see a poisoned factorization
__device__ inline double nanMax(double a, double b)
{
  return !(b <= a) ? b : a;
}
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
// block-wide integer sum; result valid in every thread
__device__ inline int blockSumInt(int v, int *sh /*[17]*/)
{
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int o = 32; o > 0; o >>= 1)
    v += __shfl_xor(v, o);
  __syncthreads();
  if (lane == 0)
    sh[wv] = v;
  __syncthreads();
  int t = 0;
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

// coherent (L1-bypassing) loads: data written by other workgroups of the same launch
__device__ inline int ldc(const int *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double ldc(const double *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// coherent (write-through) stores: results another workgroup of the same launch will read with ldc()
__device__ inline void stc(int *p, int v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void stc(double *p, double v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// "last workgroup done": true in exactly one workgroup of the launch, after every workgroup has
// published its results.  Lets the serial tail of a grid-wide pass (scan of the per-block counts,
// scalar bookkeeping) run inside the same launch.  The eight XCDs have separate L2s, so a
// device-scope fence here would write back each L2 once per workgroup; instead everything the tail
// reads is published with stc() (write-through) and read with ldc(), and the ticket only has to wait
// for those stores to complete.
// (nblocks / me: the participating workgroups and this one's index among them, when only a leading
// part of the grid takes part)
__device__ inline bool lastBlockDone(Ctrl *c, int slot, int nblocks = -1, int me = -1)
{
  if (nblocks < 0) {
    nblocks = gridDim.x;
    me = blockIdx.x;
  }
  __shared__ int s_last;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) {
    // two-level count (atomics on one address serialise at ~15 ns each across the XCDs): first the
    // counter of this workgroup's group of 32, then -- last of the group only -- the launch counter
    const int g = me >> 5, ngroups = (nblocks + 31) >> 5;
    const int gsize = min(32, nblocks - (g << 5));
    int last = 0;
    int *gc = &c->ticketGroup[slot][g & 63];
    if (ngroups > 64) {
      // (launches beyond 2048 workgroups: single level)
      int t = __hip_atomic_fetch_add(&c->ticket[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (t == nblocks - 1);
    } else if (__hip_atomic_fetch_add(gc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
      __hip_atomic_store(gc, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int t = __hip_atomic_fetch_add(&c->ticket[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (t == ngroups - 1);
    }
    if (last)
      __hip_atomic_store(&c->ticket[slot], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}
__device__ inline void scanTailBody(const Dev &D, int nbCount, int nbSum, int which, int alphaTest, int parity);
__device__ inline double randomDouble(Ctrl *c)
{
  // CoinThreadRandom::randomDouble, 32-bit LCG form [CoinUtils, not in the reference tree]
  c->seed = 1664525u * c->seed + 1013904223u;
  return ((double)c->seed) / 4294967296.0;
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

__global__ void k_btran_t(Dev D, const double *cvec, const double *y, double *t, int iter, int wide = 0)
{
  if (iter && D.ctrl->state != RUN)
    return;
  if (wide) {
    // long columns: a wave per column-slot (fixed 64-way tree)
    const int sc = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sc < D.ctrl->k) {
      const int col = D.slotCol[sc];
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
    return;
  }
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
// COHERENT: run by the last workgroup of the launch that produced the counts (ldc loads)
template <bool COHERENT> __device__ inline void scanBlocksBody(const Dev &D, int nb, int what, int nSell)
{
  Ctrl *c = D.ctrl;
  __shared__ int shi[17];
  __shared__ double shd[16];
  __shared__ int s_base;
  if (threadIdx.x == 0)
    s_base = 0;
  __syncthreads();
  double vmin = 1.0e31;
  double bytes = 0.0;
  const int T = blockDim.x, tid = threadIdx.x;
  const int per = (nb + T - 1) / T;
  if (per <= 8) {
    // one round of loads: thread t owns `per` consecutive blocks, everything it needs is requested
    // before anything is used (the coherent loads are slow and would otherwise queue up serially)
    const int lo = min(nb, tid * per), hi = min(nb, lo + per);
    int cnt[8];
    double mn[8], sm[8], smin[4], sbytes[4];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int bb = lo + u;
      cnt[u] = bb < hi ? (COHERENT ? ldc(&D.blockCount[bb]) : D.blockCount[bb]) : 0;
      mn[u] = (what == 0 && bb < hi) ? (COHERENT ? ldc(&D.blockMin[bb]) : D.blockMin[bb]) : 1.0e31;
      sm[u] = (what == 0 && bb < hi) ? (COHERENT ? ldc(&D.blockSum[bb]) : D.blockSum[bb]) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int bb = tid + u * T;
      smin[u] = (what == 0 && bb < nSell) ? D.sellMin[bb] : 1.0e31;
      sbytes[u] = (what == 0 && bb < nSell) ? D.sellBytes[bb] : 0.0;
    }
    int local = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      local += cnt[u];
      vmin = fmin(vmin, mn[u]);
      bytes += sm[u];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      vmin = fmin(vmin, smin[u]);
      bytes += sbytes[u];
    }
    if (what == 0)
      for (int bb = tid + 4 * T; bb < nSell; bb += T) {
        vmin = fmin(vmin, D.sellMin[bb]);
        bytes += D.sellBytes[bb];
      }
    const int lane = tid & 63, wv = tid >> 6, nw = T >> 6;
    int v = local;
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(v, o);
      if (lane >= o)
        v += t;
    }
    if (lane == 63)
      shi[wv] = v;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < nw; i++) {
      if (i < wv)
        base += shi[i];
      tot += shi[i];
    }
    int o = base + v - local;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (lo + u < hi)
        D.blockOffset[lo + u] = o;
      o += cnt[u];
    }
    if (tid == 0)
      s_base = tot;
  } else {
    for (int b0 = 0; b0 < nb; b0 += blockDim.x) {
      int b = b0 + threadIdx.x;
      int cnt = (b < nb) ? (COHERENT ? ldc(&D.blockCount[b]) : D.blockCount[b]) : 0;
      if (what == 0 && b < nb) {
        vmin = fmin(vmin, (COHERENT ? ldc(&D.blockMin[b]) : D.blockMin[b]));
        bytes += (COHERENT ? ldc(&D.blockSum[b]) : D.blockSum[b]);
      }
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
    if (what == 0)
      for (int b = threadIdx.x; b < nSell; b += blockDim.x) {
        vmin = fmin(vmin, D.sellMin[b]);
        bytes += D.sellBytes[b];
      }
  }
  if (what == 0) {
    vmin = blockMin(vmin, shd);
    bytes = blockSum(bytes, shd);
  }
  if (threadIdx.x == 0) {
    if (what == 0) {
      c->numberCandidates = s_base;
      c->upperTheta = vmin;
      c->dcArrive = 0;  // (as in k_cand_scatter's own scan)
      if (c->dcWide > 0)
        c->dcWide = 0;
      // algorithmic bytes of this pricing launch: per scanned column 12*len+4 (+20 per
      // emitted nonzero), plus status 1*n, pi 8*m, one extra colStart
      if (c->lastPriceByRow) {
        // B_row: 12 B per visited row entry and per pi nonzero, 8 per touched column, 20 per emitted one
        c->statRowBytes += bytes;
        c->statRowLaunches += 1.0;
      } else {
        c->statPriceBytes += bytes + (double)(D.lastColumn - D.firstColumn) + 8.0 * D.m + 4.0;
      }
      c->statPriceLaunches += 1.0;
    } else if (what == 1) {
      c->numberFlips = s_base;
    } else {
      c->numberAppend = s_base;
    }
  }
}
__global__ void __launch_bounds__(1024) k_scan_blocks(Dev D, int nb, int what, int iter, int nSell = 0)
{
  if (iter && D.ctrl->state != RUN)
    return;
  scanBlocksBody<false>(D, nb, what, nSell);
}

// nSell >= 0: no separate scan launch -- every workgroup that has candidates sums the counts of the
// workgroups before it and takes the min over all first-pass ratios itself (a few thousand L2 reads,
// all in parallel); workgroup 0 also leaves the totals in the control block for the ratio test.
__global__ void __launch_bounds__(PRICE_BLOCK) k_cand_scatter(Dev D, int nbRows, int nSell = -1)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ int shi[17];
  __shared__ double shd[16];
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
  int offset;
  double upperTheta;
  if (nSell >= 0) {
    offset = 0;
    upperTheta = 1.0e31;
    if (total > 0 || blockIdx.x == 0) {  // uniform per workgroup
      const int nb = gridDim.x, me = blockIdx.x, tid = threadIdx.x;
      int pre = 0, all = 0;
      double vmin = 1.0e31, bytes = 0.0;
      if (nb <= 8 * PRICE_BLOCK && nSell <= 8 * PRICE_BLOCK) {
        int cnt[8];
        double mn[8], sm[8], smn[8], sby[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int b = tid + u * PRICE_BLOCK;
          cnt[u] = b < nb ? D.blockCount[b] : 0;
          mn[u] = b < nb ? D.blockMin[b] : 1.0e31;
          sm[u] = (me == 0 && b < nb) ? D.blockSum[b] : 0.0;
          smn[u] = b < nSell ? D.sellMin[b] : 1.0e31;
          sby[u] = (me == 0 && b < nSell) ? D.sellBytes[b] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int b = tid + u * PRICE_BLOCK;
          all += cnt[u];
          if (b < me)
            pre += cnt[u];
          vmin = fmin(vmin, fmin(mn[u], smn[u]));
          bytes += sm[u] + sby[u];
        }
      } else {
        for (int b = tid; b < nb; b += PRICE_BLOCK) {
          const int cnt = D.blockCount[b];
          all += cnt;
          if (b < me)
            pre += cnt;
          vmin = fmin(vmin, D.blockMin[b]);
          if (me == 0)
            bytes += D.blockSum[b];
        }
        for (int b = tid; b < nSell; b += PRICE_BLOCK) {
          vmin = fmin(vmin, D.sellMin[b]);
          if (me == 0)
            bytes += D.sellBytes[b];
        }
      }
      offset = blockSumInt(pre, shi);
      upperTheta = blockMin(vmin, shd);
      if (me == 0) {
        all = blockSumInt(all, shi);
        bytes = blockSum(bytes, shd);
        if (tid == 0) {
          c->numberCandidates = all;
          c->upperTheta = upperTheta;
          c->dcArrive = 0;  // k_dual_column_wide: grid-barrier arrivals of this pivot, and whether the pivot's list is its
          if (c->dcWide > 0)
            c->dcWide = 0;
          // algorithmic bytes of this pricing launch, as in scanBlocksBody
          if (c->lastPriceByRow) {
            // B_row: 12 B per visited row entry and per pi nonzero, 8 per touched column, 20 per emitted one
            c->statRowBytes += bytes;
            c->statRowLaunches += 1.0;
          } else {
            c->statPriceBytes += bytes + (double)(D.lastColumn - D.firstColumn) + 8.0 * D.m + 4.0;
          }
          c->statPriceLaunches += 1.0;
        }
      }
    }
  } else {
    offset = D.blockOffset[blockIdx.x];
    upperTheta = c->upperTheta;
  }
  int cls = 3;
  int o = -1;
  if (flag) {
    o = offset + rank;
    D.candSeq[o] = seq;
    D.candAlpha[o] = alpha;
    // breakpoint of the coarse ratio passes (ClpSimplexDual.cpp:4384 / :4412) against theta0
    const double tol = c->dualTolerance;
    const double djv = D.dj[seq];
    const double range = D.upper[seq] - D.lower[seq];
    const double x = (alpha < 0.0) ? (djv - tol) / alpha : (djv + tol) / alpha;
    const double theta0 = fmax(10.0 * upperTheta, 1.0e-7);
    cls = (x <= theta0 * 8.0) ? 0 : ((x <= theta0 * 256.0) ? 1 : ((x <= theta0 * 16384.0) ? 2 : 3));
    D.candLive[o] = (unsigned char)cls;
    // what the ratio test reads per candidate, gathered here by the whole chip instead of by its one workgroup
    D.candDj[o] = djv;
    D.candRange[o] = range;
  }
  // per-class counts of this block (summed by the ratio test when it needs them) and, per candidate,
  // how many candidates of each class precede it inside the block: the ratio test's working set (all
  // candidates of class <= J) is then placed by  prefix[block] + rank  with no serial pass
  __shared__ int shc[PRICE_BLOCK / 64][3];
  int before[3];
  const int lane = threadIdx.x & 63, wvi = threadIdx.x >> 6;
  for (int j = 0; j < 3; j++) {
    unsigned long long mk = __ballot(cls == j);
    before[j] = (int)__popcll(mk & ((1ull << lane) - 1ull));
    if (lane == 0)
      shc[wvi][j] = (int)__popcll(mk);
  }
  __syncthreads();
  if (flag) {
    for (int j = 0; j < 3; j++)
      for (int w = 0; w < wvi; w++)
        before[j] += shc[w][j];
    const int r0 = before[0], r1 = r0 + before[1], r2 = r1 + before[2];
    D.candRk[o] = r0 | (r1 << 10) | (r2 << 20);
    D.candBlk[o] = (int)blockIdx.x;
  }
  if (threadIdx.x < 3) {
    int s = 0;
    for (int w = 0; w < PRICE_BLOCK / 64; w++)
      s += shc[w][threadIdx.x];
    D.classBlock[3 * blockIdx.x + threadIdx.x] = s;
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
  double thru, incr, ut, sumBad, bestPivot, ut2;
  int bestIdx;
  double bestDj, bestAlpha;  // dj and signed alpha of the argmax (travel with it when F_BESTV)
};
enum { F_THRU = 1, F_INCR = 2, F_UT = 4, F_BAD = 8, F_BEST = 16, F_UT2 = 32, F_BESTV = 64 };
// butterfly exchange inside a wave: stages 0-3 stay inside a 16-lane row (DPP quad_perm xor 1,
// xor 2, row_half_mirror, row_mirror), stages 4-5 cross rows.  Both partners combine the same two
// values with a commutative operation, so every lane ends with bit-identical results and no
// broadcast is needed; the tree is fixed => deterministic.
template <int STAGE> __device__ inline int xchgI(int v)
{
  if constexpr (STAGE == 0)
    return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);
  else if constexpr (STAGE == 1)
    return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);
  else if constexpr (STAGE == 2)
    return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);
  else if constexpr (STAGE == 3)
    return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false);
  else if constexpr (STAGE == 4)
    return __shfl_xor(v, 16);
  else
    return __shfl_xor(v, 32);
}
template <int STAGE> __device__ inline double xchgD(double v)
{
  if constexpr (STAGE < 4)
    return __hiloint2double(xchgI<STAGE>(__double2hiint(v)), xchgI<STAGE>(__double2loint(v)));
  else
    return __shfl_xor(v, STAGE == 4 ? 16 : 32);
}
template <int F, int STAGE> __device__ inline void dcStage(DcAcc &a)
{
  if constexpr (F & F_THRU)
    a.thru += xchgD<STAGE>(a.thru);
  if constexpr (F & F_INCR)
    a.incr += xchgD<STAGE>(a.incr);
  if constexpr (F & F_BAD)
    a.sumBad += xchgD<STAGE>(a.sumBad);
  if constexpr (F & F_UT)
    a.ut = fmin(a.ut, xchgD<STAGE>(a.ut));
  if constexpr (F & F_UT2)
    a.ut2 = fmin(a.ut2, xchgD<STAGE>(a.ut2));
  if constexpr (F & F_BEST) {
    double ov = xchgD<STAGE>(a.bestPivot);
    int ok = xchgI<STAGE>(a.bestIdx);
    double od = 0.0, oa = 0.0;
    if constexpr (F & F_BESTV) {
      od = xchgD<STAGE>(a.bestDj);
      oa = xchgD<STAGE>(a.bestAlpha);
    }
    if (ok >= 0 && (a.bestIdx < 0 || ov > a.bestPivot || (ov == a.bestPivot && ok < a.bestIdx))) {
      a.bestPivot = ov;
      a.bestIdx = ok;
      if constexpr (F & F_BESTV) {
        a.bestDj = od;
        a.bestAlpha = oa;
      }
    }
  }
}
// all-lanes result of the combined reduction (sum, sum, min, sum, argmax-first, min) inside one
// wave: no LDS, no barrier
template <int F> __device__ inline void dcReduceWave(DcAcc &a)
{
  dcStage<F, 0>(a);
  dcStage<F, 1>(a);
  dcStage<F, 2>(a);
  dcStage<F, 3>(a);
  dcStage<F, 4>(a);
  dcStage<F, 5>(a);
}
// the same across a workgroup with a single LDS exchange
template <int F> __device__ inline void dcReduce(DcAcc &a, double (*shd)[16], int *shk)
{
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  dcReduceWave<F>(a);
  __syncthreads();
  if (lane == 0) {
    shd[0][wv] = a.thru;
    shd[1][wv] = a.incr;
    shd[2][wv] = a.ut;
    shd[3][wv] = a.sumBad;
    shd[4][wv] = a.bestPivot;
    shd[5][wv] = a.ut2;
    shd[6][wv] = a.bestDj;
    shd[7][wv] = a.bestAlpha;
    shk[wv] = a.bestIdx;
  }
  __syncthreads();
  a.thru = a.incr = a.sumBad = 0.0;
  a.ut = shd[2][0];
  a.ut2 = shd[5][0];
  a.bestPivot = shd[4][0];
  a.bestIdx = shk[0];
  a.bestDj = shd[6][0];
  a.bestAlpha = shd[7][0];
  for (int i = 0; i < nw; i++) {
    if constexpr (F & F_THRU)
      a.thru += shd[0][i];
    if constexpr (F & F_INCR)
      a.incr += shd[1][i];
    if constexpr (F & F_BAD)
      a.sumBad += shd[3][i];
    if constexpr (F & F_UT)
      a.ut = fmin(a.ut, shd[2][i]);
    if constexpr (F & F_UT2)
      a.ut2 = fmin(a.ut2, shd[5][i]);
    if constexpr (F & F_BEST) {
      if (i && shk[i] >= 0 && (a.bestIdx < 0 || shd[4][i] > a.bestPivot || (shd[4][i] == a.bestPivot && shk[i] < a.bestIdx))) {
        a.bestPivot = shd[4][i];
        a.bestIdx = shk[i];
        a.bestDj = shd[6][i];
        a.bestAlpha = shd[7][i];
      }
    }
  }
}

// ---- the same reduction over the workgroups of a launch (k_dual_column_wide): per pass every workgroup leaves its partial
// {sum thruThis, sum increaseInThis, min upperTheta, sum bad pivots, best |alpha| (first wins) with its dj / alpha / list
// position, min breakpoint of the swapped} -- the dualColumnResult of the reference's blocked ratio test
// (src/AbcSimplexDual.hpp:22-40, per-block pass src/AbcSimplexDual.cpp:1450-1528) -- then one grid barrier, then EVERY
// workgroup combines all partials with the same fixed tree (the combine loop of :1623-1634), so all of them take the same
// decisions without a broadcast.  The partial is also what a column-sharded run would all-reduce per pass.
#define DCW_BLOCKS 128
#define DCW_THREADS 256
#define DCW_PART 10  // doubles per partial
// all workgroups of the launch are resident (128 x 256 threads); the counter is zeroed by k_cand_scatter earlier in the
// pivot's chain.  Producer: plain stores -> __syncthreads -> lane 0 release fence -> drained -> relaxed arrive; consumer:
// relaxed polls -> one acquire fence -> __syncthreads -> plain loads (MI355X guide, inter-workgroup visibility).  The spin
// is bounded: a launch that cannot complete reports EXIT_NO_INCOMING with an error mark instead of hanging the device.
__device__ inline bool dcGridBarrier(Ctrl *c, int no, int G)
{
  __shared__ int shOk;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&c->dcArrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 0;
    for (int spin = 0; spin < (1 << 22); spin++) {
      if (__hip_atomic_load(&c->dcArrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= no * G) {
        ok = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    shOk = ok;
  }
  __syncthreads();
  return shOk != 0;
}
template <int F> __device__ inline bool dcReduceGrid(DcAcc &a, double (*shd)[16], int *shk, const Dev &D, int &barrierNo)
{
  dcReduce<F>(a, shd, shk);
  const int G = gridDim.x, tid = threadIdx.x;
  barrierNo++;
  double *part = D.dcPart + (size_t)(barrierNo & 1) * DCW_BLOCKS * DCW_PART;  // two sets: a fast workgroup is at most one pass ahead
  if (tid == 0) {
    double *p = part + (size_t)blockIdx.x * DCW_PART;
    p[0] = a.thru;
    p[1] = a.incr;
    p[2] = a.ut;
    p[3] = a.sumBad;
    p[4] = a.bestPivot;
    p[5] = a.ut2;
    p[6] = a.bestDj;
    p[7] = a.bestAlpha;
    p[8] = (double)a.bestIdx;
  }
  const bool ok = dcGridBarrier(D.ctrl, barrierNo, G);
  DcAcc b = { 0.0, 0.0, 1.0e50, 0.0, a.bestPivot, 1.0e50, -1, 0.0, 0.0 };
  if (tid < G) {
    const double *p = part + (size_t)tid * DCW_PART;
    b.thru = p[0];
    b.incr = p[1];
    b.ut = p[2];
    b.sumBad = p[3];
    b.bestPivot = p[4];
    b.ut2 = p[5];
    b.bestDj = p[6];
    b.bestAlpha = p[7];
    b.bestIdx = (int)p[8];
  }
  dcReduce<F>(b, shd, shk);
  a = b;
  return ok;
}

// CPT > 0: every thread keeps CPT candidates (alpha, dj, range, state) in registers, a pass is
// pure ALU + one reduction; CPT == 0: candidates stay in global memory (very long rows).
// WIDE: the launch has several workgroups (k_dual_column_wide); candidates are dealt over all its threads, every reduction
// is grid-wide (dcReduceGrid) and every workgroup follows the same decisions.
#define DC_CC 4                   // candidates per lane of the compacted final batch (one wave)
#define DC_COMPACT (DC_CC * 64)
template <int CPT, bool ONEWAVE, bool MAPPED = false, bool WIDE = false>
__device__ __forceinline__ bool dualColumnImpl(const Dev &D, const int *map = nullptr, int count = -1, double tauGuard = 0.0, void *ccScratch = nullptr)
{
  Ctrl *c = D.ctrl;
  __shared__ double shd[8][16];
  __shared__ int shk[16];
  // (tid / nthr: this thread's place among all threads that share the candidate list)
  const int tid = WIDE ? (int)(blockIdx.x * blockDim.x + threadIdx.x) : (int)threadIdx.x;
  const int nthr = ONEWAVE ? 64 : (WIDE ? (int)(gridDim.x * blockDim.x) : (int)blockDim.x);
  int barrierNo = 0;
  bool gridOk = true;
  const int nc = MAPPED ? count : c->numberCandidates;  // MAPPED: a prefiltered working set (see k_dual_column)
  const double acceptablePivot = c->acceptablePivot;
  const double dualTolerance = c->dualTolerance;
  const double newTolerance = dualTolerance;
  const double absDualOut = fabs(c->dualOut);
  auto reduce = [&](DcAcc &a, auto fields) {
    constexpr int F = decltype(fields)::value;
    if constexpr (ONEWAVE)
      dcReduceWave<F>(a);
    else if constexpr (WIDE)
      gridOk = dcReduceGrid<F>(a, shd, shk, D, barrierNo) && gridOk;
    else
      dcReduce<F>(a, shd, shk);
  };
  constexpr int R = CPT > 0 ? CPT : 1;
  double ra[R], rd[R], rr[R], rq[R];
  int rt[R], ri[R];
  // a candidate's breakpoint (:4392/:4410 and :4448/:4455 compute the same quotient every pass);
  // 1e50 (never the minimum) when |alpha| is below the acceptable pivot
  auto breakpoint = [&](double alpha, double djv) {
    if (alpha < 0.0)
      return (-alpha >= acceptablePivot) ? (djv - newTolerance) / alpha : 1.0e50;
    return (alpha >= acceptablePivot) ? (djv + newTolerance) / alpha : 1.0e50;
  };
  bool rl[R];
  if constexpr (CPT > 0) {
    // alpha, dj and range of every candidate this thread owns, one round of loads (the snapshots
    // k_cand_scatter left by candidate index; dj is not touched between there and here)
#pragma unroll
    for (int q = 0; q < R; q++) {
      const int pos = tid + q * nthr;
      rl[q] = pos < nc;
      ri[q] = rl[q] ? (MAPPED ? map[pos] : pos) : 0;  // original candidate index: the list order for ties
    }
#pragma unroll
    for (int q = 0; q < R; q++) {
      ra[q] = D.candAlpha[ri[q]];
      rd[q] = D.candDj[ri[q]];
      rr[q] = D.candRange[ri[q]];
    }
#pragma unroll
    for (int q = 0; q < R; q++) {
      rt[q] = -1;
      if (rl[q]) {
        rq[q] = breakpoint(ra[q], rd[q]);
      } else {
        ri[q] = -1;
        ra[q] = rd[q] = rr[q] = 0.0;
        rq[q] = 1.0e50;
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
          body(ri[q], ra[q], rd[q], rr[q], rq[q], rl[q], rt[q]);
      }
    } else {
      for (int i = tid; i < nc; i += nthr) {
        bool live = D.candLive[i] != 0;
        int tag = D.candTag[i];
        double alpha = D.candAlpha[i], djv = D.candDj[i];
        body(i, alpha, djv, D.candRange[i], breakpoint(alpha, djv), live, tag);
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
  long long dbgT0 = wall_clock64();
  int dbgPasses = 0, dbgTries = 0;
  while (tentativeTheta < 1.0e22) {
    dbgPasses++;
    // a prefiltered run is only valid while theta stays below the threshold every excluded
    // candidate is known to exceed; otherwise the caller redoes the test on the full list
    if (MAPPED && tentativeTheta >= tauGuard)
      return false;
    // ---- coarse pass (:4355-4417)
    // ut2: smallest breakpoint among the candidates this pass swaps -- the first upperTheta of
    // the inner loop below, should the pivot turn out to be in this batch (:4441-4462)
    DcAcc acc = { 0.0, 0.0, 1.0e50, 0.0, acceptablePivot, 1.0e50, -1, 0.0, 0.0 };
    forEach([&](int i, double alpha, double oldValue, double range, double ratio, bool &live, int &tag) {
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
          acc.ut2 = fmin(acc.ut2, ratio);
        } else {
          acc.ut = fmin(acc.ut, ratio);
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
          acc.ut2 = fmin(acc.ut2, ratio);
        } else {
          acc.ut = fmin(acc.ut, ratio);
        }
      }
    });
    reduce(acc, std::integral_constant<int, F_THRU | F_INCR | F_UT | F_BEST | F_UT2>());
    if (WIDE && !gridOk)
      break;
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
      const long long dbgT1 = wall_clock64();
      forEach([&](int, double, double, double, double, bool &live, int &tag) { live = (tag == passId); });
      if constexpr (CPT == 0)
        __syncthreads();
      // the inner loop (:4436-4638) over whatever holds the batch: `fe` visits its candidates, `rd` reduces over them
      // `pass(a2)` swaps what upperTheta reaches out of the batch and leaves the reduced sums / minimum / best pivot in a2
      auto tries = [&](auto &&pass, double nextUt) {
        int iTry;
        const int MAXTRY = 100;
        for (iTry = 0; iTry < MAXTRY; iTry++) {
          passId++;
          dbgTries++;
          // smallest remaining breakpoint of the live set (:4441-4462); it was computed by the pass
          // that produced the live set (the coarse pass, or the previous trip's removal pass)
          upperTheta = nextUt;
          badSumPivots = 0;
          upperTheta *= 1.0000000001;
          DcAcc a2 = { 0.0, 0.0, 1.0e50, 0.0, acceptablePivot, 1.0e50, -1, 0.0, 0.0 };
          pass(a2);
          if (WIDE && !gridOk)
            break;
          nextUt = a2.ut;
          thruThis = a2.thru;
          increaseInThis = a2.incr;
          bestPivot = a2.bestPivot;
          bestIdx = a2.bestIdx;
          double sumBadPivots = a2.sumBad;
          if (bestIdx < 0)
            bestPivot = acceptablePivot;
          seqIdx = bestIdx;
          if (bestIdx >= 0)
            theta = a2.bestDj / a2.bestAlpha;  // dj / alpha of the chosen candidate (:4566)
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
      };
      auto passAll = [&](DcAcc &a2) {
      forEach([&](int i, double alpha, double djv, double range, double ratio, bool &live, int &tag) {
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
              a2.bestDj = djv;
              a2.bestAlpha = alpha;
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
          } else {
            a2.ut = fmin(a2.ut, ratio);
          }
        });
        reduce(a2, std::integral_constant<int, F_THRU | F_INCR | F_UT | F_BAD | F_BEST | F_BESTV>());
      };
      bool compacted = false;
      if constexpr (!ONEWAVE && !WIDE && CPT > 0 && CPT <= 8) if (ccScratch != nullptr) {
        // The batch is usually a few dozen candidates out of a working set of thousands, and the loop above makes ~9 trips
        // (profiles/r04_dc_probe_after.txt), each a workgroup-wide reduction.  A batch of at most DC_COMPACT candidates is
        // moved into the registers of wave 0 (slot = exclusive scan of the per-thread counts: a fixed order, so the sums
        // are the same on every run), the trips become register butterflies of one wave -- no LDS exchange, no barrier --
        // and the tags go back to their owners for the cost shifting below.  (Ties of "largest |alpha|" are broken by
        // the candidate's list position, which travels with it: the slot order does not matter.)
        // (LDS: the caller's working-set index array, no longer needed once the candidates are in registers)
        double *ccA = (double *)ccScratch, *ccD = ccA + DC_COMPACT, *ccR = ccD + DC_COMPACT, *ccQ = ccR + DC_COMPACT, *ccOutD = ccQ + DC_COMPACT;
        int *ccI = (int *)(ccOutD + 2), *ccT = ccI + DC_COMPACT, *ccScan = ccT + DC_COMPACT, *ccOutI = ccScan + 18;
        int mine = 0;
        forEach([&](int, double, double, double, double, bool &live, int &) { mine += live ? 1 : 0; });
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
        int incl = mine;
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(incl, o);
          if (lane >= o)
            incl += t;
        }
        __syncthreads();
        if (lane == 63)
          ccScan[wv] = incl;
        __syncthreads();
        int base = incl - mine, total = 0;
        for (int w = 0; w < nw; w++) {
          if (w < wv)
            base += ccScan[w];
          total += ccScan[w];
        }
        if (threadIdx.x == 0) {
          c->dbgCc[total <= DC_COMPACT ? 0 : 1]++;
          c->dbgCc[2] += total;
        }
        if (total <= DC_COMPACT) {  // uniform
          int slot = base;
          forEach([&](int i, double alpha, double djv, double range, double ratio, bool &live, int &) {
            if (live) {
              ccA[slot] = alpha;
              ccD[slot] = djv;
              ccR[slot] = range;
              ccQ[slot] = ratio;
              ccI[slot] = i;
              slot++;
            }
          });
          __syncthreads();
          if (threadIdx.x < 64) {
            // (the batch stays in LDS: wave 0 keeps only each slot's state in registers -- the kernel is at its register limit)
            int ct[DC_CC];
            bool cl[DC_CC];
#pragma unroll
            for (int q = 0; q < DC_CC; q++) {
              ct[q] = passId;
              cl[q] = lane + 64 * q < total;
            }
            // One trip on the compacted batch: the swap test per lane, then the (usually one or two) swapped candidates are
            // visited in slot order by every lane alike (uniform LDS reads) -- the only wave-wide reduction left is the minimum
            // over the remaining breakpoints.
            auto passC = [&](DcAcc &a2) {
              double utLocal = 1.0e50;
#pragma unroll
              for (int q = 0; q < DC_CC; q++) {
                const int sl = min(lane + 64 * q, DC_COMPACT - 1);
                const double alphaQ = ccA[sl], djQ = ccD[sl];
                const double value = djQ - upperTheta * alphaQ;
                const bool swap = cl[q] && ((alphaQ < 0.0) ? (value >= 0.0) : (value <= 0.0));
                if (cl[q] && !swap)
                  utLocal = fmin(utLocal, ccQ[sl]);
                unsigned long long mask = __ballot(swap);
                if (swap) {
                  cl[q] = false;
                  ct[q] = passId;
                }
                while (mask) {
                  const int l = __builtin_ctzll(mask);
                  mask &= mask - 1;
                  const int ss = l + 64 * q;  // (wave-uniform: every lane reads the same slot)
                  const double alpha = ccA[ss], djv = ccD[ss], range = ccR[ss];
                  const int idx = ccI[ss];
                  const double val = djv - upperTheta * alpha;
                  const double badDj = (alpha < 0.0) ? -djv - dualTolerance : djv - dualTolerance;
                  const double absAlpha = fabs(alpha);
                  if (absAlpha > a2.bestPivot || (absAlpha == a2.bestPivot && a2.bestIdx >= 0 && idx < a2.bestIdx)) {
                    a2.bestPivot = absAlpha;
                    a2.bestIdx = idx;
                    a2.bestDj = djv;
                    a2.bestAlpha = alpha;
                  }
                  if (absAlpha < acceptablePivot && upperTheta < 1.0e20) {
                    if (alpha < 0.0) {
                      if (val > dualTolerance)
                        a2.sumBad += (range < 1.0e20) ? val * range : 1.0e20;
                    } else {
                      if (val < -dualTolerance)
                        a2.sumBad += (range < 1.0e20) ? -(val * range) : 1.0e20;
                    }
                  }
                  a2.thru += range * fabs(alpha);
                  a2.incr += badDj * range;
                }
              }
              DcAcc mn = { 0.0, 0.0, utLocal, 0.0, 0.0, 1.0e50, -1, 0.0, 0.0 };
              dcReduceWave<F_UT>(mn);
              a2.ut = mn.ut;
            };
            tries(passC, acc.ut2);
#pragma unroll
            for (int q = 0; q < DC_CC; q++)
              if (lane + 64 * q < total)
                ccT[lane + 64 * q] = ct[q];
            if (lane == 0) {
              ccOutD[0] = theta;
              ccOutI[0] = seqIdx;
              ccOutI[1] = iFlip;
              ccOutI[2] = sid[0];
              ccOutI[3] = sid[1];
              ccOutI[4] = modifyCosts;
              ccOutI[5] = badSumPivots;
              ccOutI[6] = lastIdx;
              ccOutI[7] = passId;
              ccOutI[8] = dbgTries;
            }
          }
          __syncthreads();
          theta = ccOutD[0];
          seqIdx = ccOutI[0];
          iFlip = ccOutI[1];
          sid[0] = ccOutI[2];
          sid[1] = ccOutI[3];
          modifyCosts = ccOutI[4];
          badSumPivots = ccOutI[5];
          lastIdx = ccOutI[6];
          passId = ccOutI[7];
          dbgTries = ccOutI[8];
          slot = base;
          forEach([&](int, double, double, double, double, bool &live, int &tag) {
            if (live)
              tag = ccT[slot++];
          });
          compacted = true;
        }
      }
      if (!compacted)
        tries(passAll, acc.ut2);
      if (tid == 0 && !ONEWAVE) {
        c->dbgCc[3] += wall_clock64() - dbgT1;
        c->dbgCc[4] += dbgT1 - dbgT0;
        c->dbgCc[5] += WIDE ? 1 : 0;
      }
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
  if (WIDE && (!gridOk || c->numberIterations == c->debugDcTimeoutAt)) {  // (debugDcTimeoutAt: fault injection, -1 off)
    // a grid barrier timed out (never seen; the spin is bounded so that a fault cannot hang the device): no pivot, and the host is told
    if (tid == 0) {
      c->dcWide = -1;
      c->sequenceIn = -1;
      c->alpha = 0.0;
      c->bestPossible = 0.0;
      c->state = EXIT_NO_INCOMING;
    }
    return true;
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
      forEach([&](int i, double alpha, double djv, double, double, bool &, int &tag) {
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
      DcAcc a3 = { (double)changed, 0.0, 1.0e50, 0.0, 0.0, 1.0e50, -1, 0.0, 0.0 };
      reduce(a3, std::integral_constant<int, F_THRU>());
      if (tid == 0)
        c->numberChanged += (int)a3.thru;
    }
  }
  // "things look bad" (:4776-4784); badFree is the general branch's (k_free_scan), 0.0 on every other path
  if ((badSumPivots || fabs(theta * c->badFree) > 10.0 * dualTolerance) && c->pivots) {
    sequenceIn = -1;
    if (tid == 0)
      c->acceptablePivotBase = -c->acceptablePivotBase;
  }
  // (the entering column is unpacked by an extra workgroup of k_dj_flags, off this kernel's critical path)
  if (tid == 0) {
    c->dbg[ONEWAVE ? 0 : 1]++;
    c->dbg[2] += dbgPasses;
    c->dbg[3] += dbgTries;
    c->dbg[4] += nc;
    c->dbg[ONEWAVE ? 7 : 8] += wall_clock64() - dbgT0;
    c->dbg[MAPPED ? 5 : 6]++;
    if (MAPPED && !ONEWAVE) {
      // would the smallest class (breakpoints up to 8 theta0) have been enough for this pivot?
      const double theta0dbg = fmax(10.0 * c->upperTheta, 1.0e-7);
      c->dbgCc[6]++;
      if (tentativeTheta < 8.0 * theta0dbg)
        c->dbgCc[7]++;
    }
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

// =============================================================================================
// Option free_nonbasic: what the general branch of ClpSimplexDual::dualColumn0 ("some free or super basic",
// src/ClpSimplexDual.cpp:4058-4179) does beyond the fast branch the pricing kernels fuse -- for atUpperBound / atLowerBound
// variables the two branches build the same candidate list, so only the isFree / superBasic nonbasics are left to do, and
// those are few: Dev::freeList holds the sequences that had one of the two statuses at the last status check (rows first, then
// columns: the order the reference meets them in), Ctrl::freeCount of them (0 while moreSpecialOptions_ & 8 says there are none).
// One workgroup, after the candidate list is complete and before the ratio test:
//   * a free variable "worth keeping" (dj beyond the dual tolerance, or |alpha| above max(10 x acceptablePivot, 1e-5)) is given
//     fake bounds one dualBound wide starting at its value, if that value allows (:4124-4140) -- it is an ordinary
//     atUpper / atLower variable from the next pivot on -- the others feed badFree (:4108);
//   * the kept one with the largest |alpha| above the acceptable pivot (the first of equals) comes in whatever the ratios say
//     ("always choose", :4321): sequenceIn / alpha / theta = dj / alpha and the tail of dualColumn (:4786-4850) are written here and
//     the ratio-test kernels return at once (Ctrl::freeChosen).
// badFree goes to the ratio test's own "things look bad" test (:4776-4784) through Ctrl::badFree.
// =============================================================================================
__global__ void __launch_bounds__(256) k_free_scan(Dev D)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int count = c->freeCount;
  const int tid = threadIdx.x;
  if (count <= 0) {
    if (tid == 0) {
      c->freeChosen = 0;
      c->badFree = 0.0;
    }
    return;
  }
  __shared__ double shA[4], shB[4];
  __shared__ int shE[4];
  const double acceptablePivot = c->acceptablePivot;
  const double dualTolerance = c->dualTolerance;
  const double dualBound = c->dualBound;
  const double tentativeTheta = 1.0e25;
  const int seqOut = c->sequenceOut;
  double bestAbs = acceptablePivot, badFree = 0.0;
  int bestE = -1;
  for (int e = tid; e < count; e += blockDim.x) {
    const int seq = D.freeList[e];
    const unsigned char st8 = D.status[seq];
    const int st = st8 & 7;
    if ((st != ST_FREE && st != ST_SUPER) || seq == seqOut)
      continue;
    const double alpha = seq >= D.n ? D.rho[seq - D.n] : D.alphaCol[seq];
    if (alpha == 0.0)
      continue;  // not in the packed tableau row
    const double oldValue = D.dj[seq];
    bool keep;
    if (oldValue > dualTolerance || oldValue < -dualTolerance) {
      keep = true;
    } else if (fabs(alpha) > fmax(10.0 * acceptablePivot, 1.0e-5)) {
      keep = true;
    } else {
      keep = false;
      badFree = fmax(badFree, fabs(alpha));
    }
    if (!keep)
      continue;
    if (fabs(alpha) > bestAbs) {  // (this thread meets its entries in list order: a later equal one does not replace)
      bestAbs = fabs(alpha);
      bestE = e;
    }
    // give fake bounds if possible
    const double value = D.sol[seq];
    if (2.0 * fabs(value) < dualBound) {
      unsigned char nst = (unsigned char)((st8 & ~(7 | 24)) | (FAKE_BOTH << 3));
      if (oldValue - tentativeTheta * alpha > dualTolerance) {
        // pretend coming in from upper bound
        D.upper[seq] = value;
        D.lower[seq] = value - dualBound;
        nst = (unsigned char)(nst | ST_UPPER);
      } else {
        // pretend coming in from lower bound
        D.lower[seq] = value;
        D.upper[seq] = value + dualBound;
        nst = (unsigned char)(nst | ST_LOWER);
      }
      D.status[seq] = nst;
    }
  }
  // largest |alpha|, the first in list order among equals; largest badFree
  const int lane = tid & 63, wv = tid >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    const double oa = __shfl_xor(bestAbs, o);
    const int oe = __shfl_xor(bestE, o);
    const double ob = __shfl_xor(badFree, o);
    if (oe >= 0 && (bestE < 0 || oa > bestAbs || (oa == bestAbs && oe < bestE))) {
      bestAbs = oa;
      bestE = oe;
    }
    badFree = fmax(badFree, ob);
  }
  if (lane == 0) {
    shA[wv] = bestAbs;
    shE[wv] = bestE;
    shB[wv] = badFree;
  }
  __syncthreads();
  if (tid != 0)
    return;
  for (int w = 1; w < (int)(blockDim.x >> 6); w++) {
    if (shE[w] >= 0 && (bestE < 0 || shA[w] > bestAbs || (shA[w] == bestAbs && shE[w] < bestE))) {
      bestAbs = shA[w];
      bestE = shE[w];
    }
    badFree = fmax(badFree, shB[w]);
  }
  c->badFree = badFree;
  if (bestE < 0) {
    c->freeChosen = 0;
    return;
  }
  // a free variable comes in: the rest of dualColumn without the ratio passes (:4321, :4776-4850)
  c->freeChosen = 1;
  c->freeEntered++;
  c->badSumPivots = 0;
  c->modifyCosts = 0;
  int sequenceIn = D.freeList[bestE];
  const double alphaIn = sequenceIn >= D.n ? D.rho[sequenceIn - D.n] : D.alphaCol[sequenceIn];
  const double theta = D.dj[sequenceIn] / alphaIn;
  if (fabs(theta * badFree) > 10.0 * dualTolerance && c->pivots) {
    sequenceIn = -1;
    c->acceptablePivotBase = -c->acceptablePivotBase;
  }
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

#define DC_CPT 8
#define DC_THREADS 512
#define DC_SMALL (8 * 64)
#define DC_WS_CAP (DC_CPT * DC_THREADS)      // largest working set kept in registers
#define DC_NB_PER 8                          // compaction blocks per thread of the class-prefix scan
#define DC_NB_MAX (DC_NB_PER * DC_THREADS)   // beyond this many blocks (N > 1M) the working set is not used
// One launch, 512 threads (256 VGPRs per lane available: no spills).
//  * typical sparse tableau row (<= 512 candidates): wave 0 alone, candidates in registers,
//    reductions are register butterflies -- no LDS, no barrier;
//  * long candidate lists (dense rows, up to ~n/2 entries): only the few dozen candidates with the
//    smallest breakpoints ever take part in the passes, the rest only bound theta from above.
//    k_cand_scatter classified every candidate by its breakpoint against theta0 * {2^3, 2^8, 2^14};
//    the largest class prefix that fits in registers becomes the working set, the test runs on it,
//    and is repeated on the full list only if theta ever reaches the class threshold (exact either
//    way).
// The working set of the ratio test, built over the whole chip (round 3).  With a dense pi the candidate list holds
// every nonbasic column of the right sign -- 10^5 entries -- and the single-workgroup ratio test spent 62 of its 112 us
// walking that list once just to pick out the few thousand candidates whose breakpoints are near theta0.  Here every
// workgroup repeats the cheap part (class totals over the compaction blocks -> the class prefix J that fits the
// working set; exclusive prefix of the per-block counts, both from k_cand_scatter's classBlock) and then places its
// share of the candidates at  prefix[block] + rank-in-block  in D.wsIdxG -- the same positions, hence the same
// order, the ratio test used to compute for itself.  c->wsJ / c->wsCount tell k_dual_column what it got.
#define WS_THREADS 256
__global__ void __launch_bounds__(WS_THREADS) k_dc_working_set(Dev D, int nbClass)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN || c->freeChosen)
    return;
  const int nc = c->numberCandidates;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (nc <= DC_SMALL || nbClass > DC_NB_MAX) {
    if (blockIdx.x == 0 && tid == 0) {
      c->wsJ = -1;
      c->wsCount = 0;
    }
    return;
  }
  __shared__ int s_pre[DC_NB_MAX];
  __shared__ int shCls[WS_THREADS / 64][3];
  __shared__ int shw[WS_THREADS / 64];
  // class totals
  int cum[3] = { 0, 0, 0 };
  for (int bb = tid; bb < nbClass; bb += WS_THREADS) {
    cum[0] += D.classBlock[3 * bb];
    cum[1] += D.classBlock[3 * bb + 1];
    cum[2] += D.classBlock[3 * bb + 2];
  }
  for (int j = 0; j < 3; j++) {
    for (int o = 32; o > 0; o >>= 1)
      cum[j] += __shfl_xor(cum[j], o);
    if (lane == 0)
      shCls[wv][j] = cum[j];
  }
  __syncthreads();
  for (int j = 0; j < 3; j++) {
    cum[j] = 0;
    for (int w = 0; w < WS_THREADS / 64; w++)
      cum[j] += shCls[w][j];
  }
  cum[1] += cum[0];
  cum[2] += cum[1];
  int J = -1;
  for (int j = 0; j < 3; j++)
    if (cum[j] <= DC_WS_CAP && cum[j] < nc)
      J = j;
  if (blockIdx.x == 0 && tid == 0) {  // development counters (CLPGPU_DEBUG_STATS): cumulative class sizes of the candidate lists
    c->dbgDc[4] += cum[0];
    c->dbgDc[5] += cum[1];
    c->dbgDc[6] += cum[2];
  }
  if (J < 0 || cum[J] <= 0) {
    if (blockIdx.x == 0 && tid == 0) {
      c->wsJ = -1;
      c->wsCount = 0;
    }
    return;
  }
  // exclusive prefix over the blocks of their (class <= J) counts: thread t owns a contiguous run of blocks
  const int per = (nbClass + WS_THREADS - 1) / WS_THREADS;
  const int b0 = tid * per, b1 = min(b0 + per, nbClass);
  int mine = 0;
  for (int bb = b0; bb < b1; bb++)
    mine += D.classBlock[3 * bb] + (J >= 1 ? D.classBlock[3 * bb + 1] : 0) + (J >= 2 ? D.classBlock[3 * bb + 2] : 0);
  int v = mine;
  for (int o = 1; o < 64; o <<= 1) {
    int u = __shfl_up(v, o);
    if (lane >= o)
      v += u;
  }
  if (lane == 63)
    shw[wv] = v;
  __syncthreads();
  int run = v - mine;
  for (int i = 0; i < wv; i++)
    run += shw[i];
  for (int bb = b0; bb < b1; bb++) {
    s_pre[bb] = run;
    run += D.classBlock[3 * bb] + (J >= 1 ? D.classBlock[3 * bb + 1] : 0) + (J >= 2 ? D.classBlock[3 * bb + 2] : 0);
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) {
    c->wsJ = J;
    c->wsCount = cum[J];
  }
  const int shift = 10 * J;
  for (int i = blockIdx.x * WS_THREADS + tid; i < nc; i += gridDim.x * WS_THREADS) {
    if ((int)D.candLive[i] <= J)
      D.wsIdxG[s_pre[D.candBlk[i]] + ((D.candRk[i] >> shift) & 1023)] = i;
  }
}

__global__ void __launch_bounds__(DC_THREADS) k_dual_column(Dev D, int nbClass, int wide = 0)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN || c->freeChosen)  // (freeChosen: k_free_scan brought a free variable in, "always choose" :4321)
    return;
  const long long dcT0 = wall_clock64();
  const int nc = c->numberCandidates;
  const int tid = threadIdx.x;
  if (!nc) {
    if (tid == 0) {
      c->sequenceIn = -1;
      c->alpha = 0.0;
      c->bestPossible = 0.0;
      c->state = EXIT_NO_INCOMING;
    }
    return;
  }
  // wide > 0: k_dual_column_wide follows in the chain (DCW_BLOCKS workgroups): a list too long for one workgroup's registers --
  // the working set's guard failed, or there is no working set -- is left to it instead of being walked in global memory by
  // this one workgroup (1-3 ms at 10^5 candidates, profiles/r04_dc_probe_after.txt); wide == 2 (test knob) defers every list
  const bool canDefer = wide > 0 && nc <= 16 * DCW_BLOCKS * DCW_THREADS;
  if (wide == 2 && canDefer) {
    if (tid == 0)
      c->dcWide = 1;
    return;
  }
  if (nc <= DC_SMALL) {
    if (tid < 64) {
      if (nc <= 4 * 64)
        dualColumnImpl<4, true>(D);
      else
        dualColumnImpl<8, true>(D);
    }
    return;
  }
  __shared__ __attribute__((aligned(16))) int wsIdx[DC_WS_CAP];  // (>= the 10.5 KB the batch compaction of dualColumnImpl borrows)
  __shared__ int s_done;
  if (nbClass <= DC_NB_MAX) {
    // the working set (candidates of breakpoint class <= J, in list order) was compacted over the whole chip by
    // k_dc_working_set; only its indices are read here
    const int J = c->wsJ;
    if (J >= 0 && c->wsCount > 0) {
      const int ws = c->wsCount;
      const double theta0 = fmax(10.0 * c->upperTheta, 1.0e-7);
      const double tau = theta0 * (J == 0 ? 8.0 : (J == 1 ? 256.0 : 16384.0));
      for (int i = tid; i < ws; i += DC_THREADS)
        wsIdx[i] = D.wsIdxG[i];
      if (tid == 0)
        s_done = 0;
      __syncthreads();
      bool ok;
      const long long dcT1 = wall_clock64();
      if (ws <= DC_SMALL) {
        if (tid < 64) {
          ok = dualColumnImpl<8, true, true>(D, wsIdx, ws, tau);
          if (tid == 0)
            s_done = ok ? 1 : 0;
        }
        __syncthreads();
        ok = s_done != 0;
      } else {
        // (working sets beyond DC_CPT x DC_THREADS = 4096 candidates are not built: round 4 kept sets up to 8192 in registers at
        // sixteen per thread, which made the kernel spill 464 bytes per lane into scratch -- every pass then waited on scratch
        // traffic, 60 against 45 us per call; such pivots take a smaller class and, if its guard fails, the wide kernel)
        ok = dualColumnImpl<DC_CPT, false, true>(D, wsIdx, ws, tau, wsIdx);
      }
      if (!ok && tid == 0)
        c->dbgDc[7] += 1 + 1000000LL * J;  // fall-backs to the full list (+ 1e6 x the class the working set had)
      if (ok) {
        if (tid == 0) {
          const long long dt = wall_clock64() - dcT0;
          c->dbgDc[0]++;
          c->dbgDc[1] += dt;
          c->dbgDc[2] += dcT1 - dcT0;
          if (dt > c->dbgDc[3])
            c->dbgDc[3] = dt;
        }
        return;
      }
      __syncthreads();
    }
  }
  if (nc <= DC_CPT * (int)blockDim.x) {
    dualColumnImpl<DC_CPT, false>(D, nullptr, -1, 0.0, wsIdx);
  } else if (canDefer) {
    if (tid == 0)
      c->dcWide = 1;
  } else {
    dualColumnImpl<0, false>(D);
  }
}

// The ratio test over DCW_BLOCKS workgroups for the lists k_dual_column left alone (c->dcWide): the candidates are dealt over
// all threads of the launch (4 / 8 / 16 per thread, in registers), the logic is dualColumnImpl's unchanged, every pass ends in
// one grid-wide reduction of the per-workgroup partials (dcReduceGrid) -- the blocked ratio test of the reference's parallel
// engine (src/AbcSimplexDual.cpp:1450-1528, combine :1623-1634).  Returns at once on every other pivot.
__global__ void __launch_bounds__(DCW_THREADS) k_dual_column_wide(Dev D)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN || c->dcWide != 1 || c->freeChosen)
    return;
  const int nc = c->numberCandidates;
  const int nthr = gridDim.x * blockDim.x;
  if (nc <= 4 * nthr)
    dualColumnImpl<4, false, false, true>(D);
  else if (nc <= 8 * nthr)
    dualColumnImpl<8, false, false, true>(D);
  else
    dualColumnImpl<16, false, false, true>(D);
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
        int sc = D.cslot[q];  // col-slot of the basic entry, kept with it in the row copy
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

// ClpDualRowSteepest::unrollWeights (:1022): w is still intact when the host asks for this
__global__ void k_unroll_weights(Dev D)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m && D.w[p] != 0.0)
    D.weights[p] = D.altWeights[p];
}

// ---- plug-in level ClpDualRowPivot calls (clpgpu_update_weights / clpgpu_update_primal): the same
// arithmetic as the fused iteration kernels (k_rho_finish3's norm partials, k_ftran_scatter3's weight
// update, k_fix_house's list appends), as stand-alone launches on host-supplied vectors
// per-256-row partials of sum pi^2 (ClpDualRowSteepest::updateWeights :443-460)
__global__ void __launch_bounds__(256) k_plugin_norm(Dev D, const double *piRow)
{
  __shared__ double sh[16];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double sq = 0.0;
  if (i < D.m) {
    double v = piRow[i];
    sq = v * v;
  }
  double s = blockSum(sq, sh);
  if (threadIdx.x == 0)
    D.normPartial[blockIdx.x] = s;
}
// DSE weight update on the support of the updated column w (D.w, D.tau by basis position):
// :516-538 with model alpha = the ratio test's alpha; old weights saved for unrollWeights (:1022)
__global__ void __launch_bounds__(256) k_plugin_weights(Dev D, int pivotRow, double modelAlpha, int nbNorm)
{
  __shared__ double shd[16];
  double acc = 0.0;
  for (int b = threadIdx.x; b < nbNorm; b += blockDim.x)
    acc += D.normPartial[b];
  acc = blockSum(acc, shd);
  const double norm = acc / (modelAlpha * modelAlpha);
  const double multiplier = 2.0 / modelAlpha;
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= D.m)
    return;
  const double x1 = D.w[p], x2 = D.tau[p];
  if (x1 != 0.0) {
    double devex = D.weights[p];
    D.altWeights[p] = devex;
    if (p == pivotRow) {
      devex = (norm < DEVEX_TRY_NORM) ? DEVEX_TRY_NORM : norm;
    } else {
      devex += x1 * (x1 * norm + x2 * multiplier);
      if (devex < DEVEX_TRY_NORM)
        devex = DEVEX_TRY_NORM;
    }
    D.weights[p] = devex;
  }
}
// new entries of the infeasibility list at the absolute offsets scanTailBody left
__global__ void __launch_bounds__(256) k_append_scatter_abs(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->numberAppend == 0)
    return;
  __shared__ int shi[17];
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int flag = (p < D.m) ? D.appendFlag[p] : 0;
  int total;
  int rank = blockRank(flag, total, shi);
  if (flag)
    D.infIndex[D.blockOffset[blockIdx.x] + rank] = p;
}
// ClpDualRowSteepest::saveWeights mode 6 (:937-957): every weight becomes `allowed` (the reference
// assigns the bound, not the clamped value)
__global__ void k_weights_scale_back(Dev D, double allowed)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m)
    D.weights[p] = allowed;
}

// =============================================================================================
// Primal update -- ClpDualRowSteepest::updatePrimalSolution (src/ClpDualRowSteepest.cpp:630-763)
// x_B -= ratio * vec ; refresh squared infeasibilities ; new entries are appended to the list in
// ascending position order (count / scan / scatter keeps CoinIndexedVector's insertion order).
// which: 0 -> vec = w, ratio = ctrl.movement ; 1 -> vec = x3 (flip FTRAN), ratio = 1
// =============================================================================================
// bid / nblk: this workgroup's index and the number of workgroups doing the primal update
__device__ inline void primalUpdateBody(const Dev &D, int which, int bid, int nblk)
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
  int p = bid * blockDim.x + threadIdx.x;
  if (which == 0) {
    // ClpSimplexDual::flipBounds (:6345-6401), after the backwards check of the scalar block
    for (int f = p; f < c->numberFlips; f += nblk * blockDim.x) {
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
    // the leaving row stays on the list with a tiny value (:705-706); done by its owner here
    if (which == 0 && c->pivotRule && p == c->pivotRow && D.infeas[p] != 0.0)
      D.infeas[p] = REALLY_TINY;
  }
  int total;
  blockRank(append, total, shi);
  double s = blockSum(changeObj, shd);
  if (threadIdx.x == 0) {
    stc(&D.blockCount[bid], total);
    stc(&D.blockSum[bid], s);
  }
  // serial tail (offsets of the appends, objective change) in the last workgroup to finish
  if (which == 0 && lastBlockDone(D.ctrl, 2, nblk, bid))
    scanTailBody(D, nblk, nblk, 0, 0, -1);
}

__global__ void __launch_bounds__(256) k_primal_update(Dev D, int which)
{
  primalUpdateBody(D, which, blockIdx.x, gridDim.x);
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

__device__ inline void rank1Body(const Dev &D, int parity, int bx, int by, int gx, int gy)
{
  const Ctrl *c = D.ctrl;
  // parity >= 0: forked beside the rest of the pivot -- gated by the go flag the FTRAN tail set
  if (parity >= 0 ? !c->updGo[parity] : c->state != RUN)
    return;
  const int k = parity >= 0 ? c->updK : c->k;
  const double dir = (double)c->directionOut, alpha = c->alpha;
  // bx * 256 + thread = column j (coalesced along the row), by strides rows
  for (int j = bx * blockDim.x + threadIdx.x; j < k; j += gx * blockDim.x) {
    const double gj = dir * D.rhoSlot[j] / alpha;
    for (int i = by; i < k; i += gy) {
      double wi = D.slotC[i];  // w by column-slot, as the FTRAN sweep left it (== w[slotPos[i]])
      if (wi != 0.0)
        D.Minv[(size_t)i * D.ld + j] -= wi * gj;
    }
  }
}

__global__ void __launch_bounds__(256) k_rank1(Dev D, int parity = -1)
{
  rank1Body(D, parity, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}
// primal update and rank-1 sweep of the nucleus inverse in one launch: both only need what the
// FTRAN tail left behind and touch disjoint data.  Workgroups [0, nPrimal) do the primal update (and
// its serial tail), the rest the gx x gy tiles of the rank-1 sweep.
__global__ void __launch_bounds__(256) k_primal_rank1(Dev D, int parity, int nPrimal, int gx, int gy)
{
  if ((int)blockIdx.x < nPrimal) {
    primalUpdateBody(D, 0, blockIdx.x, nPrimal);
  } else {
    const int id = blockIdx.x - nPrimal;
    rank1Body(D, parity, id % gx, id / gx, gx, gy);
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
    int b = c->slotRowIn, last = k - 1;
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
  int se = D.cslot[e], sb = D.cslot[b];
  D.ccol[e] = cb;
  D.relem[e] = vb;
  D.csrToCsc[e] = pb;
  D.cscToCsr[pb] = e;
  D.cslot[e] = sb;
  D.ccol[b] = ce;
  D.relem[b] = ve;
  D.csrToCsc[b] = pe;
  D.cscToCsr[pe] = b;
  D.cslot[b] = se;
}
// LU mode with the compact eta file: the slot of the pivot's position -- the one it has, or the next free one, which the
// housekeeping's bookkeeping (tid 0, behind the row-copy moves) is about to give it
__device__ inline int luEnteringSlot(const Dev &D, const Ctrl *c)
{
  if (!c->luCompactOn)
    return -1;
  const int q = D.lu->cslotOfPos[c->pivotRow];
  return q >= 0 ? q : c->luCompactCount;
}
// col-slot the entering structural is about to get (houseBody's bookkeeping: case 0 takes the leaving
// column's slot, case 1 the new last slot)
__device__ inline int enteringSlot(const Ctrl *c)
{
  return c->updateCase == 0 ? c->slotColOut : c->k;
}

// wide mode: one of the two column moves of houseBody, one thread per entry (distinct rows)
__global__ void __launch_bounds__(256) k_house_col(Dev D, int which)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int seq = which ? c->sequenceIn : c->sequenceOut;
  if (seq >= D.n)
    return;
  const int p = D.colStart[seq] + blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= D.colStart[seq + 1])
    return;
  const int r = D.row[p];
  const int e = D.cscToCsr[p];
  if (which == 0) {
    const int b = D.rowStart[r] + D.basicCount[r] - 1;
    rowCopySwap(D, e, b);
    D.basicCount[r] -= 1;
  } else {
    const int b = D.rowStart[r] + D.basicCount[r];
    rowCopySwap(D, e, b);
    D.cslot[b] = D.luMode ? luEnteringSlot(D, c) : enteringSlot(c);
    D.basicCount[r] += 1;
  }
}

// ClpSimplexProgress::cycle (src/ClpSolve.cpp:4726-4825) on the control block's ring of the last 12 (in, out, way)
// triples: > 0 = a cycle of that length (100: irregular repeats), 0 = none; the triple of this pivot is appended.
__device__ inline int cycleStep(Ctrl *c, int seqIn, int seqOut, int directionIn, int directionOut)
{
  const int head = c->cycHead;
#define CYC(i) (((head) + (i)) % 12)
    int matched = 0;
    int outs[12];
#pragma unroll
    for (int i = 1; i < 12; i++)
      outs[i] = c->cycOut[CYC(i)];
#pragma unroll
    for (int i = 1; i < 12; i++)
      if (seqIn == outs[i] && !matched)
        matched = -1;
    if (matched && c->cycIn[CYC(0)] >= 0) {
      matched = 0;
      int nMatched = 0;
      const int way0 = c->cycWay[CYC(0)], in0 = c->cycIn[CYC(0)], out0 = c->cycOut[CYC(0)];
      for (int kk = 1; kk < 12 - 4; kk++) {
        if (in0 == c->cycIn[CYC(kk)] && out0 == c->cycOut[CYC(kk)] && way0 == c->cycWay[CYC(kk)]) {
          nMatched++;
          const int end = 12 - kk;
          int j;
          for (j = 1; j < end; j++)
            if (c->cycIn[CYC(j + kk)] != c->cycIn[CYC(j)] || c->cycOut[CYC(j + kk)] != c->cycOut[CYC(j)] ||
                c->cycWay[CYC(j + kk)] != c->cycWay[CYC(j)])
              break;
          if (j == end) {
            matched = kk;
            break;
          }
        }
      }
      if (matched <= 0 && nMatched > 1)
        matched = 100;
    }
    // drop the oldest, append this pivot: the oldest slot becomes the newest, the head moves on
    c->cycIn[head] = seqIn;
    c->cycOut[head] = seqOut;
    c->cycWay[head] = 1 - directionIn + 4 * (1 - directionOut);
    c->cycHead = (head + 1) % 12;
#undef CYC
  return matched;
}

__device__ void houseBody(Dev D, int skipColumns = 0)
{
  Ctrl *c = D.ctrl;
  const int tid = threadIdx.x;
  const int seqIn = c->sequenceIn, seqOut = c->sequenceOut, pivotRow = c->pivotRow;
  const int n = D.n;
  // ---- row copy partition: leaving structural goes to the nonbasic part, entering to the basic
  // (skipColumns: already done by k_house_col)
  if (!skipColumns && seqOut < n) {
    for (int p = D.colStart[seqOut] + tid; p < D.colStart[seqOut + 1]; p += blockDim.x) {
      int r = D.row[p];
      int e = D.cscToCsr[p];
      int b = D.rowStart[r] + D.basicCount[r] - 1;
      rowCopySwap(D, e, b);
      D.basicCount[r] -= 1;
    }
  }
  __syncthreads();
  if (!skipColumns && seqIn < n) {
    for (int p = D.colStart[seqIn] + tid; p < D.colStart[seqIn + 1]; p += blockDim.x) {
      int r = D.row[p];
      int e = D.cscToCsr[p];
      int b = D.rowStart[r] + D.basicCount[r];
      rowCopySwap(D, e, b);
      D.cslot[b] = D.luMode ? luEnteringSlot(D, c) : enteringSlot(c);
      D.basicCount[r] += 1;
    }
  }
  // ---- a structural leaves and the last col-slot moves into its place (case 2): the entries of that
  // column carry their column's slot in the row copy
  if (!D.luMode && c->updateCase == 2 && c->slotColOut != c->k - 1) {
    const int colLast = D.slotCol[c->k - 1], a = c->slotColOut;
    for (int p = D.colStart[colLast] + tid; p < D.colStart[colLast + 1]; p += blockDim.x)
      D.cslot[D.cscToCsr[p]] = a;
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
  const int ucase = D.luMode ? -1 : c->updateCase;
  if (ucase < 0) {
    // LU mode: the factorization's maps are frozen until the next refactorization; the compact eta file's are not (device_state.h):
    // the pivot's position gets a slot if it had none (k_lu_pf_append has filled the slot's column), the entering structural its
    // position
    const LuDev &L = *D.lu;
    if (L.cslotOfPos[pivotRow] < 0 && c->luCompactCount < L.ldc) {
      const int q = c->luCompactCount;
      L.cslotOfPos[pivotRow] = q;
      L.posOfCslot[q] = pivotRow;
      c->luCompactCount = q + 1;
    }
    if (seqOut < n)
      L.posOfBasicCol[seqOut] = -1;
    if (seqIn < n)
      L.posOfBasicCol[seqIn] = pivotRow;
  } else if (ucase == 0) {
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
  {
    // "making real progress" (:2096-2100): read by ClpSimplexProgress::looping at the next status check
    int flag = 0;
    if (D.upper[seqIn] > 1.0e20 && D.lower[seqIn] < -1.0e20)
      flag |= 2;
    if (D.upper[seqOut] - D.lower[seqOut] < 1.0e-12)
      flag |= 1;
    if (flag)
      c->progressFlag |= flag;
  }
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
    // length of the ratio-test candidate list; bit 30: the tableau row was priced by row (diagnostics)
    r->reserved = c->numberCandidates | (c->lastPriceByRow << 30);
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
  // ---- small cycles (ClpSimplex.cpp:2397-2431; ClpSimplexProgress::cycle, ClpSolve.cpp:4726-4825): the last 12
  // (in, out, way) triples; a repeat of the oldest with everything after it repeating too is a cycle of
  // that length, two irregular repeats count as 100.  Kept as a ring (entry i of the reference's arrays is
  // ring slot (head + i) % 12) so that a pivot costs 11 loads and 3 stores instead of shifting the arrays.
  {
    const int matched = cycleStep(c, seqIn, seqOut, c->directionIn, c->directionOut);
    if (matched > 0) {
      for (int i = 0; i < 12; i++) {
        c->cycIn[i] = c->cycOut[i] = -1;
        c->cycWay[i] = 0;
      }
      c->cycHead = 0;
      const double random = randomDouble(c);
      const int extra = (int)(9.999 * random);
      const int off[10] = { 1, 1, 1, 1, 2, 2, 2, 3, 3, 4 };
      if (c->pivots > matched) {
        c->forceFactorization = max(1, matched - off[extra]);
      } else {
        // "need to reject something": the leaving variable is flagged (:2418-2428)
        D.status[seqOut] = (unsigned char)(D.status[seqOut] | FLAGGED_BIT);
      }
      c->state = EXIT_REFACTOR;
      return;
    }
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
  if (c->state == RUN && !D.luMode && c->k + 2 >= c->kcap)
    c->state = EXIT_REFACTOR;  // nucleus storage nearly full: host regrows it at the refactorization
  // a stepped run (clpgpu_dual_steps) stops on its pivot BEFORE acting on the refactorization decision, the way
  // ClpSimplex::housekeeping returns on hitMaximumIterations() (src/ClpSimplex.cpp:2391) ahead of :2435-2450; unlike an
  // iteration limit the run can resume, so the decision taken above (and the cycle record) is kept for the resumption:
  // a stepped run and an unstepped one make the same pivots and the same refactorizations.
  if (c->stepLimit >= 0 && c->numberIterations >= c->stepLimit && (c->state == RUN || c->state == EXIT_REFACTOR)) {
    c->pendingState = c->state;
    c->state = EXIT_STEP_LIMIT;
  }
}


// =============================================================================================
// CHUZR: head (scalar), list scan over the whole chip, final selection
// =============================================================================================

// ---- CHUZR split in three so the list scan uses the whole chip ---------------------------------
// secondCall: the call of src/ClpDualRowSteepest.cpp:338-346 -- everything again (the touch-up of the last pivot row, another random
// number), with largestDualError_ 0, i.e. without the changed tolerance
__device__ void chuzrPreBody(Dev D, bool secondCall = false)
{
  Ctrl *c = D.ctrl;
  if (!secondCall && c->stepLimit >= 0 && c->numberIterations >= c->stepLimit) {
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
    int toleranceChanged = 0;
    if (!secondCall && c->numberIterations < c->lastBadIteration + 200) {
      if (c->largestDualError > c->largestPrimalError) {
        tolerance *= fmin(c->largestDualError / c->largestPrimalError, 1000.0);
        toleranceChanged = 1;
      } else if (c->debugToleranceFactor > 0.0) {
        // fault injection (option debug_tolerance_factor): which of two rounding-noise errors is larger is not reproducible between two
        // factorizations, so tests arm this branch with a factor of their own
        tolerance *= c->debugToleranceFactor;
        toleranceChanged = 1;
      }
    }
    int number = c->numberInfeasible;
    // numberWanted (src/ClpDualRowSteepest.cpp:258-278): how many entries above the tolerance one call looks at
    int numberWanted;
    if (c->steepestMode < 2) {
      numberWanted = number + 1;
    } else if (c->steepestMode == 2) {
      numberWanted = max(c->chuzrFloor, number / 8);
    } else {
      double ratio = (double)c->factorElements / (double)D.m;
      numberWanted = max(c->chuzrFloor, number / 8);
      if (ratio < 1.0) {
        numberWanted = max(c->chuzrFloor, number / 20);
      } else if (ratio > 10.0) {
        ratio = number * (ratio / 80.0);
        if (ratio > number)
          numberWanted = number + 1;
        else
          numberWanted = max(c->chuzrFloor, (int)ratio);
      }
    }
    if (c->largestPrimalError > 1.0e-3)
      numberWanted = number + 1;  // "be safe"
    if (numberWanted <= number)
      c->chuzrPartialScans++;
    double dstart = ((double)number) * randomDouble(c);
    c->chuzrNumber = number;
    c->chuzrStart = (int)dstart;
    c->chuzrWanted = numberWanted;
    c->chuzrTolChanged = toleranceChanged;
  } else {
    if (c->largestPrimalError > 1.0e-8)
      tolerance *= c->largestPrimalError / 1.0e-8;
    c->chuzrNumber = D.m;
    c->chuzrStart = 0;
    c->chuzrWanted = D.m + 1;
    c->chuzrTolChanged = 0;
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
  if (c->presetRowPlus1 > 0) {
    // the host's free-first entry of dualRow chose this pivot's row (src/ClpSimplexDual.cpp:3005-3055): the pivot-rule object is
    // not asked -- no random start, no touch-up of the last pivot row -- and the list scan finds nothing to look at
    if (c->stepLimit >= 0 && c->numberIterations >= c->stepLimit) {
      c->state = EXIT_STEP_LIMIT;
      return;
    }
    c->chuzrNumber = 0;
    c->chuzrStart = 0;
    c->chuzrLast = -1;
    c->chuzrTolerance = 0.0;
    c->preDone = 1;
    return;
  }
  chuzrPreBody(D);
}

#define CHZ_ITEMS 2
// ---- the list scan of ClpDualRowSteepest::pivotRow in ITS order, by one workgroup (src/ClpDualRowSteepest.cpp:279-335) ----
// Used where the order decides more than ties: the span of a partial scan (modes 2 / 3) in which the numberWanted-th entry above the
// tolerance falls, the rest of such a scan from the first span that holds a flagged candidate or the last pivot row, and the second
// call of :338-346.  The reference's loop carries two things from entry to entry -- `largest` (the ratio of the row chosen so far) and
// numberWanted -- and both are scans over the list in rank order (rank 0 = the random start):
//   largest before e   = max of value / weight over the earlier entries that could be chosen (above the tolerance, not flagged,
//                        really infeasible), an exclusive prefix maximum;
//   tickets used by e  = 1 for an entry above the tolerance, except 0 for a flagged one that passes `value > largest * weight` (it
//                        hands its ticket back, :321-324) and 0 for the last pivot row when it is put off by `continue` (:303-305);
//   e is looked at     iff the tickets used before it are fewer than numberWanted; the chosen row is the first largest ratio among
//                        the entries looked at (`value > largest * weight` is strict).
// A chunk of CHZ_CHUNK entries is loaded level by level (list -> row -> basic variable -> value and bounds) and scanned slab by slab
// (256 consecutive ranks, one per thread): wave scans by shuffles, the four waves through LDS, the running values in registers.
#define CHZ_CHUNK_ITEMS 4
#define CHZ_CHUNK (256 * CHZ_CHUNK_ITEMS)
struct ChzStage {
  double best;  // ratio of the chosen row so far (the reference's `largest`)
  int bestKey, bestRow, remaining, pad;
};
// ranks [rankBegin, rankEnd) of the list, continuing from a choice made on the ranks before them (carryKey < 0: none)
__device__ inline void chuzrOrderedScan(const Dev &D, ChzStage *S, double tolerance, int number, int start, int last, int wanted,
                                        int rankBegin = 0, int rankEnd = 2147483647, double carryBest = 0.0, int carryKey = -1, int carryRow = -1)
{
  __shared__ double sMax[2][4], sLastC, sBv[4];
  __shared__ int sSum[2][4], sLastRank, sBk[4], sBr[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  __syncthreads();
  if (t == 0) {
    S->best = carryBest;
    S->bestKey = carryKey;
    S->bestRow = carryRow;
    S->remaining = wanted;
  }
  __syncthreads();
  if (rankEnd > number)
    rankEnd = number;
  for (int base = rankBegin; base < rankEnd; base += CHZ_CHUNK) {
    int iRow[CHZ_CHUNK_ITEMS], iSeq[CHZ_CHUNK_ITEMS];
    double value[CHZ_CHUNK_ITEMS], weight[CHZ_CHUNK_ITEMS];
#pragma unroll
    for (int q = 0; q < CHZ_CHUNK_ITEMS; q++) {
      const int rank = base + q * 256 + t;
      int i = start + rank;
      if (i >= number)
        i -= number;
      iRow[q] = rank < rankEnd ? D.infIndex[i] : -1;
    }
#pragma unroll
    for (int q = 0; q < CHZ_CHUNK_ITEMS; q++) {
      const int r = iRow[q] >= 0 ? iRow[q] : 0;
      value[q] = D.infeas[r];
      weight[q] = D.weights[r];
      iSeq[q] = D.pivotVariable[r];
    }
    unsigned char fl[CHZ_CHUNK_ITEMS];  // 1 above the tolerance, 2 flagged, 4 really infeasible, 8 the last pivot row
#pragma unroll
    for (int q = 0; q < CHZ_CHUNK_ITEMS; q++) {
      const unsigned char st = D.status[iSeq[q]];
      const double sv = D.sol[iSeq[q]], up = D.upper[iSeq[q]], lo = D.lower[iSeq[q]];
      unsigned char f = 0;
      if (iRow[q] >= 0 && value[q] > tolerance) {
        f = 1;
        if (st & FLAGGED_BIT)
          f |= 2;
        if (sv > up + tolerance || sv < lo - tolerance)
          f |= 4;
        if (iRow[q] == last)
          f |= 8;
      }
      fl[q] = f;
      weight[q] = fmin(weight[q], 1.0e50);
    }
    const double carryL = S->best;
    const int remaining = S->remaining;
    double runL = carryL;  // `largest` in front of the slab (the same in every thread)
    int runUsed = 0;       // tickets used in front of the slab
    double best = 0.0;
    int bestKey = -1, bestRow = -1;
    if (t == 0)
      sLastRank = -1;
#pragma unroll
    for (int q = 0; q < CHZ_CHUNK_ITEMS; q++) {
      const int rank = base + q * 256 + t, par = q & 1;
      const unsigned char f = fl[q];
      // what this entry would make `largest` if it were chosen (the last pivot row joins below)
      const double cand = ((f & 15) == 5) ? value[q] / weight[q] : -1.0;
      double incl = cand;
      for (int o = 1; o < 64; o <<= 1) {
        const double up_ = __shfl_up(incl, o);
        if (lane >= o)
          incl = fmax(incl, up_);
      }
      double excl = __shfl_up(incl, 1);
      if (lane == 0)
        excl = -1.0;
      if (lane == 63)
        sMax[par][wv] = incl;
      __syncthreads();
      double Lb = fmax(runL, excl);
      for (int w = 0; w < wv; w++)
        Lb = fmax(Lb, sMax[par][w]);
      // the last pivot row: put off (`continue`, no ticket) when its scaled value cannot win, otherwise it competes with value * 1e-10
      int skipTicket = 0;
      double mine = cand;
      if (f & 8) {
        if (value[q] > Lb * weight[q]) {
          if (value[q] * 1.0e-10 < Lb * weight[q]) {
            skipTicket = 1;
          } else if ((f & 6) == 4) {
            mine = (value[q] * 1.0e-10) / weight[q];
            sLastC = mine;
            sLastRank = rank;
          }
        }
      }
      __syncthreads();
      if (sLastRank >= 0 && rank > sLastRank)
        Lb = fmax(Lb, sLastC);
      // tickets
      int d = (f & 1) ? 1 : 0;
      if ((f & 3) == 3 && !(f & 8) && value[q] > Lb * weight[q])
        d = 0;  // flagged and it would have been chosen: numberWanted++ (:321-324)
      if ((f & 11) == 11) {  // the last pivot row, flagged: the reference tests `continue` first, then flagged
        if (value[q] > Lb * weight[q] && !skipTicket)
          d = 0;
      }
      if (skipTicket)
        d = 0;
      int sincl = d;
      for (int o = 1; o < 64; o <<= 1) {
        const int up_ = __shfl_up(sincl, o);
        if (lane >= o)
          sincl += up_;
      }
      if (lane == 63)
        sSum[par][wv] = sincl;
      __syncthreads();
      int usedBefore = runUsed + sincl - d;
      for (int w = 0; w < wv; w++)
        usedBefore += sSum[par][w];
      // looked at iff fewer than numberWanted tickets were used before it; chosen: first largest ratio, strictly above what came before
      if (mine >= 0.0 && usedBefore < remaining && !skipTicket && (bestKey < 0 || mine > best)) {  // (this thread's ranks ascend)
        best = mine;
        bestKey = rank;
        bestRow = iRow[q];
      }
      // running values for the next slab
      runL = fmax(fmax(fmax(runL, sMax[par][0]), fmax(sMax[par][1], sMax[par][2])), sMax[par][3]);
      if (sLastRank >= 0 && sLastRank >= base + q * 256 && sLastRank < base + (q + 1) * 256)
        runL = fmax(runL, sLastC);
      runUsed += sSum[par][0] + sSum[par][1] + sSum[par][2] + sSum[par][3];
    }
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
      sBv[wv] = best;
      sBk[wv] = bestKey;
      sBr[wv] = bestRow;
    }
    __syncthreads();
    if (t == 0) {
      for (int i = 1; i < 4; i++)
        if (sBk[i] >= 0 && (bestKey < 0 || sBv[i] > best || (sBv[i] == best && sBk[i] < bestKey))) {
          best = sBv[i];
          bestKey = sBk[i];
          bestRow = sBr[i];
        }
      // value > largest * weight (:296): a later chunk only wins with a strictly larger ratio
      if (bestKey >= 0 && best > S->best) {
        S->best = best;
        S->bestKey = bestKey;
        S->bestRow = bestRow;
      }
      S->remaining = runUsed >= remaining ? 0 : remaining - runUsed;
    }
    __syncthreads();
    if (S->remaining <= 0)
      break;
  }
}

template <bool COHERENT> __device__ inline void chuzrFinalBody(const Dev &D, int nblocks, int wide, ChzStage *S);
// fuseFinal >= 0: the last workgroup to finish also makes the final selection and builds the BTRAN
// t-vector (no separate launch); the value is the wide-row flag of that stage
__global__ void __launch_bounds__(256) k_chuzr_scan(Dev D, int fuseFinal = -1)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double shv[4];
  __shared__ int shk[4], shr[4];
  __shared__ ChzStage stage;
  const double tolerance = c->chuzrTolerance;
  const int number = c->chuzrNumber, start = c->chuzrStart, last = c->chuzrLast;
  double best = 0.0;
  int bestKey = -1, bestRow = -1;
  if (c->pivotRule != 0 && c->chuzrWanted <= number) {
    // partial scan (modes 2 / 3): only the first numberWanted entries above the tolerance, counted from the random start, are looked
    // at.  Every workgroup takes a span of RANKS (rank 0 = the random start) and reports the best of its span as if all of it counted,
    // the number of entries above the tolerance in it, and whether a flagged candidate or the last pivot row is among them; the final
    // selection finds the workgroup the cut falls into, takes the spans before it whole and walks that one span again in order
    // (chuzrFinalBody).  Spans with the two exceptions send the final selection to the ordered walk of the whole list.
    const int base = blockIdx.x * (256 * CHZ_ITEMS);
    int iRow[CHZ_ITEMS], iSeq[CHZ_ITEMS];
    double value[CHZ_ITEMS], rawWeight[CHZ_ITEMS];
#pragma unroll
    for (int q = 0; q < CHZ_ITEMS; q++) {
      const int rank = base + q * 256 + threadIdx.x;
      int i = start + rank;
      if (i >= number)
        i -= number;
      iRow[q] = rank < number ? D.infIndex[i] : -1;
    }
#pragma unroll
    for (int q = 0; q < CHZ_ITEMS; q++) {
      const int r = iRow[q] >= 0 ? iRow[q] : 0;
      value[q] = D.infeas[r];
      rawWeight[q] = D.weights[r];
      iSeq[q] = D.pivotVariable[r];
    }
    int above = 0, special = 0;
#pragma unroll
    for (int q = 0; q < CHZ_ITEMS; q++) {
      const unsigned char st = D.status[iSeq[q]];
      const double sv = D.sol[iSeq[q]], up = D.upper[iSeq[q]], lo = D.lower[iSeq[q]];
      if (iRow[q] >= 0 && value[q] > tolerance) {
        above++;
        if ((st & FLAGGED_BIT) || iRow[q] == last) {
          special = 1;
        } else if (sv > up + tolerance || sv < lo - tolerance) {
          const double ratio = value[q] / fmin(rawWeight[q], 1.0e50);
          const int rank = base + q * 256 + threadIdx.x;
          if (ratio > best || (ratio == best && bestKey >= 0 && rank < bestKey)) {
            best = ratio;
            bestKey = rank;
            bestRow = iRow[q];
          }
        }
      }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) {
      double ov = __shfl_down(best, o);
      int ok = __shfl_down(bestKey, o);
      int orow = __shfl_down(bestRow, o);
      above += __shfl_down(above, o);
      special |= __shfl_down(special, o);
      if (ok >= 0 && (bestKey < 0 || ov > best || (ov == best && ok < bestKey))) {
        best = ov;
        bestKey = ok;
        bestRow = orow;
      }
    }
    __shared__ int sha[4], shs[4];
    if (lane == 0) {
      shv[wv] = best;
      shk[wv] = bestKey;
      shr[wv] = bestRow;
      sha[wv] = above;
      shs[wv] = special;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 1; i < 4; i++) {
        above += sha[i];
        special |= shs[i];
        if (shk[i] >= 0 && (bestKey < 0 || shv[i] > best || (shv[i] == best && shk[i] < bestKey))) {
          best = shv[i];
          bestKey = shk[i];
          bestRow = shr[i];
        }
      }
      stc(&D.chzBest[blockIdx.x], best);
      stc(&D.chzKey[blockIdx.x], bestKey);
      stc(&D.chzRow[blockIdx.x], bestRow);
      stc(&D.chzCnt[blockIdx.x], above | (special ? (1 << 30) : 0));
    }
    if (fuseFinal >= 0 && lastBlockDone(D.ctrl, 3))
      chuzrFinalBody<true>(D, gridDim.x, fuseFinal, &stage);
    return;
  }
  const int base = blockIdx.x * (256 * CHZ_ITEMS);
  // the chain list entry -> row -> basic variable -> its value and bounds is three dependent loads
  // deep: every level is requested for all of this thread's items before anything is used
  int idx[CHZ_ITEMS], iRow[CHZ_ITEMS], iSeq[CHZ_ITEMS];
  double value[CHZ_ITEMS], rawWeight[CHZ_ITEMS], sv[CHZ_ITEMS], up[CHZ_ITEMS], lo[CHZ_ITEMS];
  unsigned char st[CHZ_ITEMS];
  const bool steepest = c->pivotRule != 0;
#pragma unroll
  for (int q = 0; q < CHZ_ITEMS; q++) {
    idx[q] = base + q * 256 + threadIdx.x;
    iRow[q] = idx[q] < number ? (steepest ? D.infIndex[idx[q]] : idx[q]) : -1;
  }
#pragma unroll
  for (int q = 0; q < CHZ_ITEMS; q++) {
    const int r = iRow[q] >= 0 ? iRow[q] : 0;
    value[q] = steepest ? D.infeas[r] : 0.0;
    rawWeight[q] = steepest ? D.weights[r] : 1.0;
    iSeq[q] = D.pivotVariable[r];
  }
#pragma unroll
  for (int q = 0; q < CHZ_ITEMS; q++) {
    st[q] = D.status[iSeq[q]];
    sv[q] = D.sol[iSeq[q]];
    up[q] = D.upper[iSeq[q]];
    lo[q] = D.lower[iSeq[q]];
  }
#pragma unroll
  for (int q = 0; q < CHZ_ITEMS; q++) {
    if (iRow[q] < 0)
      continue;
    const int i = idx[q];
    if (steepest) {
      double v = value[q];
      if (v > tolerance) {
        double weight = fmin(rawWeight[q], 1.0e50);
        if (iRow[q] == last)
          v *= 1.0e-10;
        if (!(st[q] & FLAGGED_BIT)) {
          if (sv[q] > up[q] + tolerance || sv[q] < lo[q] - tolerance) {
            double ratio = v / weight;
            int rank = i - start;
            if (rank < 0)
              rank += number;
            if (ratio > best || (ratio == best && bestKey >= 0 && rank < bestKey)) {
              best = ratio;
              bestKey = rank;
              bestRow = iRow[q];
            }
          }
        }
      }
    } else {
      double infeas = fmax(sv[q] - up[q], lo[q] - sv[q]);
      if (infeas > tolerance && !(st[q] & FLAGGED_BIT)) {
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
    stc(&D.chzBest[blockIdx.x], best);
    stc(&D.chzKey[blockIdx.x], bestKey);
    stc(&D.chzRow[blockIdx.x], bestRow);
  }
  if (fuseFinal >= 0 && lastBlockDone(D.ctrl, 3))
    chuzrFinalBody<true>(D, gridDim.x, fuseFinal, &stage);
}

// iteration BTRAN, back end: rho[i] = slack part or sum of the gemvT partials, flush tiny, piNeg,
// rhoSlot (unpruned, for the nucleus update) and the per-block partial of sum rho^2 (DSE norm)
__global__ void __launch_bounds__(256) k_rho_finish3(Dev D, int wide = 0, int nbCols = -1, int nSlots = 0)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double sh[16];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double sq = 0.0;
  if (i < D.m) {
    double v;
    int sr = D.luMode ? -2 : D.slotOfRow[i];
    if (sr == -2) {
      v = D.lu->y[i];  // LU mode: the BTRAN result by row (k_lu_bt_gather / k_lu_bt_back)
    } else if (sr >= 0) {
      // y_R = Minv^T t with t given as a short list: read only those rows of Minv
      const int tc = c->tCount;
      v = 0.0;
      if (wide) {
        // per-chunk partials of Minv^T t from k_gemvT_partial2, summed in chunk order
        const int nchunk = (c->k + 63) >> 6;
        for (int ch = 0; ch < nchunk; ch++)
          v += D.partial[(size_t)ch * D.ld + sr];
      } else {
        for (int q = 0; q < tc; q++)
          v += D.Minv[(size_t)D.tIndex[q] * D.ld + sr] * D.tValue[q];
      }
      D.rhoSlot[sr] = v;
    } else {
      int p = D.posOfSlack[i];
      v = (p >= 0) ? D.vecC[p] * -1.0 : 0.0;
    }
    if (fabs(v) <= c->zeroTolerance)
      v = 0.0;
    D.rho[i] = v;
    D.piNeg[i] = -v;
    if (sr >= 0)
      D.rhoSlotF[sr] = v;
    sq = v * v;
  }
  // bitmap of the nonzero rows of pi, one 64-bit word per wave (the pricing kernel keeps it in LDS)
  unsigned long long mask = __ballot(sq != 0.0);
  if ((threadIdx.x & 63) == 0 && i < D.m + 63)
    D.piBits[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = mask;
  double s = blockSum(sq, sh);
  if (threadIdx.x == 0)
    D.normPartial[blockIdx.x] = s;
  if (nbCols >= 0) {
    // first ratio pass for the row (slack) part of the tableau row, as the pricing kernel does it
    // for the columns (ClpPackedMatrix.cpp:1007-1090); candidate counts of the compaction blocks:
    // this block's rows here, the column blocks are zeroed for the pricing kernel's atomics
    __shared__ int shi[17];
    int flag = 0;
    double ratio = 1.0e31;
    if (i < D.m) {
      const double dualT = -c->dualTolerance;
      const double value = D.rho[i];
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
    int total;
    blockRank(flag, total, shi);
    double bmin = blockMin(ratio, sh);
    if (threadIdx.x == 0) {
      D.blockCount[blockIdx.x] = total;
      D.blockMin[blockIdx.x] = bmin;
      D.blockSum[blockIdx.x] = 0.0;
    }
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < nbCols; g += gridDim.x * blockDim.x) {
      D.blockCount[gridDim.x + g] = 0;
      D.blockMin[gridDim.x + g] = 1.0e31;
      D.blockSum[gridDim.x + g] = 0.0;
    }
    // the per-workgroup slots of the by-column kernels: the by-row form leaves them untouched
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < nSlots; g += gridDim.x * blockDim.x) {
      D.sellMin[g] = 1.0e31;
      D.sellBytes[g] = 0.0;
    }
  }
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

// =============================================================================================
// Row pricing, v2: sliced-ELL (SELL-64) sweep.  One wave owns a slice of 64 columns; entry t of
// lane l sits at sellStart[slice] + 64*t + l, so every step is one fully coalesced 512 B (elements)
// + 256 B (row indices) wave transaction, 8 steps in flight per lane.  Each lane still adds its own
// column's products in ascending entry order: the result is bit-identical to the reference's scalar
// loop (ClpPackedMatrix.cpp:1872-1886) and to the v1 kernel.  Fused first ratio pass as in v1.
// =============================================================================================
#define SELL_U 8
#define SELL_LONG 128  // columns longer than this are priced by a wave each instead of a SELL lane
#define SELL_BITS_MAX 8192  // 64-bit words of the pi bitmap kept in LDS (rows <= 524288)
// PIPE: software-pipelined loads; NT: non-temporal matrix loads; BITS: gather pi only where the
// row's bit is set (pi is sparse for most pivots: the gather traffic scales with nnz(pi)/m)
template <bool PIPE, bool NT, bool BITS, bool COND = false>
// dbg (clpgpu_debug_price_bench only; 0 in the chain): switches parts of the kernel off so that their cost can be measured apart --
// 1 no candidate-count atomics, 2 tableau row / flags stored in SELL order (coalesced) instead of by column, 4 no status / dj
// gathers, 8 no matrix sweep, 16 no gather of pi
__device__ inline void priceSellBody(Dev D, unsigned long long *bits, int countCols = 0, int dbg = 0)
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
  bool sparsePi = false;
  if (BITS) {
    int pop = 0;
    for (int w = threadIdx.x; w < nwords; w += blockDim.x) {
      unsigned long long word = D.piBits[w];
      bits[w] = word;
      pop += __popcll(word);
    }
    if (COND) {
      // nnz(pi) decides between the conditional-fetch loop and the plain stream (uniform per launch)
      __shared__ int shPop[4];
      for (int o = 32; o > 0; o >>= 1)
        pop += __shfl_xor(pop, o);
      if ((threadIdx.x & 63) == 0)
        shPop[threadIdx.x >> 6] = pop;
      __syncthreads();
      pop = shPop[0] + shPop[1] + shPop[2] + shPop[3];
      sparsePi = 12 * (long long)pop < (long long)D.m;
    } else {
      __syncthreads();
    }
  }
  auto piAt = [&](int r) -> double {
    if (dbg & 16)
      return 1.0;
    if (BITS) {
      return ((bits[r >> 6] >> (r & 63)) & 1ull) ? D.piNeg[r] : 0.0;
    } else {
      return D.piNeg[r];
    }
  };
  double ratio = 1.0e31, bytes = 0.0;
  int jOut = -1, flagOut = 0;  // this lane's column and its results, for the windowed write-out below
  double valueOut = 0.0;
  if (slice < D.numSlices) {
    const int idx = slice * 64 + lane;
    const int j = D.sellCol[idx];
    int len = 0, wanted = 0;
    int hits = -1;  // elements really fetched (conditional form); -1: all of them
    if (j >= 0) {
      wanted = (dbg & 4) ? 2 : (D.status[j] & 3) - 1;
      if (wanted)
        len = D.sellLen[idx];
    }
    int maxLen = (dbg & 8) ? 0 : len;
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
      } else if (COND && sparsePi) {
        // pi is sparse: stream the row indices, fetch element and pi only where the row's bit is
        // set.  Skipped products are exact zeros in the unconditional loop, so the sum is the same.
        hits = 0;
        for (int t = 0; t < maxLen; t += SELL_U) {
          int r[SELL_U];
          double e[SELL_U], pv[SELL_U];
          bool hit[SELL_U];
#pragma unroll
          for (int u = 0; u < SELL_U; u++)
            r[u] = rp[(t + u) * 64];
#pragma unroll
          for (int u = 0; u < SELL_U; u++) {
            hit[u] = (t + u < len) && ((bits[r[u] >> 6] >> (r[u] & 63)) & 1ull);
            e[u] = 0.0;
            pv[u] = 0.0;
            if (hit[u]) {
              e[u] = ep[(t + u) * 64];
              pv[u] = D.piNeg[r[u]];
            }
          }
#pragma unroll
          for (int u = 0; u < SELL_U; u++)
            if (hit[u]) {
              value += pv[u] * e[u];
              hits++;
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
    if (dbg & 8)
      value = len > 0 ? 1.0 + 1.0e-3 * lane : 0.0;
    if (j >= 0) {
      int flag = 0;
      if (wanted) {
        // bytes this column really streams: every row index (4 B), the element (8 B) only where it is fetched --
        // all of them in the unconditional forms, those under a set bit of pi in the conditional one
        bytes = 4.0 * len + 8.0 * (hits < 0 ? len : hits) + 4.0;
        if (fabs(value) > zeroTolerance) {
          bytes += 20.0;
          if (wanted > 0) {
            double mult = (wanted == 1) ? -1.0 : 1.0;
            double alpha = value * mult;
            if (alpha > 0.0) {
              double oldValue = ((dbg & 4) ? 1.0 : D.dj[j]) * mult;
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
      jOut = j;
      valueOut = value;
      flagOut = flag;
      if (!D.sellWindowed) {
        const int at = ((dbg & 2) && idx < D.n) ? idx : j;
        D.alphaCol[at] = value;
        D.candFlag[D.m + at] = (unsigned char)flag;
        // candidate count of the column's compaction block (integer atomic: order independent)
        if (flag && countCols && !(dbg & 1))
          atomicAdd(&D.blockCount[((D.m + PRICE_BLOCK - 1) / PRICE_BLOCK) + ((j - D.firstColumn) / PRICE_BLOCK)], 1);
      }
    }
  }
  if (D.sellWindowed) {
    // Windowed SELL copy (buildSell): this workgroup's four slices hold exactly the columns of ONE compaction block of the
    // N-wide kernels -- keys j0 .. j0 + 255, sorted by length inside the window only -- so the tableau row and the flags
    // go out as two coalesced stores through LDS, and the block's candidate count is one integer per workgroup: no
    // 200 000 scattered 8-byte stores and no per-candidate atomics (measured apart with clpgpu_debug_price_bench: -10 us and
    // -2 us of the 68 us launch, profiles/r04_price_probe.txt).  Columns of the window that live elsewhere (long columns:
    // priceLongBody writes and counts them itself) keep the 0xFF mark and are left alone.
    __shared__ double shAlpha[PRICE_BLOCK];
    __shared__ unsigned char shFlag[PRICE_BLOCK];
    __shared__ int shCount[17];
    const int j0 = D.firstColumn + (D.sellWinBase + (int)blockIdx.x) * PRICE_BLOCK;
    shFlag[threadIdx.x] = 0xFF;
    __syncthreads();
    if (jOut >= 0) {
      shAlpha[jOut - j0] = valueOut;
      shFlag[jOut - j0] = (unsigned char)flagOut;
    }
    __syncthreads();
    const unsigned char f = shFlag[threadIdx.x];
    if (f != 0xFF) {
      D.alphaCol[j0 + threadIdx.x] = shAlpha[threadIdx.x];
      D.candFlag[D.m + j0 + threadIdx.x] = f;
    }
    const int total = blockSumInt(f == 1 ? 1 : 0, shCount);
    if (threadIdx.x == 0 && total && countCols && !(dbg & 1))
      atomicAdd(&D.blockCount[((D.m + PRICE_BLOCK - 1) / PRICE_BLOCK) + D.sellWinBase + (int)blockIdx.x], total);
  }
  double bmin = blockMin(ratio, shd);
  double bsum = blockSum(bytes, shd);
  if (threadIdx.x == 0) {
    D.sellMin[blockIdx.x] = bmin;
    D.sellBytes[blockIdx.x] = bsum;
  }
}

// variant: 1 plain, 2 bitmap, 3 pipelined+bitmap, 4 pipelined+nt+bitmap, 5 nt+bitmap
// long columns (see SELL_LONG): a workgroup strides one CSC column (coalesced) and reduces with a
// fixed tree (64-way per wave, then the four waves in order) -- deterministic, equal to the
// reference's sequential sum to rounding.  A power-law LP has a few columns with tens of thousands
// of entries; one wave per column would leave that tail to a single wave.  Same fused first ratio
// pass and the same per-workgroup outputs as the SELL body.
__device__ inline void priceLongBody(const Dev &D, int blk, int countCols)
{
  const Ctrl *c = D.ctrl;
  __shared__ double shd[16];
  const double dualT = -c->dualTolerance;
  const double acceptablePivot = c->acceptablePivot;
  const double zeroTolerance = c->zeroTolerance;
  double ratio = 1.0e31, bytes = 0.0;
  if (blk < D.numLong) {  // uniform per workgroup
    const int j = D.longCol[blk];
    const int wanted = (D.status[j] & 3) - 1;
    double value = 0.0;
    int flag = 0;
    if (wanted) {
      const int start = D.colStart[j], end = D.colStart[j + 1];
      double acc = 0.0;
      for (int p = start + threadIdx.x; p < end; p += blockDim.x)
        acc += D.piNeg[D.row[p]] * D.elem[p];
      value = blockSum(acc, shd);
      bytes = 12.0 * (end - start) + 4.0;
      if (fabs(value) > zeroTolerance) {
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
                ratio = (oldValue - dualT) / alpha;
            }
          }
        }
      } else {
        value = 0.0;
      }
    }
    if (threadIdx.x == 0) {
      D.alphaCol[j] = value;
      D.candFlag[D.m + j] = (unsigned char)flag;
      if (flag && countCols)
        atomicAdd(&D.blockCount[((D.m + PRICE_BLOCK - 1) / PRICE_BLOCK) + ((j - D.firstColumn) / PRICE_BLOCK)], 1);
    }
  }
  if (threadIdx.x == 0) {
    D.sellMin[blockIdx.x] = ratio;
    D.sellBytes[blockIdx.x] = bytes;
  }
}

// =============================================================================================
// Row pricing BY ROW for sparse pi -- ClpPackedMatrix::transposeTimes' other branch
// (src/ClpPackedMatrix.cpp:727-754 chooses it when nnz(pi) <= factor * m; transposeTimesByRow :1307,
// gutsOfTransposeTimesByRowGE3 :5176-5225: accumulate pi_i * row_i into a dense scratch).
// A by-column sweep streams every row index of the matrix (40 MB at config 4) however sparse pi is;
// by row only the rows in supp(pi) are read: 12 B x sum of their lengths.
// Bit-identical to the by-column result without floating-point atomics:
//   pass 1 (k_price_sell's extra workgroups): a wave per nonzero row of pi walks the nonbasic part of
//     that row in the row copy; every entry draws an integer ticket from its column's touch counter
//     (order independent) and, for the first ROW_SLOTS tickets, leaves (row, pi_i * a_ij) in the
//     column's slot of that number;
//   pass 2 (k_price_row_finish, one thread per column): the column's <= ROW_SLOTS contributions are put
//     in ascending row order and added -- the by-column loop's sum with its exact-zero terms left out;
//     a column with more contributors (dense pi) recomputes its own dot product by column, in CSC
//     order.  Then the same fused first ratio pass, candidate flags and per-block counts as the
//     by-column kernels.
// Which branch runs is decided on the device from nnz(pi) (popcount of the bitmap) against rowMax;
// every workgroup of both launches takes the same decision.
// =============================================================================================
#define ROW_SLOTS 8  // contributors a column keeps individually in by-row pricing
__device__ inline int piPopcount256(const Dev &D)
{
  __shared__ int shPopRow[4];
  const int nwords = (D.m + 63) >> 6;
  int pop = 0;
  for (int w = threadIdx.x; w < nwords; w += blockDim.x)
    pop += __popcll(D.piBits[w]);
  for (int o = 32; o > 0; o >>= 1)
    pop += __shfl_xor(pop, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
    shPopRow[threadIdx.x >> 6] = pop;
  __syncthreads();
  return shPopRow[0] + shPopRow[1] + shPopRow[2] + shPopRow[3];
}
// pass 1: workgroup blk (of cdiv(m, 256)) owns 256 rows = four bitmap words, a wave per word
__device__ inline void priceRowScatterBody(const Dev &D, int blk, int fullRows)
{
  const int lane = threadIdx.x & 63;
  const int w = blk * 4 + (threadIdx.x >> 6);
  const int nwords = (D.m + 63) >> 6;
  unsigned long long word = w < nwords ? D.piBits[w] : 0ull;
  int entries = 0;
  while (word) {
    const int bit = __ffsll((long long)word) - 1;
    word &= word - 1ull;
    const int i = w * 64 + bit;
    const double v = D.piNeg[i];
    const int e = D.rowStart[i + 1];
    // the nonbasic part of the row (the engine keeps the row copy partitioned [basic | nonbasic]; the
    // plug-in call walks whole rows, its caller's status array decides); two strides of 64 entries in flight
    for (int q0 = D.rowStart[i] + (fullRows ? 0 : D.basicCount[i]); q0 < e; q0 += 128) {
      int jj[2];
      double aa[2];
      unsigned char st[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int q = q0 + 64 * u + lane;
        jj[u] = q < e ? D.ccol[q] : -1;
        aa[u] = q < e ? D.relem[q] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 2; u++)
        st[u] = jj[u] >= 0 ? D.status[jj[u]] : 1;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (jj[u] >= D.priceFirst && jj[u] < D.priceLast && (st[u] & 3) != 1) {
          // the ticket is order independent; the first ROW_SLOTS contributors leave (row, product) in
          // the column's slots, pass 2 puts them in row order
          const int old = atomicAdd(&D.touchCol[jj[u]], 1);
          if (old < ROW_SLOTS) {
            const size_t at = (size_t)jj[u] * ROW_SLOTS + old;
            D.touchRow[at] = i;
            D.touchVal[at] = v * aa[u];
          }
          entries++;
        }
      }
    }
  }
  // algorithmic bytes of this pass (B_row): 12 B per entry visited + 12 B per pi nonzero
  __shared__ int shEnt[4];
  for (int o = 32; o > 0; o >>= 1)
    entries += __shfl_xor(entries, o);
  if (lane == 0)
    shEnt[threadIdx.x >> 6] = entries;
  __syncthreads();
  if (threadIdx.x == 0) {
    int pops = 0;
    for (int u = 0; u < 4; u++)
      if (blk * 4 + u < nwords)
        pops += __popcll(D.piBits[blk * 4 + u]);
    D.blockSum[blk] = 12.0 * (shEnt[0] + shEnt[1] + shEnt[2] + shEnt[3]) + 12.0 * pops;
  }
}
// pass 2: one thread per column of this GPU's range, compaction-block aligned (the k_price scheme:
// per-block count / min ratio / bytes stored directly, no atomics)
__global__ void __launch_bounds__(PRICE_BLOCK) k_price_row_finish(Dev D, int nbRows, int rowMax)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int pop = rowMax > 0 ? piPopcount256(D) : 0;
  const bool byRow = rowMax > 0 && pop <= rowMax;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    c->lastPriceByRow = byRow ? 1 : 0;
  if (!byRow)
    return;
  __shared__ double shd[16];
  __shared__ int shi[17];
  const double dualT = -c->dualTolerance;
  const double acceptablePivot = c->acceptablePivot;
  const double zeroTolerance = c->zeroTolerance;
  const int j = D.firstColumn + (int)blockIdx.x * PRICE_BLOCK + threadIdx.x;
  int flag = 0;
  double ratio = 1.0e31, bytes = 0.0;
  if (j < D.lastColumn) {
    double value = 0.0;
    const int cnt = (j >= D.priceFirst && j < D.priceLast) ? D.touchCol[j] : 0;
    if (cnt) {
      D.touchCol[j] = 0;
      if (cnt <= ROW_SLOTS) {
        // the contributors' products, added in ascending row order: the by-column loop's sum with its
        // exact-zero terms left out (0.0 + x == x, x + 0.0 == x)
        int rr[ROW_SLOTS];
        double vv[ROW_SLOTS];
        const size_t at = (size_t)j * ROW_SLOTS;
#pragma unroll
        for (int u = 0; u < ROW_SLOTS; u++) {
          rr[u] = u < cnt ? D.touchRow[at + u] : 0x7fffffff;
          vv[u] = u < cnt ? D.touchVal[at + u] : 0.0;
        }
        // odd-even transposition sort on the row index (ROW_SLOTS rounds, registers only)
#pragma unroll
        for (int round = 0; round < ROW_SLOTS; round++) {
#pragma unroll
          for (int u = round & 1; u + 1 < ROW_SLOTS; u += 2) {
            if (rr[u + 1] < rr[u]) {
              int tr = rr[u];
              rr[u] = rr[u + 1];
              rr[u + 1] = tr;
              double tv = vv[u];
              vv[u] = vv[u + 1];
              vv[u + 1] = tv;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < ROW_SLOTS; u++)
          if (u < cnt)
            value += vv[u];
      } else {
        // more contributors than slots (dense pi): the column's own dot product, eight entries per trip
        const int pe = D.colStart[j + 1];
        for (int p0 = D.colStart[j]; p0 < pe; p0 += 8) {
          int r8[8];
          double e8[8], p8[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            r8[u] = p0 + u < pe ? D.row[p0 + u] : 0;
            e8[u] = p0 + u < pe ? D.elem[p0 + u] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; u++)
            p8[u] = D.piNeg[r8[u]];
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (p0 + u < pe)
              value += p8[u] * e8[u];
        }
      }
      bytes = 8.0;  // the touched scratch entry
      const int wanted = (D.status[j] & 3) - 1;
      if (fabs(value) > zeroTolerance) {
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
  int total;
  blockRank(flag, total, shi);
  double bmin = blockMin(ratio, shd);
  double bsum = blockSum(bytes, shd);
  if (threadIdx.x == 0) {
    D.blockCount[nbRows + blockIdx.x] = total;
    D.blockMin[nbRows + blockIdx.x] = bmin;
    D.blockSum[nbRows + blockIdx.x] = bsum;
  }
}
// plug-in form (clpgpu_price_row): what k_rho_finish3 leaves for the pricing launches -- the first
// ratio pass of the row (slack) part with its per-block counts, neutral values in the column blocks
// and in the per-workgroup slots of the by-column kernel
__global__ void __launch_bounds__(PRICE_BLOCK) k_price_row_init(Dev D, int nbCols, int nSlots)
{
  const Ctrl *c = D.ctrl;
  __shared__ double sh[16];
  __shared__ int shi[17];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int flag = 0;
  double ratio = 1.0e31;
  if (i < D.m) {
    const double dualT = -c->dualTolerance;
    const double value = D.rho[i];
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
  unsigned long long mask = __ballot(i < D.m && D.rho[i < D.m ? i : 0] != 0.0);
  if ((threadIdx.x & 63) == 0 && i < D.m + 63)
    D.piBits[i >> 6] = mask;
  int total;
  blockRank(flag, total, shi);
  double bmin = blockMin(ratio, sh);
  if (threadIdx.x == 0) {
    D.blockCount[blockIdx.x] = total;
    D.blockMin[blockIdx.x] = bmin;
    D.blockSum[blockIdx.x] = 0.0;
  }
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < nbCols; g += gridDim.x * blockDim.x) {
    D.blockCount[gridDim.x + g] = 0;
    D.blockMin[gridDim.x + g] = 1.0e31;
    D.blockSum[gridDim.x + g] = 0.0;
  }
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < nSlots; g += gridDim.x * blockDim.x) {
    D.sellMin[g] = 1.0e31;
    D.sellBytes[g] = 0.0;
  }
}

// workgroups [0, nSellBlocks) sweep the SELL slices, the next numLong the long columns; the last
// nRowBlocks (= cdiv(m, 256), only when rowMax > 0) are pass 1 of the by-row form.  rowMax > 0: nnz(pi)
// <= rowMax sends the launch by row (the by-column workgroups return), otherwise by column.
// Row pricing by column with pi in LDS (round 5; the dense-pi regime the mature solve lives in).
// The gather of pi -- 8 bytes out of a 64-byte sector, one per matrix entry, 10^7 per launch at config 4 -- is what held
// k_price_sell at a quarter of the HBM rate while pi was a global array served by L2 (profiles/r04_price_probe_final.txt:
// 30 of 62 us).  Here the rows are cut into D.jdsTiles tiles of D.jdsTileRows (a pi tile of <= 134 KB in LDS) and one
// 1024-thread workgroup per CU walks the tiles: wave w owns slice (w & 3) of window blockIdx.x + gridDim.x * (w >> 2) --
// the windows are the 256-key compaction blocks of the windowed SELL copy, so the write-out is the coalesced one of
// priceSellBody -- and keeps each column's partial sum in a register from tile to tile.  Rows ascend inside a column, so
// the sum is the reference's sequential sum (ClpPackedMatrix.cpp:1872-1886), bit for bit.
// Streams: a slice's entries are stored tile after tile as a JAGGED sequence (device_state.h, buildJds): inside a tile the
// lanes are re-ordered by their entry count in that tile, the lanes of a step are a prefix of the wave and nothing is
// padded; the running sum changes lanes between tiles with one ds_bpermute.  Two steps share a record (one 4-byte and one
// 16-byte load per lane).  The loads do not depend on pi: a ring of two half-batches keeps RING steps of the stream in
// flight ACROSS the tile barriers (raw s_barrier + lgkmcnt only; the straight-line, all-lanes load code keeps the
// compiler's vmcnt counts exact), the next tile's first steps are requested while this tile is still being summed.
// pi tiles arrive by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction, no VGPR round trip), issued by inline
// asm so that the compiler does not drain the ring for them; the wait is placed by hand.
// A step without an entry reads a permanent 0.0 behind the tile: adding +-0.0 to the running sum is exact (the sum
// starts from +0.0 and cannot become -0.0), so the dependent chain per step is one add.
// Measured stand-alone (tools/price_lds_bench.hip, profiles/r05_price_lds_microbench*.txt): 31.7 us per launch at config 4
// against 66.9 us for the same windows with pi gathered from L2; plain streaming of the same bytes takes 19.8 us.
#define PL_THREADS 1024
#define PL_HP 2              // step pairs per half of the ring: RING = 4 PL_HP steps in flight per lane
#define PL_MAX_TILE_ROWS 16768
typedef __attribute__((address_space(3))) unsigned char pl_lds_byte;

__device__ __forceinline__ void plBarrier()
{
  // this wave's LDS traffic is done; global loads stay in flight across the barrier
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// one 1 KB wave-chunk of pi straight into LDS: lane l's 16 bytes land at ldsDst + 16 l (ldsDst wave-uniform)
__device__ __forceinline__ void plGlds16(const double *gsrc, unsigned ldsDst)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(ldsDst)
               : "memory");
}

struct PlHalf {
  unsigned r[PL_HP];
  double2 e[PL_HP];
};

// PL_HP consecutive step pairs of a segment whose lanes hold `cnt` entries; `off` = wave-uniform record position of step
// t0 (even), advanced past them.  Every lane loads: a lane without a record re-reads the first record of the pair.
__device__ __forceinline__ void plLoadHalf(PlHalf &h, const unsigned *__restrict__ rowPair, const double2 *__restrict__ elemPair, int cnt, int t0,
                                           unsigned &off, unsigned lane)
{
#pragma unroll
  for (int u = 0; u < PL_HP; u++) {
    const bool act = cnt > t0 + 2 * u;
    const unsigned k = (unsigned)__popcll(__ballot(act));
    const unsigned *rp = rowPair + off;  // wave-uniform
    const double2 *ep = elemPair + off;
    const unsigned at = act ? lane : 0u;
    h.r[u] = rp[at];
    h.e[u] = ep[at];
    off += k;
  }
}

__device__ __forceinline__ double plConsumeHalf(const PlHalf &h, const double *piTile, int cnt, int t0, double acc, unsigned zero, int maxCnt)
{
  if (t0 >= maxCnt)  // wave-uniform: a padding half (the tile's steps are rounded up to the ring), nothing to add
    return acc;
  double pa[PL_HP], pb[PL_HP];
#pragma unroll
  for (int u = 0; u < PL_HP; u++) {
    pa[u] = piTile[(cnt > t0 + 2 * u) ? (h.r[u] & 0xFFFFu) : zero];
    pb[u] = piTile[(cnt > t0 + 2 * u + 1) ? (h.r[u] >> 16) : zero];
  }
#pragma unroll
  for (int u = 0; u < PL_HP; u++) {
    acc = acc + pa[u] * h.e[u].x;
    acc = acc + pb[u] * h.e[u].y;
  }
  return acc;
}

// countCols as in k_price_sell; the launch has min(256, cdiv(jdsWindows, 4)) workgroups and
// jdsTileRows * 8 + 16 + 4 * 256 * 9 + 16 * 12 + 64 bytes of dynamic LDS
__global__ void __launch_bounds__(PL_THREADS) k_price_lds(Dev D, int countCols)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  extern __shared__ __attribute__((aligned(16))) unsigned char plSmem[];
  constexpr int NW = PL_THREADS / 64, NQ = PL_THREADS / 256, RING = 4 * PL_HP;
  // (read now: behind the memory-clobbering barriers below these loads would sit, unhidden, in front of the write-out)
  const double zeroTolerance = c->zeroTolerance, dualT = -c->dualTolerance, acceptablePivot = c->acceptablePivot;
  const int tileRows = D.jdsTileRows, T = D.jdsTiles;
  const int tileBytes = tileRows * 8 + 16;  // + the permanent zero behind the tile
  const unsigned zero = (unsigned)tileRows;
  const double *piTile = (const double *)plSmem;
  double *shAlpha = (double *)(plSmem + tileBytes);
  unsigned char *shFlag = (unsigned char *)(shAlpha + NQ * 256);
  int *shCnt = (int *)(shFlag + NQ * 256);
  double *shMin = (double *)(shCnt + NW);
  double *shBytes = shMin + NW;
  const int tid = threadIdx.x;
  const unsigned lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int quad = wv >> 2;
  const int nChunks = (tileRows * 8) >> 10;  // 1 KB wave-chunks per tile (jdsTileRows is a multiple of 128)
  const unsigned ldsBase = (unsigned)(size_t)(pl_lds_byte *)plSmem;
  if (tid == 0)
    *(double *)(plSmem + (size_t)tileRows * 8) = 0.0;
  // a workgroup takes four windows at a time (one per wave quad); more than 4 * gridDim.x windows: further rounds
  for (int window0 = (int)blockIdx.x; window0 < D.jdsWindows; window0 += 4 * (int)gridDim.x) {
    const int window = window0 + (int)gridDim.x * quad;
    const bool live = window < D.jdsWindows;
    const int slice = live ? window * 4 + (wv & 3) : 0;
    // per-tile lane metadata, fetched two tiles ahead: the load is then older than every ring load in flight when its
    // value is first needed (a wait for a YOUNGER load would drain the ring: vmcnt counts in order)
    const unsigned char *metaCnt = D.jdsCnt + (size_t)slice * T * 64 + lane, *metaSrc = D.jdsSrc + (size_t)slice * T * 64 + lane;
    int cntCur = metaCnt[0];
    int rawCnt1 = metaCnt[min(1, T - 1) * 64], rawSrc1 = metaSrc[min(1, T - 1) * 64];
    unsigned off = (unsigned)D.jdsSegStart[slice];
    if (!live)
      cntCur = 0;
    off = __builtin_amdgcn_readfirstlane(off);
    int maxCur = __builtin_amdgcn_readfirstlane(cntCur);
    PlHalf H0, H1;
    plLoadHalf(H0, D.jdsRowPair, D.jdsElemPair, cntCur, 0, off, lane);
    plLoadHalf(H1, D.jdsRowPair, D.jdsElemPair, cntCur, 2 * PL_HP, off, lane);
    // what the write-out needs of this lane's home column, requested now (a chain of three dependent loads -- home position /
    // column key -> status -> reduced cost -- at the tail of the kernel would add its full latency to the launch)
    const int homePos = live ? (int)D.jdsHome[(size_t)slice * 64 + lane] : (int)lane;
    const int j = live ? D.jdsCol[(size_t)slice * 64 + lane] : -1;
    const int wanted = j >= 0 ? (D.status[j] & 3) - 1 : 0;
    const double djHome = j >= 0 ? D.dj[j] : 0.0;
    const int lenHome = j >= 0 ? D.sellLen[(size_t)slice * 64 + lane] : 0;
    double acc = 0.0;
    for (int tau = 0; tau < T; tau++) {
      // ---- pi tile tau into LDS
      if (tau > 0 || window0 != (int)blockIdx.x)
        plBarrier();  // every wave is done reading the previous tile (or the staging area of the previous round)
      {
        const double *src = D.piNeg + (size_t)tau * tileRows;  // (piNeg is zero-padded to jdsTiles * jdsTileRows entries)
        for (int ch = wv; ch < nChunks; ch += NW)
          plGlds16(src + (size_t)ch * 128 + lane * 2, __builtin_amdgcn_readfirstlane(ldsBase + (unsigned)ch * 1024u));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      plBarrier();
      const int rawCnt2 = metaCnt[min(tau + 2, T - 1) * 64], rawSrc2 = metaSrc[min(tau + 2, T - 1) * 64];
      const bool haveNext = live && tau + 1 < T;
      const int cntNext = haveNext ? rawCnt1 : 0, srcNext = haveNext ? rawSrc1 : (int)lane;
      // the ring holds steps t0 .. t0 + RING - 1 of this tile; the tile's step count is padded to a multiple of RING
      // (steps with no active lane), so the roles of H0 / H1 are the same at every tile boundary
      const int maxPad = max(RING, (maxCur + RING - 1) / RING * RING);
      for (int t0 = 0; t0 < maxPad; t0 += RING) {
        const bool more = t0 + RING < maxPad;  // wave-uniform
        const int cntL = more ? cntCur : cntNext, tL = more ? t0 + RING : 0;
        // (the scheduling barriers pin the issue order consume / refill / consume / refill: the loads of a refill stay
        // in flight while the other half is consumed)
        acc = plConsumeHalf(H0, piTile, cntCur, t0, acc, zero, maxCur);
        __builtin_amdgcn_sched_barrier(0);
        plLoadHalf(H0, D.jdsRowPair, D.jdsElemPair, cntL, tL, off, lane);
        __builtin_amdgcn_sched_barrier(0);
        acc = plConsumeHalf(H1, piTile, cntCur, t0 + 2 * PL_HP, acc, zero, maxCur);
        __builtin_amdgcn_sched_barrier(0);
        plLoadHalf(H1, D.jdsRowPair, D.jdsElemPair, cntL, tL + 2 * PL_HP, off, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tau + 1 < T) {
        acc = __shfl(acc, srcNext);
        cntCur = cntNext;
        maxCur = __builtin_amdgcn_readfirstlane(cntNext);
        rawCnt1 = rawCnt2;
        rawSrc1 = rawSrc2;
      }
    }
    // ---- back to the home order of the slice; fused first ratio pass (ClpPackedMatrix.cpp:1799-1993) as in priceSellBody
    double value = __shfl(acc, homePos);
    int flag = 0;
    double ratio = 1.0e31, bytes = 0.0;
    if (j >= 0) {
      if (wanted) {
        // SURVEY 8d's B_col for the scanned column: 12 B per entry + colStart; + 20 per emitted nonzero
        bytes = 12.0 * (double)lenHome + 4.0;
        if (fabs(value) > zeroTolerance) {
          bytes += 20.0;
          if (wanted > 0) {
            const double mult = (wanted == 1) ? -1.0 : 1.0;
            const double alpha = value * mult;
            if (alpha > 0.0) {
              const double oldValue = djHome * mult;
              const double v2 = oldValue - 1.0e15 * alpha;
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
      } else {
        value = 0.0;  // basic / fixed columns are not priced (their sums are computed and dropped)
      }
    }
    const int wtid = tid & 255;  // position inside the window's four waves
    shFlag[quad * 256 + wtid] = 0xFF;
    plBarrier();
    const int j0 = D.firstColumn + (D.sellWinBase + window) * PRICE_BLOCK;
    if (j >= 0) {
      shAlpha[quad * 256 + (j - j0)] = value;
      shFlag[quad * 256 + (j - j0)] = (unsigned char)flag;
    }
    for (int o = 32; o > 0; o >>= 1) {
      ratio = fmin(ratio, __shfl_xor(ratio, o));
      bytes += __shfl_xor(bytes, o);
    }
    const int wcount = (int)__popcll(__ballot(flag != 0));
    if (lane == 0) {
      shCnt[wv] = wcount;
      shMin[wv] = ratio;
      shBytes[wv] = bytes;
    }
    plBarrier();
    if (live) {
      const unsigned char f = shFlag[quad * 256 + wtid];
      if (f != 0xFF) {
        D.alphaCol[j0 + wtid] = shAlpha[quad * 256 + wtid];
        D.candFlag[D.m + j0 + wtid] = f;
      }
      if (wtid == 0) {
        const int q4 = quad * 4;
        const int total = shCnt[q4] + shCnt[q4 + 1] + shCnt[q4 + 2] + shCnt[q4 + 3];
        if (total && countCols)
          atomicAdd(&D.blockCount[((D.m + PRICE_BLOCK - 1) / PRICE_BLOCK) + D.sellWinBase + window], total);
        D.sellMin[window] = fmin(fmin(shMin[q4], shMin[q4 + 1]), fmin(shMin[q4 + 2], shMin[q4 + 3]));
        D.sellBytes[window] = ((shBytes[q4] + shBytes[q4 + 1]) + shBytes[q4 + 2]) + shBytes[q4 + 3];
      }
    }
  }
  if (blockIdx.x == gridDim.x - 1 && wv == NW - 1) {
    // how many pivots had a dense pi: the host chooses between this kernel and k_price_sell (+ by-row form) by it.  (Counted by the
    // wave most likely to have had no window, after its work: at the head of workgroup 0 these loads delayed the whole launch.)
    int pop = 0;
    for (int w = (int)lane; w < ((D.m + 63) >> 6); w += 64)
      pop += __popcll(D.piBits[w]);
    for (int o = 32; o > 0; o >>= 1)
      pop += __shfl_xor(pop, o);
    if (lane == 0) {
      if (12LL * pop >= (long long)D.m)
        D.ctrl->statDensePi += 1.0;
      D.ctrl->lastPriceByRow = 0;
    }
  }
}

__global__ void __launch_bounds__(256) k_price_sell(Dev D, int variant, int countCols = 0, int nSellBlocks = 1 << 30, int nColBlocks = 1 << 30,
                                                    int rowMax = 0, int fullRows = 0, int dbg = 0)
{
  if (D.ctrl->state != RUN)
    return;
  // dbg & 32: the launch prices the long columns only (the SELL windows were priced by k_price_lds)
  if ((dbg & 32) && (int)blockIdx.x < nSellBlocks)
    return;
  bool byRow = false;
  if (rowMax > 0 || (blockIdx.x == 0 && !(dbg & 32))) {
    const int pop = piPopcount256(D);
    byRow = rowMax > 0 && pop <= rowMax;
    if (blockIdx.x == 0 && threadIdx.x == 0 && 12LL * pop >= (long long)D.m)
      D.ctrl->statDensePi += 1.0;
  }
  if ((int)blockIdx.x >= nColBlocks) {
    if (byRow)
      priceRowScatterBody(D, (int)blockIdx.x - nColBlocks, fullRows);
    return;
  }
  if (byRow)
    return;
  if ((int)blockIdx.x >= nSellBlocks) {
    priceLongBody(D, (int)blockIdx.x - nSellBlocks, countCols);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned long long sellBits[];  // (m+63)/64 words for variants >= 2
  switch (variant) {
  case 2:
    priceSellBody<false, false, true>(D, sellBits, countCols);
    break;
  case 3:
    priceSellBody<true, false, true>(D, sellBits, countCols);
    break;
  case 4:
    priceSellBody<true, true, true>(D, sellBits, countCols);
    break;
  case 5:
    priceSellBody<false, true, true>(D, sellBits, countCols);
    break;
  case 6:
    priceSellBody<false, false, true, true>(D, sellBits, countCols, dbg);
    break;
  default:
    priceSellBody<false, false, false>(D, sellBits, countCols);
    break;
  }
}

// Row pricing for long columns (dense or few-column LPs): one lane per column cannot fill the chip
// when n < ~10^5, so here a whole wave strides one CSC column (coalesced) and reduces.  The per-column
// sum is then a fixed 64-way tree instead of the reference's sequential order: deterministic, but
// only equal to the sequential sum to rounding (used when the mean column length is >= 256).
#define WIDE_BLOCKS 4096
__global__ void __launch_bounds__(256) k_price_wide(Dev D, int denseColumns = 0, int countCols = 0)
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
      if (denseColumns) {
        // full, row-ordered columns: the row index is the position, no index stream
        for (int p = start + lane; p < end; p += 64)
          acc += D.piNeg[p - start] * D.elem[p];
      } else {
        for (int p = start + lane; p < end; p += 64)
          acc += D.piNeg[D.row[p]] * D.elem[p];
      }
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
      if (flag && countCols)
        atomicAdd(&D.blockCount[((D.m + PRICE_BLOCK - 1) / PRICE_BLOCK) + ((j - D.firstColumn) / PRICE_BLOCK)], 1);
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
__global__ void __launch_bounds__(PRICE_BLOCK) k_cand_count(Dev D, int nbRows, int recomputeRatio, int nSell)
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
    stc(&D.blockCount[blockIdx.x], total);
    stc(&D.blockMin[blockIdx.x], bmin);
    stc(&D.blockSum[blockIdx.x], 0.0);
  }
  // nSell >= 0: the last workgroup to finish also scans the block counts.  (Only pays for small
  // grids -- across ~1000 workgroups the ticket + coherent-load latency exceeds a launch.)
  if (nSell >= 0 && lastBlockDone(D.ctrl, 0))
    scanBlocksBody<true>(D, gridDim.x, 0, nSell);
}


// =============================================================================================
// Column-sharded engine (one process per GPU): every rank prices, compacts, updates and
// flip-tests only its own contiguous column range (plus the rows, which are replicated) and the ranks
// exchange what the replicated part of the pivot needs -- the reduce of ABOCA_LITE's chunks
// (src/ClpPackedMatrix.cpp:1848-1854: min upperTheta, counts summed, lists concatenated in chunk order)
// and of Abc's blocked ratio test (src/AbcSimplexDual.cpp:1623-1634), done as ONE all-gather of the
// per-rank candidate lists {sequence, alpha, dj, range} with a {count, min ratio} header: the merged
// list in rank order IS the single-GPU list, so the replicated ratio test takes bit-identical
// decisions.  (A literal all-reduce of the {theta, thru, increase, best pivot} struct per ratio-test
// pass would put ~11 dependent collectives of ~10 us each into a 150 us pivot.)  After the ratio test
// the bound flips each rank found in its range travel the same way.  Reduced costs are owned by the
// rank that owns the column; nothing between two refactorizations reads a non-owned one.
// Record layouts in doubles (integers are exact): candidates [count, minRatio | seq, alpha, dj, range]*,
// flips [count, 0 | key, movement, objective term, column start, column length]*.
// =============================================================================================
#define SHARD_HDR 2
// own-column part of the local candidate list -> send buffer (one workgroup)
__global__ void __launch_bounds__(256) k_shard_pack_cands(Dev D, int nbRows, double *send, int cap)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ int shi[17];
  int local = 0;
  for (int b = threadIdx.x; b < nbRows; b += blockDim.x)
    local += D.blockCount[b];
  const int nRow = blockSumInt(local, shi);  // (every workgroup of the launch: 196 counts)
  const int cnt = c->numberCandidates - nRow;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    send[0] = (double)cnt;
    send[1] = c->upperTheta;
    c->shardRowCands = nRow;
  }
  // (the records over the whole launch: one workgroup took 0.2 ms for the 10^5 candidates of a mature pivot)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt && i < cap; i += gridDim.x * blockDim.x) {
    double *r = send + SHARD_HDR + 4 * (size_t)i;
    r[0] = (double)D.candSeq[nRow + i];
    r[1] = D.candAlpha[nRow + i];
    r[2] = D.candDj[nRow + i];
    r[3] = D.candRange[nRow + i];
  }
}
// every rank's list behind the (replicated) row candidates, in rank order; totals and the global min ratio
__global__ void __launch_bounds__(256) k_shard_merge_cands(Dev D, const double *recv, int nranks, int cap)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const size_t stride = SHARD_HDR + 4 * (size_t)cap;
  const int nRow = c->shardRowCands;
  int base = nRow, total = nRow;
  bool overflow = false;
  double vmin = 1.0e31;
  const int me = blockIdx.y;  // the rank whose records this workgroup row copies
  for (int r = 0; r < nranks; r++) {
    const int cnt = (int)recv[r * stride];
    if (cnt > cap)
      overflow = true;
    if (r < me)
      base += cnt;
    total += cnt;
    vmin = fmin(vmin, recv[r * stride + 1]);
  }
  if (overflow) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
      c->state = EXIT_SHARD_OVERFLOW;
    return;
  }
  const int cnt = (int)recv[me * stride];
  const double *src = recv + me * stride + SHARD_HDR;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
    const int seq = (int)src[4 * (size_t)i];
    D.candSeq[base + i] = seq;
    D.candAlpha[base + i] = src[4 * (size_t)i + 1];
    D.candDj[base + i] = src[4 * (size_t)i + 2];
    D.candRange[base + i] = src[4 * (size_t)i + 3];
    // the ratio test reads and shifts the reduced costs of candidates: this rank's copy of a column it
    // does not own is brought up to date here (the owner's value, bit for bit)
    D.dj[seq] = src[4 * (size_t)i + 2];
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    c->numberCandidates = total;
    c->upperTheta = vmin;
  }
}
// classes / ranks of the merged list for the ratio test's working-set shortcut, exactly as k_cand_scatter
// leaves them on one GPU (there the "blocks" are 256 keys; any partition of the list in list order gives the
// same working set): 256 merged candidates per workgroup
__global__ void __launch_bounds__(PRICE_BLOCK) k_shard_classes(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int i = blockIdx.x * PRICE_BLOCK + threadIdx.x;
  const bool have = i < c->numberCandidates;
  int cls = 3;
  if (have) {
    const double tol = c->dualTolerance, alpha = D.candAlpha[i], djv = D.candDj[i];
    const double x = (alpha < 0.0) ? (djv - tol) / alpha : (djv + tol) / alpha;
    const double theta0 = fmax(10.0 * c->upperTheta, 1.0e-7);
    cls = (x <= theta0 * 8.0) ? 0 : ((x <= theta0 * 256.0) ? 1 : ((x <= theta0 * 16384.0) ? 2 : 3));
    D.candLive[i] = (unsigned char)cls;
  }
  __shared__ int shc[PRICE_BLOCK / 64][3];
  int before[3];
  const int lane = threadIdx.x & 63, wvi = threadIdx.x >> 6;
  for (int j = 0; j < 3; j++) {
    unsigned long long mk = __ballot(cls == j);
    before[j] = (int)__popcll(mk & ((1ull << lane) - 1ull));
    if (lane == 0)
      shc[wvi][j] = (int)__popcll(mk);
  }
  __syncthreads();
  if (have) {
    for (int j = 0; j < 3; j++)
      for (int w = 0; w < wvi; w++)
        before[j] += shc[w][j];
    const int r0 = before[0], r1 = r0 + before[1], r2 = r1 + before[2];
    D.candRk[i] = r0 | (r1 << 10) | (r2 << 20);
    D.candBlk[i] = (int)blockIdx.x;
  }
  if (threadIdx.x < 3) {
    int s = 0;
    for (int w = 0; w < PRICE_BLOCK / 64; w++)
      s += shc[w][threadIdx.x];
    D.classBlock[3 * blockIdx.x + threadIdx.x] = s;
  }
}
// this rank's flip records -> send buffer (one workgroup)
__global__ void __launch_bounds__(256) k_shard_pack_flips(Dev D, double *send, int cap, int listCap)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int cnt = c->flipAppend;
  if (threadIdx.x == 0) {
    // (more flips than the append buffer holds have no records at all: reported as overflow)
    send[0] = (double)(cnt > listCap ? cap + 1 : cnt);
    send[1] = 0.0;
  }
  for (int i = threadIdx.x; i < cnt && i < cap && i < listCap; i += blockDim.x) {
    double *r = send + SHARD_HDR + 5 * (size_t)i;
    r[0] = (double)D.flipKey[i];
    r[1] = D.flipRecMv[i];
    r[2] = D.flipRecObj[i];
    r[3] = (double)D.flipRecStart[i];
    r[4] = (double)D.flipRecLen[i];
  }
}
// merged flip list: the row flips (every rank found the same ones) from this rank's own records, the
// column flips of every rank in rank order (one workgroup; k_flip_apply2 orders by key afterwards)
__global__ void __launch_bounds__(256) k_shard_merge_flips(Dev D, const double *recv, int rank, int nranks, int cap, int listCap)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ int shi[17];
  __shared__ int s_out;
  const size_t stride = SHARD_HDR + 5 * (size_t)cap;
  bool overflow = false;
  for (int r = 0; r < nranks; r++)
    if ((int)recv[r * stride] > cap)
      overflow = true;
  if (threadIdx.x == 0)
    s_out = 0;
  __syncthreads();
  if (!overflow) {
    for (int pass = 0; pass <= nranks; pass++) {
      // pass 0: own row flips; pass 1 + r: column flips of rank r
      const int r = pass == 0 ? rank : pass - 1;
      const int cnt = (int)recv[r * stride];
      const double *src = recv + r * stride + SHARD_HDR;
      for (int i0 = 0; i0 < cnt; i0 += blockDim.x) {
        const int i = i0 + threadIdx.x;
        int key = -1;
        if (i < cnt)
          key = (int)src[5 * (size_t)i];
        const int take = key >= 0 && ((pass == 0) == (key < D.m));
        int tot;
        const int rk = blockRank(take, tot, shi);
        const int o = s_out + rk;
        if (take && o < listCap) {
          D.flipKey[o] = key;
          D.flipRecMv[o] = src[5 * (size_t)i + 1];
          D.flipRecObj[o] = src[5 * (size_t)i + 2];
          D.flipRecStart[o] = (int)src[5 * (size_t)i + 3];
          D.flipRecLen[o] = (int)src[5 * (size_t)i + 4];
        }
        __syncthreads();
        if (threadIdx.x == 0)
          s_out += tot;
        __syncthreads();
      }
    }
    if (s_out > listCap)
      overflow = true;
  }
  if (threadIdx.x == 0) {
    if (overflow)
      c->state = EXIT_SHARD_OVERFLOW;
    else
      c->flipAppend = s_out;
  }
}

// =============================================================================================
// Fused stages: fewer grid-wide dependencies per pivot (each launch costs ~4 us on 256 CUs)
// =============================================================================================

// CHUZR final selection + the analytic front end of the BTRAN in one workgroup
// COHERENT: run by the last workgroup of k_chuzr_scan (the per-block winners are read with ldc)
template <bool COHERENT> __device__ inline void chuzrFinalBody(const Dev &D, int nblocks, int wide, ChzStage *S)
{
  Ctrl *c = D.ctrl;
  __shared__ double shv[4];
  __shared__ int shk[4], shr[4];
  __shared__ int s_ok, s_again;
  double best = 0.0;
  int bestKey = -1, bestRow = -1;
  int used = (c->chuzrNumber + 256 * CHZ_ITEMS - 1) / (256 * CHZ_ITEMS);
  if (used > nblocks)
    used = nblocks;
  // partial scan: the workgroup the numberWanted-th entry above the tolerance falls into (k_chuzr_scan); spans before it count whole,
  // that one is walked again in order, the ones behind it not at all
  __shared__ int s_cut, s_before, s_special;
  const bool partial = c->pivotRule != 0 && c->chuzrWanted <= c->chuzrNumber && c->presetRowPlus1 <= 0;
  if (partial) {
    __shared__ int s_wave[4], s_wspec[4];
    if (used <= (int)blockDim.x) {
      // one count per thread, an inclusive scan over the workgroup, the first span whose running total reaches numberWanted
      const int b = threadIdx.x, lane_ = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
      const int v = b < used ? (COHERENT ? ldc(&D.chzCnt[b]) : D.chzCnt[b]) : 0;
      const int cnt = v & ((1 << 30) - 1);
      int incl = cnt;
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane_ >= o)
          incl += up;
      }
      if (lane_ == 63)
        s_wave[wv_] = incl;
      if (threadIdx.x == 0) {
        s_cut = used;
        s_before = 0;
        s_special = 0;
      }
      __syncthreads();
      int offset = 0;
      for (int w = 0; w < wv_; w++)
        offset += s_wave[w];
      incl += offset;
      const int wanted = c->chuzrWanted;
      const bool reached = b < used && incl >= wanted && incl - cnt < wanted;  // exactly one span (counts are >= 0), if any
      if (reached) {
        s_cut = b;
        s_before = incl - cnt;
      }
      __syncthreads();
      const int cut = s_cut;
      // the first span up to the cut that holds one of the two exceptions: the spans before it still count whole, the ordered walk
      // starts there
      const bool isSpec = b < used && b <= cut && (v >> 30);
      const unsigned long long sp = __ballot(isSpec);
      if (lane_ == 0)
        s_wspec[wv_] = sp ? wv_ * 64 + (__ffsll((unsigned long long)sp) - 1) : (1 << 30);
      __syncthreads();
      const int first = min(min(s_wspec[0], s_wspec[1]), min(s_wspec[2], s_wspec[3]));
      if (b == first) {
        s_special = 1;
        s_cut = first;  // "used" below: the spans before the first exception
        s_before = incl - cnt;
      }
    } else if (threadIdx.x == 0) {
      int before = 0, cut = used, special = 0;
      for (int b = 0; b < used; b++) {
        const int v = COHERENT ? ldc(&D.chzCnt[b]) : D.chzCnt[b];
        if (v >> 30) {  // the ordered walk starts at this span
          special = 1;
          cut = b;
          break;
        }
        const int cnt = v & ((1 << 30) - 1);
        if (before + cnt >= c->chuzrWanted) {
          cut = b;
          break;
        }
        before += cnt;
      }
      s_cut = cut;
      s_before = before;
      s_special = special;
    }
    __syncthreads();
    used = s_cut;  // (uniform)
  }
  for (int b = threadIdx.x; b < used; b += blockDim.x) {
    double ov = COHERENT ? ldc(&D.chzBest[b]) : D.chzBest[b];
    int ok = COHERENT ? ldc(&D.chzKey[b]) : D.chzKey[b];
    if (ok >= 0 && (bestKey < 0 || ov > best || (ov == best && ok < bestKey))) {
      best = ov;
      bestKey = ok;
      bestRow = COHERENT ? ldc(&D.chzRow[b]) : D.chzRow[b];
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
    shv[0] = best;
    shk[0] = bestKey;
    shr[0] = bestRow;
  }
  __syncthreads();
  if (partial) {
    // (all threads) the span of the cut in order, continuing from the best of the spans before it -- or, when a flagged candidate or
    // the last pivot row sits in the scanned part, the whole list in order with the reference's own statements
    const int cutBlocks = (c->chuzrNumber + 256 * CHZ_ITEMS - 1) / (256 * CHZ_ITEMS);
    if (s_special) {
      if (threadIdx.x == 0)
        c->chuzrOrdered++;
      chuzrOrderedScan(D, S, c->chuzrTolerance, c->chuzrNumber, c->chuzrStart, c->chuzrLast, c->chuzrWanted - s_before, s_cut * (256 * CHZ_ITEMS),
                       2147483647, shv[0], shk[0], shr[0]);
    }
    else if (s_cut < cutBlocks && s_cut < nblocks)
      chuzrOrderedScan(D, S, c->chuzrTolerance, c->chuzrNumber, c->chuzrStart, c->chuzrLast, c->chuzrWanted - s_before, s_cut * (256 * CHZ_ITEMS),
                       (s_cut + 1) * (256 * CHZ_ITEMS), shv[0], shk[0], shr[0]);
    if (s_special || (s_cut < cutBlocks && s_cut < nblocks)) {
      __syncthreads();
      if (threadIdx.x == 0) {
        shv[0] = S->best;
        shk[0] = S->bestKey;
        shr[0] = S->bestRow;
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    best = shv[0];
    bestKey = shk[0];
    bestRow = shr[0];
    // "won't line up with checkPrimalSolution - do again" (src/ClpDualRowSteepest.cpp:338-346): nothing chosen under the changed
    // tolerance -> the whole call once more with largestDualError_ 0
    s_again = bestRow < 0 && c->pivotRule != 0 && c->chuzrTolChanged && c->presetRowPlus1 <= 0;
    if (s_again) {
      c->chuzrRecalls++;
      chuzrPreBody(D, true);
    }
    shr[0] = bestRow;
  }
  __syncthreads();
  if (s_again) {
    chuzrOrderedScan(D, S, c->chuzrTolerance, c->chuzrNumber, c->chuzrStart, c->chuzrLast, c->chuzrWanted);
    if (threadIdx.x == 0)
      shr[0] = S->bestRow;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int chosen = shr[0];
    if (c->presetRowPlus1 > 0) {  // dualRow's free-first entry (host, see k_chuzr_pre)
      chosen = c->presetRowPlus1 - 1;
      c->presetRowPlus1 = 0;
    }
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
      c->flipAppend = 0;
      c->flipHotCount = 0;
      c->flipDense = 0;
      c->appendGo = 0;
      c->objectiveChange = 0.0;
    }
  }
  __syncthreads();
  if (!s_ok || D.luMode)
    return;  // (LU mode: the BTRAN is the k_lu_* chain)
  // BTRAN t-vector for dir*e_p (t = c_K - A_SK^T y_S with c = dir*e_p, y_S = -c_S), kept as a short list: t has one nonzero when a
  // structural leaves, and one per basic entry of the leaving slack's row otherwise
  const double dir = (double)c->directionOut;
  const int seqOut = c->sequenceOut;
  if (wide) {
    // long rows (dense LPs): t as a dense vector by column-slot for the split GEMV^T
    // (k_gemvT_partial2); the list form below would be as long as the nucleus
    const int k = c->k;
    for (int s = threadIdx.x; s < k; s += blockDim.x)
      D.slotA[s] = 0.0;
    __syncthreads();
    if (seqOut < D.n) {
      if (threadIdx.x == 0)
        D.slotA[D.slotOfCol[seqOut]] = dir;
    } else {
      const int rOut = seqOut - D.n;
      const double y = dir * -1.0;
      const int s = D.rowStart[rOut], cnt = D.basicCount[rOut];
      for (int q = threadIdx.x; q < cnt; q += blockDim.x)
        D.slotA[D.cslot[s + q]] = 0.0 - y * D.relem[s + q];
    }
    if (threadIdx.x == 0)
      c->tCount = -1;
    return;
  }
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
      int sc = D.cslot[s + q];
      int rank = 0;
      for (int q2 = 0; q2 < cnt; q2++) {
        int sc2 = D.cslot[s + q2];
        rank += (sc2 < sc);
      }
      D.tIndex[rank] = sc;
      D.tValue[rank] = 0.0 - y * D.relem[s + q];
    }
    if (threadIdx.x == 0)
      c->tCount = cnt;
  }
}
__global__ void __launch_bounds__(256) k_chuzr_final_btran(Dev D, int nblocks, int wide = 0)
{
  if (D.ctrl->state != RUN)
    return;
  __shared__ ChzStage stage;
  chuzrFinalBody<false>(D, nblocks, wide, &stage);
}

// append scan with absolute offsets + the scalar tail of the primal / flip updates
__device__ inline void scanTailBody(const Dev &D, int nbCount, int nbSum, int which, int alphaTest, int parity)
{
  Ctrl *c = D.ctrl;
  __shared__ int shi[17];
  __shared__ double shd[16];
  __shared__ int s_base;
  const bool active = !(which == 1 && c->numberFlips == 0);
  // request this thread's first count and partial sum right away (slow coherent loads), use later
  const int cnt0 = (active && (int)threadIdx.x < nbCount) ? ldc(&D.blockCount[threadIdx.x]) : 0;
  const double sum0 = (active && (int)threadIdx.x < nbSum) ? ldc(&D.blockSum[threadIdx.x]) : 0.0;
  if (threadIdx.x == 0)
    s_base = c->numberInfeasible;
  __syncthreads();
  if (active) {
    for (int b0 = 0; b0 < nbCount; b0 += blockDim.x) {
      int b = b0 + threadIdx.x;
      int cnt = b0 == 0 ? cnt0 : ((b < nbCount) ? ldc(&D.blockCount[b]) : 0);
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
        (which == 1 ? D.blockOffset1 : D.blockOffset)[b] = base + v - cnt;
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
  if (active) {
    s += sum0;
    for (int b = threadIdx.x + blockDim.x; b < nbSum; b += blockDim.x)
      s += ldc(&D.blockSum[b]);
  }
  s = blockSum(s, shd);
  if (threadIdx.x != 0)
    return;
  if (active) {
    c->objectiveChange += s;
    if (which == 1)
      c->numberAppend1 = s_base - c->numberInfeasible;
    else {
      c->numberAppend = s_base - c->numberInfeasible;
      c->appendGo = 1;
    }
    c->numberInfeasible = s_base;
  } else {
    c->numberAppend1 = 0;
  }
  if (which == 1) {
    if (alphaTest) {
      // btran/ftran alpha accuracy test (whileIterating :1447-1501)
      double alphaNew = ldc(&c->tailAlpha);
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
      c->valueOut = ldc(&c->tailValueOut);
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
    if (D.luMode)
      return;  // (the eta file needs only w, alpha and the pivot row)
    int seqIn = c->sequenceIn, seqOut = c->sequenceOut;
    int inStruct = seqIn < D.n, outStruct = seqOut < D.n;
    c->updateCase = outStruct ? (inStruct ? 0 : 2) : (inStruct ? 1 : 3);
    c->slotColOut = outStruct ? D.slotOfCol[seqOut] : -1;
    c->rowOfSlackOut = outStruct ? -1 : (seqOut - D.n);
    c->slotRowIn = inStruct ? -1 : D.slotOfRow[seqIn - D.n];
    // the basis update of this pivot may go ahead (it runs beside the primal update, housekeeping
    // and the next CHUZR, so it must not look at the live state or the live k)
    if (parity >= 0) {
      c->updK = c->k;
      c->updGo[parity] = 1;
    }
  }
}
__global__ void __launch_bounds__(256) k_house(Dev D)
{
  if (D.ctrl->state != RUN)
    return;
  houseBody(D);
}

// row/column fix-up of the nucleus update (k_rank1_fix + k_rank1_fix2) and housekeeping, one workgroup
// fix-ups that complete the rank-1 update of the nucleus inverse for the four pivot types (see
// k_rank1); runs after k_rank1 on the basis-update branch
__device__ inline void minvFixBody(const Dev &D, int parity)
{
  Ctrl *c = D.ctrl;
  if (!c->updGo[parity])
    return;
  const int k = c->updK;
  const int ucase = c->updateCase;
  const double alpha = c->alpha;
  const double dir = (double)c->directionOut;
  const int a = c->slotColOut, b = c->slotRowIn, last = k - 1;
  for (int s = threadIdx.x; s < k; s += blockDim.x) {
    double slotFs = dir * D.rhoSlot[s] / alpha;  // g by row-slot
    double slotEs = D.slotC[s];                  // w by col-slot (== w[slotPos[s]])
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
  if (threadIdx.x == 0)
    c->updGo[parity] = 0;
}

__global__ void __launch_bounds__(256) k_minv_fix(Dev D, int parity)
{
  minvFixBody(D, parity);
}

__global__ void __launch_bounds__(256) k_fix_house(Dev D, int parity, int doFix, int wide = 0)
{
  const Ctrl *c = D.ctrl;
  if (blockIdx.x == 1) {
    // single-stream form: the fix-ups of the basis update (after k_rank1) run in their own
    // workgroup beside the housekeeping; they use the go flag and k saved by the FTRAN tail, not
    // the live values workgroup 0 is changing
    if (doFix)
      minvFixBody(D, parity);
    return;
  }
  if (blockIdx.x > 1) {
    // (gated by appendGo, not by state: workgroup 0 may raise an exit for the NEXT pivot while
    // these are still starting, and this pivot's entries must reach the list regardless)
    if (!c->appendGo)
      return;
    // new entries of the infeasibility list (ClpDualRowSteepest::updatePrimalSolution :630-744 adds
    // them as it goes: first those of the flip update, then those of the main update, each in
    // position order); the offsets come from the two scans
    __shared__ int shi[17];
    const int blk = blockIdx.x - 2;
    const int p = blk * 256 + threadIdx.x;  // list blocks are 256 positions whatever the launch width
    int total;
    if (c->numberFlips != 0 && c->numberAppend1 != 0) {
      int flag = (threadIdx.x < 256 && p < D.m) ? D.appendFlag1[p] : 0;
      int rank = blockRank(flag, total, shi);
      if (flag)
        D.infIndex[D.blockOffset1[blk] + rank] = p;
    }
    if (c->numberAppend != 0) {
      int flag = (threadIdx.x < 256 && p < D.m) ? D.appendFlag[p] : 0;
      int rank = blockRank(flag, total, shi);
      if (flag)
        D.infIndex[D.blockOffset[blk] + rank] = p;
    }
    return;
  }
  if (c->state != RUN)
    return;
  houseBody(D, wide);
  // head of the next pivot (only if this one ended normally and no exit was raised)
  // (freeHold: the host first decides whether the next pivot's row comes from dualRow's free-first entry)
  if (threadIdx.x == 0 && D.ctrl->state == RUN && !D.ctrl->freeHold)
    chuzrPreBody(D);
}


// =============================================================================================
// Flips are known as soon as theta is (they do not depend on the FTRAN), so the flip right-hand
// side joins the entering column and the DSE vector in ONE three-vector FTRAN sweep over Minv.
// =============================================================================================

// dual update + flip detection only (the weights need the FTRAN and come later)
#define FLIP_LIST_CAP 4096
#define FLIP_SLOTS 16  // contributions a row keeps individually while the flip right-hand side is assembled
__global__ void __launch_bounds__(PRICE_BLOCK) k_dj_flags(Dev D, int nbRows, int listCap = FLIP_LIST_CAP, int scatterFlips = 0, int slotCap = FLIP_SLOTS)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const double theta = c->theta;
  const double tolerance = c->dualTolerance + fmin(1.0e-2, c->largestDualError);
  const int seqIn = c->sequenceIn;
  if (blockIdx.x == gridDim.x - 1) {
    // the extra workgroup: unpack the entering column (ClpSimplex::unpackPacked :3439-3495), by row
    // and by nucleus row-slot for the FTRAN sweep -- beside the dual update instead of inside the
    // single-workgroup ratio test
    if (seqIn >= D.n) {
      if (threadIdx.x == 0) {
        const int r = seqIn - D.n;
        D.vecV1[r] = -1.0;
        const int sr = D.slotOfRow[r];
        if (sr >= 0)
          D.slotV1[sr] = -1.0;
      }
    } else if (seqIn >= 0) {
      for (int p = D.colStart[seqIn] + threadIdx.x; p < D.colStart[seqIn + 1]; p += blockDim.x) {
        const int r = D.row[p];
        const double e = D.elem[p];
        D.vecV1[r] = e;
        const int sr = D.slotOfRow[r];
        if (sr >= 0)
          D.slotV1[sr] = e;
      }
    }
    return;
  }
  int flag = 0, key = -1, seqF = -1;
  unsigned char stF = 0;
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
          stF = D.status[seq];
        }
      }
      D.candFlag[i] = (unsigned char)flag;
      key = i;
      seqF = seq;
    }
  } else {
    int j = D.firstColumn + ((int)blockIdx.x - nbRows) * PRICE_BLOCK + threadIdx.x;
    if (j < D.lastColumn) {
      double alphaI = D.alphaCol[j];
      if (alphaI != 0.0 && j != seqIn) {
        int iStatus = (D.status[j] & 3) - 1;
        // (the general column loop of updateDualsInDual, :2596-2651, has no case for a superbasic variable: its dj stays)
        if (c->freeCount > 0 && (D.status[j] & 7) == ST_SUPER)
          iStatus = 0;
        if (iStatus) {
          double value = D.dj[j] - theta * alphaI;
          D.dj[j] = value;
          double mult = (iStatus == 1) ? -1.0 : ((iStatus == 2) ? 1.0 : -1.0);
          value *= mult;
          if (value < -tolerance && iStatus > 0)
            flag = 1;
          stF = D.status[j];
        }
      }
      D.candFlag[D.m + j] = (unsigned char)flag;
      key = D.m + j;
      seqF = j;
    }
  }
  // flips are few: append them in arrival order (one atomic per wave that has any); k_flip_apply2
  // puts them in list order (rows, then columns ascending) before anything depends on the order
  unsigned long long mk = __ballot(flag);
  if (mk) {
    const int lane = threadIdx.x & 63;
    // the flip's movement, objective term and column extent (matrix_->add per flipped column,
    // ClpSimplexDual.cpp:2586 / ClpPackedMatrix.cpp:4874)
    double mv = 0.0, ob = 0.0;
    int start = 0, len = 1;
    if (flag) {
      const int iStatus = (stF & 3) - 1;
      const double mult = (iStatus == 1) ? -1.0 : 1.0;
      if (seqF >= D.n) {
        mv = mult * (D.lower[seqF] - D.upper[seqF]);
        ob = 0.0 - mv * D.cost[seqF];
      } else {
        mv = mult * (D.upper[seqF] - D.lower[seqF]);
        ob = mv * D.cost[seqF];
        start = D.colStart[seqF];
        len = D.colStart[seqF + 1] - start;
      }
    }
    int base = 0;
    if (lane == 0)
      base = atomicAdd(&c->flipAppend, (int)__popcll(mk));
    base = __shfl(base, 0);
    if (flag) {
      int o = base + __popcll(mk & ((1ull << lane) - 1ull));
      if (o < listCap) {
        // complete records: k_flip_apply2 starts from them instead of three more rounds of dependent loads
        D.flipKey[o] = key;
        D.flipRecMv[o] = mv;
        D.flipRecObj[o] = ob;
        D.flipRecStart[o] = start;
        D.flipRecLen[o] = len;
      }
    }
    if (scatterFlips) {
      // the flip right-hand side is assembled here, by the wave that found the flip: every entry of the
      // flipped column draws a ticket from its row's counter (integer, order independent) and leaves
      // (flip key, movement * element) in that slot; k_flip_apply2's row workgroups add a row's
      // contributions in key order = flip order, the sum of the reference's sequential loop
      unsigned long long rem = mk;
      while (rem) {
        const int b = __ffsll((long long)rem) - 1;
        rem &= rem - 1ull;
        const int keyB = __shfl(key, b), stB = __shfl(start, b), lnB = __shfl(len, b);
        const double mvB = __shfl(mv, b);
        if (keyB < D.m) {
          if (lane == 0) {
            const int t = atomicAdd(&D.flipTouch[keyB], 1);
            if (t < slotCap) {
              D.flipRowKey[(size_t)keyB * FLIP_SLOTS + t] = keyB;
              D.flipRowVal[(size_t)keyB * FLIP_SLOTS + t] = mvB;
            } else if (t == slotCap) {
              const int h = atomicAdd(&c->flipHotCount, 1);
              if (h < FLIP_HOT_CAP)
                D.flipHot[h] = keyB;
            }
          }
        } else {
          for (int p = stB + lane; p < stB + lnB; p += 64) {
            const int r = D.row[p];
            const double v = mvB * D.elem[p];
            const int t = atomicAdd(&D.flipTouch[r], 1);
            if (t < slotCap) {
              D.flipRowKey[(size_t)r * FLIP_SLOTS + t] = keyB;
              D.flipRowVal[(size_t)r * FLIP_SLOTS + t] = v;
            } else if (t == slotCap) {
              // a row most flipped columns share (the pivot row's neighbourhood: the candidates all meet
              // supp(rho)): k_flip_apply2's first workgroup sums it from the ordered records
              const int h = atomicAdd(&c->flipHotCount, 1);
              if (h < FLIP_HOT_CAP)
                D.flipHot[h] = r;
            }
          }
        }
      }
    }
  }
}

// Flip right-hand side, all flips at once (matrix_->add per flipped column, src/ClpPackedMatrix.cpp
// :4874).  One workgroup; thread e owns one (flip, entry) pair.  Rows hit by a single flip are stored
// directly; rows hit by several flips are collected, ordered by (row, flip) and summed in flip order,
// so the result is bit-identical to the sequential loop of the reference whatever the schedule.
// sequential form (very many flips or dense columns): flips in order, entries of one column in parallel
__device__ void flipSequential(const Dev &D, int nf)
{
  Ctrl *c = D.ctrl;
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
        const int r = seq - D.n;
        const double nv = D.flipRhs[r] + movement;
        D.flipRhs[r] = nv;
        const int sr = D.slotOfRow[r];
        if (sr >= 0)
          D.flipSlot[sr] = nv;
      }
    } else {
      double movement = mult * (D.upper[seq] - D.lower[seq]);
      if (tid == 0)
        changeObj += movement * D.cost[seq];
      for (int p = D.colStart[seq] + tid; p < D.colStart[seq + 1]; p += blockDim.x) {
        const int r = D.row[p];
        const double nv = D.flipRhs[r] + movement * D.elem[p];
        D.flipRhs[r] = nv;
        const int sr = D.slotOfRow[r];
        if (sr >= 0)
          D.flipSlot[sr] = nv;
      }
    }
    __syncthreads();
  }
  if (tid == 0)
    c->objectiveChange += changeObj;
}
#define FLIP_MAX_FLIPS 1024
#define FLIP_MAX_ENTRIES 8192
#define FLIP_MAX_COLLIDE 1024
#define FLIP_HASH_BITS 14
#define FLIP_HASH_ROW 0x3fffffff
#define FLIP_HASH_MULTI 0x40000000
// dense-column mode (every column holds all m rows in order): flip right-hand side with one thread
// per row, flips in list order -- the same adds in the same order as the sequential form
__global__ void __launch_bounds__(256) k_flip_dense(Dev D)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN || !c->flipDense)
    return;
  const int nf = c->numberFlips;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= D.m)
    return;
  double acc = 0.0;
  for (int f = 0; f < nf; f++) {
    const int seq = D.flipSeq[f];
    const double mv = D.flipMv[f];
    if (seq >= D.n) {
      if (seq - D.n == r)
        acc += mv;
    } else {
      acc += mv * D.elem[D.colStart[seq] + r];
    }
  }
  D.flipRhs[r] = acc;
  const int sr = D.slotOfRow[r];
  if (sr >= 0)
    D.flipSlot[sr] = acc;
}

// sum of cnt <= W (key, value) pairs in ascending key order, from 0.0 (odd-even transposition network in registers)
template <int W>
__device__ inline double flipOrderedSum(const int *keys, const double *vals, int cnt)
{
  int kk[W];
  double vv[W];
#pragma unroll
  for (int u = 0; u < W; u++) {
    kk[u] = u < cnt ? keys[u] : 0x7fffffff;
    vv[u] = u < cnt ? vals[u] : 0.0;
  }
#pragma unroll
  for (int round = 0; round < W; round++) {
#pragma unroll
    for (int u = round & 1; u + 1 < W; u += 2) {
      if (kk[u + 1] < kk[u]) {
        int tk = kk[u];
        kk[u] = kk[u + 1];
        kk[u + 1] = tk;
        double tv = vv[u];
        vv[u] = vv[u + 1];
        vv[u + 1] = tv;
      }
    }
  }
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < W; u++)
    if (u < cnt)
      acc += vv[u];
  return acc;
}

// more contributors than slots (a row many flipped columns share): they are taken from the flip records in
// ascending key order, one selection pass per contributor (rare; nraw <= FLIP_MAX_FLIPS)
__device__ inline double flipSelectSum(const Dev &D, int r, int cnt, int nraw)
{
  double acc = 0.0;
  int lastKey = -1;
  for (int done = 0; done < cnt; done++) {
    int bestKey = 0x7fffffff;
    double bestVal = 0.0;
    for (int f = 0; f < nraw; f++) {
      const int key = D.flipKey[f];
      if (key <= lastKey || key >= bestKey)
        continue;
      if (key < D.m) {
        if (key == r) {
          bestKey = key;
          bestVal = D.flipRecMv[f];
        }
      } else {
        int lo = D.flipRecStart[f], hi = lo + D.flipRecLen[f] - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (D.row[mid] < r)
            lo = mid + 1;
          else
            hi = mid;
        }
        if (lo == hi && D.row[lo] == r) {
          bestKey = key;
          bestVal = D.flipRecMv[f] * D.elem[lo];
        }
      }
    }
    if (bestKey == 0x7fffffff)
      break;
    acc += bestVal;
    lastKey = bestKey;
  }
  return acc;
}

// rows whose flip contributions k_dj_flags scattered (option scattered): a row's <= FLIP_SLOTS (key, value)
// pairs are put in key order -- the flip order: rows first, then columns ascending -- and added, exactly
// the adds of the reference's loop over the flipped columns (matrix_->add, ClpPackedMatrix.cpp:4874)
__device__ inline void flipRowsBody(const Dev &D, int blk, int nraw, bool useScatter, int slotCap)
{
  const int r = blk * (int)blockDim.x + threadIdx.x;
  if (r >= D.m)
    return;
  const int cnt = D.flipTouch[r];
  if (!cnt)
    return;
  D.flipTouch[r] = 0;
  if (!useScatter)
    return;  // (more flips than the record buffer orders: workgroup 0 takes the sequential form)
  double acc = 0.0;
  const size_t at = (size_t)r * FLIP_SLOTS;
  if (cnt > slotCap) {
    if (D.ctrl->flipHotCount <= FLIP_HOT_CAP)
      return;  // workgroup 0 sums this row from the ordered records
    atomicAdd((unsigned long long *)&D.ctrl->dbg[13], 1ull);
    acc = flipSelectSum(D, r, cnt, nraw);
  } else if (cnt == 1) {
    acc += D.flipRowVal[at];
  } else if (cnt <= 8) {
    acc = flipOrderedSum<8>(D.flipRowKey + at, D.flipRowVal + at, cnt);
  } else if (cnt <= FLIP_SLOTS) {
    acc = flipOrderedSum<FLIP_SLOTS>(D.flipRowKey + at, D.flipRowVal + at, cnt);
  }
  D.flipRhs[r] = acc;
  const int sr = D.slotOfRow[r];
  if (sr >= 0)
    D.flipSlot[sr] = acc;
}

// workgroup 0: the flip list in order, its scalars and -- unless k_dj_flags scattered the columns --
// the flip right-hand side; workgroups 1.. (scattered form only): the rows' contributions
__global__ void __launch_bounds__(1024) k_flip_apply2(Dev D, int nbPos, int denseColumns = 0, int listCap = FLIP_LIST_CAP, int scattered = 0, int slotCap = FLIP_SLOTS)
{
  Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int tid = threadIdx.x;
  if (blockIdx.x > 0) {
    const int nrawRows = c->flipAppend;
    if (nrawRows)
      flipRowsBody(D, (int)blockIdx.x - 1, nrawRows, nrawRows <= listCap && nrawRows <= FLIP_MAX_FLIPS, slotCap);
    return;
  }
  // counters of k_ftran_scatter3's appends (position blocks) are reset here, flips or not
  for (int b = tid; b < nbPos; b += blockDim.x)
    D.blockCount[b] = 0;
  // the first record of every thread is requested together with the count (stale beyond it)
  const int key0 = D.flipKey[tid];
  const double mv0 = D.flipRecMv[tid], ob0 = D.flipRecObj[tid];
  const int st0 = D.flipRecStart[tid], len0 = D.flipRecLen[tid];
  const int nraw = c->flipAppend;
  if (nraw == 0)
    return;  // numberFlips was zeroed by CHUZR
  const bool rowsDoRhs = scattered && nraw <= listCap && nraw <= FLIP_MAX_FLIPS;
  // ---- the flip list in reference order (rows first, then columns ascending)
  __shared__ int s_seq[FLIP_LIST_CAP];
  __shared__ int shw[17];
  __shared__ double s_mv[FLIP_MAX_FLIPS], s_ob[FLIP_MAX_FLIPS];
  __shared__ int s_start[FLIP_MAX_FLIPS + 1], s_cs[FLIP_MAX_FLIPS];
  int nf;
  const bool haveRecords = nraw <= listCap;
  if (haveRecords) {
    if (tid < nraw)
      s_seq[tid] = key0;
    for (int i = tid + blockDim.x; i < nraw; i += blockDim.x)
      s_seq[i] = D.flipKey[i];
    __syncthreads();
    int myKey[FLIP_LIST_CAP / 1024], myRank[FLIP_LIST_CAP / 1024];
#pragma unroll
    for (int q = 0; q < FLIP_LIST_CAP / 1024; q++) {
      int i = tid + q * 1024;
      myKey[q] = -1;
      myRank[q] = 0;
      if (i < nraw) {
        int key = s_seq[i], rank = 0;
        for (int j = 0; j < nraw; j++)
          rank += s_seq[j] < key;  // keys are distinct
        myKey[q] = key;
        myRank[q] = rank;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FLIP_LIST_CAP / 1024; q++) {
      if (myKey[q] >= 0) {
        int seq = myKey[q] < D.m ? D.n + myKey[q] : myKey[q] - D.m;
        s_seq[myRank[q]] = seq;
        D.flipSeq[myRank[q]] = seq;
        // the record k_dj_flags wrote for this flip moves to its place in list order
        if (myRank[q] < FLIP_MAX_FLIPS && nraw <= FLIP_MAX_FLIPS) {
          const int i = tid + q * 1024;
          s_mv[myRank[q]] = q == 0 ? mv0 : D.flipRecMv[i];
          s_ob[myRank[q]] = q == 0 ? ob0 : D.flipRecObj[i];
          s_cs[myRank[q]] = q == 0 ? st0 : D.flipRecStart[i];
          s_start[myRank[q] + 1] = q == 0 ? len0 : D.flipRecLen[i];
        }
      }
    }
    nf = nraw;
  } else {
    // more flips than the append buffer holds: ordered compaction of the flags by this workgroup
    const int N = D.m + D.n;
    const int per = (N + (int)blockDim.x - 1) / (int)blockDim.x;
    const int lo = min(N, tid * per), hi = min(N, lo + per);
    int cnt = 0;
    for (int i = lo; i < hi; i++)
      cnt += D.candFlag[i] != 0 && (i < D.m || (i - D.m >= D.firstColumn && i - D.m < D.lastColumn));
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
    int base = 0, tot = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) {
      if (i < wv)
        base += shw[i];
      tot += shw[i];
    }
    int o = base + v - cnt;
    for (int i = lo; i < hi; i++)
      if (D.candFlag[i] != 0 && (i < D.m || (i - D.m >= D.firstColumn && i - D.m < D.lastColumn)))
        D.flipSeq[o++] = i < D.m ? D.n + i : i - D.m;
    nf = tot;
    __syncthreads();
    for (int f = tid; f < nf && f < FLIP_LIST_CAP; f += blockDim.x)
      s_seq[f] = D.flipSeq[f];
  }
  if (tid == 0) {
    c->numberFlips = nf;
    c->dbg[9]++;
    c->dbg[10] += nf;
  }
  __syncthreads();
  __shared__ int s_hash[1 << FLIP_HASH_BITS];  // 0 empty, else (row + 1) | MULTI
  for (int i = tid; i < (1 << FLIP_HASH_BITS); i += blockDim.x)
    s_hash[i] = 0;
  __shared__ int s_cRow[FLIP_MAX_COLLIDE], s_cFlip[FLIP_MAX_COLLIDE], s_cSorted[FLIP_MAX_COLLIDE];
  __shared__ double s_cVal[FLIP_MAX_COLLIDE];
  __shared__ int s_nCollide, s_total;
  __shared__ double shd[16];
  bool fallback = nf > FLIP_MAX_FLIPS;
  double changeObj = 0.0;
  if (!fallback) {
    // per-flip scalars: already in LDS when k_dj_flags' records were used, otherwise (flip-list overflow) from the rim
    for (int f = tid; f < nf; f += blockDim.x) {
      if (haveRecords) {
        changeObj += s_ob[f];
      } else {
        int seq = s_seq[f];
        int iStatus = (D.status[seq] & 3) - 1;
        double mult = (iStatus == 1) ? -1.0 : 1.0;
        double mv;
        int len;
        if (seq >= D.n) {
          mv = mult * (D.lower[seq] - D.upper[seq]);
          changeObj -= mv * D.cost[seq];
          len = 1;
          s_cs[f] = 0;
        } else {
          mv = mult * (D.upper[seq] - D.lower[seq]);
          changeObj += mv * D.cost[seq];
          len = D.colStart[seq + 1] - D.colStart[seq];
          s_cs[f] = D.colStart[seq];
        }
        s_mv[f] = mv;
        s_start[f + 1] = len;
      }
    }
    if (rowsDoRhs) {
      // the right-hand side is the row workgroups' job: only the objective term is left
      double s = blockSum(changeObj, shd);
      if (tid == 0) {
        c->objectiveChange += s;
        c->dbg[14]++;
      }
      // ... and the rows more flipped columns share than a row has slots: one wave per such row; the
      // lanes look the row up in 64 flipped columns at a time (flip order) and the wave adds the 64
      // values in lane order (absent ones are + 0.0, which leaves every partial sum as it is)
      const int nHot = c->flipHotCount;
      if (nHot > 0 && nHot <= FLIP_HOT_CAP) {
        const int lane = tid & 63, wv = tid >> 6, nWaves = (int)blockDim.x >> 6;
        for (int h = wv; h < nHot; h += nWaves) {
          const int r = D.flipHot[h];
          double acc = 0.0;
          for (int base = 0; base < nf; base += 64) {
            const int f = base + lane;
            double v = 0.0;
            if (f < nf) {
              const int seq = s_seq[f];
              if (seq >= D.n) {
                if (seq - D.n == r)
                  v = s_mv[f];
              } else {
                int lo = s_cs[f], hi = lo + s_start[f + 1] - 1;
                while (lo < hi) {
                  const int mid = (lo + hi) >> 1;
                  if (D.row[mid] < r)
                    lo = mid + 1;
                  else
                    hi = mid;
                }
                if (lo == hi && D.row[lo] == r)
                  v = s_mv[f] * D.elem[lo];
              }
            }
            const int lim = min(64, nf - base);
            for (int i = 0; i < lim; i++)
              acc += __shfl(v, i);
          }
          if (lane == 0) {
            D.flipRhs[r] = acc;
            const int sr = D.slotOfRow[r];
            if (sr >= 0)
              D.flipSlot[sr] = acc;
          }
        }
        if (tid == 0)
          c->dbg[15] += nHot;
      }
      return;
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
    if (fallback && denseColumns) {
      // every column is a full, row-ordered column: the right-hand side is a row-parallel sweep
      // (k_flip_dense); hand over the movements and finish the scalar part here
      for (int f = tid; f < nf; f += blockDim.x)
        D.flipMv[f] = s_mv[f];
      double s = blockSum(changeObj, shd);
      if (tid == 0) {
        c->objectiveChange += s;
        c->flipDense = 1;
      }
      return;
    }
  }
  if (fallback) {
    if (tid == 0)
      c->dbg[12]++;
    flipSequential(D, nf);
    return;
  }
  const int total = s_total;
  if (tid == 0)
    c->dbg[11] += total;
  // phase 1: every (flip, entry) pair finds its row in an LDS hash table; a row reached by more
  // than one flip gets the MULTI mark.  (flipRhs is all zero on entry: k_ftran_scatter3 clears it.)
  int myRow[FLIP_MAX_ENTRIES / 1024], myFlip[FLIP_MAX_ENTRIES / 1024], mySlot[FLIP_MAX_ENTRIES / 1024];
  double myVal[FLIP_MAX_ENTRIES / 1024];
#pragma unroll
  for (int q = 0; q < FLIP_MAX_ENTRIES / 1024; q++) {
    int e = tid + q * 1024;
    myRow[q] = -1;
    mySlot[q] = 0;
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
      int seq = s_seq[f];
      int r;
      double v;
      if (seq >= D.n) {
        r = seq - D.n;
        v = s_mv[f];
      } else {
        int p = s_cs[f] + (e - s_start[f]);
        r = D.row[p];
        v = s_mv[f] * D.elem[p];
      }
      myRow[q] = r;
      myFlip[q] = f;
      myVal[q] = v;
      unsigned h = ((unsigned)r * 2654435761u) >> (32 - FLIP_HASH_BITS);
      while (true) {
        int old = atomicCAS(&s_hash[h], 0, r + 1);
        if (old == 0)
          break;
        if ((old & FLIP_HASH_ROW) == r + 1) {
          atomicOr(&s_hash[h], FLIP_HASH_MULTI);
          break;
        }
        h = (h + 1) & ((1u << FLIP_HASH_BITS) - 1u);
      }
      mySlot[q] = (int)h;
    }
  }
  __syncthreads();
  {
    // how many entries share their row with another flip?  dense columns collide everywhere:
    // take the sequential form instead
    double nColl = 0.0;
#pragma unroll
    for (int q = 0; q < FLIP_MAX_ENTRIES / 1024; q++)
      if (myRow[q] >= 0 && (s_hash[mySlot[q]] & FLIP_HASH_MULTI))
        nColl += 1.0;
    nColl = blockSum(nColl, shd);
    if (nColl > (double)FLIP_MAX_COLLIDE) {
      if (tid == 0)
        c->dbg[12]++;
      flipSequential(D, nf);
      return;
    }
  }
  // phase 2: single contributors store, the others queue up
#pragma unroll
  for (int q = 0; q < FLIP_MAX_ENTRIES / 1024; q++) {
    if (myRow[q] >= 0) {
      if (!(s_hash[mySlot[q]] & FLIP_HASH_MULTI)) {
        const double nv = 0.0 + myVal[q];
        D.flipRhs[myRow[q]] = nv;
        const int sr = D.slotOfRow[myRow[q]];
        if (sr >= 0)
          D.flipSlot[sr] = nv;
      } else {
        int o = atomicAdd(&s_nCollide, 1);
        s_cRow[o] = myRow[q];
        s_cFlip[o] = myFlip[q];
        s_cVal[o] = myVal[q];
      }
    }
  }
  __syncthreads();
  const int ncol = s_nCollide;
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
      double acc = 0.0;
      for (int j = i; j < ncol && s_cRow[s_cSorted[j]] == r; j++)
        acc += s_cVal[s_cSorted[j]];
      D.flipRhs[r] = acc;
      const int sr = D.slotOfRow[r];
      if (sr >= 0)
        D.flipSlot[sr] = acc;
    }
  }
  double s = blockSum(changeObj, shd);
  if (tid == 0)
    c->objectiveChange += s;
}

// three right-hand sides in one sweep over Minv: entering column, DSE vector (rho), flip rhs.
// The right-hand sides are kept by nucleus row-slot by their producers (k_dual_column, k_rho_finish3,
// k_flip_apply2), staged once per workgroup in LDS (chunks of GEMV_TILE slots); four waves then
// stream each row of Minv against them.
#define GEMV_TILE 2048
#define GEMV_WPR 2                      // waves per row of Minv
#define GEMV_RPB (16 / GEMV_WPR)        // rows per 1024-thread workgroup
__global__ void __launch_bounds__(1024) k_gemv3g(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double s1[GEMV_TILE], s2[GEMV_TILE], s3[GEMV_TILE];
  __shared__ double part[16][3];
  const int k = c->k;
  const bool doTau = c->pivotRule != 0, doFlip = c->numberFlips != 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rowInBlock = wv / GEMV_WPR, li = (wv % GEMV_WPR) * 64 + lane;
  for (int base = blockIdx.x * GEMV_RPB; base < k; base += gridDim.x * GEMV_RPB) {
    const int sc = base + rowInBlock;
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int t0 = 0; t0 < k; t0 += GEMV_TILE) {
      const int tn = min(GEMV_TILE, k - t0);
      __syncthreads();
      for (int i = threadIdx.x; i < tn; i += blockDim.x) {
        s1[i] = D.slotV1[t0 + i];
        s2[i] = doTau ? D.rhoSlotF[t0 + i] : 0.0;
        s3[i] = doFlip ? D.flipSlot[t0 + i] : 0.0;
      }
      __syncthreads();
      if (sc < k) {
        const double *Mrow = D.Minv + (size_t)sc * D.ld + t0;
#pragma unroll 4
        for (int i = li; i < tn; i += 64 * GEMV_WPR) {
          double mv = Mrow[i];
          a1 += mv * s1[i];
          a2 += mv * s2[i];
          a3 += mv * s3[i];
        }
      }
    }
    double r1 = waveSum(a1), r2 = waveSum(a2), r3 = waveSum(a3);
    if (lane == 0) {
      part[wv][0] = r1;
      part[wv][1] = r2;
      part[wv][2] = r3;
    }
    __syncthreads();
    if ((wv % GEMV_WPR) == 0 && lane < 3 && sc < k) {
      double r = part[wv][lane];
#pragma unroll
      for (int u = 1; u < GEMV_WPR; u++)
        r += part[wv + u][lane];
      double *dst = lane == 0 ? D.slotC : (lane == 1 ? D.slotD : D.slotE);
      dst[sc] = r;
    }
  }
}

// back end of the three FTRANs: w, tau and -- when there are flips -- x3 together with the primal
// update it drives (ratio 1.0, ClpSimplexDual.cpp:1535-1536)
// wide-row mode (dense LPs): the slack-row parts of the three FTRANs, one wave per row striding the
// basic partition of the row copy (fixed 64-way tree per row: deterministic, equal to the sequential
// sum to rounding)
__global__ void __launch_bounds__(256) k_slack_dots(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const bool doFlip = c->numberFlips != 0;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= D.m || D.posOfSlack[t] < 0)
    return;
  const int s = D.rowStart[t], e = s + D.basicCount[t];
  double a1 = 0.0, a2 = 0.0, a3 = 0.0;
  // four strides per trip so that each level of the column -> slot -> value chain is one round of
  // loads instead of four
  for (int q0 = s + lane; q0 < e; q0 += 256) {
    int cc[4], sc[4];
    double a[4], c1[4], c2[4], c3[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int q = q0 + 64 * u;
      cc[u] = q < e ? 0 : -1;
      sc[u] = q < e ? D.cslot[q] : 0;  // the entry's col-slot travels with it: no column -> slot lookup
      a[u] = q < e ? D.relem[q] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      c1[u] = D.slotC[sc[u]];
      c2[u] = D.slotD[sc[u]];
      c3[u] = doFlip ? D.slotE[sc[u]] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (cc[u] >= 0) {
        a1 += a[u] * c1[u];
        a2 += a[u] * c2[u];
        if (doFlip)
          a3 += a[u] * c3[u];
      }
    }
  }
  a1 = waveSum(a1);
  a2 = waveSum(a2);
  a3 = waveSum(a3);
  if (lane == 0) {
    D.rowDot[3 * (size_t)t] = a1;
    D.rowDot[3 * (size_t)t + 1] = a2;
    D.rowDot[3 * (size_t)t + 2] = a3;
  }
}

// back end shared by the two forms of the FTRAN scatter: DSE weights, flip part of the primal update,
// hand-over to the serial tail
__device__ inline void ftranScatterTail(const Dev &D, const Ctrl *c, int p, double x1, double x2, double x3, bool doFlip, double tolerance,
                                        int nbNorm, int parity, double *shd)
{
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
      D.appendFlag1[p] = append;
      if (append)
        atomicAdd(&D.blockCount[p >> 8], 1);
    }
    if (p == c->pivotRow) {
      // hand-over to the serial tail: alpha from the FTRAN, the leaving variable's value after the
      // flip update; the leaving row keeps a tiny entry on the list (:705-706)
      stc(&D.ctrl->tailAlpha, x1);
      if (doFlip) {
        stc(&D.ctrl->tailValueOut, D.sol[D.pivotVariable[p]]);
        if (c->pivotRule && D.infeas[p] != 0.0)
          D.infeas[p] = REALLY_TINY;
      }
    }
  }
  double s = blockSum(changeObj, shd);
  if (threadIdx.x == 0)
    stc(&D.blockSum[blockIdx.x], s);
  // serial tail in the last workgroup to finish: append offsets, objective change, the btran/ftran
  // alpha test and the scalar set-up of the basis update
  if (lastBlockDone(D.ctrl, 1))
    scanTailBody(D, nbNorm, gridDim.x, 1, 1, parity);
}


__global__ void __launch_bounds__(256) k_ftran_scatter3(Dev D, int nbNorm, int parity, int wide = 0)
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
      const double v1t = D.vecV1[t], rhot = D.rho[t], flipt = doFlip ? D.flipRhs[t] : 0.0;
      // four entries per trip: column, then slot, then the three slot values are each requested
      // together (the chain is three dependent loads deep); the adds stay in entry order
      if (wide) {
        a1 = D.rowDot[3 * (size_t)t];
        a2 = D.rowDot[3 * (size_t)t + 1];
        a3 = D.rowDot[3 * (size_t)t + 2];
        e = s;
      }
      for (int q = s; q < e; q += 4) {
        int cc[4], sc[4];
        double a[4], c1[4], c2[4], c3[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          cc[u] = (q + u < e) ? 0 : -1;
          sc[u] = (q + u < e) ? D.cslot[q + u] : 0;  // col-slot kept with the entry (rowCopySwap, houseBody)
          a[u] = (q + u < e) ? D.relem[q + u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          c1[u] = D.slotC[sc[u]];
          c2[u] = D.slotD[sc[u]];
          c3[u] = doFlip ? D.slotE[sc[u]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (cc[u] >= 0) {
            a1 += a[u] * c1[u];
            a2 += a[u] * c2[u];
            if (doFlip)
              a3 += a[u] * c3[u];
          }
        }
      }
      x1 = a1 - v1t;
      x2 = a2 - rhot;
      if (doFlip)
        x3 = a3 - flipt;
    }
    if (doFlip)
      D.flipRhs[t] = 0.0;  // consumed (nucleus rows were read by k_gemv3g)
  } else if (t < D.m + k) {
    int sc = t - D.m;
    p = D.slotPos[sc];
    x1 = D.slotC[sc];
    x2 = D.slotD[sc];
    x3 = D.slotE[sc];
    D.slotV1[sc] = 0.0;  // right-hand sides by slot: consumed by k_gemv3g
    if (doFlip)
      D.flipSlot[sc] = 0.0;
  }
  ftranScatterTail(D, c, p, x1, x2, x3, doFlip, tolerance, nbNorm, parity, shd);
}

__global__ void k_zero(double *p, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    p[i] = 0.0;
}
// =============================================================================================
// Newton-Schulz step on the explicit inverse (large nuclei, see clpgpu_context::refineInverse):
// R = I - C X with C = A[R,K] read from the basic part of the row copy (entries carry their column's slot),
// one workgroup per row slot; X += X R is a plain f64 GEMM (rocBLAS).  out[0] = max |R| (bit pattern).
// =============================================================================================
__global__ void __launch_bounds__(256) k_refine_residual(Dev D, int k, double *R, unsigned long long *out)
{
  __shared__ int s_slot[256];
  __shared__ double s_elem[256];
  __shared__ double s_max[4];
  const int sr = blockIdx.x;
  if (sr >= k)
    return;
  const int r = D.slotRow[sr];
  const int start = D.rowStart[r], nb = D.basicCount[r];
  double best = 0.0;
  for (int c0 = 0; c0 < k; c0 += 256) {
    const int col = c0 + threadIdx.x;
    double acc = (col == sr) ? 1.0 : 0.0;
    for (int e0 = 0; e0 < nb; e0 += 256) {
      __syncthreads();
      if (e0 + (int)threadIdx.x < nb) {
        s_slot[threadIdx.x] = D.cslot[start + e0 + threadIdx.x];
        s_elem[threadIdx.x] = D.relem[start + e0 + threadIdx.x];
      }
      __syncthreads();
      const int cnt = min(256, nb - e0);
      if (col < k)
        for (int e = 0; e < cnt; e++)
          acc -= s_elem[e] * D.Minv[(size_t)s_slot[e] * D.ld + col];
    }
    if (col < k) {
      R[(size_t)sr * D.ld + col] = acc;
      best = nanMax(best, fabs(acc));
    }
  }
  for (int o = 32; o > 0; o >>= 1)
    best = nanMax(best, __shfl_xor(best, o));
  if ((threadIdx.x & 63) == 0)
    s_max[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    best = nanMax(nanMax(s_max[0], s_max[1]), nanMax(s_max[2], s_max[3]));
    if (!(best == best))
      best = __longlong_as_double(0x7ff0000000000000LL);  // NaN -> +inf (largest under the integer max)
    atomicMax(out, (unsigned long long)__double_as_longlong(best));
  }
}

// dense form of the same step (LPs with long rows: the residual is a GEMM too): C = A[R,K] by slots into W
// (zeroed by the caller), one workgroup per column slot; max |R| of a k x k row-major matrix
__global__ void __launch_bounds__(64) k_gather_slots(Dev D, int k, double *W)
{
  const int sc = blockIdx.x;
  if (sc >= k)
    return;
  const int j = D.slotCol[sc];
  for (int p = D.colStart[j] + threadIdx.x; p < D.colStart[j + 1]; p += blockDim.x) {
    const int sr = D.slotOfRow[D.row[p]];
    if (sr >= 0)
      W[(size_t)sr * D.ld + sc] = D.elem[p];
  }
}
__global__ void __launch_bounds__(256) k_absmax_rows(Dev D, int k, const double *R, unsigned long long *out)
{
  __shared__ double s_max[4];
  const int r = blockIdx.x;
  if (r >= k)
    return;
  double best = 0.0;
  for (int c = threadIdx.x; c < k; c += 256)
    best = nanMax(best, fabs(R[(size_t)r * D.ld + c]));
  for (int o = 32; o > 0; o >>= 1)
    best = nanMax(best, __shfl_xor(best, o));
  if ((threadIdx.x & 63) == 0)
    s_max[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    best = nanMax(nanMax(s_max[0], s_max[1]), nanMax(s_max[2], s_max[3]));
    if (!(best == best))
      best = __longlong_as_double(0x7ff0000000000000LL);  // NaN -> +inf: ordered as the largest by the integer max
    atomicMax(out, (unsigned long long)__double_as_longlong(best));
  }
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
    if (key < 0) {
      info[0] = 1 + i;
      info[2] = D.perm[i];  // a row no step has pivoted on: its slack can replace the dependent column
    }
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
// ---- blocked form of the same elimination (GJ_B steps per launch group).  Every element still
// receives its updates one step at a time in step order (multiply, then subtract), so W, X, the
// pivot choices and therefore Minv and pivotVariable are bit-identical to the step-by-step kernels
// above; what changes is that W and X are read and written once per block instead of once per step.
#define GJ_B 32
// panel: pivot search, row swap, multipliers and the updates of the panel's own columns, one workgroup
__global__ void __launch_bounds__(1024) k_gj_panel(Dev D, int i0, int b, int k, int *info)
{
  __shared__ double shv[16];
  __shared__ int shk[16];
  __shared__ int s_row;
  __shared__ double s_prow[GJ_B];
  __shared__ double s_inv;
  if (info[0])
    return;
  const int tid = threadIdx.x;
  for (int s = 0; s < b; s++) {
    const int i = i0 + s;
    double best = D.ctrl->zeroTolerance;
    int key = -1;
    for (int j = i + tid; j < k; j += blockDim.x) {
      double v = fabs(D.workW[(size_t)j * D.ld + i]);
      if (v > best) {
        best = v;
        key = j;
      }
    }
    blockArgMax(best, key, shv, shk);
    if (tid == 0) {
      s_row = key;
      if (key < 0) {
        info[0] = 1 + i;
        info[2] = D.perm[i];
      }
      D.gjPiv[s] = key;
    }
    __syncthreads();
    const int iRow = s_row;
    if (iRow < 0)
      return;
    if (iRow != i) {
      if (tid < b) {
        size_t a = (size_t)i * D.ld + i0 + tid, c2 = (size_t)iRow * D.ld + i0 + tid;
        double t = D.workW[a];
        D.workW[a] = D.workW[c2];
        D.workW[c2] = t;
      } else if (tid >= 64 && tid - 64 < s) {
        int t2 = tid - 64;
        double t = D.gjL[(size_t)i * GJ_B + t2];
        D.gjL[(size_t)i * GJ_B + t2] = D.gjL[(size_t)iRow * GJ_B + t2];
        D.gjL[(size_t)iRow * GJ_B + t2] = t;
      } else if (tid == 128) {
        int t = D.perm[i];
        D.perm[i] = D.perm[iRow];
        D.perm[iRow] = t;
      }
    }
    __syncthreads();
    if (tid < b)
      s_prow[tid] = D.workW[(size_t)i * D.ld + i0 + tid];
    if (tid == 0) {
      double inv = 1.0 / D.workW[(size_t)i * D.ld + i];
      s_inv = inv;
      D.slotB[i] = inv;
    }
    __syncthreads();
    const double inv = s_inv;
    for (int r = tid; r < k; r += blockDim.x) {
      if (r == i) {
        D.gjL[(size_t)r * GJ_B + s] = 0.0;
      } else {
        double *Wr = D.workW + (size_t)r * D.ld + i0;
        double l = Wr[s] * inv;
        D.gjL[(size_t)r * GJ_B + s] = l;
        if (l != 0.0)
          for (int t = s + 1; t < b; t++)
            Wr[t] -= s_prow[t] * l;
      }
    }
    __syncthreads();
  }
}
// register-resident panel: thread t owns rows t, t+1024, ... (k <= 1024*GJ_RPT) and keeps their
// GJ_B panel values in registers for all steps.  Rows never move between threads: each row carries
// the position the reference's row interchanges would have put it at (pos), the pivot search
// compares positions for "first largest wins", and only the pivot row travels (through LDS).  The
// multipliers are stored under the row's home index as they are produced; the few rows whose final
// position differs are permuted once at the end.  Same arithmetic, same pivots as k_gj_panel.  The
// panel's own columns are dead after the block (only L, the pivots and the trailing columns are
// used), so nothing is written back to W.
#define GJ_NB 64  // outer block of the two-level form (inner panels of 4 / 8 columns)
struct GjShared {
  double shv[16];
  int shk[16];
  int row;
  int nMoved;
  double prow[GJ_B];
  int movedPos[2 * GJ_B];
  int movedPerm[2 * GJ_B];
  double movedL[2 * GJ_B][GJ_NB];
};
// where a panel leaves its multipliers and pivots: L[r * ldL + lcol0 + s], gjPiv[pcol0 + s].  The
// one-level form uses (gjL, GJ_B, 0, 0); the two-level form writes inner panel j of an outer block at
// column / pivot offset j * b of the outer block's L (gjL2, GJ_NB)
struct GjOut {
  double *L;
  int ldL, lcol0, pcol0;
};
// one elimination step with the step index a compile-time constant (keeps v[][] in registers);
// returns false when the panel turned out singular.  Two barriers per step.
template <int S, int GJ_RPT, int BB, int NT>
__device__ __forceinline__ bool gjPanelStep(const Dev &D, double (&v)[GJ_RPT][BB], int (&pos)[GJ_RPT], GjShared &sh, int i0, int k,
                                            int *info, double zeroTolerance, const GjOut &out)
{
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = i0 + S;
  double best = zeroTolerance;
  int key = -1;
#pragma unroll
  for (int q = 0; q < GJ_RPT; q++) {
    if (pos[q] >= i) {  // rows beyond k carry pos = -1
      double a = fabs(v[q][S]);
      if (a > best || (a == best && key >= 0 && pos[q] < key)) {
        best = a;
        key = pos[q];
      }
    }
  }
  // argmax, smallest position wins ties: butterfly inside the wave, partials through LDS, then the
  // same butterfly over the (<= 16) partials in every wave -- all threads end with the result
  auto combine = [&](double ov, int ok) {
    if (ok >= 0 && (key < 0 || ov > best || (ov == best && ok < key))) {
      best = ov;
      key = ok;
    }
  };
  combine(xchgD<0>(best), xchgI<0>(key));
  combine(xchgD<1>(best), xchgI<1>(key));
  combine(xchgD<2>(best), xchgI<2>(key));
  combine(xchgD<3>(best), xchgI<3>(key));
  combine(xchgD<4>(best), xchgI<4>(key));
  combine(xchgD<5>(best), xchgI<5>(key));
  if (lane == 0) {
    sh.shv[wv] = best;
    sh.shk[wv] = key;
  }
  __syncthreads();
  best = sh.shv[lane & 15];
  key = (lane & 15) < NT / 64 ? sh.shk[lane & 15] : -1;
  combine(xchgD<0>(best), xchgI<0>(key));
  combine(xchgD<1>(best), xchgI<1>(key));
  combine(xchgD<2>(best), xchgI<2>(key));
  combine(xchgD<3>(best), xchgI<3>(key));
  const int iRow = key;
  if (tid == 0) {
    if (iRow < 0) {
      info[0] = 1 + i;
      sh.row = i;
    }
    D.gjPiv[out.pcol0 + S] = iRow;
  }
  if (iRow < 0)
    return false;
  // the pivot row publishes itself and takes position i; the row that sat at i goes to iRow
#pragma unroll
  for (int q = 0; q < GJ_RPT; q++) {
    if (pos[q] == iRow) {
#pragma unroll
      for (int t = 0; t < BB; t++)
        sh.prow[t] = v[q][t];
      pos[q] = i;
    } else if (pos[q] == i) {
      pos[q] = iRow;
    }
  }
  __syncthreads();
  const double inv = 1.0 / sh.prow[S];
  if (tid == 0)
    D.slotB[i] = inv;
#pragma unroll
  for (int q = 0; q < GJ_RPT; q++) {
    const int r = tid + q * NT;
    if (r < k) {
      if (pos[q] == i) {
        out.L[(size_t)r * out.ldL + out.lcol0 + S] = 0.0;
      } else {
        double l = v[q][S] * inv;
        out.L[(size_t)r * out.ldL + out.lcol0 + S] = l;
        if (l != 0.0) {
#pragma unroll
          for (int t = S + 1; t < BB; t++)
            v[q][t] -= sh.prow[t] * l;
        }
      }
    }
  }
  // no barrier here: the partials and prow are next written after a barrier every wave has passed
  return true;
}
template <int S, int GJ_RPT, int BB, int NT> struct GjPanelRun {
  static __device__ __forceinline__ bool run(const Dev &D, double (&v)[GJ_RPT][BB], int (&pos)[GJ_RPT], GjShared &sh, int i0, int b,
                                             int k, int *info, double zeroTolerance, const GjOut &out)
  {
    if (S >= b)
      return true;
    if (!gjPanelStep<S, GJ_RPT, BB, NT>(D, v, pos, sh, i0, k, info, zeroTolerance, out))
      return false;
    return GjPanelRun<S + 1, GJ_RPT, BB, NT>::run(D, v, pos, sh, i0, b, k, info, zeroTolerance, out);
  }
};
template <int GJ_RPT, int BB, int NT> struct GjPanelRun<BB, GJ_RPT, BB, NT> {
  static __device__ __forceinline__ bool run(const Dev &, double (&)[GJ_RPT][BB], int (&)[GJ_RPT], GjShared &, int, int, int, int *,
                                             double, const GjOut &)
  {
    return true;
  }
};
// NT threads own GJ_RPT rows each (k <= NT * GJ_RPT); fewer, fatter waves keep the per-step control
// overhead (which is what bounds this single-workgroup kernel) low
template <int GJ_RPT, int BB, int NT>
__global__ void __launch_bounds__(NT) k_gj_panel_reg(Dev D, int i0, int b, int k, int *info, GjOut out = GjOut{ nullptr, GJ_B, 0, 0 })
{
  __shared__ GjShared sh;
  if (info[0])
    return;
  if (!out.L)
    out.L = D.gjL;
  const int tid = threadIdx.x;
  double v[GJ_RPT][BB];
  int pos[GJ_RPT];
#pragma unroll
  for (int q = 0; q < GJ_RPT; q++) {
    int r = tid + q * NT;
    pos[q] = r < k ? r : -1;
#pragma unroll
    for (int t = 0; t < BB; t++)
      v[q][t] = (r < k && t < b) ? D.workW[(size_t)r * D.ld + i0 + t] : 0.0;
  }
  if (tid == 0)
    sh.nMoved = 0;
  if (!GjPanelRun<0, GJ_RPT, BB, NT>::run(D, v, pos, sh, i0, b, k, info, D.ctrl->zeroTolerance, out)) {
    // singular: the row now at the failing position has never been a pivot row; its original index is
    // what the host needs to put that row's slack into the basis (ClpFactorization.cpp:2382-2532)
    __syncthreads();
    const int iFail = sh.row;
#pragma unroll
    for (int q = 0; q < GJ_RPT; q++)
      if (pos[q] == iFail)
        info[2] = D.perm[tid + q * NT];  // (perm is only rewritten at the end of a successful panel)
    return;
  }
  __syncthreads();
  // rows that ended at another position: move their multipliers -- this panel's and, in the two-level
  // form, those of the outer block's earlier inner panels (columns [0, lcol0): L follows its physical
  // row) -- and their perm entry there
  const int ncL = out.lcol0 + b;
#pragma unroll
  for (int q = 0; q < GJ_RPT; q++) {
    int r = tid + q * NT;
    if (r < k && pos[q] != r) {
      int slot = atomicAdd(&sh.nMoved, 1);
      sh.movedPos[slot] = pos[q];
      sh.movedPerm[slot] = D.perm[r];
      for (int t = 0; t < ncL; t++)
        sh.movedL[slot][t] = out.L[(size_t)r * out.ldL + t];
    }
  }
  __syncthreads();
  const int nMoved = sh.nMoved;
  for (int e = tid; e < nMoved; e += NT)
    D.perm[sh.movedPos[e]] = sh.movedPerm[e];
  for (int e = tid; e < nMoved * ncL; e += NT) {
    int slot = e / ncL, t = e - slot * ncL;
    out.L[(size_t)sh.movedPos[slot] * out.ldL + t] = sh.movedL[slot][t];
  }
}
// the block's row swaps applied to everything outside the panel: W columns >= i0+b and all of X
__global__ void k_gj_rowswaps(Dev D, int i0, int b, int k, int *info)
{
  if (info[0])
    return;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int nW = k - (i0 + b);
  double *base;
  int c;
  if (t < nW) {
    base = D.workW;
    c = i0 + b + t;
  } else if (t < nW + k) {
    base = D.workX;
    c = t - nW;
  } else {
    return;
  }
  for (int s = 0; s < b; s++) {
    int i = i0 + s, iRow = D.gjPiv[s];
    if (iRow != i) {
      size_t a = (size_t)i * D.ld + c, a2 = (size_t)iRow * D.ld + c;
      double v = base[a];
      base[a] = base[a2];
      base[a2] = v;
    }
  }
}
// U[s][c]: value of pivot row i0+s in column c at the time of step s
__global__ void k_gj_upanel(Dev D, int i0, int b, int k, int *info)
{
  if (info[0])
    return;
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int nW = k - (i0 + b);
  const double *base;
  int c;
  if (t < nW) {
    base = D.workW;
    c = i0 + b + t;
  } else if (t < nW + k) {
    base = D.workX;
    c = t - nW;
  } else {
    return;
  }
  double u[GJ_B];
#pragma unroll
  for (int s = 0; s < GJ_B; s++) {
    u[s] = 0.0;
    if (s < b) {
      double a = base[(size_t)(i0 + s) * D.ld + c];
      const double *Lr = D.gjL + (size_t)(i0 + s) * GJ_B;
#pragma unroll
      for (int s2 = 0; s2 < s; s2++) {
        double l = Lr[s2];
        if (l != 0.0)
          a -= u[s2] * l;
      }
      u[s] = a;
      D.gjU[(size_t)s * (2 * D.ld) + t] = a;
    }
  }
}
// every element outside the panel: a -= U[s][c] * L[r][s] for s = 0..b-1 in order (skipping r == i0+s)
#define GJ_ROWS 32
__global__ void __launch_bounds__(256) k_gj_trail(Dev D, int i0, int b, int k, int *info)
{
  if (info[0])
    return;
  __shared__ double sL[GJ_ROWS][GJ_B];
  const int r0 = blockIdx.y * GJ_ROWS;
  for (int e = threadIdx.x; e < GJ_ROWS * GJ_B; e += blockDim.x) {
    int rr = e / GJ_B, s = e % GJ_B;
    sL[rr][s] = (r0 + rr < k && s < b) ? D.gjL[(size_t)(r0 + rr) * GJ_B + s] : 0.0;
  }
  __syncthreads();
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int nW = k - (i0 + b);
  double *base;
  int c;
  if (t < nW) {
    base = D.workW;
    c = i0 + b + t;
  } else if (t < nW + k) {
    base = D.workX;
    c = t - nW;
  } else {
    return;
  }
  double u[GJ_B];
#pragma unroll
  for (int s = 0; s < GJ_B; s++)
    u[s] = (s < b) ? D.gjU[(size_t)s * (2 * D.ld) + t] : 0.0;
  const int rEnd = min(GJ_ROWS, k - r0);
  for (int rr = 0; rr < rEnd; rr++) {
    const int r = r0 + rr;
    double a = base[(size_t)r * D.ld + c];
    bool changed = false;
#pragma unroll
    for (int s = 0; s < GJ_B; s++) {
      double l = sL[rr][s];
      if (l != 0.0 && r != i0 + s) {
        a -= u[s] * l;
        changed = true;
      }
    }
    if (changed)
      base[(size_t)r * D.ld + c] = a;
  }
}

// =============================================================================================
// Two-level, in-place form of the same Gauss-Jordan re-inversion, for nuclei that are genuinely
// large and dense (k >= option refactor_min_k; the wantToGoDense tail of
// CoinAbcBaseFactorization1.cpp:2409-2462 factored by CoinAbcDgetrf, AbcSimplexParallel.cpp:2491,
// with CoinAbcDgemm, CoinAbcHelperFunctions.cpp:1658, as the trailing update).
//   * in place: M = workW is k x k.  Column s of the identity side only comes alive when row s
//     becomes a pivot row, and the W column it eliminates dies at the same moment, so the new X column
//     takes the dead W column's place (columns [0, I0) finished X columns, [I0, I0+nb) the outer panel,
//     [I0+nb, k) live W columns): half the columns of the augmented form, k live columns at all times.
//     The X columns come out in pivot order; k_gj2_finish puts column s at perm[s].
//   * two levels: an outer block of nb <= GJ_NB = 64 pivots is factored as inner register-resident
//     panels of 8 (k <= 4096) or 4 columns; each inner panel's swaps / U rows / rank-b update are
//     applied to the remaining columns of the outer panel only (L2-resident), and the k x k rest of the
//     matrix is read and written once per outer block by a rank-nb update
//         M[:, rest] -= L[:, 0:nb] * U[0:nb, rest]
//     on the matrix cores: v_mfma_f64_16x16x4_f64, 64 x 64 tile per workgroup, one 16 x 64 strip per wave.
// The vector form of the outer update (k_gj2_trail_vec, option refactor_mode 2) performs the
// one-level kernels' operations in the same order and reproduces their bits; the MFMA form fuses the
// four products of a k-step, so it equals them to rounding only -- it is used where the reference itself
// switches to its dense LAPACK-style kernels.
// =============================================================================================
typedef double gj_v4d __attribute__((ext_vector_type(4)));

// row interchanges of steps [0, b) of a panel at i0, columns [c0, c1) except [skip0, skip1)
__global__ void k_gj2_rowswaps(Dev D, int i0, int b, int *info, int c0, int c1, int skip0, int skip1, int pcol0)
{
  if (info[0])
    return;
  int c = c0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= c1 || (c >= skip0 && c < skip1))
    return;
  double *M = D.workW;
  for (int s = 0; s < b; s++) {
    int i = i0 + s, iRow = D.gjPiv[pcol0 + s];
    if (iRow != i) {
      size_t a = (size_t)i * D.ld + c, a2 = (size_t)iRow * D.ld + c;
      double v = M[a];
      M[a] = M[a2];
      M[a2] = v;
    }
  }
}
// the outer panel's own columns become the X columns its pivots create: after the interchanges pivot
// row s sits at position I0 + s, so column I0 + s starts as that unit vector
__global__ void k_gj2_unit(Dev D, int I0, int nb, int k, int *info)
{
  if (info[0])
    return;
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= k * nb)
    return;
  int r = e / nb, s = e - r * nb;
  D.workW[(size_t)r * D.ld + I0 + s] = (r == I0 + s) ? 1.0 : 0.0;
}
// U[s][c - c0]: value of pivot row i0+s in column c at the time of step s (columns [c0, c1))
template <int MAXB>
__global__ void __launch_bounds__(256) k_gj2_upanel(Dev D, int i0, int b, int *info, int c0, int c1, const double *L, int ldL, int lcol0,
                                                     double *U, int ldU)
{
  if (info[0])
    return;
  __shared__ double sLp[MAXB][MAXB + 1];  // multipliers of the pivot rows among themselves
  for (int e = threadIdx.x; e < b * b; e += blockDim.x) {
    int s = e / b, s2 = e - s * b;
    sLp[s][s2] = L[(size_t)(i0 + s) * ldL + lcol0 + s2];
  }
  __syncthreads();
  int c = c0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= c1)
    return;
  const double *M = D.workW;
  double u[MAXB];
#pragma unroll
  for (int s = 0; s < MAXB; s++) {
    u[s] = 0.0;
    if (s < b) {
      double a = M[(size_t)(i0 + s) * D.ld + c];
#pragma unroll
      for (int s2 = 0; s2 < s; s2++) {
        double l = sLp[s][s2];
        if (l != 0.0)
          a -= u[s2] * l;
      }
      u[s] = a;
      U[(size_t)s * ldU + (c - c0)] = a;
    }
  }
}
// vector form of the rank-b update, columns [c0, c1), all rows: a -= U[s][c] * L[r][s] for s = 0..b-1 in
// order (multiply, then subtract: the arithmetic of k_gj_trail), 32 rows per workgroup
template <int MAXB>
__global__ void __launch_bounds__(256) k_gj2_trail_vec(Dev D, int i0, int b, int k, int *info, int c0, int c1, const double *L, int ldL,
                                                        int lcol0, const double *U, int ldU)
{
  if (info[0])
    return;
  __shared__ double sL[GJ_ROWS][MAXB];
  const int r0 = blockIdx.y * GJ_ROWS;
  for (int e = threadIdx.x; e < GJ_ROWS * MAXB; e += blockDim.x) {
    int rr = e / MAXB, s = e % MAXB;
    sL[rr][s] = (r0 + rr < k && s < b) ? L[(size_t)(r0 + rr) * ldL + lcol0 + s] : 0.0;
  }
  __syncthreads();
  int c = c0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= c1)
    return;
  double *M = D.workW;
  double u[MAXB];
#pragma unroll
  for (int s = 0; s < MAXB; s++)
    u[s] = (s < b) ? U[(size_t)s * ldU + (c - c0)] : 0.0;
  const int rEnd = min(GJ_ROWS, k - r0);
  for (int rr = 0; rr < rEnd; rr++) {
    const int r = r0 + rr;
    double a = M[(size_t)r * D.ld + c];
    bool changed = false;
#pragma unroll
    for (int s = 0; s < MAXB; s++) {
      double l = sL[rr][s];
      if (l != 0.0 && r != i0 + s) {
        a -= u[s] * l;
        changed = true;
      }
    }
    if (changed)
      M[(size_t)r * D.ld + c] = a;
  }
}
// matrix-core form of the outer update over all k columns: M -= L[:, 0:nb] * U[0:nb, :].
// v_mfma_f64_16x16x4_f64: lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15] (one f64 each) and four
// results D[(l >> 4) + 4 v][l & 15], v = 0..3.  A = -L so that D = A B + C is the update.  A pivot
// row's own step is excluded by its zero multiplier (the panel stores L[I0 + s][s] = 0).
__global__ void __launch_bounds__(256) k_gj2_trail_mfma(Dev D, int I0, int nb, int k, int *info, const double *L, int ldL, const double *U,
                                                        int ldU)
{
  if (info[0])
    return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rowA = blockIdx.y * 64 + wv * 16 + (lane & 15);   // A operand: this lane's row of L
  const int kq = lane >> 4;                                     // its k index inside a 4-step
  const int colB = blockIdx.x * 64 + (lane & 15);              // B / C / D: this lane's column in sub-tile 0
  const int rowC = blockIdx.y * 64 + wv * 16 + (lane >> 4);    // C / D: row of result v is rowC + 4 v
  double *M = D.workW;
  gj_v4d acc[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c = colB + 16 * j;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int r = rowC + 4 * v;
      acc[j][v] = (r < k && c < k) ? M[(size_t)r * D.ld + c] : 0.0;
    }
  }
  const double *Lrow = L + (size_t)(rowA < k ? rowA : 0) * ldL;
  const bool rowOk = rowA < k;
  for (int k0 = 0; k0 < nb; k0 += 4) {
    const int kk = k0 + kq;
    const bool kOk = kk < nb;
    const double a = (rowOk && kOk) ? -Lrow[kk] : 0.0;
    double bv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int c = colB + 16 * j;
      bv[j] = (kOk && c < k) ? U[(size_t)kk * ldU + c] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
      acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[j], acc[j], 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c = colB + 16 * j;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int r = rowC + 4 * v;
      if (r < k && c < k)
        M[(size_t)r * D.ld + c] = acc[j][v];
    }
  }
}
// Minv = D^-1 X with the X columns put back from pivot order: column s belongs to the row that was
// pivot s, whose original (local) index is perm[s]
__global__ void __launch_bounds__(256) k_gj2_finish(Dev D, int k)
{
  for (int r = blockIdx.y; r < k; r += gridDim.y) {
    double inv = D.slotB[r];
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < k; j += gridDim.x * blockDim.x)
      D.Minv[(size_t)r * D.ld + D.perm[j]] = D.workW[(size_t)r * D.ld + j] * inv;
  }
}

// col-slots of the basic entries of the row copy, rebuilt after a refactorization renumbered the slots
__global__ void k_cslot_rebuild(Dev D)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.m)
    return;
  const int s = D.rowStart[i], e = s + D.basicCount[i];
  for (int q = s; q < e; q++)
    D.cslot[q] = D.slotOfCol[D.ccol[q]];
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
__global__ void k_primal_rhs(Dev D, double *rhs, int wide = 0)
{
  if (wide) {
    // long rows: a wave per row (fixed 64-way tree)
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i < D.m) {
      double acc = 0.0;
      for (int q = D.rowStart[i] + lane; q < D.rowStart[i + 1]; q += 64)
        acc += D.relem[q] * D.sol[D.ccol[q]];
      acc = waveSum(acc);
      if (lane == 0)
        rhs[i] = -acc + D.sol[D.n + i];
    }
    return;
  }
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D.m) {
    double acc = 0.0;
    for (int q = D.rowStart[i]; q < D.rowStart[i + 1]; q++)
      acc += D.relem[q] * D.sol[D.ccol[q]];
    rhs[i] = -acc + D.sol[D.n + i];
  }
}
// max |(A x)_i - s_i| per block (largestPrimalError of computePrimals)
__global__ void __launch_bounds__(256) k_primal_residual(Dev D, int wide = 0)
{
  __shared__ double sh[16];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double r = 0.0;
  if (wide) {
    // long rows: a wave per row, 4 rows per workgroup
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row < D.m) {
      double acc = 0.0;
      for (int q = D.rowStart[row] + lane; q < D.rowStart[row + 1]; q += 64)
        acc += D.relem[q] * D.sol[D.ccol[q]];
      acc = waveSum(acc);
      if (lane == 0)
        r = fabs(acc - D.sol[D.n + row]);
    }
  } else if (i < D.m) {
    double acc = 0.0;
    for (int q = D.rowStart[i]; q < D.rowStart[i + 1]; q++)
      acc += D.relem[q] * D.sol[D.ccol[q]];
    r = fabs(acc - D.sol[D.n + i]);
  }
  // max via min of negatives (a NaN residual is reported as +inf: comparisons would drop it)
  if (!(r == r))
    r = __longlong_as_double(0x7ff0000000000000LL);
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
// iterative refinement of the resync solves: x_B += dx, x_B -= dx, basic reduced costs by position, y += a x
__global__ void k_add_basic(Dev D, const double *x)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m)
    D.sol[D.pivotVariable[p]] += x[p];
}
__global__ void k_sub_basic(Dev D, const double *x)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m)
    D.sol[D.pivotVariable[p]] -= x[p];
}
__global__ void k_basic_djs(Dev D, double *out)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m)
    out[p] = D.dj[D.pivotVariable[p]];
}
__global__ void k_axpy(double *y, const double *x, double a, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    y[i] += a * x[i];
}
// computeDuals helpers (src/ClpSimplex.cpp:1164)
__global__ void k_basic_costs(Dev D, double *cB)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m)
    cB[p] = D.cost[D.pivotVariable[p]];
}
__global__ void k_djs(Dev D, const double *y, int wide = 0)
{
  if (wide) {
    // long columns: a wave per structural column; the slack part keeps a thread per row
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w < D.n) {
      double acc = 0.0;
      for (int p = D.colStart[w] + lane; p < D.colStart[w + 1]; p += 64)
        acc += y[D.row[p]] * D.elem[p];
      acc = waveSum(acc);
      if (lane == 0)
        D.dj[w] = D.cost[w] + acc * -1.0;
    } else {
      const int t = D.n + (w - D.n) * 64 + lane;
      if (t < D.N)
        D.dj[t] = y[t - D.n] + D.cost[t];
    }
    return;
  }
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
__global__ void k_weights_to_seq(Dev D, double *bySeq)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m)
    bySeq[D.pivotVariable[p]] = D.weights[p];
}
__global__ void k_weights_from_seq(Dev D, const double *bySeq, int initialize)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m) {
    double wgt = 1.0;
    if (!initialize) {
      wgt = bySeq[D.pivotVariable[p]];
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
// parity hook for the cycle detector: feeds a sequence of pivots through cycleStep on a scratch control block
__global__ void k_test_cycle(Ctrl *scratch, int n, const int *in, const int *out, const int *wayIn, const int *wayOut, int *matched)
{
  if (threadIdx.x || blockIdx.x)
    return;
  for (int i = 0; i < 12; i++) {
    scratch->cycIn[i] = scratch->cycOut[i] = -1;
    scratch->cycWay[i] = 0;
  }
  scratch->cycHead = 0;
  for (int i = 0; i < n; i++)
    matched[i] = cycleStep(scratch, in[i], out[i], wayIn[i], wayOut[i]);
}
}  // namespace clpgpu
#include "gemm_kernel.hip"
#include "lu_kernels.hip"

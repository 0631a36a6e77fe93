// lu_kernels.hip -- device side of the LU basis factorization:
//
//   B0 = [ slack singletons | Markowitz front (sparse L, U) | dense tail S ]      frozen at a refactorization
//   B_t^-1 = E_t ... E_1 B0^-1                                                      product-form eta file between them
//
// What each piece stands in for (reference, /root/reference):
//   * FTRAN through L, the dense tail and U:   CoinAbcBaseFactorization::updateColumn       src/CoinAbcBaseFactorization3.cpp:68
//     (updateColumnL :1030ff, updateColumnU :1491-1596), FT variant updateColumnFT :2173
//   * BTRAN through U^T, the dense tail and L^T: updateColumnTranspose                       src/CoinAbcBaseFactorization4.cpp:3216
//   * dense tail:  factorDense / CoinAbcDgetrs (src/CoinAbcBaseFactorization2.cpp:976, src/CoinAbcDenseFactorization.cpp:412-470)
//     -- here the tail's INVERSE, produced on the matrix cores (k_gj2_trail_mfma), applied as one GEMV per solve
//   * eta file:    the product-form update of CoinAbcDenseFactorization::replaceColumn (src/CoinAbcDenseFactorization.cpp:334-370:
//     "elements_ + (maximumRows_+numberPivots_)*..." one dense eta per pivot) and its use in updateColumn :412-470
//     / updateColumnTranspose :634-700; the reference's sparse engine uses Forrest-Tomlin row etas
//     (src/CoinAbcBaseFactorization4.cpp:1634-1900) for the same job
//
// The triangular factors of the front are applied through their EXPLICIT sparse inverses (lu_host.hip builds L^-1 and
// U11^-1 [I | -U12] and their transposes; on the bench LP they hold < 2x the entries of L and U), every one a gather-form
// operator: out[tgt] = (src[s] - sum val * vec[idx]) / div with all items independent -- one pass over the whole chip per
// solve, one wave per row, fixed reduction tree, no floating-point atomics.  (A first version walked level schedules of
// the triangular factors in one workgroup per right-hand side: 16 levels, 760 us; profiles/r03_lu_v1_* -> r03_lu_v2_*.)
//
// Product form without a serial chain: with eta_j = (w_j - e_pj)/alpha_j, H = [eta_1 .. eta_t], P = [p_1 .. p_t] and
// N[j][i] = eta_i[p_j] (i < j), the scalars s_j = (E_{j-1}..E_1 x0)[p_j] solve (I + N) s = x0[P]; the engine keeps
// G = (I + N)^-1 (unit lower triangular, one new row per pivot: G[t][:] = -n_t^T G) so that
//     FTRAN  x = x0 - H (G x0[P])                    BTRAN  y = B0^-T (c - P (G^T (H^T c)))
// are a GEMV over H, two small GEMVs over G and one solve with B0 each -- no t-step dependency chain.
#pragma once

namespace clpgpu {

// the factorization's descriptor lives in device memory (its sizes and pointers change at every refactorization
// while the captured launch graphs stay valid)
#define LUD (*D.lu)
#define LU_TCAP_MAX 2048  // capacity of the eta file (rows of G, LDS staging of its vectors)
#define LUG_ROWS 4        // rows of the dense tail (inverse or its transpose) one wave carries through a GEMV
// (two strips of 128 columns per trip below: four strips -- 16 KB in flight per wave -- measured SLOWER in round 6 on all three streams of
// this form, k_lu_gemv3 75 against 68 us, k_lu_gemvT 58 against 55, k_lu_eta_apply 32 against 30: they are not short of bytes in flight)

// One sparse operator application in gather form: out[tgt] = (srcv[src] - sum val * vec[idx]) / div, every item
// independent of the others.  The triangular factors of the front are applied through their EXPLICIT sparse
// inverses (lu_host.hip builds them: on these LPs L^-1 and U11^-1 [I | -U12] hold < 2x the entries of L and U),
// so a solve with the front is one such pass over the whole chip instead of a level-by-level dependency chain.
// The dot products of one item per wave (items it = 4 * workgroup + wave), NV vectors at once: a[u] = -sum val * load(idx, u),
// valid in lane 0.  Rows of the explicit inverses run from one entry to thousands, and a pass is only as fast as its longest
// row: a row longer than LU_LONG_ROW is summed by the WHOLE workgroup (256 strided partial sums, a butterfly per wave, the four
// waves in order -- a fixed tree, deterministic) instead of by its one wave in len / 64 dependent trips.  Every thread of the
// workgroup must call this (it has barriers); items beyond nItems count as empty rows.
#define LU_LONG_ROW 192
template <int NV, class F> __device__ inline void luRowDots(const LuTri &T, int it, F load, double (&a)[NV])
{
  __shared__ int shE0[4], shLen[4];
  __shared__ double shPart[4][NV][4];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int e0 = 0, len = 0;
  if (it < T.nItems) {
    e0 = T.entStart[it];
    len = T.entStart[it + 1] - e0;
  }
  if (lane == 0) {
    shE0[wv] = e0;
    shLen[wv] = len;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NV; u++)
    a[u] = 0.0;
  if (len <= LU_LONG_ROW) {
    for (int e = e0 + lane; e < e0 + len; e += 64) {
      const int idx = T.entIdx[e];
      const double val = T.entVal[e];
#pragma unroll
      for (int u = 0; u < NV; u++)
        a[u] -= val * load(idx, u);
    }
    if (len > 1) {
#pragma unroll
      for (int u = 0; u < NV; u++)
        a[u] = waveSum(a[u]);
    }
  }
  bool any = false;
  for (int q = 0; q < 4; q++) {
    const int lq = shLen[q];
    if (lq <= LU_LONG_ROW)
      continue;  // (uniform over the workgroup)
    any = true;
    const int b = shE0[q];
    double p[NV];
#pragma unroll
    for (int u = 0; u < NV; u++)
      p[u] = 0.0;
    for (int e = b + (int)threadIdx.x; e < b + lq; e += 256) {
      const int idx = T.entIdx[e];
      const double val = T.entVal[e];
#pragma unroll
      for (int u = 0; u < NV; u++)
        p[u] -= val * load(idx, u);
    }
#pragma unroll
    for (int u = 0; u < NV; u++) {
      p[u] = waveSum(p[u]);
      if (lane == 0)
        shPart[q][u][wv] = p[u];
    }
  }
  if (any) {
    __syncthreads();
    if (len > LU_LONG_ROW) {
#pragma unroll
      for (int u = 0; u < NV; u++)
        a[u] = ((shPart[wv][u][0] + shPart[wv][u][1]) + shPart[wv][u][2]) + shPart[wv][u][3];
    }
  }
}

// ---- FTRAN, front half: [y_F ; tail rhs] = L^-1 v for the three right-hand sides (given by row).
// item = local nucleus row; target < k: a front row (work vector), >= k: tail slot target - k (the GEMV's input)
__global__ void __launch_bounds__(256) k_lu_fwd(Dev D, int chain, const double *v0, const double *v1, const double *v2, double *t0,
                                                double *t1, double *t2)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  const LuTri T = LUD.Lf;
  // one wave per item: rows of the explicit inverses run from one entry to thousands
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  bool l0 = v0 != nullptr, l1 = v1 != nullptr, l2 = v2 != nullptr;
  if (chain) {
    l1 = c->pivotRule != 0;
    l2 = c->numberFlips != 0;
  }
  double acc[3];
  luRowDots<3>(T, it, [&](int idx, int u) -> double { return u == 0 ? (l0 ? v0[idx] : 0.0) : (u == 1 ? (l1 ? v1[idx] : 0.0) : (l2 ? v2[idx] : 0.0)); }, acc);
  if (it >= T.nItems || lane != 0)
    return;
  double a0 = acc[0], a1 = acc[1], a2 = acc[2];
  const int src = T.src[it];
  a0 = l0 ? v0[src] + a0 : 0.0;
  a1 = l1 ? v1[src] + a1 : 0.0;
  a2 = l2 ? v2[src] + a2 : 0.0;
  const int tgt = T.tgt[it], k = LUD.k, kpad = LUD.kpad;
  if (tgt < k) {
    LUD.wr[tgt] = a0;
    LUD.wr[kpad + tgt] = a1;
    LUD.wr[2 * kpad + tgt] = a2;
  } else {
    t0[tgt - k] = a0;
    if (t1)
      t1[tgt - k] = a1;
    if (t2)
      t2[tgt - k] = a2;
  }
}

// ---- FTRAN, back half: x_F = U11^-1 (y_F - U12 x_T) as one pass (entries index y by local row, or x_T by
// k + tail column slot); the tail columns copy x_T; both scatter by basis position and by local column
__global__ void __launch_bounds__(256) k_lu_bwd(Dev D, int chain, const double *x0, const double *x1, const double *x2, int live0,
                                                int live1, int live2)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  const LuTri T = LUD.Ub;
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int k = LUD.k, k2 = LUD.k2, kpad = LUD.kpad;
  bool l0 = live0 != 0, l1 = live1 != 0, l2 = live2 != 0;
  if (chain) {
    l0 = true;
    l1 = c->pivotRule != 0;
    l2 = c->numberFlips != 0;
  }
  int col;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  const double *w0 = LUD.wr, *w1 = LUD.wr + kpad, *w2 = LUD.wr + 2 * kpad;
  {
    double acc[3];
    luRowDots<3>(T, it, [&](int idx, int u) -> double {
      if (idx < k)
        return u == 0 ? (l0 ? w0[idx] : 0.0) : (u == 1 ? (l1 ? w1[idx] : 0.0) : (l2 ? w2[idx] : 0.0));
      return u == 0 ? (l0 ? x0[idx - k] : 0.0) : (u == 1 ? (l1 ? x1[idx - k] : 0.0) : (l2 ? x2[idx - k] : 0.0));
    }, acc);
    a0 = acc[0];
    a1 = acc[1];
    a2 = acc[2];
  }
  if (it < T.nItems) {
    if (lane != 0)
      return;
    const int src = T.src[it];
    const double dv = T.div[it];
    a0 = l0 ? (w0[src] + a0) / dv : 0.0;
    a1 = l1 ? (w1[src] + a1) / dv : 0.0;
    a2 = l2 ? (w2[src] + a2) / dv : 0.0;
    col = T.tgt[it];
  } else if (it < T.nItems + k2) {
    if (lane != 0)
      return;
    const int tc = it - T.nItems;
    col = LUD.tailCol[tc];
    a0 = l0 ? x0[tc] : 0.0;
    a1 = l1 ? x1[tc] : 0.0;
    a2 = l2 ? x2[tc] : 0.0;
  } else {
    return;
  }
  const int pos = LUD.posOfCol[col];
  const size_t m = (size_t)D.m;
  if (l0) {
    LUD.xc[col] = a0;
    LUD.x0[pos] = a0;
  }
  if (l1) {
    LUD.xc[kpad + col] = a1;
    LUD.x0[m + pos] = a1;
  }
  if (l2) {
    LUD.xc[2 * kpad + col] = a2;
    LUD.x0[2 * m + pos] = a2;
  }
}

// ---- FTRAN, slack positions: x0[i] = A[i, K0] x_K0 - v[i] over the rows whose slack was basic at the
// refactorization (their U rows); one thread per row and right-hand side
__global__ void __launch_bounds__(256) k_lu_slack(Dev D, int chain, const double *v0, const double *v1, const double *v2, int live0,
                                                  int live1, int live2)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  const int r = blockIdx.y;
  const double *v = r == 0 ? v0 : (r == 1 ? v1 : v2);
  bool live = (r == 0 ? live0 : (r == 1 ? live1 : live2)) != 0;
  if (chain && r == 1)
    live = c->pivotRule != 0;
  if (chain && r == 2)
    live = c->numberFlips != 0;
  if (!live)
    return;
  // 8 lanes per row (fixed 8-way tree): rows are short on the uniform LPs, long on the power-law ones
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  int s = g >> 3;
  const int sub = g & 7;
  if (chain && c->luCompactOn) {
    // compact eta file: the rows of the positions that got a slot since the refactorization (their slack left), nothing else
    const int q = LUD.ncs0 + s;
    s = q < c->luCompactCount ? LUD.sRowOf[LUD.posOfCslot[q]] : LUD.ns;
    if (s < 0)
      s = LUD.ns;
  }
  double acc = 0.0;
  if (s < LUD.ns) {
    const double *xc = LUD.xc + (size_t)r * LUD.kpad;
    for (int e = LUD.sRowStart[s] + sub; e < LUD.sRowStart[s + 1]; e += 8)
      acc += LUD.sRowVal[e] * xc[LUD.sRowCol[e]];
  }
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  acc += __shfl_xor(acc, 4);
  if (s < LUD.ns && sub == 0) {
    const int i = LUD.sRowIndex[s];
    LUD.x0[(size_t)r * D.m + i] = acc - v[i];
  }
}

// ---- product form, FTRAN side: s = G x0[P] for the three right-hand sides; one wave per row of G.
// x0[P] (a scattered gather) is staged once per workgroup in LDS; the rows of G then stream against it.
__global__ void __launch_bounds__(256) k_lu_pf_s(Dev D, int chain, int live0, int live1, int live2)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  const int t = c->pivots;
  if ((int)blockIdx.x * 4 >= t)
    return;
  bool l0 = live0 != 0, l1 = live1 != 0, l2 = live2 != 0;
  if (chain) {
    l0 = true;
    l1 = c->pivotRule != 0;
    l2 = c->numberFlips != 0;
  }
  __shared__ double xp[3 * LU_TCAP_MAX];
  {
    const double *x0 = LUD.x0, *x1 = LUD.x0 + D.m, *x2 = LUD.x0 + 2 * (size_t)D.m;
    for (int i = threadIdx.x; i < t; i += blockDim.x) {
      const int p = LUD.P[i];
      xp[i] = l0 ? x0[p] : 0.0;
      xp[LU_TCAP_MAX + i] = l1 ? x1[p] : 0.0;
      xp[2 * LU_TCAP_MAX + i] = l2 ? x2[p] : 0.0;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < t; j += gridDim.x * 4) {
    const double *Grow = LUD.G + (size_t)j * LUD.tcap;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int i = lane; i <= j; i += 64) {
      const double g = Grow[i];
      a0 += g * xp[i];
      a1 += g * xp[LU_TCAP_MAX + i];
      a2 += g * xp[2 * LU_TCAP_MAX + i];
    }
    a0 = waveSum(a0);
    a1 = waveSum(a1);
    a2 = waveSum(a2);
    if (lane == 0) {
      LUD.s[j] = a0;
      LUD.s[LUD.tcap + j] = a1;
      LUD.s[2 * LUD.tcap + j] = a2;
    }
  }
}

// x = x0 - H s for one position and the three right-hand sides (s staged in LDS by the caller)
__device__ inline void luPfApply(const Dev &D, int t, int p, const double *s0, const double *s1, const double *s2, double &x1, double &x2,
                                 double &x3)
{
  const double *Hp = LUD.H + p;
  const size_t m = (size_t)D.m;
  int j = 0;
  // eight independent loads per trip: one thread per position leaves few waves per CU, so the HBM stream
  // needs its parallelism from within the thread
  for (; j + 8 <= t; j += 8) {
    double h[8];
#pragma unroll
    for (int u = 0; u < 8; u++)
      h[u] = Hp[(size_t)(j + u) * m];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      x1 -= h[u] * s0[j + u];
      x2 -= h[u] * s1[j + u];
      x3 -= h[u] * s2[j + u];
    }
  }
  for (; j < t; j++) {
    const double h = Hp[(size_t)j * m];
    x1 -= h * s0[j];
    x2 -= h * s1[j];
    x3 -= h * s2[j];
  }
}

// The same for the 256 positions base .. base + 255 of one workgroup, as the chain needs it: wave w takes the etas of quarter w of
// the file for ALL 256 positions (four per lane), 8 etas x 4 positions = 32 loads in flight per lane, and the four partial sums of
// a position are added in wave order -- 64 KB of H in flight per workgroup instead of 16 (one thread per position and eight loads
// in flight left this stream at 0.35-0.42 of the HBM peak).  d1..d3 = -(H s) at position base + threadIdx.x.
// (Hsrc / ld / limit: the full file H with stride m over the m positions by default; the compact copy Hc with stride ldc over its slots)
__device__ inline void luPfApplyWg(const Dev &D, int t, int base, int ppb, const double *s0, const double *s1, const double *s2, double *part /*[4][3][256]*/,
                                   double &d1, double &d2, double &d3, const double *Hsrc = nullptr, size_t ld = 0, int limit = -1)
{
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t m = Hsrc ? ld : (size_t)D.m;
  if (!Hsrc)
    Hsrc = LUD.H;
  if (limit < 0)
    limit = D.m;
  const int j0 = (int)(((long long)t * w) / 4), j1 = (int)(((long long)t * (w + 1)) / 4);
  double acc[4][3];
  bool live[4];
  const double *Hp[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    acc[q][0] = acc[q][1] = acc[q][2] = 0.0;
    const int p = base + lane + 64 * q;
    live[q] = lane + 64 * q < ppb && p < limit;  // ppb positions per workgroup (the launch balances m over the CUs)
    Hp[q] = Hsrc + (live[q] ? p : 0);
  }
  int j = j0;
  // (one workgroup per CU at this grid: registers are not what limits residency, bytes in flight are -- 8 etas x 4 positions)
  for (; j + 8 <= j1; j += 8) {
    double h[8][4];
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int q = 0; q < 4; q++)
        h[u][q] = live[q] ? Hp[q][(size_t)(j + u) * m] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const double a = s0[j + u], b = s1[j + u], cc = s2[j + u];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        acc[q][0] -= h[u][q] * a;
        acc[q][1] -= h[u][q] * b;
        acc[q][2] -= h[u][q] * cc;
      }
    }
  }
  for (; j + 4 <= j1; j += 4) {
    double h[4][4];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int q = 0; q < 4; q++)
        h[u][q] = live[q] ? Hp[q][(size_t)(j + u) * m] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const double a = s0[j + u], b = s1[j + u], cc = s2[j + u];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        acc[q][0] -= h[u][q] * a;
        acc[q][1] -= h[u][q] * b;
        acc[q][2] -= h[u][q] * cc;
      }
    }
  }
  for (; j < j1; j++) {
    const double a = s0[j], b = s1[j], cc = s2[j];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double h = live[q] ? Hp[q][(size_t)j * m] : 0.0;
      acc[q][0] -= h * a;
      acc[q][1] -= h * b;
      acc[q][2] -= h * cc;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int v = 0; v < 3; v++)
      part[(w * 3 + v) * 256 + lane + 64 * q] = acc[q][v];
  __syncthreads();
  const int tid = threadIdx.x;
  d1 = ((part[(0 * 3 + 0) * 256 + tid] + part[(1 * 3 + 0) * 256 + tid]) + part[(2 * 3 + 0) * 256 + tid]) + part[(3 * 3 + 0) * 256 + tid];
  d2 = ((part[(0 * 3 + 1) * 256 + tid] + part[(1 * 3 + 1) * 256 + tid]) + part[(2 * 3 + 1) * 256 + tid]) + part[(3 * 3 + 1) * 256 + tid];
  d3 = ((part[(0 * 3 + 2) * 256 + tid] + part[(1 * 3 + 2) * 256 + tid]) + part[(2 * 3 + 2) * 256 + tid]) + part[(3 * 3 + 2) * 256 + tid];
}

// generic form (refactorization boundary, plug-in calls): out_r = x0_r - H s_r
__global__ void __launch_bounds__(256) k_lu_pf_apply(Dev D, double *o0, double *o1, double *o2)
{
  extern __shared__ double lds[];
  const int t = D.ctrl->pivots;
  double *s0 = lds, *s1 = lds + t, *s2 = lds + 2 * t;
  for (int j = threadIdx.x; j < t; j += blockDim.x) {
    s0[j] = LUD.s[j];
    s1[j] = o1 ? LUD.s[LUD.tcap + j] : 0.0;
    s2[j] = o2 ? LUD.s[2 * LUD.tcap + j] : 0.0;
  }
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= D.m)
    return;
  double x1 = LUD.x0[p], x2 = o1 ? LUD.x0[(size_t)D.m + p] : 0.0, x3 = o2 ? LUD.x0[2 * (size_t)D.m + p] : 0.0;
  luPfApply(D, t, p, s0, s1, s2, x1, x2, x3);
  o0[p] = x1;
  if (o1)
    o1[p] = x2;
  if (o2)
    o2[p] = x3;
}

// ---- product form, BTRAN side.  g = H^T c: chain form c = dir * e_r (a gather of row r of H);
// generic form: one wave per eta over the dense c
__global__ void __launch_bounds__(256) k_lu_pf_gdot(Dev D, const double *cvec)
{
  const int t = D.ctrl->pivots;
  const int lane = threadIdx.x & 63;
  for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < t; j += gridDim.x * 4) {
    const double *Hj = LUD.H + (size_t)j * D.m;
    double a = 0.0;
    for (int p = lane; p < D.m; p += 64)
      a += Hj[p] * cvec[p];
    a = waveSum(a);
    if (lane == 0)
      LUD.g[j] = a;
  }
}

// d = G^T g: one wave per column i of G = row i of the transposed copy GT (contiguous), lanes over the etas j >= i;
// chain form: g_j = dir * eta_j[r], a scattered load per eta that every wave issues alongside its GT loads
// chain == 2: the workgroup that finishes last also builds c' (k_lu_cprime's chain form) -- one launch less per pivot
__global__ void __launch_bounds__(256) k_lu_pf_d(Dev D, int chain)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  const int t = c->pivots;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i < t) {
    const double dir = (double)c->directionOut;
    const int r = c->pivotRow;
    const double *GTrow = LUD.GT + (size_t)i * LUD.tcap;
    double acc = 0.0;
    for (int j = i + lane; j < t; j += 64) {
      const double gj = chain ? dir * LUD.H[(size_t)j * D.m + r] : LUD.g[j];
      acc += GTrow[j] * gj;
    }
    acc = waveSum(acc);
    if (lane == 0) {
      if (chain == 2)
        stc(&LUD.d[i], acc);
      else
        LUD.d[i] = acc;
    }
  }
  if (chain != 2)
    return;
  if (!lastBlockDone(D.ctrl, 4))
    return;
  // c' = dir e_r - P d: every position of c' is written by one thread in eta order (the chains of etas on one position)
  if (threadIdx.x == 0)
    LUD.cp[c->pivotRow] = (double)c->directionOut;
  __syncthreads();
  for (int j = threadIdx.x; j < t; j += blockDim.x) {
    if (LUD.nextSame[j] >= 0)
      continue;
    int head = j, len = 1;
    while (LUD.prevSame[head] >= 0) {
      head = LUD.prevSame[head];
      len++;
    }
    double sum = 0.0;
    int q = head;
    for (int u = 0; u < len; u++) {
      sum += ldc(&LUD.d[q]);
      q = LUD.nextSame[q];
    }
    const int p = LUD.P[j];
    LUD.cp[p] = LUD.cp[p] - sum;
  }
}

// c' = c - P d as a dense vector by position (luCp).  Etas that replaced the same position form a chain
// (prevSame); the last one of a chain subtracts the chain's d values in eta order, so every position is
// written by one thread.  chain form: c = dir * e_r.  One workgroup.
__global__ void __launch_bounds__(1024) k_lu_cprime(Dev D, int chain, const double *cvec)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  const int t = c->pivots;
  if (chain) {
    if (threadIdx.x == 0)
      LUD.cp[c->pivotRow] = (double)c->directionOut;
  } else {
    for (int p = threadIdx.x; p < D.m; p += blockDim.x)
      LUD.cp[p] = cvec[p];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < t; j += blockDim.x) {
    if (LUD.nextSame[j] >= 0)
      continue;
    // walk back to the head of the chain, then add forward (eta order)
    int head = j, len = 1;
    while (LUD.prevSame[head] >= 0) {
      head = LUD.prevSame[head];
      len++;
    }
    double sum = 0.0;
    int q = head;
    for (int u = 0; u < len; u++) {
      sum += LUD.d[q];
      q = LUD.nextSame[q];
    }
    const int p = LUD.P[j];
    LUD.cp[p] = LUD.cp[p] - sum;
  }
}

// y_i = -c'[i] on the frozen slack rows, t_c = c'[pos(c)] + sum_{i in S0} a_ic c'[i] for every nucleus column
__global__ void __launch_bounds__(256) k_lu_bt_gather(Dev D, int chain, double *y)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  // one wave per nucleus column: a column holds a few dozen entries in slack rows on the uniform LPs, thousands
  // on the power-law (Netlib-shaped) ones; fixed 64-way tree
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (id < LUD.k) {
    double a0 = 0.0, a1 = 0.0;
    const int e1 = LUD.sColStart[id + 1];
    int e = LUD.sColStart[id] + lane;
    for (; e + 64 < e1; e += 128) {
      a0 += LUD.sColVal[e] * LUD.cp[LUD.sColRow[e]];
      a1 += LUD.sColVal[e + 64] * LUD.cp[LUD.sColRow[e + 64]];
    }
    if (e < e1)
      a0 += LUD.sColVal[e] * LUD.cp[LUD.sColRow[e]];
    const double acc = waveSum(a0 + a1);
    if (lane == 0)
      LUD.tcv[id] = LUD.cp[LUD.posOfCol[id]] + acc;
  }
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < LUD.ns) {
    const int i = LUD.sRowIndex[g];
    (chain ? LUD.y : y)[i] = 0.0 - LUD.cp[i];
  }
}

// z_F = U11^-T t_F and the tail's right-hand side t_T - U12^T z_F, one pass over the transposed explicit inverse
// (every entry indexes t by local column).  target < k: local row of a front pivot (work vector), >= k: tail column slot
__global__ void __launch_bounds__(256) k_lu_bt_front(Dev D, int chain, double *zt)
{
  if (chain && D.ctrl->state != RUN)
    return;
  const LuTri T = LUD.Utf;
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const double *tcv = LUD.tcv;
  double dots[1];
  luRowDots<1>(T, it, [&](int idx, int) -> double { return tcv[idx]; }, dots);
  if (it >= T.nItems || lane != 0)
    return;
  double acc = (tcv[T.src[it]] + dots[0]) / T.div[it];
  const int tgt = T.tgt[it], k = LUD.k;
  if (tgt < k)
    LUD.wr[tgt] = acc;
  else
    zt[tgt - k] = acc;
}

// y_T = S^-T z_T: the tail inverse is frozen between refactorizations, so its transpose is kept as well
// (k_lu_transpose_tail, once per refactorization) and the BTRAN streams contiguous rows like the FTRAN does:
// one wave per tail row slot, result straight into the work vector by local row
__global__ void __launch_bounds__(1024) k_lu_gemvT(Dev D, int chain, const double *zt)
{
  if (chain && D.ctrl->state != RUN)
    return;
  const int k2 = LUD.k2;
  const int lane = threadIdx.x & 63;
  // four rows per wave, 16-byte loads, two strips of 128 columns per trip: 8 x 1 KB in flight per wave (one row per wave with two
  // 8-byte loads in flight left the HBM stream at 0.52 of its peak, profiles/r03_bench_line_default.json)
  const int wpb = blockDim.x >> 6;  // (waves per workgroup: the launch decides -- 128-thread workgroups spread the ~1 400 waves of a tail of
  // 5 600 evenly over the CUs; with 256 threads 353 workgroups land one or two to a CU and the stream runs 15 % slower, round 6)
  for (int ts0 = (blockIdx.x * wpb + (threadIdx.x >> 6)) * LUG_ROWS; ts0 < k2; ts0 += gridDim.x * wpb * LUG_ROWS) {
    double acc[LUG_ROWS];
    const double *row[LUG_ROWS];
#pragma unroll
    for (int r = 0; r < LUG_ROWS; r++) {
      acc[r] = 0.0;
      row[r] = LUD.MinvT + (size_t)min(ts0 + r, k2 - 1) * D.ld;
    }
    for (int i = 2 * lane; i < k2; i += 256) {
      const int ib = i + 128;
      const bool vb = ib < k2, ya = i + 1 < k2, yb = ib + 1 < k2;
      double2 ma[LUG_ROWS], mb[LUG_ROWS];
#pragma unroll
      for (int r = 0; r < LUG_ROWS; r++) {
        ma[r] = *reinterpret_cast<const double2 *>(row[r] + i);
        mb[r] = vb ? *reinterpret_cast<const double2 *>(row[r] + ib) : make_double2(0.0, 0.0);
      }
      double za0, za1, zb0, zb1;
      if (chain) {  // (uniform) the chain's vector is D.slotA: 16-byte pairs, as in k_lu_gemv3; a caller's vector is read entry by entry
        const double2 pa = *reinterpret_cast<const double2 *>(zt + i), pb = vb ? *reinterpret_cast<const double2 *>(zt + ib) : make_double2(0.0, 0.0);
        za0 = pa.x, za1 = ya ? pa.y : 0.0, zb0 = pb.x, zb1 = yb ? pb.y : 0.0;
      } else {
        za0 = zt[i], za1 = ya ? zt[i + 1] : 0.0, zb0 = vb ? zt[ib] : 0.0, zb1 = yb ? zt[ib + 1] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < LUG_ROWS; r++) {
        acc[r] += ma[r].x * za0;
        acc[r] += (ya ? ma[r].y : 0.0) * za1;
        acc[r] += mb[r].x * zb0;
        acc[r] += (yb ? mb[r].y : 0.0) * zb1;
      }
    }
#pragma unroll
    for (int r = 0; r < LUG_ROWS; r++) {
      const double sum = waveSum(acc[r]);
      if (lane == 0 && ts0 + r < k2)
        LUD.wr[LUD.tailRow[ts0 + r]] = sum;
    }
  }
}
// x_T = S^-1 v_T for the three FTRAN right-hand sides in one sweep of the tail inverse.  One wave per FOUR rows:
// the three vectors (slotV1 / rhoSlotF / flipSlot, from L2) are loaded once per column and used against four
// matrix rows, so the kernel issues 7 loads per 12 multiply-adds instead of 4 per 3; skip rules as in k_gemv3g
__global__ void __launch_bounds__(1024) k_lu_gemv3(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int k2 = LUD.k2;
  const bool doTau = c->pivotRule != 0, doFlip = c->numberFlips != 0;
  const int lane = threadIdx.x & 63;
  const double *v1 = D.slotV1, *v2 = D.rhoSlotF, *v3 = D.flipSlot;
  const int wpb = blockDim.x >> 6;  // (waves per workgroup: the launch decides)
  for (int sc0 = (blockIdx.x * wpb + (threadIdx.x >> 6)) * LUG_ROWS; sc0 < k2; sc0 += gridDim.x * wpb * LUG_ROWS) {
    double a1[LUG_ROWS], a2[LUG_ROWS], a3[LUG_ROWS];
    const double *row[LUG_ROWS];
#pragma unroll
    for (int r = 0; r < LUG_ROWS; r++) {
      a1[r] = a2[r] = a3[r] = 0.0;
      row[r] = D.Minv + (size_t)min(sc0 + r, k2 - 1) * D.ld;
    }
    // 16-byte loads, two strips of 128 columns per trip: 8 x 1 KB of the matrix in flight per wave
    for (int i = 2 * lane; i < k2; i += 256) {
      const int ib = i + 128;
      const bool vb = ib < k2, ya = i + 1 < k2, yb = ib + 1 < k2;
      double2 ma[LUG_ROWS], mb[LUG_ROWS];
#pragma unroll
      for (int r = 0; r < LUG_ROWS; r++) {
        ma[r] = *reinterpret_cast<const double2 *>(row[r] + i);
        mb[r] = vb ? *reinterpret_cast<const double2 *>(row[r] + ib) : make_double2(0.0, 0.0);
      }
      // the three vectors as 16-byte pairs (i and ib are even, the slot vectors carry two spare entries): 6 instead of 12 vector
      // loads beside the 8 matrix loads of a trip; entries past k2 are masked (they may hold anything)
      double xs[3][4];  // [vector][strip a: i, i + 1; strip b: ib, ib + 1]
      const double2 zero2 = make_double2(0.0, 0.0);
      const double2 p1a = *reinterpret_cast<const double2 *>(v1 + i), p1b = vb ? *reinterpret_cast<const double2 *>(v1 + ib) : zero2;
      const double2 p2a = doTau ? *reinterpret_cast<const double2 *>(v2 + i) : zero2, p2b = (doTau && vb) ? *reinterpret_cast<const double2 *>(v2 + ib) : zero2;
      const double2 p3a = doFlip ? *reinterpret_cast<const double2 *>(v3 + i) : zero2, p3b = (doFlip && vb) ? *reinterpret_cast<const double2 *>(v3 + ib) : zero2;
      xs[0][0] = p1a.x, xs[0][1] = ya ? p1a.y : 0.0, xs[0][2] = p1b.x, xs[0][3] = yb ? p1b.y : 0.0;
      xs[1][0] = p2a.x, xs[1][1] = ya ? p2a.y : 0.0, xs[1][2] = p2b.x, xs[1][3] = yb ? p2b.y : 0.0;
      xs[2][0] = p3a.x, xs[2][1] = ya ? p3a.y : 0.0, xs[2][2] = p3b.x, xs[2][3] = yb ? p3b.y : 0.0;
#pragma unroll
      for (int r = 0; r < LUG_ROWS; r++) {
        const double mv[4] = { ma[r].x, ya ? ma[r].y : 0.0, mb[r].x, yb ? mb[r].y : 0.0 };
#pragma unroll
        for (int u = 0; u < 4; u++) {
          a1[r] += mv[u] * xs[0][u];
          a2[r] += mv[u] * xs[1][u];
          a3[r] += mv[u] * xs[2][u];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < LUG_ROWS; r++) {
      const double r1 = waveSum(a1[r]), r2 = waveSum(a2[r]), r3 = waveSum(a3[r]);
      if (lane == 0 && sc0 + r < k2) {
        D.slotC[sc0 + r] = r1;
        D.slotD[sc0 + r] = r2;
        D.slotE[sc0 + r] = r3;
      }
    }
  }
}
// MinvT[ts][tc] = Minv[tc][ts], 32 x 32 tiles through LDS
__global__ void __launch_bounds__(256) k_lu_transpose_tail(Dev D, int k2, double *out)
{
  __shared__ double tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int row = by + r, col = bx + tx;
    tile[r][tx] = (row < k2 && col < k2) ? D.Minv[(size_t)row * D.ld + col] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int row = bx + r, col = by + tx;
    if (row < k2 && col < k2)
      out[(size_t)row * D.ld + col] = tile[tx][r];
  }
}

// y = L^-T z (z_F by front row and y_T by tail row in the work vector), result by row; c' is cleared again
__global__ void __launch_bounds__(256) k_lu_bt_back(Dev D, int chain, double *y)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  const LuTri T = LUD.Ltb;
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  double *yout = chain ? LUD.y : y;
  // c' back to zero (nothing below reads it): chain form at the pivot row and the eta positions, generic form everywhere
  {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (chain) {
      if (g == 0)
        LUD.cp[c->pivotRow] = 0.0;
      if (g < c->pivots)
        LUD.cp[LUD.P[g]] = 0.0;
    } else if (g < D.m) {
      LUD.cp[g] = 0.0;
    }
  }
  const double *wr = LUD.wr;
  double dots[1];
  luRowDots<1>(T, it, [&](int idx, int) -> double { return wr[idx]; }, dots);
  if (it < T.nItems && lane == 0)
    yout[LUD.rowOfLocal[T.tgt[it]]] = wr[T.src[it]] + dots[0];
}

// ---- the update: a new eta (column t of H), its position, and row t of G.
// Workgroups [0, gm): eta = (w - e_r) / alpha over the m positions; the rest: a wave per column of the new row of G.
// n_t[j] = eta_j[r] (j < t); G[t][i] = -sum_{j >= i} n_t[j] G[j][i]; G[t][t] = 1.
// chain == 2: the first gm workgroups also make the pivot's primal update (primalUpdateBody: the same span of positions, the same w)
__global__ void __launch_bounds__(256) k_lu_pf_append(Dev D, int chain, int gm)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  if (chain == 2 && (int)blockIdx.x < gm)
    primalUpdateBody(D, 0, blockIdx.x, gm);
  const int t = c->pivots;
  if (t >= LUD.tcap)
    return;
  const int r = c->pivotRow;
  const double alpha = c->alpha;
  if ((int)blockIdx.x < gm) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < D.m) {
      double v = D.w[p];
      if (p == r)
        v -= 1.0;
      v = v / alpha;
      LUD.H[(size_t)t * D.m + p] = v;
      if (c->luCompactOn) {
        // the compact copy: the slot of this position; the pivot's position gets the next free slot when it has none yet (the maps are
        // written by the housekeeping kernel behind this one, so every thread of this launch reads the same state)
        int q = LUD.cslotOfPos[p];
        if (q < 0 && p == r)
          q = c->luCompactCount;
        if (q >= 0)
          LUD.Hc[(size_t)q * LUD.tcap + t] = v;
      }
    }
    if (p == 0) {
      LUD.P[t] = r;
      const int prev = LUD.lastOfPos[r];
      LUD.prevSame[t] = prev;
      LUD.nextSame[t] = -1;
      if (prev >= 0)
        LUD.nextSame[prev] = t;
      LUD.lastOfPos[r] = t;
      LUD.G[(size_t)t * LUD.tcap + t] = 1.0;
      LUD.GT[(size_t)t * LUD.tcap + t] = 1.0;
    }
    return;
  }
  if (c->luCompactOn && LUD.cslotOfPos[r] < 0) {
    // the slack of row r leaves a position that had no slot: its column of H so far becomes the new slot's column
    const int j = (blockIdx.x - gm) * blockDim.x + threadIdx.x;
    if (j < t)
      LUD.Hc[(size_t)c->luCompactCount * LUD.tcap + j] = LUD.H[(size_t)j * D.m + r];
  }
  // row t of G (and column t of GT): one wave per column i < t
  const int i = (blockIdx.x - gm) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= t)
    return;
  const double *GTrow = LUD.GT + (size_t)i * LUD.tcap;
  double acc = 0.0;
  for (int j = i + lane; j < t; j += 64)
    acc += LUD.H[(size_t)j * D.m + r] * GTrow[j];
  acc = waveSum(acc);
  if (lane == 0) {
    LUD.G[(size_t)t * LUD.tcap + i] = 0.0 - acc;
    LUD.GT[(size_t)i * LUD.tcap + t] = 0.0 - acc;
  }
}

// LU mode with the compact eta file, at a refactorization: every basic entry of the row copy gets its column's slot
__global__ void k_cslot_rebuild_lu(Dev D)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.m)
    return;
  const int s = D.rowStart[i], e = s + D.basicCount[i];
  for (int q = s; q < e; q++) {
    const int pos = LUD.posOfBasicCol[D.ccol[q]];
    D.cslot[q] = pos >= 0 ? LUD.cslotOfPos[pos] : -1;
  }
}

// compact eta file: x0 -= Hc s over the slots in use, for the three right-hand sides, in place (x0 by position).  Hc is slot-major
// (slot q holds its t eta entries contiguously): a (slots x t) matrix against the three s vectors -- the shape of k_lu_gemv3, and its
// form: one wave per four slots, 16-byte loads, two strips of 128 etas per trip.
__global__ void __launch_bounds__(1024) k_lu_eta_apply(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int count = c->luCompactCount, t = c->pivots;
  const bool doTau = c->pivotRule != 0, doFlip = c->numberFlips != 0;
  const int lane = threadIdx.x & 63;
  const size_t ld = (size_t)LUD.tcap, m = (size_t)D.m;
  const double *v1 = LUD.s, *v2 = LUD.s + LUD.tcap, *v3 = LUD.s + 2 * LUD.tcap;
  const int wpb = blockDim.x >> 6;
  for (int q0 = (blockIdx.x * wpb + (threadIdx.x >> 6)) * LUG_ROWS; q0 < count; q0 += gridDim.x * wpb * LUG_ROWS) {
    double a1[LUG_ROWS], a2[LUG_ROWS], a3[LUG_ROWS];
    const double *row[LUG_ROWS];
#pragma unroll
    for (int r = 0; r < LUG_ROWS; r++) {
      a1[r] = a2[r] = a3[r] = 0.0;
      row[r] = LUD.Hc + (size_t)min(q0 + r, count - 1) * ld;
    }
    for (int i = 2 * lane; i < t; i += 256) {
      const int ib = i + 128;
      const bool vb = ib < t, ya = i + 1 < t, yb = ib + 1 < t;
      double2 ma[LUG_ROWS], mb[LUG_ROWS];
#pragma unroll
      for (int r = 0; r < LUG_ROWS; r++) {
        ma[r] = *reinterpret_cast<const double2 *>(row[r] + i);
        mb[r] = vb ? *reinterpret_cast<const double2 *>(row[r] + ib) : make_double2(0.0, 0.0);
      }
      double xs[3][4];
      const double2 zero2 = make_double2(0.0, 0.0);
      const double2 p1a = *reinterpret_cast<const double2 *>(v1 + i), p1b = vb ? *reinterpret_cast<const double2 *>(v1 + ib) : zero2;
      const double2 p2a = doTau ? *reinterpret_cast<const double2 *>(v2 + i) : zero2, p2b = (doTau && vb) ? *reinterpret_cast<const double2 *>(v2 + ib) : zero2;
      const double2 p3a = doFlip ? *reinterpret_cast<const double2 *>(v3 + i) : zero2, p3b = (doFlip && vb) ? *reinterpret_cast<const double2 *>(v3 + ib) : zero2;
      xs[0][0] = p1a.x, xs[0][1] = ya ? p1a.y : 0.0, xs[0][2] = p1b.x, xs[0][3] = yb ? p1b.y : 0.0;
      xs[1][0] = p2a.x, xs[1][1] = ya ? p2a.y : 0.0, xs[1][2] = p2b.x, xs[1][3] = yb ? p2b.y : 0.0;
      xs[2][0] = p3a.x, xs[2][1] = ya ? p3a.y : 0.0, xs[2][2] = p3b.x, xs[2][3] = yb ? p3b.y : 0.0;
#pragma unroll
      for (int r = 0; r < LUG_ROWS; r++) {
        const double mv[4] = { ma[r].x, ya ? ma[r].y : 0.0, mb[r].x, yb ? mb[r].y : 0.0 };
#pragma unroll
        for (int u = 0; u < 4; u++) {
          a1[r] += mv[u] * xs[0][u];
          a2[r] += mv[u] * xs[1][u];
          a3[r] += mv[u] * xs[2][u];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < LUG_ROWS; r++) {
      const double r1 = waveSum(a1[r]), r2 = waveSum(a2[r]), r3 = waveSum(a3[r]);
      if (lane == 0 && q0 + r < count) {
        const int p = LUD.posOfCslot[q0 + r];
        const double y1 = LUD.x0[p] - r1, y2 = doTau ? LUD.x0[m + p] - r2 : 0.0, y3 = doFlip ? LUD.x0[2 * m + p] - r3 : 0.0;
        LUD.x0[p] = y1;
        if (doTau)
          LUD.x0[m + p] = y2;
        if (doFlip)
          LUD.x0[2 * m + p] = y3;
        // the same three values packed by slot: one 32-byte gather per row entry for the slack rows (k_ftran_scatter3_lu)
        *reinterpret_cast<double4 *>(LUD.xK + 4 * (size_t)(q0 + r)) = make_double4(y1, y2, y3, 0.0);
      }
    }
  }
}

// LU mode: x = x0 - H s per basis position (x0 from the k_lu_* sweeps, s from k_lu_pf_s), then the same back end.
// compact: k_lu_eta_apply has already applied the eta file at the positions with a slot; the others hold the slack of their own row
// since the refactorization and their value comes from that row of B x = v: x_i = A[i, K] x_K - v_i over the row copy's basic part
// (8 lanes per position, fixed tree), with x_K read at the basic columns' positions.
__global__ void __launch_bounds__(256) k_ftran_scatter3_lu(Dev D, int nbNorm, int parity, int ppb, int compact = 0)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double shd[16];
  __shared__ double sS[3 * LU_TCAP_MAX];
  const int t = c->pivots;
  const bool doFlip = c->numberFlips != 0, doTau = c->pivotRule != 0;
  const double tolerance = c->primalTolerance;
  if (!compact) {
    for (int j = threadIdx.x; j < t; j += blockDim.x) {
      sS[j] = LUD.s[j];
      sS[LU_TCAP_MAX + j] = doTau ? LUD.s[LUD.tcap + j] : 0.0;
      sS[2 * LU_TCAP_MAX + j] = doFlip ? LUD.s[2 * LUD.tcap + j] : 0.0;
    }
    __syncthreads();
  }
  // ppb <= 256 positions per workgroup, chosen by the launch so that one round of workgroups covers the m positions on all CUs
  // (256 positions each left 60 of the 256 CUs without work at m = 50 000)
  const int tt = blockIdx.x * ppb + threadIdx.x;
  int p = -1;
  double x1 = 0.0, x2 = 0.0, x3 = 0.0;
  __shared__ double sPart[4 * 3 * 256];
  double d1 = 0.0, d2 = 0.0, d3 = 0.0;
  if (compact) {
    // one thread per position of the workgroup's span; the entries of a row eight at a time.  An entry carries its column's slot
    // (D.cslot, kept with the row copy as under the explicit inverse), and the three results sit packed by slot: one 32-byte gather
    // per entry instead of a column -> position lookup and three 8-byte gathers
    const int i = blockIdx.x * ppb + threadIdx.x;
    if ((int)threadIdx.x < ppb && i < D.m && LUD.cslotOfPos[i] < 0) {
      const int s = D.rowStart[i], e = s + D.basicCount[i];
      const double4 *xK = reinterpret_cast<const double4 *>(LUD.xK);
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      for (int q = s; q < e; q += 8) {
        double el[8];
        int sl[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const bool in = q + u < e;
          el[u] = in ? D.relem[q + u] : 0.0;
          sl[u] = in ? D.cslot[q + u] : 0;
        }
        double4 y[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
          y[u] = xK[sl[u]];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          a0 += el[u] * y[u].x;
          a1 += el[u] * y[u].y;
          a2 += el[u] * y[u].z;
        }
      }
      sPart[threadIdx.x] = a0 - D.vecV1[i];
      sPart[256 + threadIdx.x] = doTau ? a1 - D.rho[i] : 0.0;
      sPart[512 + threadIdx.x] = doFlip ? a2 - D.flipRhs[i] : 0.0;
    }
  } else {
    luPfApplyWg(D, t, blockIdx.x * ppb, ppb, sS, sS + LU_TCAP_MAX, sS + 2 * LU_TCAP_MAX, sPart, d1, d2, d3);
  }
  if ((int)threadIdx.x < ppb && tt < D.m) {
    p = tt;
    if (compact && LUD.cslotOfPos[p] < 0) {
      x1 = sPart[threadIdx.x];
      x2 = sPart[256 + threadIdx.x];
      x3 = sPart[512 + threadIdx.x];
    } else {
      x1 = LUD.x0[p] + d1;
      x2 = doTau ? LUD.x0[(size_t)D.m + p] + d2 : 0.0;
      x3 = doFlip ? LUD.x0[2 * (size_t)D.m + p] + d3 : 0.0;
    }
    if (doFlip)
      D.flipRhs[tt] = 0.0;  // consumed by k_lu_fwd / k_lu_slack
  }
  ftranScatterTail(D, c, p, x1, x2, x3, doFlip, tolerance, nbNorm, parity, shd);
}

// the dense tail as the re-inversion's input: S entries (tail row slot, tail column slot, value)
__global__ void k_lu_scatter_tail(Dev D, const int *sRow, const int *sCol, const double *sVal, int nnz)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nnz)
    D.workW[(size_t)sRow[e] * D.ld + sCol[e]] = sVal[e];
}

__global__ void k_lu_reset(Dev D)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < D.m) {
    LUD.lastOfPos[p] = -1;
    LUD.cp[p] = 0.0;
  }
}

}  // namespace clpgpu

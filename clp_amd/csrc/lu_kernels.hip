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
// The triangular parts are GATHER-form level schedules (every item is out[tgt] = (src[s] - sum val*vec[idx]) / div,
// items of a level are independent): deterministic, no floating-point atomics.  On the bench LP (nucleus 11 000,
// front 4 500 pivots, 6-8 levels) one 1024-thread workgroup per right-hand side runs a sweep in a few microseconds.
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

__device__ inline void luSweep(const LuTri T, double *out, const double *srcv, const double *vec)
{
  for (int l = 0; l < T.nLevels; l++) {
    const int a = T.levelStart[l], b = T.levelStart[l + 1];
    for (int it = a + threadIdx.x; it < b; it += blockDim.x) {
      double acc = srcv[T.src[it]];
      const int e0 = T.entStart[it], e1 = T.entStart[it + 1];
      for (int e = e0; e < e1; e++)
        acc -= T.entVal[e] * vec[T.entIdx[e]];
      out[T.tgt[it]] = acc / T.div[it];
    }
    __syncthreads();
  }
}

// ---- FTRAN, front half: gather the right-hand sides by nucleus row, L forward, hand the tail part to the GEMV.
// One workgroup per right-hand side.  chain = 1: the three vectors of the pivot (entering column, pruned rho,
// flip rhs) with the skip rules of k_gemv3g; chain = 0: the vectors given.
__global__ void __launch_bounds__(1024) k_lu_fwd(Dev D, int chain, const double *v0, const double *v1, const double *v2, double *t0,
                                                 double *t1, double *t2)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  const int r = blockIdx.x;
  const double *v = r == 0 ? v0 : (r == 1 ? v1 : v2);
  double *tout = r == 0 ? t0 : (r == 1 ? t1 : t2);
  if (!tout)
    return;
  const int k = LUD.k, k2 = LUD.k2;
  bool live = v != nullptr;
  if (chain && r == 1)
    live = c->pivotRule != 0;
  if (chain && r == 2)
    live = c->numberFlips != 0;
  if (!live) {
    for (int ts = threadIdx.x; ts < k2; ts += blockDim.x)
      tout[ts] = 0.0;
    return;
  }
  double *wr = LUD.wr + (size_t)r * LUD.kpad;
  for (int lr = threadIdx.x; lr < k; lr += blockDim.x)
    wr[lr] = v[LUD.rowOfLocal[lr]];
  __syncthreads();
  luSweep(LUD.Lf, wr, wr, wr);
  for (int ts = threadIdx.x; ts < k2; ts += blockDim.x)
    tout[ts] = wr[LUD.tailRow[ts]];
}

// ---- FTRAN, back half: tail solution in, U backward, scatter by basis position
__global__ void __launch_bounds__(1024) k_lu_bwd(Dev D, int chain, const double *x0, const double *x1, const double *x2, int live0,
                                                 int live1, int live2)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  const int r = blockIdx.x;
  const double *xt = r == 0 ? x0 : (r == 1 ? x1 : x2);
  bool live = (r == 0 ? live0 : (r == 1 ? live1 : live2)) != 0;
  if (chain && r == 1)
    live = c->pivotRule != 0;
  if (chain && r == 2)
    live = c->numberFlips != 0;
  if (!live)
    return;
  const int k = LUD.k, k2 = LUD.k2;
  double *wr = LUD.wr + (size_t)r * LUD.kpad;
  double *xc = LUD.xc + (size_t)r * LUD.kpad;
  for (int tc = threadIdx.x; tc < k2; tc += blockDim.x)
    xc[LUD.tailCol[tc]] = xt[tc];
  __syncthreads();
  luSweep(LUD.Ub, xc, wr, xc);
  double *x0pos = LUD.x0 + (size_t)r * D.m;
  for (int cc = threadIdx.x; cc < k; cc += blockDim.x)
    x0pos[LUD.posOfCol[cc]] = xc[cc];
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
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= LUD.ns)
    return;
  const double *xc = LUD.xc + (size_t)r * LUD.kpad;
  const int i = LUD.sRowIndex[s];
  double acc = 0.0;
  for (int e = LUD.sRowStart[s]; e < LUD.sRowStart[s + 1]; e++)
    acc += LUD.sRowVal[e] * xc[LUD.sRowCol[e]];
  LUD.x0[(size_t)r * D.m + i] = acc - v[i];
}

// ---- product form, FTRAN side: s = G x0[P] for the three right-hand sides; one wave per row of G
__global__ void __launch_bounds__(256) k_lu_pf_s(Dev D, int chain, int live0, int live1, int live2)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  const int t = c->pivots;
  bool l0 = live0 != 0, l1 = live1 != 0, l2 = live2 != 0;
  if (chain) {
    l0 = true;
    l1 = c->pivotRule != 0;
    l2 = c->numberFlips != 0;
  }
  const int lane = threadIdx.x & 63;
  const double *x0 = LUD.x0, *x1 = LUD.x0 + D.m, *x2 = LUD.x0 + 2 * (size_t)D.m;
  for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < t; j += gridDim.x * 4) {
    const double *Grow = LUD.G + (size_t)j * LUD.tcap;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int i = lane; i <= j; i += 64) {
      const double g = Grow[i];
      const int p = LUD.P[i];
      if (l0)
        a0 += g * x0[p];
      if (l1)
        a1 += g * x1[p];
      if (l2)
        a2 += g * x2[p];
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
  for (; j + 4 <= t; j += 4) {
    const double h0 = Hp[(size_t)j * m], h1 = Hp[(size_t)(j + 1) * m], h2 = Hp[(size_t)(j + 2) * m], h3 = Hp[(size_t)(j + 3) * m];
    x1 -= h0 * s0[j];
    x2 -= h0 * s1[j];
    x3 -= h0 * s2[j];
    x1 -= h1 * s0[j + 1];
    x2 -= h1 * s1[j + 1];
    x3 -= h1 * s2[j + 1];
    x1 -= h2 * s0[j + 2];
    x2 -= h2 * s1[j + 2];
    x3 -= h2 * s2[j + 2];
    x1 -= h3 * s0[j + 3];
    x2 -= h3 * s1[j + 3];
    x3 -= h3 * s2[j + 3];
  }
  for (; j < t; j++) {
    const double h = Hp[(size_t)j * m];
    x1 -= h * s0[j];
    x2 -= h * s1[j];
    x3 -= h * s2[j];
  }
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

// d = G^T g: workgroup b owns columns [64 b, 64 b + 64); its four waves take the rows j = w (mod 4), the four
// partial sums are added in wave order
__global__ void __launch_bounds__(256) k_lu_pf_d(Dev D, int chain)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  const int t = c->pivots;
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  if (blockIdx.x * 64 >= t)
    return;
  const double dir = (double)c->directionOut;
  const int r = c->pivotRow;
  double acc = 0.0;
  for (int j = blockIdx.x * 64 + wv; j < t; j += 4) {
    const double gj = chain ? dir * LUD.H[(size_t)j * D.m + r] : LUD.g[j];
    if (i <= j)
      acc += LUD.G[(size_t)j * LUD.tcap + i] * gj;
  }
  part[wv][lane] = acc;
  __syncthreads();
  if (wv == 0 && i < t)
    LUD.d[i] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
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
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id < LUD.k) {
    double acc = LUD.cp[LUD.posOfCol[id]];
    for (int e = LUD.sColStart[id]; e < LUD.sColStart[id + 1]; e++)
      acc += LUD.sColVal[e] * LUD.cp[LUD.sColRow[e]];
    LUD.tcv[id] = acc;
  } else if (id < LUD.k + LUD.ns) {
    const int i = LUD.sRowIndex[id - LUD.k];
    (chain ? LUD.y : y)[i] = 0.0 - LUD.cp[i];
  }
}

// U^T forward over the front, then the tail's right-hand side (by tail column slot) for the GEMV^T
__global__ void __launch_bounds__(1024) k_lu_bt_front(Dev D, int chain, double *zt)
{
  if (chain && D.ctrl->state != RUN)
    return;
  double *wr = LUD.wr;
  luSweep(LUD.Utf, wr, LUD.tcv, wr);
  luSweep(LUD.UtT, zt, LUD.tcv, wr);
}

// y_T = Minv^T z_T from the per-chunk partials of k_gemvT_partial, left by tail row slot in the work vector
__global__ void __launch_bounds__(256) k_lu_gemvT_final(Dev D, int chain)
{
  if (chain && D.ctrl->state != RUN)
    return;
  const int k2 = LUD.k2;
  const int sr = blockIdx.x * blockDim.x + threadIdx.x;
  if (sr >= k2)
    return;
  const int nchunk = (k2 + 63) >> 6;
  double acc = 0.0;
  for (int ch = 0; ch < nchunk; ch++)
    acc += D.partial[(size_t)ch * D.ld + sr];
  LUD.wr[LUD.tailRow[sr]] = acc;
}

// L^T backward, result by row; the sparse c' is cleared again
__global__ void __launch_bounds__(1024) k_lu_bt_back(Dev D, int chain, double *y)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  double *wr = LUD.wr;
  luSweep(LUD.Ltb, wr, wr, wr);
  const int k = LUD.k;
  double *yout = chain ? LUD.y : y;
  for (int lr = threadIdx.x; lr < k; lr += blockDim.x)
    yout[LUD.rowOfLocal[lr]] = wr[lr];
  if (chain) {
    const int t = c->pivots;
    if (threadIdx.x == 0)
      LUD.cp[c->pivotRow] = 0.0;
    __syncthreads();
    for (int j = threadIdx.x; j < t; j += blockDim.x)
      LUD.cp[LUD.P[j]] = 0.0;
  } else {
    for (int p = threadIdx.x; p < D.m; p += blockDim.x)
      LUD.cp[p] = 0.0;
  }
}

// ---- the update: a new eta (column t of H), its position, and row t of G.
// Workgroups [0, gm): eta = (w - e_r) / alpha over the m positions; the rest: 64 columns of the new row of G each.
// n_t[j] = eta_j[r] (j < t); G[t][i] = -sum_{j >= i} n_t[j] G[j][i]; G[t][t] = 1.
__global__ void __launch_bounds__(256) k_lu_pf_append(Dev D, int chain, int gm)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
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
      LUD.H[(size_t)t * D.m + p] = v / alpha;
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
    }
    return;
  }
  const int b = blockIdx.x - gm;
  if (b * 64 >= t)
    return;
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = b * 64 + lane;
  double acc = 0.0;
  for (int j = b * 64 + wv; j < t; j += 4) {
    const double nj = LUD.H[(size_t)j * D.m + r];
    if (i <= j)
      acc += nj * LUD.G[(size_t)j * LUD.tcap + i];
  }
  part[wv][lane] = acc;
  __syncthreads();
  if (wv == 0 && i < t)
    LUD.G[(size_t)t * LUD.tcap + i] = 0.0 - (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]);
}

// LU mode: x = x0 - H s per basis position (x0 from the k_lu_* sweeps, s from k_lu_pf_s), then the same back end
#define LU_TCAP_MAX 2048
__global__ void __launch_bounds__(256) k_ftran_scatter3_lu(Dev D, int nbNorm, int parity)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  __shared__ double shd[16];
  __shared__ double sS[3 * LU_TCAP_MAX];
  const int t = c->pivots;
  const bool doFlip = c->numberFlips != 0, doTau = c->pivotRule != 0;
  const double tolerance = c->primalTolerance;
  for (int j = threadIdx.x; j < t; j += blockDim.x) {
    sS[j] = LUD.s[j];
    sS[LU_TCAP_MAX + j] = doTau ? LUD.s[LUD.tcap + j] : 0.0;
    sS[2 * LU_TCAP_MAX + j] = doFlip ? LUD.s[2 * LUD.tcap + j] : 0.0;
  }
  __syncthreads();
  const int tt = blockIdx.x * blockDim.x + threadIdx.x;
  int p = -1;
  double x1 = 0.0, x2 = 0.0, x3 = 0.0;
  if (tt < D.m) {
    p = tt;
    x1 = LUD.x0[p];
    x2 = doTau ? LUD.x0[(size_t)D.m + p] : 0.0;
    x3 = doFlip ? LUD.x0[2 * (size_t)D.m + p] : 0.0;
    luPfApply(D, t, p, sS, sS + LU_TCAP_MAX, sS + 2 * LU_TCAP_MAX, x1, x2, x3);
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

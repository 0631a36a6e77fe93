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

// One sparse operator application in gather form: out[tgt] = (srcv[src] - sum val * vec[idx]) / div, every item
// independent of the others.  The triangular factors of the front are applied through their EXPLICIT sparse
// inverses (lu_host.hip builds them: on these LPs L^-1 and U11^-1 [I | -U12] hold < 2x the entries of L and U),
// so a solve with the front is one such pass over the whole chip instead of a level-by-level dependency chain.
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
  if (it >= T.nItems)
    return;
  bool l0 = v0 != nullptr, l1 = v1 != nullptr, l2 = v2 != nullptr;
  if (chain) {
    l1 = c->pivotRule != 0;
    l2 = c->numberFlips != 0;
  }
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  const int e0 = T.entStart[it], e1 = T.entStart[it + 1];
  for (int e = e0 + lane; e < e1; e += 64) {
    const int idx = T.entIdx[e];
    const double val = T.entVal[e];
    if (l0)
      a0 -= val * v0[idx];
    if (l1)
      a1 -= val * v1[idx];
    if (l2)
      a2 -= val * v2[idx];
  }
  if (e1 - e0 > 1) {
    a0 = waveSum(a0);
    a1 = waveSum(a1);
    a2 = waveSum(a2);
  }
  if (lane != 0)
    return;
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
  if (it < T.nItems) {
    const double *w0 = LUD.wr, *w1 = LUD.wr + kpad, *w2 = LUD.wr + 2 * kpad;
    const int e0 = T.entStart[it], e1 = T.entStart[it + 1];
    for (int e = e0 + lane; e < e1; e += 64) {
      const int idx = T.entIdx[e];
      const double val = T.entVal[e];
      if (idx < k) {
        if (l0)
          a0 -= val * w0[idx];
        if (l1)
          a1 -= val * w1[idx];
        if (l2)
          a2 -= val * w2[idx];
      } else {
        if (l0)
          a0 -= val * x0[idx - k];
        if (l1)
          a1 -= val * x1[idx - k];
        if (l2)
          a2 -= val * x2[idx - k];
      }
    }
    if (e1 - e0 > 1) {
      a0 = waveSum(a0);
      a1 = waveSum(a1);
      a2 = waveSum(a2);
    }
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
  const int s = g >> 3, sub = g & 7;
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
__global__ void __launch_bounds__(256) k_lu_pf_d(Dev D, int chain)
{
  const Ctrl *c = D.ctrl;
  if (chain && c->state != RUN)
    return;
  const int t = c->pivots;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= t)
    return;
  const double dir = (double)c->directionOut;
  const int r = c->pivotRow;
  const double *GTrow = LUD.GT + (size_t)i * LUD.tcap;
  double acc = 0.0;
  for (int j = i + lane; j < t; j += 64) {
    const double gj = chain ? dir * LUD.H[(size_t)j * D.m + r] : LUD.g[j];
    acc += GTrow[j] * gj;
  }
  acc = waveSum(acc);
  if (lane == 0)
    LUD.d[i] = acc;
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
  if (it >= T.nItems)
    return;
  const double *tcv = LUD.tcv;
  double acc = 0.0;
  const int e0 = T.entStart[it], e1 = T.entStart[it + 1];
  for (int e = e0 + lane; e < e1; e += 64)
    acc -= T.entVal[e] * tcv[T.entIdx[e]];
  if (e1 - e0 > 1)
    acc = waveSum(acc);
  if (lane != 0)
    return;
  acc = (tcv[T.src[it]] + acc) / T.div[it];
  const int tgt = T.tgt[it], k = LUD.k;
  if (tgt < k)
    LUD.wr[tgt] = acc;
  else
    zt[tgt - k] = acc;
}

// y_T = S^-T z_T: the tail inverse is frozen between refactorizations, so its transpose is kept as well
// (k_lu_transpose_tail, once per refactorization) and the BTRAN streams contiguous rows like the FTRAN does:
// one wave per tail row slot, result straight into the work vector by local row
__global__ void __launch_bounds__(256) k_lu_gemvT(Dev D, int chain, const double *zt)
{
  if (chain && D.ctrl->state != RUN)
    return;
  const int k2 = LUD.k2;
  const int lane = threadIdx.x & 63;
  for (int ts = blockIdx.x * 4 + (threadIdx.x >> 6); ts < k2; ts += gridDim.x * 4) {
    const double *row = LUD.MinvT + (size_t)ts * D.ld;
    double a0 = 0.0, a1 = 0.0;
    int i = lane;
    for (; i + 64 < k2; i += 128) {
      a0 += row[i] * zt[i];
      a1 += row[i + 64] * zt[i + 64];
    }
    if (i < k2)
      a0 += row[i] * zt[i];
    const double acc = waveSum(a0 + a1);
    if (lane == 0)
      LUD.wr[LUD.tailRow[ts]] = acc;
  }
}
// x_T = S^-1 v_T for the three FTRAN right-hand sides in one sweep of the tail inverse.  One wave per FOUR rows:
// the three vectors (slotV1 / rhoSlotF / flipSlot, from L2) are loaded once per column and used against four
// matrix rows, so the kernel issues 7 loads per 12 multiply-adds instead of 4 per 3; skip rules as in k_gemv3g
#define LUG_ROWS 4
__global__ void __launch_bounds__(256) k_lu_gemv3(Dev D)
{
  const Ctrl *c = D.ctrl;
  if (c->state != RUN)
    return;
  const int k2 = LUD.k2;
  const bool doTau = c->pivotRule != 0, doFlip = c->numberFlips != 0;
  const int lane = threadIdx.x & 63;
  const double *v1 = D.slotV1, *v2 = D.rhoSlotF, *v3 = D.flipSlot;
  for (int sc0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LUG_ROWS; sc0 < k2; sc0 += gridDim.x * 4 * LUG_ROWS) {
    double a1[LUG_ROWS], a2[LUG_ROWS], a3[LUG_ROWS];
    const double *row[LUG_ROWS];
#pragma unroll
    for (int r = 0; r < LUG_ROWS; r++) {
      a1[r] = a2[r] = a3[r] = 0.0;
      row[r] = D.Minv + (size_t)min(sc0 + r, k2 - 1) * D.ld;
    }
    for (int i = lane; i < k2; i += 64) {
      const double x1 = v1[i], x2 = doTau ? v2[i] : 0.0, x3 = doFlip ? v3[i] : 0.0;
      double mv[LUG_ROWS];
#pragma unroll
      for (int r = 0; r < LUG_ROWS; r++)
        mv[r] = row[r][i];
#pragma unroll
      for (int r = 0; r < LUG_ROWS; r++) {
        a1[r] += mv[r] * x1;
        a2[r] += mv[r] * x2;
        a3[r] += mv[r] * x3;
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
  if (it >= T.nItems)
    return;
  const double *wr = LUD.wr;
  double acc = 0.0;
  const int e0 = T.entStart[it], e1 = T.entStart[it + 1];
  for (int e = e0 + lane; e < e1; e += 64)
    acc -= T.entVal[e] * wr[T.entIdx[e]];
  if (e1 - e0 > 1)
    acc = waveSum(acc);
  if (lane == 0)
    yout[LUD.rowOfLocal[T.tgt[it]]] = wr[T.src[it]] + acc;
}

// ---- the update: a new eta (column t of H), its position, and row t of G.
// Workgroups [0, gm): eta = (w - e_r) / alpha over the m positions; the rest: a wave per column of the new row of G.
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
      LUD.GT[(size_t)t * LUD.tcap + t] = 1.0;
    }
    return;
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

// LU mode: x = x0 - H s per basis position (x0 from the k_lu_* sweeps, s from k_lu_pf_s), then the same back end
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

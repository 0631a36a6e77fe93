// lu_host.hip -- host side of the LU factorization mode (included by engine.hip after clpgpu_context):
// builds B0 = [slack singletons | Markowitz front | dense tail] at a refactorization boundary and drives the
// generic (non-chain) solves.  See lu_front.h (host Markowitz) and lu_kernels.hip (device sweeps, eta file).
//
// ClpFactorization::factorize (src/ClpFactorization.cpp:1649) -> CoinAbcBaseFactorization::factor
// (src/CoinAbcBaseFactorization1.cpp:372: preProcess, factorSparse, factorDense, cleanup/permute).

// one gather-form level schedule, flattened for the device
struct LuTriHost {
  std::vector<int> levelStart, tgt, src, entStart, entIdx;
  std::vector<double> entVal, div;
  // items given unordered: item i has level lev[i], entries [es[i], es[i+1]) of (idx, val)
  void build(int nItems, const std::vector<int> &lev, const std::vector<int> &tg, const std::vector<int> &sr, const std::vector<double> &dv,
             const std::vector<int> &es, const std::vector<int> &ei, const std::vector<double> &ev)
  {
    int nLev = 0;
    for (int i = 0; i < nItems; i++)
      nLev = std::max(nLev, lev[i] + 1);
    levelStart.assign(nLev + 1, 0);
    for (int i = 0; i < nItems; i++)
      levelStart[lev[i] + 1]++;
    for (int l = 0; l < nLev; l++)
      levelStart[l + 1] += levelStart[l];
    std::vector<int> at(levelStart.begin(), levelStart.end() - 1), order(nItems);
    for (int i = 0; i < nItems; i++)
      order[at[lev[i]]++] = i;
    tgt.resize(nItems);
    src.resize(nItems);
    div.resize(nItems);
    entStart.assign(nItems + 1, 0);
    entIdx.clear();
    entVal.clear();
    entIdx.reserve(ei.size());
    entVal.reserve(ev.size());
    for (int o = 0; o < nItems; o++) {
      const int i = order[o];
      tgt[o] = tg[i];
      src[o] = sr[i];
      div[o] = dv[i];
      for (int e = es[i]; e < es[i + 1]; e++) {
        entIdx.push_back(ei[e]);
        entVal.push_back(ev[e]);
      }
      entStart[o + 1] = (int)entIdx.size();
    }
  }
};

int clpgpu_context::luUploadTri(const LuTriHost &h, LuTri &d, int slot)
{
  const int nItems = (int)h.tgt.size();
  int rc = 0;
  d.nLevels = (int)h.levelStart.size() - 1;
  d.nItems = nItems;
  int *ip = nullptr;
  double *dp = nullptr;
  rc |= luBuf[slot + 0].put(this, h.levelStart, ip);
  d.levelStart = ip;
  rc |= luBuf[slot + 1].put(this, h.tgt, ip);
  d.tgt = ip;
  rc |= luBuf[slot + 2].put(this, h.src, ip);
  d.src = ip;
  rc |= luBuf[slot + 3].put(this, h.entStart, ip);
  d.entStart = ip;
  rc |= luBuf[slot + 4].put(this, h.entIdx, ip);
  d.entIdx = ip;
  rc |= luBuf[slot + 5].put(this, h.entVal, dp);
  d.entVal = dp;
  rc |= luBuf[slot + 6].put(this, h.div, dp);
  d.div = dp;
  return rc;
}

int clpgpu_context::factorizeLu(const std::vector<int> &kcol, const std::vector<int> &rrows, const std::vector<int> &localOfRow)
{
  const int k = (int)kcol.size();
  const auto t0 = std::chrono::steady_clock::now();
  int rc = 0;
  // ---- the nucleus C = A[R, K] by columns, local rows
  std::vector<int> cStart(k + 1, 0), cRow;
  std::vector<double> cVal;
  // ... and the frozen slack part: entries of the basic structurals in rows whose slack is basic
  std::vector<int> sColStart(k + 1, 0), sColRow;
  std::vector<double> sColVal;
  for (int c = 0; c < k; c++) {
    const int j = kcol[c];
    for (int p = colStart[j]; p < colStart[j + 1]; p++) {
      const int lr = localOfRow[row[p]];
      if (lr >= 0) {
        cRow.push_back(lr);
        cVal.push_back(elem[p]);
      } else {
        sColRow.push_back(row[p]);
        sColVal.push_back(elem[p]);
      }
    }
    cStart[c + 1] = (int)cRow.size();
    sColStart[c + 1] = (int)sColRow.size();
  }
  // ---- Markowitz front on the host (skipped outright when the nucleus is dense already)
  LuFront &F = luF;
  const double nnzC = (double)cRow.size();
  if (luStopDensity > 0.0 && nnzC <= luStopDensity * (double)k * (double)k) {
    luFrontFactor(k, cStart.data(), cRow.data(), cVal.data(), luStopDensity, luMinTail, luThreshold, 1.0e-11, F);
  } else {
    F = LuFront();
    F.k = k;
    F.k2 = k;
    F.lStart.assign(1, 0);
    F.uStart.assign(1, 0);
    F.tailRow.resize(k);
    F.tailCol.resize(k);
    for (int i = 0; i < k; i++)
      F.tailRow[i] = F.tailCol[i] = i;
    for (int c = 0; c < k; c++)
      for (int p = cStart[c]; p < cStart[c + 1]; p++) {
        F.sRow.push_back(cRow[p]);
        F.sCol.push_back(c);
        F.sVal.push_back(cVal[p]);
      }
  }
  const int nF = F.nF, k2 = F.k2;
  // this factorization's own count in CoinFactorization's convention: L and U of the front (pivots included), the dense block, the
  // frozen slack part of the basic structurals (option steepest_elements 1)
  luOwnElements = (long long)F.lVal.size() + (long long)F.uVal.size() + nF + (long long)k2 * k2 + (long long)sColVal.size();
  const auto t1 = std::chrono::steady_clock::now();
  // ---- dense tail: S -> S^-1 on the device (MFMA re-inversion)
  rc = allocNucleus(k2);
  if (rc)
    return rc;
  std::vector<int> perm(k2);
  if (k2) {
    int info[4];
    rc |= prepareWork(k2);
    int *dSRow = nullptr, *dSCol = nullptr;
    double *dSVal = nullptr;
    rc |= luBuf[LB_SROW].put(this, F.sRow, dSRow);
    rc |= luBuf[LB_SCOL].put(this, F.sCol, dSCol);
    rc |= luBuf[LB_SVAL].put(this, F.sVal, dSVal);
    if (rc)
      return rc;
    const int snz = (int)F.sVal.size();
    if (snz)
      hipLaunchKernelGGL(k_lu_scatter_tail, dim3(cdiv(snz, 256)), dim3(256), 0, stream, D, (const int *)dSRow, (const int *)dSCol,
                         (const double *)dSVal, snz);
    rc = invertWork(k2, perm, info);
    if (rc)
      return rc;
    if (info[0]) {
      setError("factorize: singular tail at step %d of %d (nucleus %d, front %d)", info[0] - 1, k2, k, nF);
      lastSingularColumn = kcol[F.tailCol[info[0] - 1]];
      lastSingularRow = (info[2] >= 0 && info[2] < k2) ? rrows[F.tailRow[info[2]]] : -1;
      return -1;
    }
  }
  // ---- polish: the explicit inverse of an ill-conditioned tail carries a residual of cond(S) * eps; Newton-Schulz
  // steps X += X (I - S X) square it (the bases of the bench LP pass through condition numbers of 1e10 and more)
  hLu.k2 = k2;  // (refineInverse sizes its work from the descriptor)
  // One GEMM measures max |I - S X|; below the tolerance nothing else happens.  A Newton-Schulz step (one more
  // GEMM) squares the residual, so after a step from r the next measurement is only taken when r^2 is still
  // above the tolerance.
  for (int step = 0; k2 && step < luPolish; step++) {
    const int prc = refineInverse(true, luPolishTolerance);
    luLastResidual = lastResidual;
    if (prc != 0)
      break;  // good enough already (3), not finite / too far (1, 2): keep what we have
    numberPolishSteps++;
    if (lastResidual * lastResidual < luPolishTolerance) {
      luLastResidual = lastResidual * lastResidual;  // (estimate: the step squares it)
      break;
    }
  }
  // (the GEMM library probes pointers with runtime calls whose failures stay behind as the thread's "last error":
  // the engine's own launches up to here were checked by invertWork)
  {
    hipError_t pending = hipGetLastError();
    if (pending != hipSuccess && logLevel > 0)
      fprintf(stderr, "clpgpu: (cleared after the tail polish: %s)\n", hipGetErrorString(pending));
  }
  if (k2)
    hipLaunchKernelGGL(k_lu_transpose_tail, dim3(cdiv(k2, 32), cdiv(k2, 32)), dim3(256), 0, stream, D, k2, D.workX);
  hLu.MinvT = D.workX;  // (free until the next refactorization; allocNucleus drops the graphs when it moves)
  const auto t2 = std::chrono::steady_clock::now();
  // ---- positions: front pivot f puts column fcol[f] at the position of row frow[f]; the tail as the
  // dense pivoting decided
  std::vector<int> posOfCol(k), rowOfLocal(rrows);
  for (int f = 0; f < nF; f++)
    posOfCol[F.fcol[f]] = rrows[F.frow[f]];
  for (int tc = 0; tc < k2; tc++)
    posOfCol[F.tailCol[tc]] = rrows[F.tailRow[perm[tc]]];
  for (int i = 0; i < m; i++)
    if (localOfRow[i] < 0)
      pivotVariable[i] = n + i;
  for (int c = 0; c < k; c++)
    pivotVariable[posOfCol[c]] = kcol[c];
  // ---- the front's triangular factors as EXPLICIT sparse inverses, each applied in one gather pass:
  //   L^-1  (rows; unit diagonal)            y = L^-1 v          -> front rows y_F, tail rows the GEMV's input
  //   X = U11^-1 [ I | -U12 ]  (rows)        x_F = X [y_F ; x_T]
  // and their transposes for the BTRAN.  On the bench LP they hold < 2x the entries of L and U (measured:
  // nucleus 11 000: L 44 k -> L^-1 75 k; U 46 k -> X 91 k); a front whose inverses fill in beyond
  // luInverseFillCap falls back to the explicit inverse of the whole nucleus for this refactorization.
  std::vector<int> pivOfRow(k, -1), pivOfCol(k, -1), tailSlotOfRow(k, -1), tailSlotOfCol(k, -1);
  for (int f = 0; f < nF; f++) {
    pivOfRow[F.frow[f]] = f;
    pivOfCol[F.fcol[f]] = f;
  }
  for (int ts = 0; ts < k2; ts++) {
    tailSlotOfRow[F.tailRow[ts]] = ts;
    tailSlotOfCol[F.tailCol[ts]] = ts;
  }
  const size_t fillCap = (size_t)luInverseFillCap;
  // L by row (entries in pivot order)
  std::vector<int> lcnt(k + 1, 0);
  for (size_t e = 0; e < F.lRow.size(); e++)
    lcnt[F.lRow[e] + 1]++;
  for (int i = 0; i < k; i++)
    lcnt[i + 1] += lcnt[i];
  std::vector<int> lrp(F.lRow.size()), lat(lcnt.begin(), lcnt.end() - 1);
  std::vector<double> lrv(F.lRow.size());
  for (int f = 0; f < nF; f++)
    for (int e = F.lStart[f]; e < F.lStart[f + 1]; e++) {
      const int lr = F.lRow[e];
      lrp[lat[lr]] = F.frow[f];
      lrv[lat[lr]] = F.lVal[e];
      lat[lr]++;
    }
  // sparse accumulator
  std::vector<double> spa(k + k2, 0.0);
  std::vector<int> spaMark(k + k2, 0), spaList;
  auto spaAdd = [&](int j, double v) {
    if (!spaMark[j]) {
      spaMark[j] = 1;
      spaList.push_back(j);
    }
    spa[j] += v;
  };
  // rows of L^-1 - I: y_r = v_r + sum_j linv[r][j] v_j
  std::vector<std::vector<std::pair<int, double>>> linv(k);
  size_t fill = 0;
  bool tooMuch = false;
  auto lrow = [&](int r) {
    if (lcnt[r + 1] == lcnt[r])
      return;
    for (int e = lcnt[r]; e < lcnt[r + 1]; e++) {
      const int p = lrp[e];
      const double mult = lrv[e];
      spaAdd(p, -mult);
      for (const auto &pr : linv[p])
        spaAdd(pr.first, -mult * pr.second);
    }
    // (entries stay in the order the recurrence produced them: deterministic, and the gather kernels do not care)
    linv[r].reserve(spaList.size());
    for (int j : spaList) {
      if (spa[j] != 0.0)
        linv[r].push_back({ j, spa[j] });
      spa[j] = 0.0;
      spaMark[j] = 0;
    }
    spaList.clear();
    fill += linv[r].size();
  };
  for (int f = 0; f < nF && fill <= fillCap; f++)
    lrow(F.frow[f]);
  for (int ts = 0; ts < k2 && fill <= fillCap; ts++)
    lrow(F.tailRow[ts]);
  // rows of X: x_f = sum xinv[f][j] * (j < k ? y[local row j] : x_T[j - k])
  std::vector<std::vector<std::pair<int, double>>> xinv(nF);
  for (int f = nF - 1; f >= 0 && fill <= fillCap; f--) {
    const double ip = 1.0 / F.fpiv[f];
    spaAdd(F.frow[f], ip);
    for (int e = F.uStart[f]; e < F.uStart[f + 1]; e++) {
      const int cc = F.uCol[e];
      const double u = F.uVal[e] * ip;
      const int f2 = pivOfCol[cc];
      if (f2 >= 0) {
        for (const auto &pr : xinv[f2])
          spaAdd(pr.first, -u * pr.second);
      } else {
        spaAdd(k + tailSlotOfCol[cc], -u);
      }
    }
    // (entries stay in the order the recurrence produced them: deterministic, and the gather kernels do not care)
    xinv[f].reserve(spaList.size());
    for (int j : spaList) {
      if (spa[j] != 0.0)
        xinv[f].push_back({ j, spa[j] });
      spa[j] = 0.0;
      spaMark[j] = 0;
    }
    spaList.clear();
    fill += xinv[f].size();
  }
  tooMuch = fill > fillCap;
  if (tooMuch) {
    if (logLevel > 0)
      fprintf(stderr, "clpgpu: LU front of %d pivots: explicit inverses exceed %zu entries, explicit nucleus inverse instead\n", nF, fillCap);
    return -7;  // factorizeOnce falls back to the explicit inverse of the whole nucleus
  }
  LuTriHost Lf, Ub, Utf, Ltb;
  auto flat = [&](LuTriHost &h, int nItems, const std::vector<int> &tg, const std::vector<int> &sr, const std::vector<double> &dv,
                  const std::vector<int> &es, const std::vector<int> &ei, const std::vector<double> &ev) {
    h.levelStart.assign(2, 0);
    h.levelStart[1] = nItems;
    h.tgt = tg;
    h.src = sr;
    h.div = dv;
    h.entStart = es;
    h.entIdx = ei;
    h.entVal = ev;
  };
  {
    // forward: item = local row r; source vectors are given BY ROW (global row indices)
    std::vector<int> tg(k), sr(k), es(k + 1, 0), ei;
    std::vector<double> dv(k, 1.0), ev;
    for (int r = 0; r < k; r++) {
      tg[r] = tailSlotOfRow[r] >= 0 ? k + tailSlotOfRow[r] : r;
      sr[r] = rrows[r];
      for (const auto &pr : linv[r]) {
        ei.push_back(rrows[pr.first]);
        ev.push_back(-pr.second);
      }
      es[r + 1] = (int)ei.size();
    }
    flat(Lf, k, tg, sr, dv, es, ei, ev);
  }
  {
    // backward: item = front pivot f; out = (y[frow f] - sum val * src[idx]) / piv with val = -piv * xinv (off-diagonal)
    std::vector<int> tg(nF), sr(nF), es(nF + 1, 0), ei;
    std::vector<double> dv(nF), ev;
    for (int f = 0; f < nF; f++) {
      tg[f] = F.fcol[f];
      sr[f] = F.frow[f];
      double diag = 0.0;
      for (const auto &pr : xinv[f])
        if (pr.first == F.frow[f])
          diag = pr.second;
      dv[f] = 1.0 / diag;
      for (const auto &pr : xinv[f])
        if (pr.first != F.frow[f]) {
          ei.push_back(pr.first);
          ev.push_back(-pr.second / diag);
        }
      es[f + 1] = (int)ei.size();
    }
    flat(Ub, nF, tg, sr, dv, es, ei, ev);
  }
  {
    // BTRAN front: z_f = sum_f' xinv[f'][frow f] t[fcol f'] ; tail: zt[tc] = t[tailCol tc] + sum_f' xinv[f'][k + tc] t[fcol f']
    // (transposes of the rows above; item order: front pivots, then tail column slots)
    const int nI = nF + k2;
    std::vector<int> cnt(nI + 1, 0);
    auto itemOf = [&](int j) { return j < k ? pivOfRow[j] : nF + (j - k); };
    for (int f = 0; f < nF; f++)
      for (const auto &pr : xinv[f])
        if (pr.first != F.frow[f])
          cnt[itemOf(pr.first) + 1]++;
    for (int i = 0; i < nI; i++)
      cnt[i + 1] += cnt[i];
    std::vector<int> ei(cnt[nI]), at(cnt.begin(), cnt.end() - 1), tg(nI), sr(nI);
    std::vector<double> ev(cnt[nI]), dv(nI, 1.0);
    for (int f = 0; f < nF; f++) {
      double diag = 0.0;
      for (const auto &pr : xinv[f])
        if (pr.first == F.frow[f])
          diag = pr.second;
      tg[f] = F.frow[f];
      sr[f] = F.fcol[f];
      dv[f] = 1.0 / diag;
    }
    for (int tc = 0; tc < k2; tc++) {
      tg[nF + tc] = k + tc;
      sr[nF + tc] = F.tailCol[tc];
    }
    for (int f = 0; f < nF; f++)
      for (const auto &pr : xinv[f])
        if (pr.first != F.frow[f]) {
          const int it = itemOf(pr.first);
          ei[at[it]] = F.fcol[f];
          // out = (t[src] - sum val t[idx]) / div  must equal  diag_it * t[src] + sum xinv * t[idx]
          ev[at[it]] = -pr.second * dv[it];
          at[it]++;
        }
    flat(Utf, nI, tg, sr, dv, cnt, ei, ev);
  }
  {
    // BTRAN back: y_r = z_r + sum_r' linv[r'][r] z_r'   (transposed rows of L^-1 - I); item = local row r
    std::vector<int> cnt(k + 1, 0);
    for (int r2 = 0; r2 < k; r2++)
      for (const auto &pr : linv[r2])
        cnt[pr.first + 1]++;
    for (int i = 0; i < k; i++)
      cnt[i + 1] += cnt[i];
    std::vector<int> ei(cnt[k]), at(cnt.begin(), cnt.end() - 1), tg(k), sr(k);
    std::vector<double> ev(cnt[k]), dv(k, 1.0);
    for (int r = 0; r < k; r++)
      tg[r] = sr[r] = r;
    for (int r2 = 0; r2 < k; r2++)
      for (const auto &pr : linv[r2]) {
        ei[at[pr.first]] = r2;
        ev[at[pr.first]] = -pr.second;
        at[pr.first]++;
      }
    flat(Ltb, k, tg, sr, dv, cnt, ei, ev);
  }
  luLastInverseFill = (long)fill;
  // ---- frozen slack rows (their U rows) by row
  std::vector<int> sRowIndex, sRowOf(m, -1);
  for (int i = 0; i < m; i++)
    if (localOfRow[i] < 0) {
      sRowOf[i] = (int)sRowIndex.size();
      sRowIndex.push_back(i);
    }
  const int ns = (int)sRowIndex.size();
  std::vector<int> sRowStart(ns + 1, 0);
  for (int e = 0; e < (int)sColRow.size(); e++)
    sRowStart[sRowOf[sColRow[e]] + 1]++;
  for (int s = 0; s < ns; s++)
    sRowStart[s + 1] += sRowStart[s];
  std::vector<int> sRowCol(sColRow.size()), at(sRowStart.begin(), sRowStart.end() - 1);
  std::vector<double> sRowVal(sColRow.size());
  for (int c = 0; c < k; c++)
    for (int e = sColStart[c]; e < sColStart[c + 1]; e++) {
      const int s = sRowOf[sColRow[e]];
      sRowCol[at[s]] = c;
      sRowVal[at[s]] = sColVal[e];
      at[s]++;
    }
  // ---- upload
  LuDev &L = hLu;
  L.k = k;
  L.nF = nF;
  L.k2 = k2;
  L.ns = ns;
  L.kpad = (k + 63) & ~63;
  int tcapWant = std::min(luMaxPivots, LU_TCAP_MAX - 1) + 1;
  rc |= luUploadTri(Lf, L.Lf, LB_TRI + 0);
  rc |= luUploadTri(Ub, L.Ub, LB_TRI + 7);
  rc |= luUploadTri(Utf, L.Utf, LB_TRI + 14);
  rc |= luUploadTri(Ltb, L.Ltb, LB_TRI + 28);
  int *ip = nullptr;
  double *dp = nullptr;
  rc |= luBuf[LB_ROWOFLOCAL].put(this, rowOfLocal, ip);
  L.rowOfLocal = ip;
  rc |= luBuf[LB_POSOFCOL].put(this, posOfCol, ip);
  L.posOfCol = ip;
  rc |= luBuf[LB_TAILROW].put(this, F.tailRow, ip);
  L.tailRow = ip;
  rc |= luBuf[LB_TAILCOL].put(this, F.tailCol, ip);
  L.tailCol = ip;
  rc |= luBuf[LB_SROWINDEX].put(this, sRowIndex, ip);
  L.sRowIndex = ip;
  rc |= luBuf[LB_SROWSTART].put(this, sRowStart, ip);
  L.sRowStart = ip;
  rc |= luBuf[LB_SROWCOL].put(this, sRowCol, ip);
  L.sRowCol = ip;
  rc |= luBuf[LB_SROWVAL].put(this, sRowVal, dp);
  L.sRowVal = dp;
  rc |= luBuf[LB_SCOLSTART].put(this, sColStart, ip);
  L.sColStart = ip;
  rc |= luBuf[LB_SCOLROW].put(this, sColRow, ip);
  L.sColRow = ip;
  rc |= luBuf[LB_SCOLVAL].put(this, sColVal, dp);
  L.sColVal = dp;
  // work vectors and the eta file (sized once per problem / capacity)
  void *vp = nullptr;
  rc |= luBuf[LB_WR].need(this, sizeof(double) * 3 * (size_t)L.kpad, vp);
  L.wr = (double *)vp;
  rc |= luBuf[LB_XC].need(this, sizeof(double) * 3 * (size_t)L.kpad, vp);
  L.xc = (double *)vp;
  rc |= luBuf[LB_TCV].need(this, sizeof(double) * (size_t)L.kpad, vp);
  L.tcv = (double *)vp;
  rc |= luBuf[LB_X0].need(this, sizeof(double) * 3 * (size_t)m, vp);
  L.x0 = (double *)vp;
  rc |= luBuf[LB_CP].need(this, sizeof(double) * (size_t)m, vp);
  L.cp = (double *)vp;
  rc |= luBuf[LB_Y].need(this, sizeof(double) * (size_t)m, vp);
  L.y = (double *)vp;
  if (tcapWant > L.tcap || !luBuf[LB_H].p) {
    L.tcap = tcapWant;
    dropGraph();  // launch extents of the eta-file kernels follow the capacity
    rc |= luBuf[LB_H].need(this, sizeof(double) * (size_t)L.tcap * (size_t)m, vp);
    rc |= luBuf[LB_G].need(this, sizeof(double) * (size_t)L.tcap * (size_t)L.tcap, vp);
    rc |= luBuf[LB_GT].need(this, sizeof(double) * (size_t)L.tcap * (size_t)L.tcap, vp);
    rc |= luBuf[LB_P].need(this, sizeof(int) * (size_t)L.tcap, vp);
    rc |= luBuf[LB_PREV].need(this, sizeof(int) * (size_t)L.tcap, vp);
    rc |= luBuf[LB_NEXT].need(this, sizeof(int) * (size_t)L.tcap, vp);
    rc |= luBuf[LB_S].need(this, sizeof(double) * 3 * (size_t)L.tcap, vp);
    rc |= luBuf[LB_GV].need(this, sizeof(double) * (size_t)L.tcap, vp);
    rc |= luBuf[LB_DV].need(this, sizeof(double) * (size_t)L.tcap, vp);
  }
  L.H = (double *)luBuf[LB_H].p;
  L.G = (double *)luBuf[LB_G].p;
  L.GT = (double *)luBuf[LB_GT].p;
  L.P = (int *)luBuf[LB_P].p;
  L.prevSame = (int *)luBuf[LB_PREV].p;
  L.nextSame = (int *)luBuf[LB_NEXT].p;
  L.s = (double *)luBuf[LB_S].p;
  L.g = (double *)luBuf[LB_GV].p;
  L.d = (double *)luBuf[LB_DV].p;
  rc |= luBuf[LB_LASTOFPOS].need(this, sizeof(int) * (size_t)m, vp);
  L.lastOfPos = (int *)vp;
  // compact eta file (device_state.h): slots of the positions that hold a structural now, in position order
  {
    std::vector<int> cslotOfPos(m, -1), posOfBasicCol(n, -1);
    int count = 0;
    for (int p = 0; p < m; p++)
      if (pivotVariable[p] < n) {
        cslotOfPos[p] = count++;
        posOfBasicCol[pivotVariable[p]] = p;
      }
    const int ldc = std::min(m, (count + L.tcap + 63) & ~63);
    std::vector<int> posOfCslot(ldc, -1);
    for (int p = 0; p < m; p++)
      if (cslotOfPos[p] >= 0)
        posOfCslot[cslotOfPos[p]] = p;
    rc |= luBuf[LB_CSLOT].put(this, cslotOfPos, ip);
    L.cslotOfPos = ip;
    rc |= luBuf[LB_POSOFCSLOT].put(this, posOfCslot, ip);
    L.posOfCslot = ip;
    rc |= luBuf[LB_POSOFBASICCOL].put(this, posOfBasicCol, ip);
    L.posOfBasicCol = ip;
    L.ldc = ldc;
    L.ncs0 = count;
    rc |= luBuf[LB_SROWOF].put(this, sRowOf, ip);
    L.sRowOf = ip;
    L.Hc = nullptr;
    L.xK = nullptr;
    if (luCompactEta) {
      rc |= luBuf[LB_HC].need(this, sizeof(double) * (size_t)L.tcap * (size_t)ldc, vp);
      L.Hc = (double *)vp;
      rc |= luBuf[LB_XK].need(this, sizeof(double) * 4 * (size_t)ldc, vp);
      L.xK = (double *)vp;
    }
    hCtrl->luCompactCount = count;
    hCtrl->luCompactOn = (luCompactEta && L.Hc && L.xK) ? 1 : 0;
  }
  if (rc)
    return rc;
  if (checkLaunches("factorizeLu (uploads)"))
    return -99;
  // the descriptor itself sits in device memory at a fixed address: captured launch graphs stay valid
  if (!dLu) {
    rc |= dalloc(dLu, 1);
    dropGraph();
  }
  rc |= h2d(dLu, &hLu, 1);
  D.lu = dLu;
  D.luMode = 1;
  hipLaunchKernelGGL(k_lu_reset, dim3(cdiv(m, 256)), dim3(256), 0, stream, D);
  if (checkLaunches("factorizeLu (reset)"))
    return -99;
  // ---- the explicit-inverse bookkeeping is switched off: no row or column has a slot
  if (!luSlotsCleared) {
    std::vector<int> minusM(m, -1), minusN(n, -1);
    rc |= h2d(D.slotOfRow, minusM.data(), m);
    rc |= h2d(D.slotOfCol, minusN.data(), n);
    rc |= h2d(D.posOfSlack, minusM.data(), m);
    luSlotsCleared = true;
  }
  rc |= h2d(D.pivotVariable, pivotVariable.data(), m);
  rc |= rebuildRowCopyIfNeeded();
  if (hCtrl->luCompactOn)  // the basic entries of the row copy carry their column's compact slot (as they carry the col-slot under the explicit inverse)
    hipLaunchKernelGGL(k_cslot_rebuild_lu, dim3(cdiv(m, 256)), dim3(256), 0, stream, D);
  luActive = true;
  kNucleus = k;
  pivots = 0;
  hCtrl->k = k2;
  hCtrl->pivots = 0;
  hCtrl->kcap = kcap;
  rc |= checkLaunches("factorizeLu");
  const auto t3 = std::chrono::steady_clock::now();
  luFrontSeconds += std::chrono::duration<double>(t1 - t0).count();
  luInvertSeconds += std::chrono::duration<double>(t2 - t1).count();
  luBuildSeconds += std::chrono::duration<double>(t3 - t2).count();
  luFactorizations++;
  // length of the eta file until the next scheduled refactorization: a pivot with t etas streams 8 m t bytes of H,
  // a refactorization costs R seconds -> the cost per pivot R / T + a T / 2 is least at T = sqrt(2 R / a)
  // (a = seconds per eta and pivot at the rate the eta kernel reaches); option "lu_max_pivots" caps it
  {
    // R from a cost MODEL of this refactorization, not from the clock: a measured R makes the refactorization points -- and with
    // them the whole pivot sequence of an ill-conditioned LP -- differ from run to run (two runs of config 4 with the same options
    // stood 38 000 objective units apart at pivot 16 000, profiles/r04_objective_race.md).  Fitted to the measured phases on the
    // MI355X box (profiles/r03_lu_final_kernel_stats.txt: k = 11 438, tail 7 162 -> host front 25 ms, tail inversion + polish
    // 179 ms, build + upload 8 ms): host front ~ 2.2 us per nucleus column, tail ~ 4.9e-13 k2^3 s, 2 ms fixed.
    const double R = 2.0e-3 + 2.2e-6 * (double)k + 4.9e-13 * (double)k2 * (double)k2 * (double)k2;
    luRefactorSeconds = R;
    // (with the compact copy a pivot streams the etas over the ~k structural positions only)
    const double a = 8.0 * (double)(hCtrl->luCompactOn ? std::min(m, k + 512) : m) / 3.0e12;
    int T = (int)sqrt(2.0 * luRefactorSeconds / a);
    T = std::min(std::max(T, luMinPivots), std::min(luMaxPivots, hLu.tcap - 1));
    luEtaLimit = luAdaptive ? T : std::min(luMaxPivots, hLu.tcap - 1);
  }
  luLastFront = nF;
  luLastTail = k2;
  if (logLevel > 1)
    fprintf(stderr, "clpgpu: LU factorization: nucleus %d = front %d (L %zu, U %zu nz; explicit inverses %ld nz) + dense tail %d (S %zu nz); host %.1f ms, "
                    "inversion %.1f ms (max |I - S X| %.2g after %d Newton-Schulz steps so far), build+upload %.1f ms\n",
            k, nF, F.lRow.size(), F.uCol.size(), luLastInverseFill, k2, F.sVal.size(),
            1e3 * std::chrono::duration<double>(t1 - t0).count(), 1e3 * std::chrono::duration<double>(t2 - t1).count(), luLastResidual, numberPolishSteps,
            1e3 * std::chrono::duration<double>(t3 - t2).count());
  return rc;
}

// generic solves in LU mode (refactorization boundaries, plug-in calls): one or two right-hand sides by row in,
// results by basis position out
int clpgpu_context::luFtran(const double *v0, const double *v1, double *o0, double *o1)
{
  const int k2 = hLu.k2, ns = hLu.ns;
  const int nrhs = v1 ? 2 : 1;
  const int gm = cdiv(m, 256);
  hipLaunchKernelGGL(k_lu_fwd, dim3(cdiv(m, 4)), dim3(256), 0, stream, D, 0, v0, v1, (const double *)nullptr, D.slotA, v1 ? D.slotB : (double *)nullptr,
                     (double *)nullptr);
  if (k2)
    hipLaunchKernelGGL(k_gemv2, dim3(cdiv(k2, 4)), dim3(256), 0, stream, D, (const double *)D.slotA, v1 ? (const double *)D.slotB : (const double *)nullptr,
                       D.slotC, v1 ? D.slotD : (double *)nullptr, 0);
  hipLaunchKernelGGL(k_lu_bwd, dim3(cdiv(m, 4)), dim3(256), 0, stream, D, 0, (const double *)D.slotC, (const double *)D.slotD, (const double *)nullptr, 1,
                     v1 ? 1 : 0, 0);
  if (ns)
    hipLaunchKernelGGL(k_lu_slack, dim3(cdiv(m, 32), nrhs), dim3(256), 0, stream, D, 0, v0, v1, (const double *)nullptr, 1, v1 ? 1 : 0, 0);
  hipLaunchKernelGGL(k_lu_pf_s, dim3(128), dim3(256), 0, stream, D, 0, 1, v1 ? 1 : 0, 0);
  hipLaunchKernelGGL(k_lu_pf_apply, dim3(cdiv(m, 256)), dim3(256), sizeof(double) * 3 * (size_t)hLu.tcap, stream, D, o0, o1, (double *)nullptr);
  return 0;
}

int clpgpu_context::luBtran(const double *cPos, double *yRow)
{
  const int k = hLu.k, k2 = hLu.k2, ns = hLu.ns, tcap = hLu.tcap;
  hipLaunchKernelGGL(k_lu_pf_gdot, dim3(64), dim3(256), 0, stream, D, cPos);
  hipLaunchKernelGGL(k_lu_pf_d, dim3(cdiv(tcap, 4)), dim3(256), 0, stream, D, 0);
  hipLaunchKernelGGL(k_lu_cprime, dim3(1), dim3(1024), 0, stream, D, 0, cPos);
  hipLaunchKernelGGL(k_lu_bt_gather, dim3(cdiv(m, 4)), dim3(256), 0, stream, D, 0, yRow);
  hipLaunchKernelGGL(k_lu_bt_front, dim3(cdiv(m, 4)), dim3(256), 0, stream, D, 0, D.slotA);
  if (k2)
    hipLaunchKernelGGL(k_lu_gemvT, dim3(cdiv(k2, 16)), dim3(256), 0, stream, D, 0, (const double *)D.slotA);
  hipLaunchKernelGGL(k_lu_bt_back, dim3(cdiv(m, 4)), dim3(256), 0, stream, D, 0, yRow);
  return 0;
}

// the LU-mode stretch of one pivot's chain: BTRAN (after CHUZR) ...
void clpgpu_context::luLaunchBtran()
{
  const int k = hLu.k, ns = hLu.ns, tcap = hLu.tcap, kc = kcap;
  if (luFold & 2) {
    KL("k_lu_pf_d", k_lu_pf_d, dim3(cdiv(tcap, 4)), dim3(256), 0, stream, D, 2);  // + c' in the workgroup that finishes last
  } else {
    KL("k_lu_pf_d", k_lu_pf_d, dim3(cdiv(tcap, 4)), dim3(256), 0, stream, D, 1);
    KL("k_lu_cprime", k_lu_cprime, dim3(1), dim3(1024), 0, stream, D, 1, (const double *)nullptr);
  }
  KL("k_lu_bt_gather", k_lu_bt_gather, dim3(cdiv(m, 4)), dim3(256), 0, stream, D, 1, (double *)nullptr);
  KL("k_lu_bt_front", k_lu_bt_front, dim3(cdiv(m, 4)), dim3(256), 0, stream, D, 1, D.slotA);
  KL("k_lu_gemvT", k_lu_gemvT, dim3(cdiv(kc, LUG_ROWS * (luGemvThreads >> 6))), dim3(luGemvThreads), 0, stream, D, 1, (const double *)D.slotA);
  KL("k_lu_bt_back", k_lu_bt_back, dim3(cdiv(m, 4)), dim3(256), 0, stream, D, 1, (double *)nullptr);
}
// ... and the three FTRANs (entering column, rho, flip rhs) up to the scatter with the eta file applied
void clpgpu_context::luLaunchFtran(int gm, int parity)
{
  const int ns = hLu.ns, kc = kcap;
  KL("k_lu_fwd", k_lu_fwd, dim3(cdiv(m, 4)), dim3(256), 0, stream, D, 1, (const double *)D.vecV1, (const double *)D.rho, (const double *)D.flipRhs, D.slotV1,
     D.rhoSlotF, D.flipSlot);
  KL("k_lu_gemv3", k_lu_gemv3, dim3(cdiv(kc, LUG_ROWS * (luGemvThreads >> 6))), dim3(luGemvThreads), 0, stream, D);
  KL("k_lu_bwd", k_lu_bwd, dim3(cdiv(m, 4)), dim3(256), 0, stream, D, 1, (const double *)D.slotC, (const double *)D.slotD, (const double *)D.slotE, 1, 1, 1);
  // (compact eta file: only the positions whose slack left since the refactorization still need their B0 value -- for s = G x0[P];
  // the untouched slack positions get their final value from their own rows in k_ftran_scatter3_lu)
  KL("k_lu_slack", k_lu_slack, dim3(hCtrl->luCompactOn ? cdiv(hLu.tcap, 32) : cdiv(m, 32), 3), dim3(256), 0, stream, D, 1, (const double *)D.vecV1, (const double *)D.rho,
       (const double *)D.flipRhs, 1, 1, 1);
  KL("k_lu_pf_s", k_lu_pf_s, dim3(luPfsBlocks), dim3(256), 0, stream, D, 1, 1, 1, 1);
  {
    // positions per workgroup: one round of workgroups over the 256 CUs (m = 50 000: 250 workgroups of 200 positions, not 196 of 256)
    const int compact = hCtrl->luCompactOn;
    const int ppb = compact ? luScatterPpb : std::min(256, std::max(64, (cdiv(m, 256) + 7) & ~7));
    if (compact)  // x0 -= Hc s over the slots (the structural positions): 8 (k + conversions) t bytes instead of 8 m t
      KL("k_lu_eta_apply", k_lu_eta_apply, dim3(4096 / (luGemvThreads >> 6)), dim3(luGemvThreads), 0, stream, D);
    KL("k_ftran_scatter3_lu", k_ftran_scatter3_lu, dim3(cdiv(m, ppb)), dim3(256), 0, stream, D, gm, parity, ppb, compact);
  }
}

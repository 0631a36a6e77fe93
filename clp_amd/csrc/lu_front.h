// lu_front.h -- host side of the sparse part of the basis factorization (N2): a right-looking
// Markowitz LU of the nucleus C = A[R,K] with threshold pivoting that STOPS when the active
// submatrix has filled in, leaving a dense Schur complement ("tail") for the matrix cores.
//
// What it stands in for (reference, file:line under /root/reference):
//   CoinAbcBaseFactorization::factorSparse   src/CoinAbcBaseFactorization2.cpp:18   (Markowitz search over
//       rows / columns of increasing count, threshold test against the largest element, fill-in)
//   CoinAbcBaseFactorization::pivotColumnSingleton / pivotRowSingleton  ...1.cpp:2589 (count-1 pivots first)
//   wantToGoDense / factorDense              ...1.cpp:2409-2462, ...2.cpp:976      (switch to a dense tail when
//       the remaining block is dense enough; here the dense tail is inverted on MFMA, engine.hip)
// This is a from-scratch implementation of the published algorithm (Markowitz 1957; Suhl & Suhl 1990),
// not a transliteration: row-wise value storage, column-wise pattern, count buckets, row-relative
// threshold.  Runs on the host at refactorization boundaries only (SURVEY.md section 7 step 6 allows a
// CPU factor + upload); everything per pivot is on the device (lu_kernels.hip).
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace clpgpu {

struct LuFront {
  // input size
  int k = 0;
  // front pivots in elimination order
  int nF = 0;
  std::vector<int> frow, fcol;   // local nucleus row / column of pivot f
  std::vector<double> fpiv;      // pivot value
  // L by pivot (column form): multipliers of pivot f on later rows
  std::vector<int> lStart, lRow;
  std::vector<double> lVal;
  // U by pivot (row form): entries of pivot row f on later columns
  std::vector<int> uStart, uCol;
  std::vector<double> uVal;
  // tail = what is left of the active submatrix
  int k2 = 0;
  std::vector<int> tailRow, tailCol;         // local nucleus rows / columns, ascending
  std::vector<int> sRow, sCol;               // S entries as (tail slot row, tail slot col)
  std::vector<double> sVal;
  long fill = 0;
  double seconds = 0.0;
};

// progress print of the elimination (environment LUDBG=1), read once
static inline bool luFrontDebug()
{
  static const bool on = getenv("LUDBG") != nullptr;
  return on;
}

// C given by columns: column c holds rows cRow[cStart[c] .. cStart[c+1]) (local nucleus rows 0..k-1).
// stopDensity: stop when nnz(active) > stopDensity * nActive^2; minTail: never look for pivots once the
// active block is this small; threshold u: |pivot| >= u * max|row|.
static inline void luFrontFactor(int k, const int *cStart, const int *cRow, const double *cVal, double stopDensity, int minTail,
                                 double threshold, double absTol, LuFront &F)
{
  F = LuFront();
  F.k = k;
  std::vector<std::vector<int>> rIdx(k), cIdx(k);
  std::vector<std::vector<double>> rVal(k);
  long actNnz = 0;
  for (int c = 0; c < k; c++)
    for (int p = cStart[c]; p < cStart[c + 1]; p++) {
      const int r = cRow[p];
      rIdx[r].push_back(c);
      rVal[r].push_back(cVal[p]);
      cIdx[c].push_back(r);
      actNnz++;
    }
  // count buckets (doubly linked), rows and columns separately
  std::vector<int> rHead(k + 2, -1), cHead(k + 2, -1), rNext(k, -1), rPrev(k, -1), cNext(k, -1), cPrev(k, -1);
  std::vector<char> rAct(k, 1), cAct(k, 1);
  auto link = [](std::vector<int> &head, std::vector<int> &next, std::vector<int> &prev, int i, int count) {
    prev[i] = -1;
    next[i] = head[count];
    if (head[count] >= 0)
      prev[head[count]] = i;
    head[count] = i;
  };
  auto unlink = [](std::vector<int> &head, std::vector<int> &next, std::vector<int> &prev, int i, int count) {
    if (prev[i] >= 0)
      next[prev[i]] = next[i];
    else
      head[count] = next[i];
    if (next[i] >= 0)
      prev[next[i]] = prev[i];
  };
  for (int i = 0; i < k; i++) {
    link(rHead, rNext, rPrev, i, (int)rIdx[i].size());
    link(cHead, cNext, cPrev, i, (int)cIdx[i].size());
  }
  std::vector<int> where(k, -1);
  std::vector<double> rowMax(k, -1.0);
  auto getRowMax = [&](int i) {
    if (rowMax[i] < 0.0) {
      double mx = 0.0;
      for (double v : rVal[i])
        mx = std::max(mx, fabs(v));
      rowMax[i] = mx;
    }
    return rowMax[i];
  };
  F.lStart.push_back(0);
  F.uStart.push_back(0);
  int nAct = k;
  int minCount = 1;
  const int maxTrials = 4;
  while (nAct > minTail) {
    if ((double)actNnz > stopDensity * (double)nAct * (double)nAct)
      break;
    // ---- Markowitz search: counts 1, 2, ... ; columns then rows of each count
    int bi = -1, bj = -1;
    double bestCost = 1.0e300, bestAbs = 0.0;
    int trials = 0;
    while (minCount <= nAct && rHead[minCount] < 0 && cHead[minCount] < 0)
      minCount++;
    // (a count can also drop below minCount after an elimination: handled by resetting below)
    for (int count = minCount; count <= nAct && trials < maxTrials; count++) {
      if (bestCost <= (double)(count - 1) * (double)(count - 1))
        break;
      for (int j = cHead[count]; j >= 0 && trials < maxTrials; j = cNext[j]) {
        // column j: the row with the fewest entries among those passing the threshold test
        bool any = false;
        for (int i : cIdx[j]) {
          double a = 0.0;
          const std::vector<int> &ri = rIdx[i];
          for (size_t q = 0; q < ri.size(); q++)
            if (ri[q] == j) {
              a = fabs(rVal[i][q]);
              break;
            }
          if (a < absTol || a < threshold * getRowMax(i))
            continue;
          const double cost = (double)(ri.size() - 1) * (double)(count - 1);
          if (cost < bestCost || (cost == bestCost && a > bestAbs)) {
            bestCost = cost;
            bestAbs = a;
            bi = i;
            bj = j;
          }
          any = true;
        }
        if (any)
          trials++;
      }
      if (bestCost <= (double)(count - 1) * (double)(count - 1))
        break;
      for (int i = rHead[count]; i >= 0 && trials < maxTrials; i = rNext[i]) {
        const double mx = getRowMax(i);
        bool any = false;
        for (size_t q = 0; q < rIdx[i].size(); q++) {
          const double a = fabs(rVal[i][q]);
          if (a < absTol || a < threshold * mx)
            continue;
          const int j = rIdx[i][q];
          const double cost = (double)(count - 1) * (double)(cIdx[j].size() - 1);
          if (cost < bestCost || (cost == bestCost && a > bestAbs)) {
            bestCost = cost;
            bestAbs = a;
            bi = i;
            bj = j;
          }
          any = true;
        }
        if (any)
          trials++;
      }
    }
    if (luFrontDebug() && (F.nF % 250 == 0)) {
      static double tl = 0;
      struct timespec ts;
      clock_gettime(CLOCK_MONOTONIC, &ts);
      const double tn = ts.tv_sec + 1e-9 * ts.tv_nsec;
      fprintf(stderr, "step %d nAct %d actNnz %ld bestCost %.0f minCount %d us/step %.1f\n", F.nF, nAct, actNnz, bestCost, minCount,
              (tn - tl) * 1e6 / 250);
      tl = tn;
    }
    if (bi < 0)
      break;  // nothing acceptable among the sparse candidates: the rest goes to the dense tail
    // ---- eliminate with pivot (bi, bj)
    const int pi = bi, pj = bj;
    std::vector<int> &prI = rIdx[pi];
    std::vector<double> &prV = rVal[pi];
    double piv = 0.0;
    for (size_t q = 0; q < prI.size(); q++)
      if (prI[q] == pj) {
        piv = prV[q];
        prI[q] = prI.back();
        prV[q] = prV.back();
        prI.pop_back();
        prV.pop_back();
        break;
      }
    // row pi leaves the column patterns
    unlink(rHead, rNext, rPrev, pi, (int)prI.size() + 1);
    rAct[pi] = 0;
    for (int c : prI) {
      std::vector<int> &ci = cIdx[c];
      const int old = (int)ci.size();
      for (size_t q = 0; q < ci.size(); q++)
        if (ci[q] == pi) {
          ci[q] = ci.back();
          ci.pop_back();
          break;
        }
      unlink(cHead, cNext, cPrev, c, old);
      link(cHead, cNext, cPrev, c, old - 1);
      if (old - 1 < minCount)
        minCount = std::max(1, old - 1);
    }
    // column pj leaves
    unlink(cHead, cNext, cPrev, pj, (int)cIdx[pj].size());
    cAct[pj] = 0;
    actNnz -= (long)prI.size() + 1;
    for (int i : cIdx[pj]) {
      if (i == pi)
        continue;
      std::vector<int> &ri = rIdx[i];
      std::vector<double> &rv = rVal[i];
      const int oldCount = (int)ri.size();
      double aij = 0.0;
      for (size_t q = 0; q < ri.size(); q++)
        if (ri[q] == pj) {
          aij = rv[q];
          ri[q] = ri.back();
          rv[q] = rv.back();
          ri.pop_back();
          rv.pop_back();
          break;
        }
      actNnz--;
      const double mult = aij / piv;
      F.lRow.push_back(i);
      F.lVal.push_back(mult);
      for (size_t q = 0; q < ri.size(); q++)
        where[ri[q]] = (int)q;
      for (size_t q = 0; q < prI.size(); q++) {
        const int c = prI[q];
        const double delta = mult * prV[q];
        if (where[c] >= 0) {
          rv[where[c]] -= delta;
        } else {
          ri.push_back(c);
          rv.push_back(-delta);
          const int old = (int)cIdx[c].size();
          cIdx[c].push_back(i);
          unlink(cHead, cNext, cPrev, c, old);
          link(cHead, cNext, cPrev, c, old + 1);
          actNnz++;
          F.fill++;
        }
      }
      for (size_t q = 0; q < ri.size(); q++)
        where[ri[q]] = -1;
      rowMax[i] = -1.0;
      const int newCount = (int)ri.size();
      if (newCount != oldCount) {
        unlink(rHead, rNext, rPrev, i, oldCount);
        link(rHead, rNext, rPrev, i, newCount);
      }
      if (newCount < minCount)
        minCount = std::max(1, newCount);
    }
    std::vector<int>().swap(cIdx[pj]);
    F.frow.push_back(pi);
    F.fcol.push_back(pj);
    F.fpiv.push_back(piv);
    for (size_t q = 0; q < prI.size(); q++) {
      F.uCol.push_back(prI[q]);
      F.uVal.push_back(prV[q]);
    }
    F.lStart.push_back((int)F.lRow.size());
    F.uStart.push_back((int)F.uCol.size());
    F.nF++;
    nAct--;
  }
  // ---- tail
  std::vector<int> slotOfRow(k, -1), slotOfCol(k, -1);
  for (int i = 0; i < k; i++)
    if (rAct[i]) {
      slotOfRow[i] = (int)F.tailRow.size();
      F.tailRow.push_back(i);
    }
  for (int j = 0; j < k; j++)
    if (cAct[j]) {
      slotOfCol[j] = (int)F.tailCol.size();
      F.tailCol.push_back(j);
    }
  F.k2 = (int)F.tailRow.size();
  for (int i = 0; i < k; i++)
    if (rAct[i])
      for (size_t q = 0; q < rIdx[i].size(); q++) {
        F.sRow.push_back(slotOfRow[i]);
        F.sCol.push_back(slotOfCol[rIdx[i][q]]);
        F.sVal.push_back(rVal[i][q]);
      }
}

}  // namespace clpgpu

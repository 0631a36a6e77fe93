// perturb_host.h -- ClpSimplexDual::perturb (src/ClpSimplexDual.cpp:6533-6957): cost perturbation of the nonbasic,
// non-fixed structurals.  Host code: it runs once or twice per solve on the rim arrays (at start-up, :335, or as the
// "kick" after 2(m+n) iterations, :488); the engine pushes the costs to the device afterwards.  Kept free of HIP so that
// tests/host/perturb_harness.cpp can run it on a CPU-only machine against the oracle's restatement.
//
// perturbation is ClpSimplex::perturbation_ and is updated as the reference does: entered with 50 (perturb when at most
// a quarter of the |costs| are distinct), 51-69 (fixed maximum fractions), 100 (treated as 50; the value the kick is
// entered with) or a value below 50 (10^value, "user is in charge", without the <= -10 experiments); left at 101
// (perturbed) or 100 (the costs are varied enough).  Row costs are left alone (:6745), the Cbc branches are not restated.
// Returns 1 where the reference would rather use primal (every cost zero, :6588), else 0.
#pragma once
#include <float.h>
#include <math.h>

#include <algorithm>
#include <vector>

namespace clpgpu {

struct PerturbRim {
  int m, n;
  const int *colStart;          // [n+1] of the (scaled) matrix the solve runs on
  const double *elem;
  const double *lower, *upper;  // working bounds [columns | rows]
  const unsigned char *status;  // ClpSimplex::Status in the low three bits (1 basic, 2 at upper)
  const double *objective;      // the objective as the caller gave it (before scaling), [n]
  double dualTolerance, largeValue;
  int numberIterations;
};

inline int perturbCosts(const PerturbRim &rim, int &perturbation, std::vector<double> &perturbationArray, unsigned int &seed, double *cost)
{
  const int m = rim.m, n = rim.n, N = rim.m + rim.n, numberIterations = rim.numberIterations;
  const long nnz = rim.colStart[n];
  const int *colStart = rim.colStart;
  const double *elem = rim.elem, *lower = rim.lower, *upper = rim.upper, *origObj = rim.objective;
  const unsigned char *status = rim.status;
  const double dualTolerance = rim.dualTolerance, largeValue = rim.largeValue;
  if (perturbation > 100)
    return 0;
  if (perturbation == 100)
    perturbation = 50;
  const int entered = perturbation;
  double size = 1.0e-20;       // "perturbation" of the reference
  double maxFraction = 1.0e-5; // largest fraction of a cost to move it by
  const double constantPart = 100.0 * dualTolerance;
  int longest = 0, shortest = m;
  double meanCost = 0.0;
  if (!numberIterations && perturbation >= 50) {
    // worth it? (:6562-6606) -- looks at the objective as the caller gave it, before scaling
    std::vector<double> magnitude(n);
    int nonZero = 0;
    for (int j = 0; j < n; j++) {
      magnitude[j] = fabs(origObj[j]);
      meanCost += magnitude[j];
      nonZero += magnitude[j] != 0.0;
    }
    meanCost = nonZero ? meanCost / (double)nonZero : 1.0;
    std::sort(magnitude.begin(), magnitude.end());
    int distinct = 1;
    for (int j = 1; j < n; j++)
      distinct += magnitude[j] != magnitude[j - 1];
    if (!nonZero && perturbation < 55)
      return 1;
    if (distinct * 4 > n) {
      perturbation = 100;  // the costs are varied enough
      return 0;
    }
  }
  for (int j = 0; j < n; j++) {
    const int length = colStart[j + 1] - colStart[j];
    if (lower[j] < upper[j] && length > 2) {
      longest = std::max(longest, length);
      shortest = std::min(shortest, length);
    }
  }
  if (perturbation >= 70)
    perturbation -= 20;
  if (perturbation > 50) {
    static const double fractionTable[11] = {1.0e-10, 1.0e-9, 1.0e-8, 1.0e-7, 1.0e-6, 1.0e-5, 1.0e-4, 1.0e-3, 1.0e-2, 1.0e-1, 1.0};
    maxFraction = fractionTable[std::min(perturbation - 51, 10)];
  }
  double smallestCost = 1.0e100;
  if (perturbation >= 50) {
    size = 1.0e-8;
    if (perturbation > 50 && perturbation < 60)
      size = std::max(1.0e-8, maxFraction);
    // are all the finite bounds of one magnitude (rows among themselves, columns among themselves)?
    bool oneMagnitude = true;
    double seen[2] = {0.0, 0.0};
    for (int pass = 0; pass < 2; pass++) {
      const int first = pass ? 0 : n, last = pass ? n : N;  // rows first, as the reference
      double &ref = seen[pass];
      for (int i = first; i < last; i++) {
        const double lo = lower[i], up = upper[i];
        if (lo < up) {
          const double c = fabs(cost[i]);
          size = std::max(size, c);
          if (c)
            smallestCost = std::min(smallestCost, c);
        }
        if (lo && lo > -1.0e10) {
          if (!ref)
            ref = fabs(lo);
          else if (fabs(fabs(lo) - ref) > 1.0e-7)
            oneMagnitude = false;
        }
        if (up && up < 1.0e10) {
          if (!ref)
            ref = fabs(up);
          else if (fabs(fabs(up) - ref) > 1.0e-7)
            oneMagnitude = false;
        }
      }
    }
    if (oneMagnitude) {
      // ... and the matrix one positive and one negative value (ClpPackedMatrix::rangeOfElements,
      // src/ClpPackedMatrix.cpp:5229)? then "really hit perturbation"
      double negNear = -DBL_MAX, negFar = 0.0, posNear = DBL_MAX, posFar = 0.0;
      for (long p = 0; p < nnz; p++) {
        const double v = elem[p];
        if (v > 0.0) {
          posNear = std::min(posNear, v);
          posFar = std::max(posFar, v);
        } else if (v < 0.0) {
          negNear = std::max(negNear, v);
          negFar = std::min(negFar, v);
        }
      }
      if (negNear == negFar && posNear == posFar)
        maxFraction = std::max(std::min(100.0 * maxFraction, 1.0e-3 * std::max(seen[0], seen[1])), maxFraction);
    }
    size = std::min(size, smallestCost / maxFraction);
  } else {
    maxFraction = 1.0e-1;
    size = pow(10.0, (double)perturbation);
  }
  // columns with more elements are made more expensive (:6775-6790; the "scale back" table needs
  // constantPerturbation < 99 dualTolerance, which it never is)
  static const double lengthWeight[11] = {1.0e-4, 1.0e-2, 5.0e-1, 1.0, 2.0, 5.0, 10.0, 20.0, 30.0, 40.0, 100.0};
  const double lengthFactor = longest ? 3.0 / (double)shortest : 1.0;
  const double floorValue = std::min(1.0e-2 * dualTolerance, maxFraction);
  double ceilingValue = std::max(1.0e3 * dualTolerance, maxFraction * meanCost);
  if (perturbation == 51)
    ceilingValue = std::max(dualTolerance, maxFraction);
  if (perturbationArray.empty()) {
    perturbationArray.resize(2 * (size_t)n + 1);
    for (int j = 0; j < 2 * n; j++) {
      seed = 1664525u * seed + 1013904223u;  // the generator the device's randomDouble uses (kernels.hip)
      perturbationArray[j] = ((double)seed) / 4294967296.0;
    }
  }
  double largestOnZero = 0.0, largestOnCost = 0.0;
  for (int j = 0; j < n; j++) {
    const int st = status[j] & 7;
    if (!(lower[j] < upper[j]) || st == 1)
      continue;
    const double current = cost[j];
    double delta = std::min(size, constantPart + maxFraction * (fabs(current) + 1.0e-1 * size + 1.0e-8));
    double cap = constantPart + 1.0e-1 * smallestCost;
    const double r0 = 0.5 + 0.5 * perturbationArray[2 * j], r1 = 0.5 + 0.5 * perturbationArray[2 * j + 1];
    if (lower[j] > -largeValue) {
      if (fabs(lower[j]) < fabs(upper[j])) {
        delta *= r0;
        cap *= r1;
      } else {
        delta = 0.0;
      }
    } else if (upper[j] < largeValue) {
      delta *= -r0;
      cap *= -r1;
    } else {
      delta = 0.0;
    }
    if (!delta)
      continue;
    int length = colStart[j + 1] - colStart[j];
    if (length > 3)
      length = std::max(3, (int)((double)length * lengthFactor));
    delta *= lengthWeight[std::min(length, 10)];
    delta = std::min(delta, cap);
    if (entered < 50 || entered > 60) {
      if (fabs(delta) <= dualTolerance)
        delta = 0.0;
    } else if (delta) {
      // into [floorValue, ceilingValue] by factors of ten
      if (fabs(delta) <= floorValue) {
        do
          delta *= 10.0;
        while (fabs(delta) <= floorValue);
      } else if (fabs(delta) > ceilingValue) {
        do
          delta *= 0.1;
        while (fabs(delta) > ceilingValue);
      }
    }
    if (!delta)
      continue;
    if (current)
      largestOnCost = std::max(largestOnCost, fabs(delta));
    else
      largestOnZero = std::max(largestOnZero, fabs(delta));
    cost[j] += (st == 2) ? -delta : delta;
  }
  if (largestOnZero > largestOnCost && largestOnCost) {
    // a zero cost must not end up moved further than the others (:6902-6917)
    const double limit = std::max(1.0e-8, largestOnCost);
    for (int j = 0; j < n; j++) {
      if (origObj[j])
        continue;
      double c = cost[j];
      while (fabs(c) > limit)
        c *= 0.5;
      cost[j] = c;
    }
  }
  perturbation = 101;
  return 0;
}

}  // namespace clpgpu

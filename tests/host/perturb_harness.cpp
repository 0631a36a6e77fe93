// Test harness for clp_amd/csrc/perturb_host.h (the engine's host restatement of ClpSimplexDual::perturb): reads a rim from a
// binary file, perturbs, writes the costs back; tests/test_perturb_host.py compares with the oracle's perturb() bit for bit.
#include "../../clp_amd/csrc/perturb_host.h"

#include <cstdio>
#include <cstdlib>
using namespace clpgpu;
template <class T> static bool get(FILE *f, std::vector<T> &v, long count)
{
  v.resize(count);
  return count == 0 || fread(v.data(), sizeof(T), count, f) == (size_t)count;
}
int main(int argc, char **argv)
{
  if (argc < 3)
    return 2;
  FILE *f = fopen(argv[1], "rb");
  long head[6];  // m, n, nnz, numberIterations, perturbation, seed
  double tol[2]; // dualTolerance, largeValue
  if (!f || fread(head, 8, 6, f) != 6 || fread(tol, 8, 2, f) != 2)
    return 3;
  const long m = head[0], n = head[1], nnz = head[2], N = m + n;
  std::vector<int> colStart;
  std::vector<double> elem, lower, upper, objective, cost;
  std::vector<unsigned char> status;
  if (!get(f, colStart, n + 1) || !get(f, elem, nnz) || !get(f, lower, N) || !get(f, upper, N) || !get(f, status, N) || !get(f, objective, n)
      || !get(f, cost, N))
    return 3;
  fclose(f);
  PerturbRim rim{(int)m, (int)n, colStart.data(), elem.data(), lower.data(), upper.data(), status.data(), objective.data(), tol[0], tol[1], (int)head[3]};
  int perturbation = (int)head[4];
  unsigned int seed = (unsigned int)head[5];
  std::vector<double> draws;
  long out[3];
  out[0] = perturbCosts(rim, perturbation, draws, seed, cost.data());
  out[1] = perturbation;
  out[2] = seed;
  // a second call on the perturbed rim must change nothing (perturbation_ > 100 returns at once)
  std::vector<double> again(cost);
  if (perturbation == 101 && (perturbCosts(rim, perturbation, draws, seed, again.data()) != 0 || again != cost))
    return 4;
  FILE *o = fopen(argv[2], "wb");
  fwrite(out, 8, 3, o);
  fwrite(cost.data(), 8, N, o);
  fclose(o);
  return 0;
}

// Test harness for clp_amd/csrc/lu_front.h (host Markowitz front): reads a square sparse matrix by columns, runs the
// front factorization and dumps the factors; tests/test_lu_front_host.py rebuilds the solve from them in numpy.
#include "../../clp_amd/csrc/lu_front.h"

#include <cstdio>
#include <cstdlib>
using namespace clpgpu;
template <class T> static void dump(FILE *f, const std::vector<T> &v)
{
  long n = (long)v.size();
  fwrite(&n, 8, 1, f);
  if (n)
    fwrite(v.data(), sizeof(T), n, f);
}
int main(int argc, char **argv)
{
  if (argc < 5)
    return 2;
  FILE *f = fopen(argv[1], "rb");
  long k, nnz;
  if (!f || fread(&k, 8, 1, f) != 1 || fread(&nnz, 8, 1, f) != 1)
    return 3;
  std::vector<int> cs(k + 1), cr(nnz);
  std::vector<double> cv(nnz);
  if (fread(cs.data(), 4, k + 1, f) != (size_t)(k + 1) || fread(cr.data(), 4, nnz, f) != (size_t)nnz || fread(cv.data(), 8, nnz, f) != (size_t)nnz)
    return 3;
  fclose(f);
  LuFront F;
  luFrontFactor((int)k, cs.data(), cr.data(), cv.data(), atof(argv[2]), atoi(argv[3]), 0.1, 1e-11, F);
  FILE *o = fopen(argv[4], "wb");
  dump(o, F.frow);
  dump(o, F.fcol);
  dump(o, F.fpiv);
  dump(o, F.lStart);
  dump(o, F.lRow);
  dump(o, F.lVal);
  dump(o, F.uStart);
  dump(o, F.uCol);
  dump(o, F.uVal);
  dump(o, F.tailRow);
  dump(o, F.tailCol);
  dump(o, F.sRow);
  dump(o, F.sCol);
  dump(o, F.sVal);
  fclose(o);
  printf("%d %d\n", F.nF, F.k2);
  return 0;
}

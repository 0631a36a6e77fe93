// STUB of src/ClpMatrixBase.hpp: the virtuals the adapters override, signatures as in the reference
// (tests/test_adapters.py checks each line below against the reference header when it is mounted)
#ifndef ClpMatrixBase_STUB
#define ClpMatrixBase_STUB
#include "CoinHelperFunctions.hpp"
class ClpSimplex;
class ClpModel;
class CoinIndexedVector;
class CoinPackedMatrix;
class ClpMatrixBase {
public:
  virtual ~ClpMatrixBase();
  virtual CoinPackedMatrix *getPackedMatrix() const = 0;
  virtual CoinBigIndex getNumElements() const = 0;
  virtual void times(double scalar,
    const double *COIN_RESTRICT x, double *COIN_RESTRICT y) const = 0;
  virtual void transposeTimes(double scalar,
    const double *COIN_RESTRICT x, double *COIN_RESTRICT y) const = 0;
  virtual void transposeTimes(const ClpSimplex *model, double scalar,
    const CoinIndexedVector *x,
    CoinIndexedVector *y,
    CoinIndexedVector *z) const = 0;
  virtual ClpMatrixBase *clone() const = 0;
};
#endif

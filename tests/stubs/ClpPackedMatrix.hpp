// STUB of src/ClpPackedMatrix.hpp
#ifndef ClpPackedMatrix_STUB
#define ClpPackedMatrix_STUB
#include "ClpMatrixBase.hpp"
class ClpPackedMatrix : public ClpMatrixBase {
public:
  ClpPackedMatrix(const ClpPackedMatrix &);
  virtual CoinPackedMatrix *getPackedMatrix() const;
  virtual CoinBigIndex getNumElements() const;
  virtual void times(double scalar,
    const double *x, double *y) const;
  virtual void transposeTimes(double scalar,
    const double *x, double *y) const;
  virtual void transposeTimes(const ClpSimplex *model, double scalar,
    const CoinIndexedVector *x,
    CoinIndexedVector *y,
    CoinIndexedVector *z) const;
  virtual ClpMatrixBase *clone() const;
};
#endif

// STUB (CoinUtils)
#ifndef CoinPackedMatrix_STUB
#define CoinPackedMatrix_STUB
#include "CoinHelperFunctions.hpp"
class CoinPackedMatrix {
public:
  const CoinBigIndex *getVectorStarts() const;
  const int *getIndices() const;
  const double *getElements() const;
};
#endif

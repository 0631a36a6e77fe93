// STUB (CoinUtils): the CoinIndexedVector members the adapters use (reference usage e.g.
// src/ClpPackedMatrix.cpp:780-811, src/ClpDualRowSteepest.cpp:431-461)
#ifndef CoinIndexedVector_STUB
#define CoinIndexedVector_STUB
#include "CoinHelperFunctions.hpp"
class CoinIndexedVector {
public:
  int getNumElements() const;
  void setNumElements(int value);
  int *getIndices() const;
  double *denseVector() const;
  bool packedMode() const;
  void setPackedMode(bool yesNo);
  void expand();
  int scan();
  int scanAndPack();
  void clear();
};
#endif

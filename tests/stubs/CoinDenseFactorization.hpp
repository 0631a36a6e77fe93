// STUB of CoinUtils' CoinDenseFactorization.hpp (CoinOtherFactorization; NOT in the reference tree --
// signatures follow the calls ClpFactorization makes on coinFactorizationB_, src/ClpFactorization.cpp:1683-1899,
// :2652-2665, :2750, :2827, :2946, :3012, and the in-tree twin CoinAbcAnyFactorization,
// src/CoinAbcDenseFactorization.hpp:22-433)
#ifndef CoinDenseFactorization_STUB
#define CoinDenseFactorization_STUB
#include "CoinHelperFunctions.hpp"
class CoinIndexedVector;
class CoinOtherFactorization {
public:
  CoinOtherFactorization();
  CoinOtherFactorization(const CoinOtherFactorization &other);
  virtual ~CoinOtherFactorization();
  virtual CoinOtherFactorization *clone() const = 0;
  inline int status() const { return status_; }
  inline int pivots() const { return numberPivots_; }
  inline int numberRows() const { return numberRows_; }
  virtual void getAreas(int numberRows, int numberColumns, CoinBigIndex maximumL, CoinBigIndex maximumU) = 0;
  virtual void preProcess() = 0;
  virtual int factor() = 0;
  virtual void postProcess(const int *sequence, int *pivotVariable) = 0;
  virtual void makeNonSingular(int *sequence, int numberColumns) = 0;
  virtual int replaceColumn(CoinIndexedVector *regionSparse, int pivotRow, double pivotCheck,
    bool checkBeforeModifying = false, double acceptablePivot = 1.0e-8) = 0;
  virtual int updateColumnFT(CoinIndexedVector *regionSparse, CoinIndexedVector *regionSparse2, bool noPermute = false) = 0;
  virtual int updateColumn(CoinIndexedVector *regionSparse, CoinIndexedVector *regionSparse2, bool noPermute = false) const = 0;
  virtual int updateTwoColumnsFT(CoinIndexedVector *regionSparse1, CoinIndexedVector *regionSparse2,
    CoinIndexedVector *regionSparse3, bool noPermute = false) = 0;
  virtual int updateColumnTranspose(CoinIndexedVector *regionSparse, CoinIndexedVector *regionSparse2) const = 0;
  virtual int *indices() const = 0;
  virtual int *permute() const = 0;
  virtual int numberElements() const = 0;
protected:
  int numberRows_, numberColumns_, numberGoodU_, numberPivots_, status_;
};
#endif

// STUB of src/ClpSimplex.hpp + src/ClpModel.hpp: the accessors the adapters use (line numbers of the
// reference declarations in tests/test_adapters.py)
#ifndef ClpSimplex_STUB
#define ClpSimplex_STUB
#include "ClpDualRowPivot.hpp"
#include "ClpMatrixBase.hpp"
class CoinIndexedVector;
class ClpFactorization {
public:
  inline int maximumPivots() const;
};
class ClpModel {
public:
  inline int numberRows() const;
  inline int numberColumns() const;
  inline int numberIterations() const;
  inline void setNumberIterations(int numberIterationsNew);
  inline int maximumIterations() const;
  inline double primalTolerance() const;
  inline double dualTolerance() const;
  inline void setProblemStatus(int problemStatusNew);
  inline double *primalColumnSolution() const;
  inline double *primalRowSolution() const;
  inline double *dualColumnSolution() const;
  inline double *dualRowSolution() const;
  inline double *rowLower() const;
  inline double *rowUpper() const;
  inline double *objective() const;
  inline double *columnLower() const;
  inline double *columnUpper() const;
  inline int scalingFlag() const;
  inline ClpMatrixBase *clpMatrix() const;
  inline void setObjectiveValue(double value);
  inline double optimizationDirection() const;
  inline double objectiveOffset() const;
  inline unsigned char *statusArray() const;
};
class ClpSimplex : public ClpModel {
public:
  int dual(int ifValuesPass = 0, int startFinishOptions = 0);
  int primal(int ifValuesPass = 0, int startFinishOptions = 0);
  inline ClpFactorization *factorization() const;
  inline double dualBound() const;
  inline int perturbation() const;
  inline CoinIndexedVector *rowArray(int index) const;
  inline ClpDualRowPivot *dualRowPivot() const;
  inline double zeroTolerance() const;
  inline int *pivotVariable() const;
  inline double currentDualTolerance() const;
  inline double alpha() const;
  inline int pivotRow() const;
  inline double *solutionRegion() const;
  inline double *djRegion() const;
  inline double *lowerRegion() const;
  inline double *upperRegion() const;
  inline int sequenceIn() const;
  mutable int spareIntArray_[4];
  mutable double spareDoubleArray_[4];
};
#endif

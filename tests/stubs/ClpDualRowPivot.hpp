// STUB of src/ClpDualRowPivot.hpp:23-130
#ifndef ClpDualRowPivot_STUB
#define ClpDualRowPivot_STUB
class ClpSimplex;
class CoinIndexedVector;
class ClpDualRowPivot {
public:
  virtual int pivotRow() = 0;
  virtual double updateWeights(CoinIndexedVector *input,
    CoinIndexedVector *spare,
    CoinIndexedVector *spare2,
    CoinIndexedVector *updatedColumn)
    = 0;
  virtual void updatePrimalSolution(CoinIndexedVector *input,
    double theta,
    double &changeInObjective)
    = 0;
  virtual void saveWeights(ClpSimplex *model, int mode);
  virtual void checkAccuracy();
  virtual void unrollWeights();
  virtual void clearArrays();
  virtual bool looksOptimal() const
  {
    return false;
  }
  virtual void maximumPivotsChanged() {}
  ClpDualRowPivot();
  ClpDualRowPivot(const ClpDualRowPivot &);
  virtual ~ClpDualRowPivot();
  virtual ClpDualRowPivot *clone(bool copyData = true) const = 0;
  inline int type()
  {
    return type_;
  }
protected:
  ClpSimplex *model_;
  int type_;
};
#endif

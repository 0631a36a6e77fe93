// CoinGpuFactorization -- a CoinOtherFactorization whose solves and basis updates run on libclpgpu.
// Install: ClpFactorization(const CoinOtherFactorization &) (src/ClpFactorization.hpp:73), then
// ClpSimplex::setFactorization (src/ClpSimplex.cpp:4998).  Methods are the ones ClpFactorization calls on
// coinFactorizationB_ (src/ClpFactorization.cpp:1683-1899, :2652-2665, :2750, :2827, :2946, :3012).
// CoinOtherFactorization lives in CoinUtils (CoinDenseFactorization.hpp, not in the reference tree); the
// in-tree class of the same shape is CoinAbcAnyFactorization (src/CoinAbcDenseFactorization.hpp:22-433).
#ifndef CoinGpuFactorization_H
#define CoinGpuFactorization_H

#include <memory>
#include <vector>

#include "ClpSimplex.hpp"
#include "CoinDenseFactorization.hpp"
#include "CoinIndexedVector.hpp"
#include "clpgpu.h"

class CoinGpuFactorization : public CoinOtherFactorization {
public:
  CoinGpuFactorization(std::shared_ptr< clpgpu_context > context, const ClpSimplex *model)
    : ctx_(context)
    , model_(model)
  {
  }
  CoinGpuFactorization(const CoinGpuFactorization &rhs)
    : CoinOtherFactorization(rhs)
    , ctx_(rhs.ctx_)
    , model_(rhs.model_)
    , sequence_(rhs.sequence_)
    , pivotVariable_(rhs.pivotVariable_)
  {
  }
  virtual CoinOtherFactorization *clone() const override { return new CoinGpuFactorization(*this); }
  // the model whose sequenceIn()/status the update calls refer to (Clp copies factorizations with models)
  void setModel(const ClpSimplex *model) { model_ = model; }

  // ---- factorize: getAreas / preProcess / factor / postProcess (ClpFactorization.cpp:1800-1895).
  // The engine gathers the basis from its own copy of A; only the basic set is needed, taken from the
  // model's status array.
  virtual void getAreas(int numberOfRows, int numberOfColumns, CoinBigIndex, CoinBigIndex) override
  {
    numberRows_ = numberOfRows;
    numberColumns_ = numberOfColumns;
    pivotVariable_.assign(numberOfRows, 0);
  }
  virtual void preProcess() override {}
  virtual int factor() override
  {
    status_ = clpgpu_factorize(ctx_.get(), model_->statusArray(), pivotVariable_.data());
    numberPivots_ = 0;
    numberGoodU_ = status_ == 0 ? numberRows_ : 0;
    return status_;
  }
  virtual void postProcess(const int *, int *pivotVariable) override
  {
    for (int i = 0; i < numberRows_; i++)
      pivotVariable[i] = pivotVariable_[i];
  }
  virtual void makeNonSingular(int *, int) override {}

  // ---- solves: dense regions by basis position / by row, in place
  virtual int updateColumn(CoinIndexedVector *, CoinIndexedVector *regionSparse2, bool) const override
  {
    return solve(regionSparse2, &clpgpu_ftran);
  }
  virtual int updateColumnFT(CoinIndexedVector *, CoinIndexedVector *regionSparse2, bool = false) override
  {
    return solve(regionSparse2, &clpgpu_ftran_ft);
  }
  virtual int updateTwoColumnsFT(CoinIndexedVector *, CoinIndexedVector *regionSparse2, CoinIndexedVector *regionSparse3, bool) override
  {
    // the callers hand packed vectors in and read them back packed (ClpSimplexDual::whileIterating unpacks the
    // entering column with unpackPacked, src/ClpSimplexDual.cpp:1435; ClpDualRowSteepest::updateWeights reads
    // work[i] / which[i]); the reference's dense factorization honours packedMode the same way
    // (src/CoinAbcDenseFactorization.cpp:412, :439)
    const bool packed2 = regionSparse2->packedMode(), packed3 = regionSparse3->packedMode();
    regionSparse2->expand();
    regionSparse3->expand();
    int rc = clpgpu_ftran_two_ft(ctx_.get(), regionSparse2->denseVector(), regionSparse3->denseVector());
    repack(regionSparse2, packed2);
    repack(regionSparse3, packed3);
    return rc;
  }
  virtual int updateColumnTranspose(CoinIndexedVector *, CoinIndexedVector *regionSparse2) const override
  {
    return solve(regionSparse2, &clpgpu_btran);
  }
  // ---- basis update: 0 OK / 2 singular / 3 no room / 5 max pivots (src/ClpFactorization.hpp:83)
  virtual int replaceColumn(CoinIndexedVector *, int pivotRow, double pivotCheck, bool, double acceptablePivot) override
  {
    int rc = clpgpu_replace_column(ctx_.get(), pivotRow, model_->sequenceIn(), pivotCheck, acceptablePivot);
    if (rc == 0)
      numberPivots_++;
    return rc;
  }
  virtual int *indices() const override { return NULL; }
  virtual int *permute() const override { return NULL; }  // ClpDualRowSteepest then skips permutation (:442-460)
  // read by ClpDualRowSteepest::pivotRow in mode 3 (src/ClpDualRowSteepest.cpp:262): the entries this factorization holds
  virtual int numberElements() const override
  {
    clpgpu_stats stats;
    if (clpgpu_get_stats(ctx_.get(), &stats))
      return numberRows_;
    return stats.factor_elements > 2147483647L ? 2147483647 : static_cast< int >(stats.factor_elements);
  }

private:
  int solve(CoinIndexedVector *region, int (*fn)(clpgpu_context *, double *)) const
  {
    const bool packed = region->packedMode();
    region->expand();
    int rc = fn(ctx_.get(), region->denseVector());
    repack(region, packed);
    return rc < 0 ? rc : region->getNumElements();
  }
  // expand() leaves the vector dense and clears packedMode_; a vector that came in packed goes back packed
  static void repack(CoinIndexedVector *region, bool wasPacked)
  {
    if (wasPacked)
      region->scanAndPack();
    else
      region->scan();
  }
  std::shared_ptr< clpgpu_context > ctx_;
  const ClpSimplex *model_;
  std::vector< int > sequence_, pivotVariable_;
};
#endif

// ClpGpuPackedMatrix -- ClpPackedMatrix whose hot products run on libclpgpu (include/clpgpu.h).
// Adapter for a Clp build (needs Clp + CoinUtils headers); tests/test_adapters.py compiles it against
// minimal stub headers carrying the reference's virtual signatures so that signature drift is caught.
// Reference surface: ClpMatrixBase (src/ClpMatrixBase.hpp:38-546); install with
// ClpModel::replaceMatrix(new ClpGpuPackedMatrix(...), true) (src/ClpModel.cpp:4046-4052).
#ifndef ClpGpuPackedMatrix_H
#define ClpGpuPackedMatrix_H

#include <memory>

#include "ClpPackedMatrix.hpp"
#include "ClpSimplex.hpp"
#include "CoinIndexedVector.hpp"
#include "clpgpu.h"

// one device context shared by the clones Clp makes (ClpModel.cpp:822, :885): the matrix is immutable
struct ClpGpuContextDeleter {
  void operator()(clpgpu_context *c) const { clpgpu_destroy(c); }
};
typedef std::shared_ptr< clpgpu_context > ClpGpuContextPtr;

class ClpGpuPackedMatrix : public ClpPackedMatrix {
public:
  ClpGpuPackedMatrix(const ClpPackedMatrix &rhs, ClpGpuContextPtr context)
    : ClpPackedMatrix(rhs)
    , ctx_(context)
  {
  }
  ClpGpuPackedMatrix(const ClpGpuPackedMatrix &rhs)
    : ClpPackedMatrix(rhs)
    , ctx_(rhs.ctx_)
  {
  }
  // ClpMatrixBase::clone (:361)
  virtual ClpMatrixBase *clone() const override { return new ClpGpuPackedMatrix(*this); }

  // ClpMatrixBase::times (:275) / transposeTimes (:287): y += scalar * A x, y += scalar * A^T x
  virtual void times(double scalar, const double *COIN_RESTRICT x, double *COIN_RESTRICT y) const override
  {
    if (clpgpu_times(ctx_.get(), scalar, x, y) != 0)
      ClpPackedMatrix::times(scalar, x, y);
  }
  virtual void transposeTimes(double scalar, const double *COIN_RESTRICT x, double *COIN_RESTRICT y) const override
  {
    if (clpgpu_transpose_times(ctx_.get(), scalar, x, y) != 0)
      ClpPackedMatrix::transposeTimes(scalar, x, y);
  }

  // ClpMatrixBase::transposeTimes(model, scalar, x, y, z) (:308): the row-pricing call of
  // ClpSimplexDual::whileIterating (src/ClpSimplexDual.cpp:1300) with the fused first ratio pass the
  // caller asks for through spareIntArray_[0] == 1 (:1290-1308); completion is signalled with -2
  // (src/ClpPackedMatrix.cpp:1088) and upperTheta comes back in spareDoubleArray_[0].
  virtual void transposeTimes(const ClpSimplex *model, double scalar, const CoinIndexedVector *x, CoinIndexedVector *y,
    CoinIndexedVector *z) const override
  {
    ClpSimplex *m = const_cast< ClpSimplex * >(model);
    if (scalar != -1.0 || !x->packedMode() || m->spareIntArray_[0] <= 0) {
      ClpPackedMatrix::transposeTimes(model, scalar, x, y, z);
      return;
    }
    CoinIndexedVector *candidates = model->rowArray(3);
    int numberOut = 0, numberCandidates = 0;
    double upperTheta = 0.0;
    int rc = clpgpu_price_row(ctx_.get(), x->getNumElements(), x->getIndices(), x->denseVector(), model->statusArray(),
      model->djRegion(), model->zeroTolerance(), model->currentDualTolerance(), m->spareDoubleArray_[0], &numberOut,
      z->getIndices(), z->denseVector(), &numberCandidates, candidates->getIndices(), candidates->denseVector(), &upperTheta);
    if (rc != 0) {
      ClpPackedMatrix::transposeTimes(model, scalar, x, y, z);
      return;
    }
    z->setNumElements(numberOut);
    z->setPackedMode(true);
    candidates->setNumElements(numberCandidates);
    m->spareDoubleArray_[0] = upperTheta;
    m->spareIntArray_[0] = -2;
  }

  clpgpu_context *context() const { return ctx_.get(); }

private:
  ClpGpuContextPtr ctx_;
};
#endif

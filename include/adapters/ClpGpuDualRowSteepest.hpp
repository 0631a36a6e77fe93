// ClpGpuDualRowSteepest -- ClpDualRowPivot (src/ClpDualRowPivot.hpp:23-130) on libclpgpu: CHUZR, the
// DSE weight update (with its FT FTRAN, as ClpDualRowSteepest::updateWeights does at :472) and the
// primal update run on the device state; Clp's rim arrays are handed over with clpgpu_bind_rim.
// Install: ClpSimplex::setDualRowPivotAlgorithm (src/ClpSimplex.cpp:4985) clones it and calls setModel.
#ifndef ClpGpuDualRowSteepest_H
#define ClpGpuDualRowSteepest_H

#include <memory>
#include <vector>

#include "ClpDualRowPivot.hpp"
#include "ClpSimplex.hpp"
#include "CoinIndexedVector.hpp"
#include "clpgpu.h"

class ClpGpuDualRowSteepest : public ClpDualRowPivot {
public:
  explicit ClpGpuDualRowSteepest(std::shared_ptr< clpgpu_context > context)
    : ctx_(context)
  {
    type_ = 2;  // steepest (ClpDualRowSteepest sets 2, Dantzig 1)
  }
  ClpGpuDualRowSteepest(const ClpGpuDualRowSteepest &rhs)
    : ClpDualRowPivot(rhs)
    , ctx_(rhs.ctx_)
  {
  }
  virtual ClpDualRowPivot *clone(bool = true) const override { return new ClpGpuDualRowSteepest(*this); }

  // :30 -- the engine sees the current rim first (cheap next to the PCIe cost of the other plug-in calls)
  virtual int pivotRow() override
  {
    clpgpu_bind_rim(ctx_.get(), NULL, model_->lowerRegion(), model_->upperRegion(), NULL, model_->solutionRegion(),
      model_->statusArray());
    return clpgpu_pivot_row(ctx_.get());
  }
  // :33 -- input = packed pi, updatedColumn = the unpacked entering column (rowArray_[1]); returns alpha
  virtual double updateWeights(CoinIndexedVector *input, CoinIndexedVector *, CoinIndexedVector *, CoinIndexedVector *updatedColumn) override
  {
    double alpha = 0.0;
    updatedColumn->expand();
    clpgpu_update_weights(ctx_.get(), input->getNumElements(), input->getIndices(), input->denseVector(), model_->pivotRow(),
      model_->sequenceIn(), model_->alpha(), updatedColumn->denseVector(), &alpha);
    updatedColumn->scan();
    return alpha;
  }
  // :40
  virtual void updatePrimalSolution(CoinIndexedVector *input, double theta, double &changeInObjective) override
  {
    input->expand();
    clpgpu_update_primal(ctx_.get(), input->denseVector(), model_->pivotRow(), theta, &changeInObjective);
    std::vector< double > solution(model_->numberRows() + model_->numberColumns());
    clpgpu_get_solution(ctx_.get(), solution.data());
    const int *pivotVariable = model_->pivotVariable();
    double *modelSolution = model_->solutionRegion();
    for (int i = 0; i < model_->numberRows(); i++)
      modelSolution[pivotVariable[i]] = solution[pivotVariable[i]];
    input->clear();
  }
  // :53, :60
  virtual void saveWeights(ClpSimplex *model, int mode) override
  {
    model_ = model;
    if (mode >= 2)
      clpgpu_bind_rim(ctx_.get(), NULL, model_->lowerRegion(), model_->upperRegion(), NULL, model_->solutionRegion(),
        model_->statusArray());
    clpgpu_save_weights(ctx_.get(), mode);
  }
  virtual void unrollWeights() override { clpgpu_unroll_weights(ctx_.get()); }

private:
  std::shared_ptr< clpgpu_context > ctx_;
};
#endif

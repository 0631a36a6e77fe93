// clpGpuDual -- the whole dual simplex on the device, installed the way ClpSimplex::dealWithAbc installs
// the Abc engine (src/ClpSolve.cpp:555-833): copy the model in, solve, move status and solution back,
// let Clp re-derive what it wants (:777-811).  Call next to dealWithAbc in ClpSimplex::initialSolve (:1935).
#include <vector>

#include "ClpDualRowSteepest.hpp"
#include "ClpPackedMatrix.hpp"
#include "ClpSimplex.hpp"
#include "CoinHelperFunctions.hpp"
#include "CoinPackedMatrix.hpp"
#include "clpgpu.h"

int clpGpuDual(ClpSimplex &model, int device, bool scaling)
{
  ClpPackedMatrix *clpMatrix = dynamic_cast< ClpPackedMatrix * >(model.clpMatrix());
  if (!clpMatrix || clpMatrix->getNumElements() == 0)
    return model.dual(); // other matrix types stay on the CPU path
  const CoinPackedMatrix *A = clpMatrix->getPackedMatrix(); // column ordered, no gaps
  clpgpu_context *ctx = clpgpu_create(device);
  if (!ctx)
    return model.dual(); // no MI355X: the caller keeps the reference path
  const int numberRows = model.numberRows(), numberColumns = model.numberColumns();
  if (scaling)
    clpgpu_set_option(ctx, "scaling", model.scalingFlag() > 0 ? model.scalingFlag() : 0); // before the load
  // the engine minimises: a maximisation model (optimizationDirection -1) is loaded with its costs negated, as
  // ClpSimplex::createRim does (src/ClpSimplex.cpp:3754: cost_ = direction * objective), and the duals come back
  // multiplied by the direction again (ClpSimplex::finish -> deleteRim, src/ClpSimplex.cpp:4590ff)
  const double direction = model.optimizationDirection() == 0.0 ? 1.0 : model.optimizationDirection();
  std::vector< double > cost(model.objective(), model.objective() + numberColumns);
  for (int j = 0; j < numberColumns; j++)
    cost[j] *= direction;
  clpgpu_load_problem(ctx, numberRows, numberColumns, A->getVectorStarts(), A->getIndices(), A->getElements(),
    model.columnLower(), model.columnUpper(), cost.data(), model.rowLower(), model.rowUpper());
  clpgpu_set_option(ctx, "pivot_rule", model.dualRowPivot()->type() == 2 ? 1 : 0);
  // ClpDualRowSteepest::mode_ decides how much of the infeasibility list one pivotRow() call scans (src/ClpDualRowSteepest.cpp:258-278:
  // 0 / 1 everything, 2 max(2000, number / 8), 3 -- the constructor's default -- sized by factorization()->numberElements() / rows).
  // What stands for numberElements() in mode 3 is the engine's option steepest_elements: 0 (its default, left alone here) = the entries of
  // the basic structural columns, 1 = the entries its own factorization holds (what CoinGpuFactorization::numberElements() answers at
  // the plug-in level).  1 is the closer stand-in for CoinFactorization's count once the basis has a dense tail, but it means full scans
  // there, and those were measured to stall LPs of the bench's family (102 000 pivots against 2.7 M unfinished on a
  // 7 000-row instance); a caller who wants it sets clpgpu_set_option(ctx, "steepest_elements", 1) before clpgpu_dual.
  if (const ClpDualRowSteepest *steepest = dynamic_cast< const ClpDualRowSteepest * >(model.dualRowPivot()))
    clpgpu_set_option(ctx, "steepest_mode", steepest->mode());
  clpgpu_set_option(ctx, "max_iterations", model.maximumIterations());
  clpgpu_set_option(ctx, "max_pivots", model.factorization()->maximumPivots());
  clpgpu_set_option(ctx, "dual_bound", model.dualBound());
  clpgpu_set_option(ctx, "primal_tolerance", model.primalTolerance());
  clpgpu_set_option(ctx, "dual_tolerance", model.dualTolerance());
  // ClpSimplex::perturbation_ (100 from the constructor, 50 from the clp command): start-up perturbation and the kick of
  // ClpSimplexDual.cpp:488 happen on the device side as they would in dual(); a solve that ends on perturbed costs with
  // dual infeasibilities for the true ones comes back as status 10 and is finished by primal below, as in ClpSimplex::dual
  clpgpu_set_option(ctx, "perturbation", model.perturbation());
  // "infeasible" with fake bounds still active comes back as 10 (ClpSimplex::dual, src/ClpSimplex.cpp:5800-5803) and is finished by primal below
  clpgpu_set_option(ctx, "fake_bound_cleanup", 1);
  // nonbasic free columns stay isFree as in ClpSimplexDual (free-first dualRow :3005-3055, general branch of dualColumn0 :4058-4179)
  // instead of getting bothFake bounds at start; a solve that ends primal feasible with dual infeasibilities on free variables only
  // comes back as 10 (:5619-5622) and is finished by primal below
  clpgpu_set_option(ctx, "free_nonbasic", 1);
  // "problems - try primal" (gutsOfDual, src/ClpSimplexDual.cpp:533-547): a solve whose primal infeasibilities run away while the
  // objective stands still comes back as 10 instead of escalating the dual bound further
  clpgpu_set_option(ctx, "try_primal", 1);
  if (model.statusArray())
    clpgpu_set_status(ctx, model.statusArray()); // warm start
  int problemStatus = clpgpu_dual(ctx);          // ClpSimplex::dual()
  model.setProblemStatus(problemStatus);
  model.setNumberIterations(model.numberIterations() + clpgpu_number_iterations(ctx));
  std::vector< double > solution(numberRows + numberColumns), dj(numberRows + numberColumns);
  clpgpu_get_status(ctx, model.statusArray());
  clpgpu_get_solution(ctx, solution.data());
  clpgpu_get_reduced_costs(ctx, dj.data());
  CoinMemcpyN(solution.data(), numberColumns, model.primalColumnSolution());
  CoinMemcpyN(solution.data() + numberColumns, numberRows, model.primalRowSolution());
  for (int j = 0; j < numberColumns; j++)
    model.dualColumnSolution()[j] = direction * dj[j];
  for (int i = 0; i < numberRows; i++)
    model.dualRowSolution()[i] = direction * dj[numberColumns + i]; // row dj == dual (src/ClpSimplex.cpp:1381-1386)
  // ClpModel::setObjectiveValue takes the value in the user's sense and applies offset and direction itself
  // (src/ClpModel.hpp:862-865); the device minimised direction * c x
  model.setObjectiveValue(direction * clpgpu_objective_value(ctx) - model.objectiveOffset());
  clpgpu_destroy(ctx);
  if (problemStatus == 10) // "needs primal clean-up", src/ClpSimplex.cpp:5808
    return model.primal(1);
  return problemStatus;
}

// ClpSimplexDual::strongBranching (src/ClpSimplexDual.hpp:125-131, src/ClpSimplexDual.cpp:6965) for a node LP
// that was solved on the device: same arguments, same outputs (newLower / newUpper come back as the objective
// changes, outputStatus 0 / 1 / 2, even = down, odd = up).  `ctx` is the context of that solve -- a caller that
// wants strong branching keeps it alive instead of destroying it at the end of clpGpuDual -- so the matrix and
// the optimal basis are already resident; every branch is a fastDual (src/ClpSimplexDual.cpp:7227) on the device.
int clpGpuStrongBranching(ClpSimplex &model, clpgpu_context *ctx, int numberVariables, const int *variables,
  double *newLower, double *newUpper, double **outputSolution, int *outputStatus, int *outputIterations,
  bool stopOnFirstInfeasible, bool alwaysFinish)
{
  clpgpu_set_option(ctx, "max_iterations", model.maximumIterations()); // Cbc sets a small limit for this call
  return clpgpu_strong_branching(ctx, numberVariables, variables, newLower, newUpper, outputSolution, outputStatus,
    outputIterations, stopOnFirstInfeasible ? 1 : 0, alwaysFinish ? 1 : 0);
}

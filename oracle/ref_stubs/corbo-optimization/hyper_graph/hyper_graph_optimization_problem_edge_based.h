// ORACLE (test infrastructure only): the reference only creates this object and hands it to the optimal control problem
#pragma once
#include <memory>
namespace corbo {
class BaseHyperGraphOptimizationProblem { public: using Ptr = std::shared_ptr<BaseHyperGraphOptimizationProblem>; virtual ~BaseHyperGraphOptimizationProblem() = default; };
class HyperGraphOptimizationProblemEdgeBased : public BaseHyperGraphOptimizationProblem {};
}  // namespace corbo

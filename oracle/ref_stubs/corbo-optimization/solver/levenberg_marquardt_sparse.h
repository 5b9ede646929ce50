// ORACLE (test infrastructure only): see nlp_solver_ipopt.h
#pragma once
#include <corbo-optimization/solver/nlp_solver_ipopt.h>
namespace corbo {
class LevenbergMarquardtSparse : public NlpSolverInterface {
 public:
    using Ptr = std::shared_ptr<LevenbergMarquardtSparse>;
    bool isLsqSolver() const override { return true; }
    void setIterations(int n) { iterations = n; }
    void setPenaltyWeights(double e, double i, double b) { w[0] = e; w[1] = i; w[2] = b; }
    void setWeightAdapation(double fe, double fi, double fb, double me, double mi, double mb) { a[0] = fe; a[1] = fi; a[2] = fb; a[3] = me; a[4] = mi; a[5] = mb; }
    int iterations = -1;
    double w[3] = {0, 0, 0}, a[6] = {0, 0, 0, 0, 0, 0};
};
}  // namespace corbo

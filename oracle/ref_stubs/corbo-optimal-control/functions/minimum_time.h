// ORACLE (test infrastructure only).  NOT control-box-rst: corbo::MinimumTime as a record (the reference only constructs it); its value on a single-dt grid is
// (n - 1) dt, which the reference's own MinTimeViaPointsCost also computes (src/optimal_control/min_time_via_points_cost.cpp:119-128)
#pragma once
#include <corbo-optimal-control/functions/quadratic_cost.h>
namespace corbo {
class MinimumTime : public QuadraticCostStubBase {
 public:
    explicit MinimumTime(bool lsq_form = false) { _lsq_form = lsq_form; }
    Ptr getInstance() const override { return std::make_shared<MinimumTime>(); }
    void computeNonIntegralStateTerm(int, const Eigen::Ref<const Eigen::VectorXd>&, Eigen::Ref<Eigen::VectorXd>) const override {}
};
}  // namespace corbo

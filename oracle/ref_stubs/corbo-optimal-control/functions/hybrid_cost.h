// ORACLE (test infrastructure only).  NOT control-box-rst: corbo::MinTimeQuadraticControls as a record of its constructor arguments
#pragma once
#include <corbo-optimal-control/functions/quadratic_control_cost.h>
namespace corbo {
class MinTimeQuadraticControls : public QuadraticControlCost {
 public:
    MinTimeQuadraticControls() = default;
    MinTimeQuadraticControls(const Eigen::Ref<const Eigen::MatrixXd>& R, bool integral_form = false, bool lsq_form = false) : QuadraticControlCost(R, integral_form, lsq_form) {}
    Ptr getInstance() const override { return std::make_shared<MinTimeQuadraticControls>(); }
};
}  // namespace corbo

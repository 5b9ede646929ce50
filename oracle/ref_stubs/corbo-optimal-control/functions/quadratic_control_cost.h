// ORACLE (test infrastructure only).  NOT control-box-rst: corbo::QuadraticControlCost as a record of its constructor arguments
#pragma once
#include <corbo-optimal-control/functions/quadratic_cost.h>
namespace corbo {
class QuadraticControlCost : public QuadraticCostStubBase {
 public:
    QuadraticControlCost() = default;
    QuadraticControlCost(const Eigen::Ref<const Eigen::MatrixXd>& R, bool integral_form = false, bool lsq_form = false) : _R(R) { _integral_form = integral_form; _lsq_form = lsq_form; }
    Ptr getInstance() const override { return std::make_shared<QuadraticControlCost>(); }
    void computeNonIntegralStateTerm(int, const Eigen::Ref<const Eigen::VectorXd>&, Eigen::Ref<Eigen::VectorXd>) const override {}
    Eigen::MatrixXd _R;
};
}  // namespace corbo

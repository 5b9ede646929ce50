// ORACLE (test infrastructure only).  NOT control-box-rst: the data members of corbo::QuadraticStateCost that QuadraticStateCostSE2 reads (see quadratic_cost.h)
#pragma once
#include <corbo-optimal-control/functions/quadratic_cost.h>

namespace corbo {
class QuadraticStateCost : public QuadraticCostStubBase {
 public:
    QuadraticStateCost() = default;
    QuadraticStateCost(const Eigen::Ref<const Eigen::MatrixXd>& Q, bool integral_form = false, bool lsq_form = false) : _Q(Q) { _integral_form = integral_form; _lsq_form = lsq_form; }
    virtual void computeIntegralStateControlTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& x_k, const Eigen::Ref<const Eigen::VectorXd>& u_k,
                                                 Eigen::Ref<Eigen::VectorXd> cost) const = 0;
    Eigen::MatrixXd _Q, _Q_sqrt;
    Eigen::DiagonalMatrix<double, -1> _Q_diag, _Q_diag_sqrt;
    bool _diagonal_mode = false;
};
}  // namespace corbo

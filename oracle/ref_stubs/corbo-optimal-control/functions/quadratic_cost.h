// ORACLE (test infrastructure only).  NOT control-box-rst: the DATA MEMBERS of corbo::QuadraticFormCost that the reference's QuadraticFormCostSE2
// (src/optimal_control/quadratic_cost_se2.cpp) reads, filled by plain setters; no corbo code.  The lsq-form square roots are set by the caller (diagonal weights:
// the element-wise roots; how corbo factors a full matrix is not restated, the full-matrix lsq form is not exercised).
#pragma once
#include <corbo-optimal-control/functions/stage_functions.h>

namespace corbo {
class QuadraticCostStubBase : public StageCost {
 public:
    bool hasNonIntegralTerms(int) const override { return !_integral_form; }
    bool hasIntegralTerms(int) const override { return _integral_form; }
    int getNonIntegralDtTermDimension(int) const override { return 0; }
    bool isLsqFormNonIntegralDtTerm(int) const override { return false; }
    int getNonIntegralStateTermDimension(int) const override { return _integral_form ? 0 : (_lsq_form ? _state_dim : 1); }
    bool update(int, double, ReferenceTrajectoryInterface& xref, ReferenceTrajectoryInterface& uref, ReferenceTrajectoryInterface*, bool, const Eigen::VectorXd&,
                StagePreprocessor::Ptr, const std::vector<double>&, const DiscretizationGridInterface*) override { _x_ref = &xref; _u_ref = &uref; return false; }
    void computeNonIntegralDtTerm(int, double, Eigen::Ref<Eigen::VectorXd>) const override {}
    int _state_dim = 3;
    bool _integral_form = false, _lsq_form = false, _zero_u_ref = true;
    ReferenceTrajectoryInterface* _x_ref = nullptr;
    ReferenceTrajectoryInterface* _u_ref = nullptr;
};
class QuadraticFormCost : public QuadraticCostStubBase {
 public:
    QuadraticFormCost() = default;
    QuadraticFormCost(const Eigen::Ref<const Eigen::MatrixXd>& Q, const Eigen::Ref<const Eigen::MatrixXd>& R, bool integral_form = false, bool lsq_form = false)
        : _Q(Q), _R(R) { _integral_form = integral_form; _lsq_form = lsq_form; }
    virtual void computeIntegralStateControlTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& x_k, const Eigen::Ref<const Eigen::VectorXd>& u_k,
                                                 Eigen::Ref<Eigen::VectorXd> cost) const = 0;
    Eigen::MatrixXd _Q, _R, _Q_sqrt;
    Eigen::DiagonalMatrix<double, -1> _Q_diag, _R_diag, _Q_diag_sqrt;
    bool _Q_diagonal_mode = false, _R_diagonal_mode = false;
};
}  // namespace corbo

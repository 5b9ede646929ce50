// ORACLE (test infrastructure only).  NOT control-box-rst: the data members of corbo::QuadraticFinalStateCost that QuadraticFinalStateCostSE2 reads
#pragma once
#include <corbo-core/reference_trajectory.h>
#include <memory>

namespace corbo {
class FinalStageCost {
 public:
    using Ptr = std::shared_ptr<FinalStageCost>;
    virtual ~FinalStageCost() = default;
    virtual Ptr getInstance() const = 0;
    virtual int getNonIntegralStateTermDimension(int k) const = 0;
    virtual void computeNonIntegralStateTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& x_k, Eigen::Ref<Eigen::VectorXd> cost) const = 0;
};
class QuadraticFinalStateCost : public FinalStageCost {
 public:
    QuadraticFinalStateCost() = default;
    QuadraticFinalStateCost(const Eigen::Ref<const Eigen::MatrixXd>& Qf, bool lsq_form) : _Qf(Qf), _lsq_form(lsq_form) {}
    int getNonIntegralStateTermDimension(int) const override { return _lsq_form ? 3 : 1; }
    Eigen::MatrixXd _Qf, _Qf_sqrt;
    Eigen::DiagonalMatrix<double, -1> _Qf_diag, _Qf_diag_sqrt;
    bool _lsq_form = false, _diagonal_mode = false;
    ReferenceTrajectoryInterface* _x_ref = nullptr;
};
}  // namespace corbo

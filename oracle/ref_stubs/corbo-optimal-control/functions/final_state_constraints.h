// ORACLE (test infrastructure only).  NOT control-box-rst: the data members of corbo::TerminalBall that TerminalBallSE2 reads
#pragma once
#include <corbo-core/reference_trajectory.h>
#include <memory>

namespace corbo {
class FinalStageConstraint {
 public:
    using Ptr = std::shared_ptr<FinalStageConstraint>;
    virtual ~FinalStageConstraint() = default;
    virtual Ptr getInstance() const = 0;
    virtual int getNonIntegralStateTermDimension(int k) const = 0;
    virtual void computeNonIntegralStateTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& x_k, Eigen::Ref<Eigen::VectorXd> cost) const = 0;
};
class TerminalBall : public FinalStageConstraint {
 public:
    TerminalBall() = default;
    TerminalBall(const Eigen::Ref<const Eigen::MatrixXd>& S, double gamma) : _S(S), _gamma(gamma) {}
    int getNonIntegralStateTermDimension(int) const override { return 1; }
    Eigen::MatrixXd _S;
    Eigen::DiagonalMatrix<double, -1> _S_diag;
    double _gamma = 0.0;
    bool _diagonal_mode = false;
    ReferenceTrajectoryInterface* _x_ref = nullptr;
};
}  // namespace corbo

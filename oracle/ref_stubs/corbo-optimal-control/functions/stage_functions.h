// ORACLE (test infrastructure only): the INTERFACE of corbo::StageInequalityConstraint as include/mpc_local_planner/optimal_control/stage_inequality_se2.h
// overrides it, and the two types that only appear in the signature of update(); no corbo code.
#pragma once
#include <corbo-core/reference_trajectory.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/discretization_grid_interface.h>

namespace corbo {
class StageInequalityConstraint {
 public:
    using Ptr = std::shared_ptr<StageInequalityConstraint>;
    virtual ~StageInequalityConstraint() = default;
    virtual Ptr getInstance() const = 0;
    virtual bool hasNonIntegralTerms(int k) const = 0;
    virtual bool hasIntegralTerms(int k) const = 0;
    virtual int getNonIntegralStateTermDimension(int k) const = 0;
    virtual int getNonIntegralStateDtTermDimension(int k) const = 0;
    virtual int getNonIntegralControlDeviationTermDimension(int k) const = 0;
    virtual bool update(int n, double t, ReferenceTrajectoryInterface& xref, ReferenceTrajectoryInterface& uref, ReferenceTrajectoryInterface* sref, bool single_dt,
                        const Eigen::VectorXd& x0, StagePreprocessor::Ptr stage_preprocessor, const std::vector<double>& dts, const DiscretizationGridInterface* grid) = 0;
    virtual void computeNonIntegralStateTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& x_k, Eigen::Ref<Eigen::VectorXd> cost) const = 0;
    virtual void computeNonIntegralStateDtTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& x_k, double dt_k, Eigen::Ref<Eigen::VectorXd> cost) const = 0;
    virtual void computeNonIntegralControlDeviationTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& u_k, const Eigen::Ref<const Eigen::VectorXd>& u_prev, double dt,
                                                        Eigen::Ref<Eigen::VectorXd> cost) const = 0;
};
// the INTERFACE of corbo::StageCost as include/mpc_local_planner/optimal_control/min_time_via_points_cost.h overrides it
class StageCost {
 public:
    using Ptr = std::shared_ptr<StageCost>;
    virtual ~StageCost() = default;
    virtual Ptr getInstance() const = 0;
    virtual bool hasNonIntegralTerms(int k) const = 0;
    virtual bool hasIntegralTerms(int k) const = 0;
    virtual int getNonIntegralDtTermDimension(int k) const = 0;
    virtual bool isLsqFormNonIntegralDtTerm(int k) const = 0;
    virtual int getNonIntegralStateTermDimension(int k) const = 0;
    virtual bool update(int n, double t, ReferenceTrajectoryInterface& xref, ReferenceTrajectoryInterface& uref, ReferenceTrajectoryInterface* sref, bool single_dt,
                        const Eigen::VectorXd& x0, StagePreprocessor::Ptr stage_preprocessor, const std::vector<double>& dts, const DiscretizationGridInterface* grid) = 0;
    virtual void computeNonIntegralDtTerm(int k, double dt, Eigen::Ref<Eigen::VectorXd> cost) const = 0;
    virtual void computeNonIntegralStateTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& x_k, Eigen::Ref<Eigen::VectorXd> cost) const = 0;
};
}  // namespace corbo

// ORACLE (test infrastructure only): the INTERFACE of corbo::FiniteDifferencesCollocationInterface as the reference's SE(2) collocation rules
// override it (include/mpc_local_planner/optimal_control/fd_collocation_se2.h); no corbo code.
#pragma once
#include <corbo-systems/system_dynamics_interface.h>

namespace corbo {
class FiniteDifferencesCollocationInterface {
 public:
    using Ptr = std::shared_ptr<FiniteDifferencesCollocationInterface>;
    using StateVector = Eigen::VectorXd;
    using InputVector = Eigen::VectorXd;
    virtual ~FiniteDifferencesCollocationInterface() = default;
    virtual Ptr getInstance() const = 0;
    virtual void computeEqualityConstraint(const StateVector& x1, const InputVector& u1, const StateVector& x2, double dt, const SystemDynamicsInterface& system,
                                           Eigen::Ref<Eigen::VectorXd> error) = 0;
};
// named by the reference's grid header as the default rule (full_discretization_grid_base_se2.h:199); never evaluated here
class CrankNicolsonDiffCollocation : public FiniteDifferencesCollocationInterface {
 public:
    Ptr getInstance() const override { return std::make_shared<CrankNicolsonDiffCollocation>(); }
    void computeEqualityConstraint(const StateVector&, const InputVector&, const StateVector&, double, const SystemDynamicsInterface&, Eigen::Ref<Eigen::VectorXd>) override {}
};
}  // namespace corbo

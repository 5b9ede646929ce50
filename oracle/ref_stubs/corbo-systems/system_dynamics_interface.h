// ORACLE (test infrastructure only): the INTERFACE of corbo::SystemDynamicsInterface as the reference's robot models override it
// (include/mpc_local_planner/systems/base_robot_se2.h:45-58, unicycle_robot.h, simple_car.h, kinematic_bicycle_model.h); no corbo code.
#pragma once
#include <corbo-core/types.h>

namespace corbo {
class SystemDynamicsInterface {
 public:
    using Ptr = std::shared_ptr<SystemDynamicsInterface>;
    using StateVector = Eigen::VectorXd;
    using ControlVector = Eigen::VectorXd;
    virtual ~SystemDynamicsInterface() = default;
    virtual Ptr getInstance() const = 0;
    virtual int getInputDimension() const = 0;
    virtual int getStateDimension() const = 0;
    virtual bool isContinuousTime() const = 0;
    virtual bool isLinear() const = 0;
    virtual void dynamics(const Eigen::Ref<const StateVector>& x, const Eigen::Ref<const ControlVector>& u, Eigen::Ref<StateVector> f) const = 0;
};
}  // namespace corbo

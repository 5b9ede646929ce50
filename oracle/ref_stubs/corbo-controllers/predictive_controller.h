// ORACLE (test infrastructure only).  NOT control-box-rst: the members of corbo::PredictiveController that the reference's Controller (src/controller.cpp) uses.
// step() does what the reference relies on: num_ocp_iterations x OptimalControlProblemInterface::compute (the first one a "new run"), then the state and control
// time series of the result are handed back.  The optimal control problem behind it is the stand-in of structured_optimal_control_problem.h.
#pragma once
#include <corbo-core/reference_trajectory.h>
#include <corbo-core/time_series.h>
#include <memory>

namespace corbo {
class SignalTargetInterface;
class OptimalControlProblemInterface {
 public:
    using Ptr = std::shared_ptr<OptimalControlProblemInterface>;
    virtual ~OptimalControlProblemInterface() = default;
    virtual bool initialize() = 0;
    virtual void reset() = 0;
    virtual bool compute(const Eigen::VectorXd& x, ReferenceTrajectoryInterface& xref, ReferenceTrajectoryInterface& uref, ReferenceTrajectoryInterface* sref, const Time& t,
                         bool new_run, ReferenceTrajectoryInterface* xinit, ReferenceTrajectoryInterface* uinit) = 0;
    virtual void getTimeSeries(TimeSeries::Ptr x_sequence, TimeSeries::Ptr u_sequence) = 0;
    virtual void setPreviousControlInput(const Eigen::Ref<const Eigen::VectorXd>& u_prev, double dt) = 0;
};
class ControllerInterface {
 public:
    using Ptr = std::shared_ptr<ControllerInterface>;
    virtual ~ControllerInterface() = default;
    virtual Ptr getInstance() const = 0;
    virtual void reset() = 0;
};
struct ControllerStatistics { Duration step_time; };
class PredictiveController : public ControllerInterface {
 public:
    void setOptimalControlProblem(OptimalControlProblemInterface::Ptr ocp) { _ocp = ocp; }
    OptimalControlProblemInterface::Ptr getOptimalControlProblem() { return _ocp; }
    void setNumOcpIterations(int n) { _num_ocp_iterations = n; }
    void setAutoUpdatePreviousControl(bool enable) { _auto_update_prev_control = enable; }
    virtual bool step(const Eigen::VectorXd& x, ReferenceTrajectoryInterface& xref, ReferenceTrajectoryInterface& uref, const Duration& dt, const Time& t, TimeSeries::Ptr u_sequence,
                      TimeSeries::Ptr x_sequence, SignalTargetInterface* = nullptr, ReferenceTrajectoryInterface* sref = nullptr, ReferenceTrajectoryInterface* xinit = nullptr,
                      ReferenceTrajectoryInterface* uinit = nullptr) {
        if (!_ocp) return false;
        if (!_x_ts) _x_ts = std::make_shared<TimeSeries>();
        if (!_u_ts) _u_ts = std::make_shared<TimeSeries>();
        bool success = false;
        for (int i = 0; i < _num_ocp_iterations; ++i) success = _ocp->compute(x, xref, uref, sref, t, i == 0, xinit, uinit);
        _ocp->getTimeSeries(_x_ts, _u_ts);
        if (_auto_update_prev_control && !_u_ts->isEmpty()) _ocp->setPreviousControlInput(_u_ts->getValuesMap(0), dt.toSec());
        if (u_sequence) *u_sequence = *_u_ts;
        if (x_sequence) *x_sequence = *_x_ts;
        return success;
    }
    void reset() override { if (_ocp) _ocp->reset(); }
 protected:
    OptimalControlProblemInterface::Ptr _ocp;
    TimeSeries::Ptr _x_ts, _u_ts;
    ControllerStatistics _statistics;
    int _num_ocp_iterations = 1;
    bool _auto_update_prev_control = true;
};
}  // namespace corbo

// ORACLE (test infrastructure only): the INTERFACE of corbo::DiscretizationGridInterface as the reference's grid classes override it
// (include/mpc_local_planner/optimal_control/full_discretization_grid_base_se2.h) and the types in its signatures, reduced to what
// src/optimal_control/full_discretization_grid_base_se2.cpp and finite_differences_variable_grid_se2.cpp touch.  No corbo code.
#pragma once
#include <functional>
#include <corbo-core/time_series.h>
#include <corbo-core/console.h>
#include <corbo-core/reference_trajectory.h>
#include <corbo-numerics/finite_differences_collocation.h>
#include <corbo-optimization/hyper_graph/scalar_vertex.h>
#include <corbo-optimization/hyper_graph/vector_vertex.h>
#include <corbo-systems/system_dynamics_interface.h>

namespace corbo {
class OptimizationEdgeSet {};
class BaseEdge {};
class BaseMixedEdge {};
template <class T> class Factory { public: static Factory& instance() { static Factory f; return f; } };
class StagePreprocessor { public: using Ptr = std::shared_ptr<StagePreprocessor>; };

class DiscretizationGridInterface;
// bounds + the per-cycle update of the stage functions (association of obstacles / via-points): a no-op here, the rows are exercised on their own
struct NlpFunctions {
    Eigen::VectorXd x_lb, x_ub, u_lb, u_ub;
    void checkAndInitializeBoundDimensions(int x_dim, int u_dim) {
        auto fill = [](Eigen::VectorXd& v, int n, double val) { if (v.size() != n) { v = Eigen::VectorXd(n); for (int i = 0; i < n; ++i) v[i] = val; } };
        fill(x_lb, x_dim, -CORBO_INF_DBL); fill(x_ub, x_dim, CORBO_INF_DBL); fill(u_lb, u_dim, -CORBO_INF_DBL); fill(u_ub, u_dim, CORBO_INF_DBL);
    }
    // the per-cycle update of the stage functions: forwarded to whoever registered (the optimal-control-problem stand-in hands it to the stage cost and the
    // stage inequalities, i.e. the reference's obstacle and via-point association); nobody registered: nothing to do
    std::function<bool(int, double, ReferenceTrajectoryInterface&, ReferenceTrajectoryInterface&, ReferenceTrajectoryInterface*, bool, const Eigen::VectorXd&, const std::vector<double>&,
                       const DiscretizationGridInterface*)> on_update;
    bool update(int n, double t, ReferenceTrajectoryInterface& xref, ReferenceTrajectoryInterface& uref, ReferenceTrajectoryInterface* sref, bool single_dt, const Eigen::VectorXd& x0,
                const std::vector<double>& dts, const DiscretizationGridInterface* grid) { return on_update ? on_update(n, t, xref, uref, sref, single_dt, x0, dts, grid) : false; }
};
struct GridUpdateResult {
    bool vertices_updated = false, edges_updated = false;
    bool updated() const { return vertices_updated || edges_updated; }
};

class DiscretizationGridInterface {
 public:
    using Ptr = std::shared_ptr<DiscretizationGridInterface>;
    virtual ~DiscretizationGridInterface() = default;
    virtual Ptr getInstance() const = 0;
    virtual GridUpdateResult update(const Eigen::VectorXd& x0, ReferenceTrajectoryInterface& xref, ReferenceTrajectoryInterface& uref, NlpFunctions& nlp_fun, OptimizationEdgeSet& edges,
                                    SystemDynamicsInterface::Ptr dynamics, bool new_run, const Time& t, ReferenceTrajectoryInterface* sref = nullptr, const Eigen::VectorXd* prev_u = nullptr,
                                    double prev_u_dt = 0, ReferenceTrajectoryInterface* xinit = nullptr, ReferenceTrajectoryInterface* uinit = nullptr) = 0;
    virtual double getFirstDt() const = 0;
    virtual double getFinalTime() const = 0;
    virtual bool hasConstantControls() const = 0;
    virtual bool hasSingleDt() const = 0;
    virtual bool isTimeVariableGrid() const = 0;
    virtual bool isUniformGrid() const = 0;
    virtual bool providesStateTrajectory() const = 0;
    virtual bool getFirstControlInput(Eigen::VectorXd& u0) = 0;
    virtual void getStateAndControlTimeSeries(TimeSeries::Ptr x_sequence, TimeSeries::Ptr u_sequence, double t_max = CORBO_INF_DBL) const = 0;
    virtual void clear() = 0;
    virtual bool isEmpty() const = 0;
    virtual void setN(int n, bool try_resample = true) = 0;
    virtual void setInitialDt(double dt) = 0;
    virtual double getInitialDt() const = 0;
    virtual int getInitialN() const = 0;
    virtual int getN() const = 0;
    virtual std::vector<VertexInterface*>& getActiveVertices() = 0;
    virtual void getVertices(std::vector<VertexInterface*>& vertices) = 0;
    void setModified(bool m) { _modified = m; }
    bool isModified() const { return _modified; }
    void setPreviousControl(const Eigen::VectorXd& prev_u, double prev_u_dt) { _u_prev.values() = prev_u; _u_prev_dt.value() = prev_u_dt; }
    void setLastControlRef(const Eigen::VectorXd& u_ref) { _u_ref.values() = u_ref; }
 protected:
    virtual void computeActiveVertices() = 0;
    VectorVertex _u_prev, _u_ref;
    ScalarVertex _u_prev_dt;
    bool _modified = true;
};
}  // namespace corbo

// ORACLE (test infrastructure only): the INTERFACE of corbo::DiscretizationGridInterface as the reference's grid classes override it
// (include/mpc_local_planner/optimal_control/full_discretization_grid_base_se2.h) and the types in its signatures, reduced to what
// src/optimal_control/full_discretization_grid_base_se2.cpp and finite_differences_variable_grid_se2.cpp touch.  No corbo code.
#pragma once
#include <string>
#include <initializer_list>
#include <functional>
#include <corbo-core/time_series.h>
#include <corbo-core/console.h>
#include <corbo-core/reference_trajectory.h>
#include <corbo-numerics/finite_differences_collocation.h>
#include <corbo-optimization/hyper_graph/scalar_vertex.h>
#include <corbo-optimization/hyper_graph/vector_vertex.h>
#include <corbo-systems/system_dynamics_interface.h>

namespace corbo {
// an edge as a RECORD: its kind, the grid point it belongs to and the vertices it connects (oracle/ref_wrap_edges.cpp dumps them); no evaluation
class BaseEdge {
 public:
    using Ptr = std::shared_ptr<BaseEdge>;
    virtual ~BaseEdge() = default;
    BaseEdge() = default;
    BaseEdge(const std::string& kind_, int k_, std::initializer_list<const VertexInterface*> v) : kind(kind_), k(k_), vertices(v) {}
    std::string kind, note;
    int k = -1;
    std::vector<const VertexInterface*> vertices;
};
class OptimizationEdgeSet {
 public:
    void clear() { objective.clear(); equalities.clear(); inequalities.clear(); }
    void addObjectiveEdge(BaseEdge::Ptr e) { objective.push_back(e); }
    void addEqualityEdge(BaseEdge::Ptr e) { equalities.push_back(e); }
    void addInequalityEdge(BaseEdge::Ptr e) { inequalities.push_back(e); }
    std::vector<BaseEdge::Ptr> objective, equalities, inequalities;
};
// what createEdges asks a stage function for
struct StageFunctionHandle {
    using Ptr = std::shared_ptr<StageFunctionHandle>;
    bool integral_terms = false, equality = false;
    bool hasIntegralTerms(int) const { return integral_terms; }
    bool isEqualityConstraint() const { return equality; }
};
class BaseMixedEdge {};
template <class T> class Factory { public: static Factory& instance() { static Factory f; return f; } };
class StagePreprocessor { public: using Ptr = std::shared_ptr<StagePreprocessor>; };

class DiscretizationGridInterface;
// bounds + the per-cycle update of the stage functions (association of obstacles / via-points): a no-op here, the rows are exercised on their own
struct NlpFunctions {
    Eigen::VectorXd x_lb, x_ub, u_lb, u_ub;
    // ---- for createEdges (src/optimal_control/finite_differences_grid_se2.cpp): the functions it asks about, and its three requests for edges, answered with RECORDS of the
    // vertices that were handed over (which edges corbo would really build from them -- e.g. none for a term whose vertices are all fixed -- is corbo's business)
    StageFunctionHandle::Ptr stage_cost, stage_equalities, stage_inequalities, final_stage_constraints;
    bool has_final_cost = false;
    void getNonIntegralStageFunctionEdges(int k, VectorVertex& x_k, VectorVertex& u_k, ScalarVertex& dt, VectorVertex& u_prev, ScalarVertex& dt_prev, std::vector<BaseEdge::Ptr>& cost_terms,
                                          std::vector<BaseEdge::Ptr>&, std::vector<BaseEdge::Ptr>&) {
        cost_terms.push_back(std::make_shared<BaseEdge>("non_integral_stage_functions", k, std::initializer_list<const VertexInterface*>{&x_k, &u_k, &dt, &u_prev, &dt_prev}));
    }
    BaseEdge::Ptr getFinalStateCostEdge(int k, VectorVertex& xf) { return has_final_cost ? std::make_shared<BaseEdge>("final_state_cost", k, std::initializer_list<const VertexInterface*>{&xf}) : nullptr; }
    BaseEdge::Ptr getFinalStateConstraintEdge(int k, VectorVertex& xf) {
        return final_stage_constraints ? std::make_shared<BaseEdge>("final_state_constraint", k, std::initializer_list<const VertexInterface*>{&xf}) : nullptr;
    }
    void getFinalControlDeviationEdges(int n, VectorVertex& u_ref, VectorVertex& u_prev, ScalarVertex& dt, std::vector<BaseEdge::Ptr>&, std::vector<BaseEdge::Ptr>&, std::vector<BaseEdge::Ptr>& ineq_terms) {
        ineq_terms.push_back(std::make_shared<BaseEdge>("final_control_deviation", n, std::initializer_list<const VertexInterface*>{&u_ref, &u_prev, &dt}));
    }
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

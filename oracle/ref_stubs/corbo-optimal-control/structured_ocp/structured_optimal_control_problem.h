// ORACLE (test infrastructure only).  NOT control-box-rst: corbo::StructuredOptimalControlProblem as the reference's Controller builds and drives it
// (src/controller.cpp:483-805, :166-172): a record of everything that was set, and compute() = DiscretizationGridInterface::update() -- for the grids of this
// package that is the REFERENCE's own code (cold start from the initial state trajectory, warm start, grid adaptation, obstacle / via-point association through
// NlpFunctions::update) -- followed by a "solver" the test plugs in (solve_hook: it sees the grid after update() and writes its result into the vertices).
#pragma once
#include <corbo-controllers/predictive_controller.h>
#include <corbo-optimal-control/functions/final_state_constraints.h>
#include <corbo-optimal-control/functions/final_state_cost.h>
#include <corbo-optimal-control/functions/stage_functions.h>
#include <corbo-optimization/hyper_graph/hyper_graph_optimization_problem_edge_based.h>
#include <corbo-optimization/solver/nlp_solver_ipopt.h>
#include <corbo-systems/system_dynamics_interface.h>
#include <functional>

namespace corbo {
class StructuredOptimalControlProblem : public OptimalControlProblemInterface {
 public:
    using Ptr = std::shared_ptr<StructuredOptimalControlProblem>;
    StructuredOptimalControlProblem(DiscretizationGridInterface::Ptr grid, SystemDynamicsInterface::Ptr dynamics, BaseHyperGraphOptimizationProblem::Ptr optim_prob, NlpSolverInterface::Ptr solver)
        : grid(grid), dynamics(dynamics), optim_prob(optim_prob), solver(solver) {}
    void setControlBounds(const Eigen::Vector2d& lb, const Eigen::Vector2d& ub) {
        functions.u_lb = Eigen::VectorXd(2); functions.u_ub = Eigen::VectorXd(2);
        functions.u_lb[0] = lb.x(); functions.u_lb[1] = lb.y(); functions.u_ub[0] = ub.x(); functions.u_ub[1] = ub.y();
    }
    void setStageCost(StageCost::Ptr c) { stage_cost = c; }
    void setFinalStageCost(FinalStageCost::Ptr c) { final_stage_cost = c; }
    void setFinalStageConstraint(FinalStageConstraint::Ptr c) { final_stage_constraint = c; }
    void setStageInequalityConstraint(StageInequalityConstraint::Ptr c) { stage_inequalities = c; }
    bool initialize() override {
        if (!grid || !dynamics || !optim_prob || !solver) return false;
        functions.on_update = [this](int n, double t, ReferenceTrajectoryInterface& xref, ReferenceTrajectoryInterface& uref, ReferenceTrajectoryInterface* sref, bool single_dt,
                                     const Eigen::VectorXd& x0, const std::vector<double>& dts, const DiscretizationGridInterface* g) {
            bool changed = false;
            if (stage_cost) changed |= stage_cost->update(n, t, xref, uref, sref, single_dt, x0, nullptr, dts, g);
            if (stage_inequalities) changed |= stage_inequalities->update(n, t, xref, uref, sref, single_dt, x0, nullptr, dts, g);
            return changed;
        };
        initialized = true;
        return true;
    }
    void reset() override { if (grid) grid->clear(); ++n_resets; }
    void setPreviousControlInput(const Eigen::Ref<const Eigen::VectorXd>& u, double dt) override { u_prev = Eigen::VectorXd(u); u_prev_dt = dt; }
    bool compute(const Eigen::VectorXd& x, ReferenceTrajectoryInterface& xref, ReferenceTrajectoryInterface& uref, ReferenceTrajectoryInterface* sref, const Time& t, bool new_run,
                 ReferenceTrajectoryInterface* xinit, ReferenceTrajectoryInterface* uinit) override {
        if (!initialized) return false;
        grid->update(x, xref, uref, functions, edges, dynamics, new_run, t, sref, &u_prev, u_prev_dt, xinit, uinit);
        ++n_computes;
        return solve_hook ? solve_hook(*this) : true;
    }
    void getTimeSeries(TimeSeries::Ptr x_sequence, TimeSeries::Ptr u_sequence) override { grid->getStateAndControlTimeSeries(x_sequence, u_sequence); }

    DiscretizationGridInterface::Ptr grid;
    SystemDynamicsInterface::Ptr dynamics;
    BaseHyperGraphOptimizationProblem::Ptr optim_prob;
    NlpSolverInterface::Ptr solver;
    NlpFunctions functions;
    OptimizationEdgeSet edges;
    StageCost::Ptr stage_cost;
    FinalStageCost::Ptr final_stage_cost;
    FinalStageConstraint::Ptr final_stage_constraint;
    StageInequalityConstraint::Ptr stage_inequalities;
    Eigen::VectorXd u_prev;
    double u_prev_dt = 0;
    bool initialized = false;
    int n_resets = 0, n_computes = 0;
    std::function<bool(StructuredOptimalControlProblem&)> solve_hook;
};
}  // namespace corbo

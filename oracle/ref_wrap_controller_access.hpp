// ORACLE (test infrastructure only): access to the protected members of the reference's Controller / grids / constraint classes (through derived classes), reading and
// writing the grid's vertex values, and plugging a stand-in "solver" into the optimal-control-problem stand-in.  Shared by oracle/ref_wrap_controller.cpp and
// oracle/ref_wrap_plugin.cpp.
#pragma once
#include <functional>
#include <vector>
#include <mpc_local_planner/controller.h>
#include <mpc_local_planner/optimal_control/finite_differences_variable_grid_se2.h>
#include <mpc_local_planner/optimal_control/min_time_via_points_cost.h>

namespace ref_access {
using namespace mpc_local_planner;
// protected members, read through derived classes
struct CtlAccess : Controller {
    using Controller::_grid; using Controller::_dynamics; using Controller::_solver; using Controller::_structured_ocp; using Controller::_inequality_constraint;
    using Controller::_force_reinit_new_goal_dist; using Controller::_force_reinit_new_goal_angular; using Controller::_guess_backwards_motion;
    using Controller::_force_reinit_num_steps; using Controller::_prefer_x_feedback; using Controller::_publish_ocp_results; using Controller::_print_cpu_time;
    using Controller::_num_ocp_iterations; using Controller::_auto_update_prev_control; using Controller::_x_seq_init; using Controller::_ocp_seq; using Controller::_robot_type;
    using Controller::_initial_plan_estimate_orientation;
};
struct GridAccess : FiniteDifferencesVariableGridSE2 {
    using FiniteDifferencesVariableGridSE2::_x_seq; using FiniteDifferencesVariableGridSE2::_u_seq; using FiniteDifferencesVariableGridSE2::_xf; using FiniteDifferencesVariableGridSE2::_dt;
    using FiniteDifferencesVariableGridSE2::_n_ref; using FiniteDifferencesVariableGridSE2::_dt_ref; using FiniteDifferencesVariableGridSE2::_warm_start;
    using FiniteDifferencesVariableGridSE2::_xf_fixed; using FiniteDifferencesVariableGridSE2::_dt_lb; using FiniteDifferencesVariableGridSE2::_dt_ub;
    using FiniteDifferencesVariableGridSE2::_cost_integration; using FiniteDifferencesVariableGridSE2::_fd_eval; using FiniteDifferencesVariableGridSE2::_grid_adapt;
    using FiniteDifferencesVariableGridSE2::_n_max; using FiniteDifferencesVariableGridSE2::_n_min; using FiniteDifferencesVariableGridSE2::_dt_hyst_ratio;
    using FiniteDifferencesVariableGridSE2::_u_prev; using FiniteDifferencesVariableGridSE2::_u_prev_dt;
};
struct BaseGridAccess : FiniteDifferencesGridSE2 {
    using FiniteDifferencesGridSE2::_x_seq; using FiniteDifferencesGridSE2::_u_seq; using FiniteDifferencesGridSE2::_xf; using FiniteDifferencesGridSE2::_dt;
    using FiniteDifferencesGridSE2::_n_ref; using FiniteDifferencesGridSE2::_dt_ref; using FiniteDifferencesGridSE2::_warm_start; using FiniteDifferencesGridSE2::_xf_fixed;
    using FiniteDifferencesGridSE2::_dt_lb; using FiniteDifferencesGridSE2::_dt_ub; using FiniteDifferencesGridSE2::_cost_integration; using FiniteDifferencesGridSE2::_fd_eval;
    using FiniteDifferencesGridSE2::_u_prev; using FiniteDifferencesGridSE2::_u_prev_dt;
};
struct IneqAccess : StageInequalitySE2 {
    using StageInequalitySE2::_min_obstacle_dist; using StageInequalitySE2::_obstacle_filter_force_inclusion_dist; using StageInequalitySE2::_obstacle_filter_cutoff_dist;
    using StageInequalitySE2::_enable_dynamic_obstacles; using StageInequalitySE2::_du_lb; using StageInequalitySE2::_du_ub; using StageInequalitySE2::_relevant_obstacles;
};
struct ViaAccess : MinTimeViaPointsCost {
    using MinTimeViaPointsCost::_via_points_ordered; using MinTimeViaPointsCost::_vp_position_weight; using MinTimeViaPointsCost::_vp_orientation_weight;
};

template <class G> void read_grid(G& g, std::vector<double>& x, std::vector<double>& u, double& dt, int& n) {
    n = g.getN();
    x.assign((size_t)3 * n, 0.0); u.assign((size_t)2 * (n > 1 ? n - 1 : 0), 0.0);
    for (int k = 0; k < n; ++k) { const Eigen::VectorXd& s = g.getState(k); for (int i = 0; i < 3; ++i) x[3 * k + i] = s[i]; }
    for (int k = 0; k < n - 1; ++k) for (int j = 0; j < 2; ++j) u[2 * k + j] = g._u_seq[(size_t)k].values()[j];
    dt = g.getDt();
}
template <class G> void write_grid(G& g, const std::vector<double>& x, const std::vector<double>& u, double dt, int n) {
    for (int k = 0; k < n - 1; ++k) for (int i = 0; i < 3; ++i) g._x_seq[(size_t)k].values()[i] = x[3 * k + i];
    for (int i = 0; i < 3; ++i) g._xf.values()[i] = x[3 * (n - 1) + i];
    for (int k = 0; k < n - 1; ++k) for (int j = 0; j < 2; ++j) g._u_seq[(size_t)k].values()[j] = u[2 * k + j];
    g._dt.value() = dt;
}

typedef int (*solve_cb)(int n, double* x, double* u, double* dt, const double* u_prev, double u_prev_dt);
struct GuessRecord { std::vector<double> x, u; double dt = 0; int n = 0; };
// the "solver" of the controller's optimal control problem: records the grid as update() left it, lets *cb (may be null: keep the guess) write a result into it
inline void install_solver(Controller& ctl, solve_cb* cb, GuessRecord* rec) {
    CtlAccess& c = static_cast<CtlAccess&>(ctl);
    if (!c._structured_ocp) return;
    c._structured_ocp->solve_hook = [cb, rec](corbo::StructuredOptimalControlProblem& ocp) {
        bool ok = true;
        auto run = [&](auto& g) {
            read_grid(g, rec->x, rec->u, rec->dt, rec->n);
            if (!*cb) return;
            std::vector<double> x = rec->x, u = rec->u; double dt = rec->dt;
            double up[2] = {g._u_prev.values()[0], g._u_prev.values()[1]};
            ok = (*cb)(rec->n, x.data(), u.data(), &dt, up, g._u_prev_dt.value()) != 0;
            write_grid(g, x, u, dt, rec->n);
        };
        if (auto* vg = dynamic_cast<FiniteDifferencesVariableGridSE2*>(ocp.grid.get())) run(static_cast<GridAccess&>(*vg));
        else run(static_cast<BaseGridAccess&>(*dynamic_cast<FiniteDifferencesGridSE2*>(ocp.grid.get())));
        return ok;
    };
}
}  // namespace ref_access

// mpc_problem.hpp -- host-side translation of the ABI's mpc_config (the reference's ROS
// parameter set, src/controller.cpp:225-805) into the device-side Problem<T> record.
#pragma once
#include "../../include/mpc_hip.h"
#include "mpc_core.hpp"

namespace mpc {

// what mpc_config.line_search = MPC_LS_DEFAULT means: 0 the l1 merit, 1 Ipopt's filter
constexpr int kDefaultLineSearch = 1;

template <typename T>
inline void fill_problem(const mpc_config& c, mpc::Problem<T>& P) {
    P.model = c.model;
    P.n = c.n;
    P.dt_free = c.dt_free ? 1 : 0;
    for (int i = 0; i < 3; ++i) P.xf_fixed[i] = c.xf_fixed[i] ? 1 : 0;
    P.objective = c.objective == MPC_OBJ_MIN_TIME_VIA_POINTS ? MPC_OBJ_MIN_TIME : c.objective;
    P.via = c.objective == MPC_OBJ_MIN_TIME_VIA_POINTS ? 1 : 0;
    P.n_via = P.via ? c.max_via_points : 0;
    P.vp_ordered = c.via_points_ordered ? 1 : 0;
    P.vp_wp = T(c.vp_position_weight);
    P.vp_wo = T(c.vp_orientation_weight);
    P.collocation = c.collocation;
    P.has_Qf = c.has_Qf ? 1 : 0;
    P.max_iter = c.max_iter > 0 ? c.max_iter : 100;
    P.p0 = T(c.model_params[0]);
    P.p1 = T(c.model_params[1]);
    P.dt_ref = T(c.dt_ref);
    P.dt_lb = T(c.dt_lb);
    P.dt_ub = T(c.dt_ub);
    // integral form on the fixed-dt grid (quadratic_cost_se2.cpp:54-83, left sum finite_differences_grid_se2.cpp:61-75): every stage
    // term is multiplied by the constant dt, i.e. the weights are scaled; the terminal cost is not.  (dt free + integral form: see below.)
    const double wsc = (c.integral_form && !c.dt_free) ? c.dt_ref : 1.0;
    // integral form on the variable grid (dt is a decision variable): the weights stay unscaled, the kernel multiplies by the current dt
    // and carries the state-dt / control-dt coupling of the Hessian (wave kernel, A-form slots A05 A15 A25 A56 A57)
    P.integral_form = (c.objective == MPC_OBJ_QUADRATIC && c.integral_form && c.dt_free) ? 1 : 0;
    for (int i = 0; i < 3; ++i) { P.Q[i] = T(c.Q[i] * wsc); P.Qf[i] = T(c.Qf[i]); }
    for (int j = 0; j < 2; ++j) {
        P.R[j] = T(c.R[j] * wsc);
        P.u_lb[j] = T(c.u_lb[j]);
        P.u_ub[j] = T(c.u_ub[j]);
        P.rate_on[j] = c.du_lb[j] > -1e29 ? 1 : 0;
        P.rate_on[2 + j] = c.du_ub[j] < 1e29 ? 1 : 0;
        P.rate_lim[j] = T(P.rate_on[j] ? c.du_lb[j] : 0.0);
        P.rate_lim[2 + j] = T(P.rate_on[2 + j] ? c.du_ub[j] : 0.0);
    }
    const bool f32 = sizeof(T) == 4;
    P.tol = T(c.tol > 0 ? c.tol : (f32 ? 1e-4 : 1e-8));
    P.mu_init = T(c.mu_init > 0 ? c.mu_init : 0.1);
    P.mu_init_warm = T(c.mu_init_warm > 0 ? c.mu_init_warm : (c.mu_init > 0 ? c.mu_init : 0.1));
    P.n_obst = c.max_obstacles > 0 ? c.max_obstacles : 0;
    P.n_vert = c.max_vertices > 0 ? c.max_vertices : 1;
    P.obst_rows = P.n_obst > 0 ? (c.max_obstacle_rows > 0 ? c.max_obstacle_rows : 4) : 0;
    P.footprint_kind = c.footprint_kind;
    P.d_min = T(c.min_obstacle_dist);
    P.force_incl = T(c.force_inclusion_dist);
    P.cutoff = T(c.cutoff_dist);
    P.fp_radius = T(c.footprint_kind == MPC_FOOTPRINT_CIRCLE ? c.footprint_radius : 0.0);
    for (int i = 0; i < 4; ++i) P.fp_line[i] = T(c.footprint_params[i]);
    P.dyn_obst = (c.enable_dynamic_obstacles && c.max_obstacles > 0) ? 1 : 0;
    P.fp_nv = c.footprint_kind == MPC_FOOTPRINT_POLYGON ? (c.footprint_n_vertices < 16 ? c.footprint_n_vertices : 16) : 0;
    for (int i = 0; i < 32; ++i) P.fp_poly[i] = T(i < 2 * P.fp_nv ? c.footprint_vertices[i] : 0.0);
    // TerminalBallSE2 (final_state_conditions_se2.cpp:54-64): the edge exists only with an unfixed final state (finite_differences_grid_se2.cpp:128-143)
    P.ball = (c.terminal_ball && !(c.xf_fixed[0] && c.xf_fixed[1] && c.xf_fixed[2])) ? 1 : 0;
    for (int i = 0; i < 3; ++i) P.ball_S[i] = T(c.terminal_ball_S[i]);
    P.ball_gamma = T(c.terminal_ball_gamma);
    P.n_cand = c.n_candidates > 1 ? (c.n_candidates < MPC_MAX_CANDIDATES ? c.n_candidates : MPC_MAX_CANDIDATES) : 1;
    for (int k = 0; k < 4; ++k) {
        P.cand_kind[k] = (c.n_candidates > 1 && k < P.n_cand) ? c.candidate_kind[k] : MPC_CAND_REFERENCE;
        P.cand_max_iter[k] = (c.n_candidates > 1 && c.candidate_max_iter[k] > 0) ? c.candidate_max_iter[k] : P.max_iter;
        P.cand_param[k] = T((c.n_candidates > 1 && c.candidate_param[k] > 0) ? c.candidate_param[k] : 2.0);
    }
    P.cand_blend = c.candidate_blend > 0 ? c.candidate_blend : 8;
    P.mu_init_dual = T(c.mu_init_dual > 0 ? c.mu_init_dual : 1e-3);
    P.hess_mode = c.hessian_mode == MPC_HESSIAN_CONVEXIFIED ? 1 : 0;
    // ---- cost variants
    const bool quad = c.objective == MPC_OBJ_QUADRATIC;
    P.hybrid = (quad && c.hybrid_cost_minimum_time) ? 1 : 0;
    for (int i = 0; i < 3; ++i) { P.Qo[i] = T(quad ? c.Q_offdiag[i] * wsc : 0.0); P.Qfo[i] = T(c.has_Qf ? c.Qf_offdiag[i] : 0.0); P.So[i] = T(c.terminal_ball_S_offdiag[i]); }
    P.Ro = T(quad ? c.R_offdiag * wsc : 0.0);
    const bool trapezoid = quad && c.integral_form && c.cost_integration == MPC_COST_TRAPEZOIDAL;
    // trapezoidal rule (finite_differences_grid_se2.cpp:63-68): every interval gives half of its state cost to either end, so against the left
    // sum x_0 loses half a term (a constant on the fixed grid: x_0 is not a variable) and the FINAL state gains 0.5 dt xd' Q xd.
    // Fixed grid: that is a terminal cost with weights 0.5 dt_ref Q on top of Qf.  Variable grid: handled in the kernel (P.trapz).
    P.trapz = (trapezoid && c.dt_free) ? 1 : 0;
    if (trapezoid && !c.dt_free) {
        for (int i = 0; i < 3; ++i) {
            P.Qf[i] = T((c.has_Qf ? c.Qf[i] : 0.0) + 0.5 * c.dt_ref * c.Q[i]);
            P.Qfo[i] = T((c.has_Qf ? c.Qf_offdiag[i] : 0.0) + 0.5 * c.dt_ref * c.Q_offdiag[i]);
        }
        P.has_Qf = 1;
    } else if (!c.has_Qf) {
        for (int i = 0; i < 3; ++i) P.Qf[i] = T(0);
    }
    P.pit = 1;
    P.pit_mu_min = P.tol > T(1e-6) ? P.tol : T(1e-6);      // the barrier subproblems below max(tol, 1e-6) are solved with the serial sweeps (measured: mpc_wave.hpp, pit_floor)
    P.acc_tol = T(c.acceptable_tol > 0 ? c.acceptable_tol : (c.acceptable_tol < 0 ? 0.0 : 1e-6));
    P.acc_iter = c.acceptable_iter > 0 ? c.acceptable_iter : (c.acceptable_iter < 0 ? 0 : 15);
    P.max_ticks = c.max_time_us > 0 ? 100ll * c.max_time_us : 0ll;
    P.mu_strategy = c.mu_strategy == MPC_MU_MONOTONE ? 1 : 0;
    P.line_search = c.line_search == MPC_LS_DEFAULT ? kDefaultLineSearch : (c.line_search == MPC_LS_FILTER ? 1 : 0);
    P.costx = (P.trapz || P.Ro != T(0)) ? 1 : 0;
    for (int i = 0; i < 3; ++i) if (P.Qo[i] != T(0) || P.Qfo[i] != T(0) || (P.ball && P.So[i] != T(0))) P.costx = 1;
}


}  // namespace mpc

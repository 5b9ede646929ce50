// ORACLE (test infrastructure only): C entry points around the reference's src/optimal_control/stage_inequality_se2.cpp, compiled from where it lies
// under /root/reference (second translation unit of oracle/_ref/libmpc_ref.so; see oracle/ref_wrap.cpp and the notes in oracle/ref_stubs/).
// What runs is the reference's own StageInequalitySE2::update (the obstacle association of every grid point, :50-162),
// computeNonIntegralStateTerm / computeNonIntegralStateDtTerm (clearance rows of static / moving obstacles, :164-189) and
// computeNonIntegralControlDeviationTerm (control-rate rows, :191-226) -- on POINT obstacles and the POINT footprint, whose distance is unambiguous
// (teb's other shapes are not available here).
#include <vector>

#include "ref_wrap_common.hpp"                                                        // a reference grid object holding given states
#include <mpc_local_planner/optimal_control/stage_inequality_se2.h>                   // the reference's header
#include <mpc_local_planner/optimal_control/min_time_via_points_cost.h>               // the reference's header

namespace {
// states only: controls zero, dt as given
struct StateGrid {
    Probe<mpc_local_planner::FiniteDifferencesGridSE2> g;
    corbo::NlpFunctions nlp;
    std::vector<Eigen::VectorXd> x;
    StateGrid(int n, const double* states, double dt) {
        std::vector<double> u((size_t)2 * (n > 1 ? n - 1 : 1), 0.0);
        const bool fx[3] = {true, true, true};
        fill(g, nlp, n, states, u.data(), dt, fx);
        for (int k = 0; k < n; ++k) x.push_back(g.getState(k));
    }
};
struct RowProbe : mpc_local_planner::StageInequalitySE2 {       // the association result is a protected member
    using StageInequalitySE2::_relevant_obstacles;
    using StageInequalitySE2::_relevant_dyn_obstacles;
};
}  // namespace

extern "C" {
// states [n][3]; obstacles: xy [n_obst][2], vel [n_obst][2], dynamic [n_obst].  Out, per grid point k: indices of the associated static / dynamic
// obstacles in the reference's order (rel_idx / dyn_idx [n][max_out], counts rel_cnt / dyn_cnt [n]) and the rows evaluated at the states
// (rows / dyn_rows [n][max_out]).  Returns 0, or -1 if some grid point has more than max_out associated obstacles.
int ref_associate(int n, const double* states, int n_obst, const double* obst_xy, const double* obst_vel, const int* dynamic, double min_dist, double force_incl,
                  double cutoff, int enable_dyn, double dt, int max_out, int* rel_idx, int* rel_cnt, int* dyn_idx, int* dyn_cnt, double* rows, double* dyn_rows) {
    teb_local_planner::ObstContainer obstacles;
    for (int j = 0; j < n_obst; ++j)
        obstacles.push_back(std::make_shared<teb_local_planner::PointObstacle>(obst_xy[2 * j], obst_xy[2 * j + 1], obst_vel[2 * j], obst_vel[2 * j + 1], dynamic[j] != 0));
    StateGrid grid(n, states, dt);
    RowProbe row;
    row.setObstacleVector(obstacles);
    row.setRobotFootprintModel(std::make_shared<teb_local_planner::PointRobotFootprint>());
    row.setMinimumDistance(min_dist);
    row.setObstacleFilterParameters(force_incl, cutoff);
    row.setEnableDynamicObstacles(enable_dyn != 0);
    corbo::ReferenceTrajectoryInterface xref, uref;
    row.update(n, 0.0, xref, uref, nullptr, true, grid.x[0], nullptr, std::vector<double>(), &grid.g);
    auto index_of = [&](const teb_local_planner::ObstaclePtr& o) { for (int j = 0; j < n_obst; ++j) if (obstacles[(size_t)j].get() == o.get()) return j; return -1; };
    int rc = 0;
    for (int k = 0; k < n; ++k) {
        const auto& rel = row._relevant_obstacles[(size_t)k];
        const auto& dyn = row._relevant_dyn_obstacles[(size_t)k];
        rel_cnt[k] = (int)rel.size(); dyn_cnt[k] = (int)dyn.size();
        if ((int)rel.size() > max_out || (int)dyn.size() > max_out) { rc = -1; continue; }
        for (size_t i = 0; i < rel.size(); ++i) rel_idx[k * max_out + (int)i] = index_of(rel[i]);
        for (size_t i = 0; i < dyn.size(); ++i) dyn_idx[k * max_out + (int)i] = index_of(dyn[i]);
        if (!rel.empty()) { Eigen::VectorXd c((int)rel.size()); row.computeNonIntegralStateTerm(k, grid.x[(size_t)k], c); for (int i = 0; i < c.size(); ++i) rows[k * max_out + i] = c[i]; }
        if (!dyn.empty()) { Eigen::VectorXd c((int)dyn.size()); row.computeNonIntegralStateDtTerm(k, grid.x[(size_t)k], dt, c); for (int i = 0; i < c.size(); ++i) dyn_rows[k * max_out + i] = c[i]; }
    }
    return rc;
}

// control-rate rows of grid point k: returns their number (finite lower bounds first, then finite upper bounds, :207-225); bounds beyond +-corbo::CORBO_INF_DBL
// mean "none"
int ref_control_deviation_rows(int k, const double* u_k, const double* u_prev, double dt_prev, const double* du_lb, const double* du_ub, double* out) {
    RowProbe row;
    Eigen::VectorXd lb(2), ub(2), u(2), up(2);
    for (int i = 0; i < 2; ++i) { lb[i] = du_lb[i]; ub[i] = du_ub[i]; u[i] = u_k[i]; up[i] = u_prev[i]; }
    row.setControlDeviationBounds(lb, ub);
    const double three_states[9] = {0, 0, 0, 1, 0, 0, 2, 0, 0};
    StateGrid grid(3, three_states, 0.1);
    corbo::ReferenceTrajectoryInterface xref, uref;
    row.update(3, 0.0, xref, uref, nullptr, true, grid.x[0], nullptr, std::vector<double>(), &grid.g);      // counts the finite bounds (:150-156)
    const int m = row.getNonIntegralControlDeviationTermDimension(k);
    Eigen::VectorXd c(m);
    row.computeNonIntegralControlDeviationTerm(k, u, up, dt_prev, c);
    for (int i = 0; i < m; ++i) out[i] = c[i];
    return m;
}
double ref_corbo_inf(void) { return corbo::CORBO_INF_DBL; }

// MinTimeViaPointsCost (src/optimal_control/min_time_via_points_cost.cpp, compiled from the reference): update() attaches every via-point
// (via [n_via][3]) to a grid point of the states [n][3]; out: attached[n_via] = grid index or -1 (skipped), terms[n_via] = the cost term of that via-point
// evaluated at its grid point's state (computeNonIntegralStateTerm), *dt_term = computeNonIntegralDtTerm(0, dt)
namespace {
struct ViaProbe : mpc_local_planner::MinTimeViaPointsCost { using MinTimeViaPointsCost::_vp_association; };
}
void ref_via_points(int n, const double* states, int n_via, const double* via, double w_pos, double w_orient, int ordered, double dt, int* attached, double* terms, double* dt_term) {
    mpc_local_planner::MinTimeViaPointsCost::ViaPointContainer vps;
    for (int v = 0; v < n_via; ++v) vps.emplace_back(via[3 * v], via[3 * v + 1], via[3 * v + 2]);
    StateGrid grid(n, states, dt);
    ViaProbe cost;
    cost.setViaPointContainer(vps);
    cost.setViaPointWeights(w_pos, w_orient);
    cost.setViaPointOrderedMode(ordered != 0);
    corbo::ReferenceTrajectoryInterface xref, uref;
    cost.update(n, 0.0, xref, uref, nullptr, true, grid.x[0], nullptr, std::vector<double>(), &grid.g);
    for (int v = 0; v < n_via; ++v) { attached[v] = -1; terms[v] = 0.0; }
    for (int k = 0; k < n; ++k) {
        const auto& item = cost._vp_association[(size_t)k];
        if (item.first.empty()) continue;
        Eigen::VectorXd c((int)item.first.size());
        cost.computeNonIntegralStateTerm(k, grid.x[(size_t)k], c);
        for (size_t i = 0; i < item.first.size(); ++i) {
            const int v = (int)(item.first[i] - vps.data());
            attached[v] = k; terms[v] = c[(int)i];
        }
    }
    Eigen::VectorXd d(1);
    cost.computeNonIntegralDtTerm(0, dt, d);
    *dt_term = d[0];
}
}  // extern "C"

// ORACLE (test infrastructure only): C entry points around REFERENCE code compiled from where it lies under /root/reference -- nothing of it is
// copied into this repository.  Compiled into oracle/_ref/libmpc_ref.so by `make -C oracle ref` (only where /root/reference exists):
//
//   include/mpc_local_planner/utils/math_utils.h                          normalize_theta, interpolate_angle, average_angles   (self-contained)
//   include/mpc_local_planner/systems/{unicycle_robot,simple_car,kinematic_bicycle_model}.h      the four robot models' dynamics()
//   include/mpc_local_planner/optimal_control/fd_collocation_se2.h        forward / midpoint / Crank-Nicolson collocation rows on SE(2)
//   include/mpc_local_planner/optimal_control/vector_vertex_se2.h         the vertex classes of the state variables: plus / plusUnfixed / setData / set (r05): oracle/ref_wrap_vertex.cpp
//   src/optimal_control/stage_inequality_se2.cpp (+ its header)           obstacle association, clearance rows, control-rate rows: oracle/ref_wrap_rows.cpp
//   src/optimal_control/min_time_via_points_cost.cpp (+ its header)       via-point association and cost terms, time term: oracle/ref_wrap_rows.cpp
//   src/optimal_control/full_discretization_grid_base_se2.cpp, finite_differences_variable_grid_se2.cpp (+ headers), src/utils/time_series_se2.cpp
//                                                                         the grid classes and the SE(2) time series: oracle/ref_wrap_grid.cpp
//   src/optimal_control/quadratic_cost_se2.cpp, final_state_conditions_se2.cpp (+ headers)      cost terms, final-state cost, terminal ball: oracle/ref_wrap_cost.cpp
//   src/optimal_control/finite_differences_grid_se2.cpp                  createEdges with record edges (own library): oracle/ref_wrap_edges.cpp
//   src/controller.cpp (+ controller.h)                                  configure / step / isPoseTrajectoryFeasible / publishOptimalControlResult: oracle/ref_wrap_controller.cpp
//   src/mpc_local_planner_ros.cpp (+ its header)                         costmap / message obstacles, via-points, goal heading, footprint parameters: oracle/ref_wrap_plugin.cpp
//
// The model and collocation headers are written against Eigen and control_box_rst (corbo), neither of which is in this image; oracle/ref_stubs/
// provides the INTERFACES they derive from and the few element-wise vector operations they use (see the notes there).  The arithmetic that
// runs is the reference's own statements.  tests/golden/make_ref_vectors.py records its outputs on seeded inputs (tests/golden/ref_models_collocation.npz),
// tests/test_reference_pinned.py holds the repository's oracles, the C++ facade and the host build of the kernel's core to them.
#include <vector>

#include <mpc_local_planner/utils/math_utils.h>
#include <mpc_local_planner/systems/unicycle_robot.h>
#include <mpc_local_planner/systems/simple_car.h>
#include <mpc_local_planner/systems/kinematic_bicycle_model.h>
#include <mpc_local_planner/optimal_control/fd_collocation_se2.h>

namespace {
// model ids as in include/mpc_hip.h: 0 unicycle, 1 simple car (rear wheel), 2 simple car (front wheel), 3 kinematic bicycle (p0 = lr, p1 = lf)
std::shared_ptr<mpc_local_planner::BaseRobotSE2> make_model(int model, double p0, double p1) {
    using namespace mpc_local_planner;
    switch (model) {
        case 0: return std::make_shared<UnicycleModel>();
        case 1: return std::make_shared<SimpleCarModel>(p0);
        case 2: return std::make_shared<SimpleCarFrontWheelDrivingModel>(p0);
        default: return std::make_shared<KinematicBicycleModelVelocityInput>(p0, p1);
    }
}
Eigen::VectorXd vec(const double* p, int n) { Eigen::VectorXd v(n); for (int i = 0; i < n; ++i) v[i] = p[i]; return v; }
}  // namespace

extern "C" {
double ref_normalize_theta(double th) { return mpc_local_planner::normalize_theta(th); }
double ref_interpolate_angle(double a1, double a2, double f) { return mpc_local_planner::interpolate_angle(a1, a2, f); }
double ref_average_angles(const double* a, int n) { return mpc_local_planner::average_angles(std::vector<double>(a, a + n)); }

// f = dynamics(x, u) for count samples; x [count][3], u [count][2], f [count][3]
void ref_dynamics(int model, double p0, double p1, int count, const double* x, const double* u, double* f) {
    auto m = make_model(model, p0, p1);
    for (int i = 0; i < count; ++i) {
        Eigen::VectorXd xv = vec(x + 3 * i, 3), uv = vec(u + 2 * i, 2), fv(3);
        m->dynamics(xv, uv, fv);
        for (int a = 0; a < 3; ++a) f[3 * i + a] = fv[a];
    }
}

// error = computeEqualityConstraint(x1, u1, x2, dt) of the collocation rule `method` (0 forward, 1 midpoint, 2 Crank-Nicolson differences)
void ref_collocation(int method, int model, double p0, double p1, int count, const double* x1, const double* u1, const double* x2, const double* dt, double* err) {
    using namespace mpc_local_planner;
    auto m = make_model(model, p0, p1);
    ForwardDiffCollocationSE2 fwd; MidpointDiffCollocationSE2 mid; CrankNicolsonDiffCollocationSE2 cn;
    corbo::FiniteDifferencesCollocationInterface* rule = method == 0 ? (corbo::FiniteDifferencesCollocationInterface*)&fwd : (method == 1 ? (corbo::FiniteDifferencesCollocationInterface*)&mid : &cn);
    for (int i = 0; i < count; ++i) {
        Eigen::VectorXd a = vec(x1 + 3 * i, 3), u = vec(u1 + 2 * i, 2), b = vec(x2 + 3 * i, 3), e(3);
        rule->computeEqualityConstraint(a, u, b, dt[i], *m, e);
        for (int k = 0; k < 3; ++k) err[3 * i + k] = e[k];
    }
}
}  // extern "C"

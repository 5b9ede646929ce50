// ORACLE (test infrastructure only): the reference's src/controller.cpp -- Controller::configure (configureRobotDynamics / configureGrid / configureSolver /
// configureOcp: the parameter keys, defaults and fix-ups), Controller::step (state selection, re-initialisation decision, generateInitialStateTrajectory),
// stateFeedbackCallback, publishOptimalControlResult, reset, isPoseTrajectoryFeasible -- compiled from where it lies under /root/reference and EXECUTED, together with the
// grid / cost / constraint sources of the other translation units of oracle/_ref.  Stand-ins (oracle/ref_stubs/): roscpp's parameter lookup, time, publisher and console
// macros (ros/ros.h), the message structs, base_local_planner's CostmapModel (answers come from the test), teb's PoseSE2 / point obstacle / point footprint, corbo's
// PredictiveController, StructuredOptimalControlProblem (a record + grid->update() + a solver the test plugs in), solver option records, reference-trajectory kinds.
// NOT executed: any NLP solver.  The "solver" here is a callback that sees the grid after the reference's update() -- i.e. the initial guess / warm start the reference
// would hand to Ipopt -- and writes a result into the grid.
#include <csignal>
#include <cstring>
#include <sstream>
#include <sys/wait.h>
#include <unistd.h>

#include <mpc_local_planner/controller.h>
#include <mpc_local_planner/optimal_control/fd_collocation_se2.h>
#include <mpc_local_planner/optimal_control/final_state_conditions_se2.h>
#include <mpc_local_planner/optimal_control/finite_differences_variable_grid_se2.h>
#include <mpc_local_planner/optimal_control/min_time_via_points_cost.h>
#include <mpc_local_planner/optimal_control/quadratic_cost_se2.h>
#include <mpc_local_planner/systems/kinematic_bicycle_model.h>
#include <mpc_local_planner/systems/simple_car.h>
#include <mpc_local_planner/systems/unicycle_robot.h>
#include <corbo-optimal-control/functions/hybrid_cost.h>
#include <corbo-optimal-control/functions/minimum_time.h>
#include <corbo-optimal-control/functions/quadratic_control_cost.h>
#include <corbo-optimization/solver/levenberg_marquardt_sparse.h>
#include "ref_wrap_controller_access.hpp"

namespace {
using namespace mpc_local_planner;
using namespace ref_access;

typedef double (*cost_cb)(double x, double y, double theta);

struct Handle {
    ros::ParamStore store;
    ros::NodeHandle nh;
    teb_local_planner::ObstContainer obstacles;
    teb_local_planner::RobotFootprintModelPtr robot_model;
    std::vector<teb_local_planner::PoseSE2> via_points;
    Controller ctl;
    bool configured = false;
    solve_cb solver = nullptr;
    GuessRecord guess;            // the grid as the last compute() handed it to the "solver"
    mpc_local_planner_msgs::OptimalControlResult last_msg; int n_published = 0;
    CtlAccess& acc() { return static_cast<CtlAccess&>(ctl); }
};

std::vector<std::string> split(const std::string& s, char c) { std::vector<std::string> out; std::stringstream ss(s); std::string item; while (std::getline(ss, item, c)) out.push_back(item); return out; }

// one parameter per line: key TAB type TAB value; types i d b s, nl (comma separated, each i:<int> or d:<double>), bl, nm (key:i|d:value, ...), sm (key:value, ...)
void parse_params(const char* text, ros::ParamStore& store) {
    for (const std::string& line : split(text, '\n')) {
        const auto f = split(line, '\t');
        if (f.size() < 2) continue;
        const std::string val = f.size() > 2 ? f[2] : "";
        ros::ParamValue p;
        if (f[1] == "i") { p.kind = ros::ParamValue::Int; p.i = std::stol(val); }
        else if (f[1] == "d") { p.kind = ros::ParamValue::Double; p.d = std::stod(val); }
        else if (f[1] == "b") { p.kind = ros::ParamValue::Bool; p.b = val == "1"; }
        else if (f[1] == "s") { p.kind = ros::ParamValue::String; p.s = val; }
        else if (f[1] == "nl") { p.kind = ros::ParamValue::NumList; for (const auto& e : split(val, ',')) { p.num_is_int.push_back(e[0] == 'i'); p.nums.push_back(std::stod(e.substr(2))); } }
        else if (f[1] == "bl") { p.kind = ros::ParamValue::BoolList; for (const auto& e : split(val, ',')) p.bools.push_back(e == "1"); }
        else if (f[1] == "nm") { p.kind = ros::ParamValue::NumMap; for (const auto& e : split(val, ',')) { const auto kv = split(e, ':'); p.num_map[kv[0]] = std::stod(kv[2]); p.num_map_is_int[kv[0]] = kv[1] == "i"; } }
        else if (f[1] == "sm") { p.kind = ros::ParamValue::StrMap; for (const auto& e : split(val, ',')) { const auto kv = split(e, ':'); p.str_map[kv[0]] = kv.size() > 1 ? kv[1] : ""; } }
        else continue;
        store[f[0]] = p;
    }
}

const char* model_name(const RobotDynamicsInterface* d) {
    if (dynamic_cast<const UnicycleModel*>(d)) return "unicycle";
    if (dynamic_cast<const SimpleCarFrontWheelDrivingModel*>(d)) return "simple_car_front_wheel_driving";
    if (dynamic_cast<const SimpleCarModel*>(d)) return "simple_car";
    if (dynamic_cast<const KinematicBicycleModelVelocityInput*>(d)) return "kinematic_bicycle_vel_input";
    return "?";
}
void put(std::ostringstream& o, const char* k, const Eigen::MatrixXd& m) { o << k << "="; for (int i = 0; i < m.rows(); ++i) for (int j = 0; j < m.cols(); ++j) o << (i + j ? "," : "") << m(i, j); o << "\n"; }
void put(std::ostringstream& o, const char* k, const Eigen::VectorXd& v) { o << k << "="; for (int i = 0; i < v.size(); ++i) o << (i ? "," : "") << v[i]; o << "\n"; }
}  // namespace

extern "C" {
void* ref_ctl_create(const char*, int, const double*, int, const double*, int);
void* ref_ctl_create(const char* params_text, int n_obst, const double* obst_xy, int n_via, const double* via, int estimate_orientation) {
    ros::stub_log().lines.clear();
    Handle* h = new Handle;
    parse_params(params_text, h->store);
    h->nh.store = &h->store;
    *h->nh.publish_sink = [h](const void* m) { h->last_msg = *static_cast<const mpc_local_planner_msgs::OptimalControlResult*>(m); ++h->n_published; };
    for (int i = 0; i < n_obst; ++i) h->obstacles.push_back(std::make_shared<teb_local_planner::PointObstacle>(obst_xy[2 * i], obst_xy[2 * i + 1]));
    h->robot_model = std::make_shared<teb_local_planner::PointRobotFootprint>();
    for (int i = 0; i < n_via; ++i) h->via_points.emplace_back(via[3 * i], via[3 * i + 1], via[3 * i + 2]);
    h->ctl.setInitialPlanEstimateOrientation(estimate_orientation != 0);
    h->configured = h->ctl.configure(h->nh, h->obstacles, h->robot_model, h->via_points);
    if (h->configured) install_solver(h->ctl, &h->solver, &h->guess);
    return h;
}
// configure() in a forked child, because several error paths of the reference do not return false but dereference an empty pointer (an unknown solver type:
// src/controller.cpp:552; every `return {}` of configureOcp: :97 `_ocp->initialize()`).  Returns 1 / 0 = configure()'s result, 2 = the child died of a signal;
// log: the console lines up to that point ("<level>|text").
static int g_probe_fd = -1;
static void probe_dump_log(int status) {
    std::ostringstream o;
    o << status << "\n";
    for (const auto& l : ros::stub_log().lines) o << l.first << "|" << l.second << "\n";
    const std::string s = o.str();
    ssize_t w = write(g_probe_fd, s.c_str(), s.size()); (void)w;
}
static void probe_on_signal(int) { probe_dump_log(2); _exit(2); }
int ref_ctl_probe_configure(const char* params_text, char* log, int cap) {
    int fd[2];
    if (pipe(fd) != 0) return -1;
    const pid_t pid = fork();
    if (pid == 0) {
        close(fd[0]); g_probe_fd = fd[1];
        std::signal(SIGSEGV, probe_on_signal); std::signal(SIGBUS, probe_on_signal); std::signal(SIGABRT, probe_on_signal);
        Handle* h = static_cast<Handle*>(ref_ctl_create(params_text, 0, nullptr, 0, nullptr, 1));
        probe_dump_log(h->configured ? 1 : 0);
        _exit(0);
    }
    close(fd[1]);
    std::string all; char buf[4096]; ssize_t r;
    while ((r = read(fd[0], buf, sizeof buf)) > 0) all.append(buf, (size_t)r);
    close(fd[0]);
    int st = 0; waitpid(pid, &st, 0);
    int result = all.empty() ? 2 : all[0] - '0';
    const size_t nl = all.find('\n');
    const std::string rest = nl == std::string::npos ? "" : all.substr(nl + 1);
    std::strncpy(log, rest.c_str(), (size_t)cap - 1); log[cap - 1] = 0;
    return result;
}
void ref_ctl_destroy(void* p) { delete static_cast<Handle*>(p); }
int ref_ctl_configured(void* p) { return static_cast<Handle*>(p)->configured ? 1 : 0; }
void ref_ctl_set_solver(void* p, solve_cb cb) { static_cast<Handle*>(p)->solver = cb; }
// the ROS console output since the last ref_ctl_create: "<level>|text" per line (1 info, 2 warn, 3 error)
int ref_ctl_log(char* out, int cap) {
    std::ostringstream o;
    for (const auto& l : ros::stub_log().lines) o << l.first << "|" << l.second << "\n";
    const std::string s = o.str();
    std::strncpy(out, s.c_str(), (size_t)cap - 1); out[cap - 1] = 0;
    return (int)s.size();
}
// what configure() built, "key=value" per line
int ref_ctl_dump(void* p, char* out, int cap) {
    Handle* h = static_cast<Handle*>(p);
    CtlAccess& c = h->acc();
    std::ostringstream o;
    o.precision(17);
    o << "configured=" << h->configured << "\n";
    if (c._dynamics) {
        o << "model=" << model_name(c._dynamics.get()) << "\n";
        if (auto* car = dynamic_cast<const SimpleCarModel*>(c._dynamics.get())) o << "wheelbase=" << car->getWheelbase() << "\n";
        if (auto* b = dynamic_cast<const KinematicBicycleModelVelocityInput*>(c._dynamics.get())) o << "length_rear=" << b->getLengthRear() << "\nlength_front=" << b->getLengthFront() << "\n";
    }
    if (c._grid) {
        auto* vg = dynamic_cast<FiniteDifferencesVariableGridSE2*>(c._grid.get());
        auto dump_common = [&](auto& g) {
            o << "n_ref=" << g._n_ref << "\ndt_ref=" << g._dt_ref << "\nwarm_start=" << g._warm_start << "\ndt_lb=" << g._dt_lb << "\ndt_ub=" << g._dt_ub << "\n";
            o << "xf_fixed="; for (int i = 0; i < g._xf_fixed.size(); ++i) o << (i ? "," : "") << (g._xf_fixed[i] ? 1 : 0); o << "\n";
            o << "cost_integration=" << (g._cost_integration == FullDiscretizationGridBaseSE2::CostIntegrationRule::LeftSum ? "left_sum" : "trapezoidal_rule") << "\n";
            const auto* e = g._fd_eval.get();
            o << "collocation=" << (dynamic_cast<const ForwardDiffCollocationSE2*>(e) ? "forward_differences" : dynamic_cast<const MidpointDiffCollocationSE2*>(e) ? "midpoint_differences"
                                   : dynamic_cast<const CrankNicolsonDiffCollocationSE2*>(e) ? "crank_nicolson_differences" : "corbo_crank_nicolson_not_se2") << "\n";
        };
        o << "variable_grid=" << (vg ? 1 : 0) << "\n";
        if (vg) {
            GridAccess& g = static_cast<GridAccess&>(*vg);
            dump_common(g);
            o << "grid_adaptation=" << (g._grid_adapt == FiniteDifferencesVariableGridSE2::GridAdaptStrategy::TimeBasedSingleStep ? 1 : 0) << "\nn_max=" << g._n_max << "\nn_min=" << g._n_min
              << "\ndt_hyst_ratio=" << g._dt_hyst_ratio << "\n";
        } else {
            dump_common(static_cast<BaseGridAccess&>(*dynamic_cast<FiniteDifferencesGridSE2*>(c._grid.get())));
        }
    }
    if (auto* ip = dynamic_cast<corbo::SolverIpopt*>(c._solver.get())) {
        o << "solver=ipopt\niterations=" << ip->iterations << "\nmax_cpu_time=" << ip->max_cpu_time << "\n";
        for (const auto& e : ip->numeric) o << "ipopt_numeric." << e.first << "=" << e.second << "\n";
        for (const auto& e : ip->strings) o << "ipopt_string." << e.first << "=" << e.second << "\n";
        for (const auto& e : ip->integers) o << "ipopt_integer." << e.first << "=" << e.second << "\n";
    } else if (auto* lm = dynamic_cast<corbo::LevenbergMarquardtSparse*>(c._solver.get())) {
        o << "solver=lsq_lm\niterations=" << lm->iterations << "\npenalty_weights=" << lm->w[0] << "," << lm->w[1] << "," << lm->w[2] << "\nweight_adaptation=";
        for (int i = 0; i < 6; ++i) o << (i ? "," : "") << lm->a[i];
        o << "\n";
    }
    if (c._structured_ocp) {
        auto& ocp = *c._structured_ocp;
        put(o, "u_lb", ocp.functions.u_lb); put(o, "u_ub", ocp.functions.u_ub);
        const corbo::StageCost* sc = ocp.stage_cost.get();
        if (auto* q = dynamic_cast<const QuadraticFormCostSE2*>(sc)) { o << "stage_cost=QuadraticFormCostSE2\nintegral_form=" << q->_integral_form << "\nlsq_form=" << q->_lsq_form << "\n"; put(o, "Q", q->_Q); put(o, "R", q->_R); }
        else if (auto* q = dynamic_cast<const QuadraticStateCostSE2*>(sc)) { o << "stage_cost=QuadraticStateCostSE2\nintegral_form=" << q->_integral_form << "\nlsq_form=" << q->_lsq_form << "\n"; put(o, "Q", q->_Q); }
        else if (auto* q = dynamic_cast<const corbo::MinTimeQuadraticControls*>(sc)) { o << "stage_cost=MinTimeQuadraticControls\nintegral_form=" << q->_integral_form << "\nlsq_form=" << q->_lsq_form << "\n"; put(o, "R", q->_R); }
        else if (auto* q = dynamic_cast<const corbo::QuadraticControlCost*>(sc)) { o << "stage_cost=QuadraticControlCost\nintegral_form=" << q->_integral_form << "\nlsq_form=" << q->_lsq_form << "\n"; put(o, "R", q->_R); }
        else if (auto* q = dynamic_cast<const corbo::MinimumTime*>(sc)) { o << "stage_cost=MinimumTime\nlsq_form=" << q->_lsq_form << "\n"; }
        else if (auto* q = dynamic_cast<const MinTimeViaPointsCost*>(sc)) {
            const ViaAccess& v = static_cast<const ViaAccess&>(*q);
            o << "stage_cost=MinTimeViaPointsCost\nvia_points_ordered=" << v._via_points_ordered << "\nvp_position_weight=" << v._vp_position_weight << "\nvp_orientation_weight=" << v._vp_orientation_weight << "\n";
        } else o << "stage_cost=none\n";
        if (auto* f = dynamic_cast<const QuadraticFinalStateCostSE2*>(ocp.final_stage_cost.get())) { o << "final_stage_cost=QuadraticFinalStateCostSE2\nfinal_lsq_form=" << f->_lsq_form << "\n"; put(o, "Qf", f->_Qf); }
        else o << "final_stage_cost=none\n";
        if (auto* b = dynamic_cast<const TerminalBallSE2*>(ocp.final_stage_constraint.get())) { o << "final_stage_constraint=TerminalBallSE2\ngamma=" << b->_gamma << "\n"; put(o, "S", b->_S); }
        else o << "final_stage_constraint=none\n";
    }
    if (c._inequality_constraint) {
        const IneqAccess& q = static_cast<const IneqAccess&>(*c._inequality_constraint);
        o << "min_obstacle_dist=" << q._min_obstacle_dist << "\nenable_dynamic_obstacles=" << q._enable_dynamic_obstacles << "\nforce_inclusion_dist=" << q._obstacle_filter_force_inclusion_dist
          << "\ncutoff_dist=" << q._obstacle_filter_cutoff_dist << "\n";
        put(o, "du_lb", q._du_lb); put(o, "du_ub", q._du_ub);
    }
    o << "outer_ocp_iterations=" << c._num_ocp_iterations << "\nauto_update_previous_control=" << c._auto_update_prev_control << "\nforce_reinit_new_goal_dist=" << c._force_reinit_new_goal_dist
      << "\nforce_reinit_new_goal_angular=" << c._force_reinit_new_goal_angular << "\nallow_init_with_backward_motion=" << c._guess_backwards_motion << "\nforce_reinit_num_steps="
      << c._force_reinit_num_steps << "\nprefer_x_feedback=" << c._prefer_x_feedback << "\npublish_ocp_results=" << c._publish_ocp_results << "\nprint_cpu_time=" << c._print_cpu_time << "\n";
    const std::string s = o.str();
    std::strncpy(out, s.c_str(), (size_t)cap - 1); out[cap - 1] = 0;
    return (int)s.size();
}
void ref_ctl_set_previous_control(void* p, const double* u, double dt) {
    Handle* h = static_cast<Handle*>(p);
    Eigen::VectorXd v(2); v[0] = u[0]; v[1] = u[1];
    h->ctl.getOptimalControlProblem()->setPreviousControlInput(v, dt);                 // src/mpc_local_planner_ros.cpp:384
}
void ref_ctl_state_feedback(void* p, const double* state, int dim, double stamp) {
    auto msg = std::make_shared<mpc_local_planner_msgs::StateFeedback>();
    msg->header.stamp = ros::Time(stamp);
    msg->state.assign(state, state + dim);
    static_cast<Handle*>(p)->ctl.stateFeedbackCallback(msg);
}
void ref_ctl_reset(void* p) { static_cast<Handle*>(p)->ctl.reset(); }
// Controller::step(initial_plan, vel, dt, t, u_seq, x_seq): plan [n_plan][3] (x, y, yaw); returns 1 / 0; the time series: t_out [cap], x_out [cap][3], u_out [cap][2], *n_out samples
int ref_ctl_step(void* p, int n_plan, const double* plan, const double* vel, double dt, double t, int cap, double* t_out, double* x_out, double* u_out, int* n_out) {
    Handle* h = static_cast<Handle*>(p);
    std::vector<geometry_msgs::PoseStamped> poses((size_t)n_plan);
    for (int i = 0; i < n_plan; ++i) teb_local_planner::PoseSE2(plan[3 * i], plan[3 * i + 1], plan[3 * i + 2]).toPoseMsg(poses[(size_t)i].pose);
    geometry_msgs::Twist tw; tw.linear.x = vel[0]; tw.linear.y = vel[1]; tw.angular.z = vel[2];
    auto xs = std::make_shared<corbo::TimeSeries>(), us = std::make_shared<corbo::TimeSeries>();
    const bool ok = h->ctl.step(poses, tw, dt, ros::Time(t), us, xs);
    const int m = xs->getTimeDimension() < cap ? xs->getTimeDimension() : cap;
    for (int k = 0; k < m; ++k) {
        t_out[k] = xs->getTime()[(size_t)k];
        for (int i = 0; i < 3; ++i) x_out[3 * k + i] = xs->getValuesMap(k)[i];
        if (k < us->getTimeDimension()) for (int j = 0; j < 2; ++j) u_out[2 * k + j] = us->getValuesMap(k)[j];
    }
    *n_out = xs->getTimeDimension();
    return ok ? 1 : 0;
}
// the start / goal overload (src/controller.cpp:102-109)
int ref_ctl_step_two_poses(void* p, const double* start, const double* goal, const double* vel, double dt, double t, int cap, double* t_out, double* x_out, double* u_out, int* n_out) {
    Handle* h = static_cast<Handle*>(p);
    geometry_msgs::Twist tw; tw.linear.x = vel[0]; tw.linear.y = vel[1]; tw.angular.z = vel[2];
    auto xs = std::make_shared<corbo::TimeSeries>(), us = std::make_shared<corbo::TimeSeries>();
    const bool ok = h->ctl.step(teb_local_planner::PoseSE2(start[0], start[1], start[2]), teb_local_planner::PoseSE2(goal[0], goal[1], goal[2]), tw, dt, ros::Time(t), us, xs);
    const int m = xs->getTimeDimension() < cap ? xs->getTimeDimension() : cap;
    for (int k = 0; k < m; ++k) {
        t_out[k] = xs->getTime()[(size_t)k];
        for (int i = 0; i < 3; ++i) x_out[3 * k + i] = xs->getValuesMap(k)[i];
        if (k < us->getTimeDimension()) for (int j = 0; j < 2; ++j) u_out[2 * k + j] = us->getValuesMap(k)[j];
    }
    *n_out = xs->getTimeDimension();
    return ok ? 1 : 0;
}
// the grid as the last compute() handed it to the solver (= the reference's initial guess / warm start): returns n; x [n][3], u [n-1][2]
int ref_ctl_last_guess(void* p, int cap, double* x, double* u, double* dt) {
    Handle* h = static_cast<Handle*>(p);
    const int n = h->guess.n < cap ? h->guess.n : cap;
    for (int i = 0; i < 3 * n; ++i) x[i] = h->guess.x[(size_t)i];
    for (int i = 0; i < 2 * (n - 1); ++i) u[i] = h->guess.u[(size_t)i];
    *dt = h->guess.dt;
    return h->guess.n;
}
// counters: [0] _ocp_seq, [1] grid->clear() calls through reset(), [2] compute() calls, [3] messages published, [4] grid empty now, [5] number of precompute() calls of the
// initial state trajectory; *last_sample_dt = the dt the last of them was asked for
void ref_ctl_counters(void* p, long* out, double* last_sample_dt) {
    Handle* h = static_cast<Handle*>(p);
    CtlAccess& c = h->acc();
    out[0] = (long)c._ocp_seq; out[1] = c._structured_ocp ? c._structured_ocp->n_resets : 0; out[2] = c._structured_ocp ? c._structured_ocp->n_computes : 0;
    out[3] = h->n_published; out[4] = c._grid && c._grid->isEmpty() ? 1 : 0; out[5] = (long)c._x_seq_init.sample_dts.size();
    *last_sample_dt = c._x_seq_init.sample_dts.empty() ? 0.0 : c._x_seq_init.sample_dts.back();
}
// the last published OptimalControlResult: head = [seq, dim_states, dim_controls, found, cpu_time, n_time_states, n_states, n_time_controls, n_controls]; arrays up to cap values
void ref_ctl_result_msg(void* p, double* head, int cap, double* time_states, double* states, double* time_controls, double* controls) {
    const auto& m = static_cast<Handle*>(p)->last_msg;
    head[0] = m.header.seq; head[1] = (double)m.dim_states; head[2] = (double)m.dim_controls; head[3] = m.optimal_solution_found; head[4] = m.cpu_time;
    head[5] = (double)m.time_states.size(); head[6] = (double)m.states.size(); head[7] = (double)m.time_controls.size(); head[8] = (double)m.controls.size();
    auto copy = [cap](const std::vector<double>& v, double* o) { for (size_t i = 0; i < v.size() && (int)i < cap; ++i) o[i] = v[i]; };
    copy(m.time_states, time_states); copy(m.states, states); copy(m.time_controls, time_controls); copy(m.controls, controls);
}
// Controller::isPoseTrajectoryFeasible on the grid's current trajectory; `cost` answers footprintCost (-1 = collision); calls [cap][3] = every pose asked, *n_calls
int ref_ctl_feasible(void* p, cost_cb cost, double inscribed_radius, double circumscribed_radius, double min_resolution_collision_check_angular, int look_ahead_idx, int cap,
                     double* calls, int* n_calls) {
    Handle* h = static_cast<Handle*>(p);
    base_local_planner::CostmapModel model;
    model.answer = [cost](double x, double y, double th) { return cost(x, y, th); };
    std::vector<geometry_msgs::Point> spec;
    const bool ok = h->ctl.isPoseTrajectoryFeasible(&model, spec, inscribed_radius, circumscribed_radius, min_resolution_collision_check_angular, look_ahead_idx);
    *n_calls = (int)model.calls.size();
    for (int i = 0; i < *n_calls && i < cap; ++i) { calls[3 * i] = model.calls[(size_t)i].x; calls[3 * i + 1] = model.calls[(size_t)i].y; calls[3 * i + 2] = model.calls[(size_t)i].theta; }
    return ok ? 1 : 0;
}
}  // extern "C"

// GPU test program (built and run by tests/test_gpu_parity.py): the scenario of the reference's stand-alone node
// (mpc_local_planner/src/test_mpc_optim_node.cpp:59-131 + cfg/test_mpc_optim_node.yaml): unicycle, n = 20, dt_ref = .3,
// variable grid, minimum time, xf fixed, point footprint, d_min = .5, point obstacles (-3,1), (6,2), (4,.1),
// start (0,0,0) -> goal (5,2,0), 20 Hz.  Drives the C++ Controller facade in closed loop: step -> apply u_0 for
// one period -> step (warm start).  Prints one line per cycle; exit code 0 iff every check holds.
#include <cmath>
#include <cstdio>

#include "../include/mpc_controller.hpp"
#include "../include/mpc_params.hpp"

using namespace mpc_local_planner_amd;

// second scenario: the shipped configuration (grid adaptation on): drive to the goal; n follows dt (n+-1 per cycle)
static int run_adaptive() {
    // configured the way the plugin does it: from the parameter set (here the keys of the diff-drive minimum-time example, held in a map
    // instead of the ROS parameter server) through include/mpc_params.hpp
    MapParamSource prm;
    prm.set("robot/type", "unicycle");
    prm.set("robot/unicycle/max_vel_x", 0.4); prm.set("robot/unicycle/max_vel_x_backwards", 0.2); prm.set("robot/unicycle/max_vel_theta", 0.3);
    prm.set("grid/type", "fd_grid"); prm.set("grid/grid_size_ref", 20); prm.set("grid/dt_ref", 0.3);
    prm.set("grid/xf_fixed", std::vector<bool>{true, true, true});
    prm.set("grid/variable_grid/enable", true); prm.set("grid/variable_grid/min_dt", 0.0); prm.set("grid/variable_grid/max_dt", 10.0);
    prm.set("grid/variable_grid/grid_adaptation/enable", true); prm.set("grid/variable_grid/grid_adaptation/max_grid_size", 50);
    prm.set("grid/variable_grid/grid_adaptation/dt_hyst_ratio", 0.1); prm.set("grid/variable_grid/grid_adaptation/min_grid_size", 2);   // the facade clamps 2 to 3
    prm.set("planning/objective/type", "minimum_time");
    prm.set("solver/type", "ipopt"); prm.set("solver/ipopt/iterations", 100); prm.set("solver/ipopt/ipopt_numeric_options/tol", 1e-6);
    prm.set("solver/ipopt/ipopt_string_options/linear_solver", "mumps");
    Controller ctl;
    ParamReport rep;
    if (configure_from_params(ctl, prm, rep) != PARAMS_OK) { std::printf("configure failed: %s\n", rep.error.c_str()); return 2; }
    for (const std::string& note : rep.notes) std::printf("parameter note: %s\n", note.c_str());
    PoseSE2 pose{0, 0, 0}, goal{2.0, 1.0, 0.5};
    Twist vel;
    const double period = 0.1;
    TimeSeries x_seq, u_seq;
    double u_prev[2] = {0, 0};
    int failures = 0, n_first = 0, n_min_seen = 1000, n_max_seen = 0, cycles = 0;
    bool reached = false;
    for (int cyc = 0; cyc < 400; ++cyc) {
        const double dx = goal.x - pose.x, dy = goal.y - pose.y;
        if (std::hypot(dx, dy) < 0.1 && std::fabs(normalize_theta(goal.theta - pose.theta)) < 0.1) { reached = true; break; }   // xy / yaw goal tolerance
        ctl.setPreviousControlInput(u_prev, cyc == 0 ? 0.0 : period);
        const bool ok = ctl.step(pose, goal, vel, period, cyc * period, u_seq, x_seq);
        ++cycles;
        if (!ok) { ++failures; ctl.reset(); u_prev[0] = u_prev[1] = 0; continue; }
        const int n = ctl.gridSize();
        if (cyc == 0) n_first = n;
        n_min_seen = n < n_min_seen ? n : n_min_seen; n_max_seen = n > n_max_seen ? n : n_max_seen;
        if (x_seq.size() != n || u_seq.size() != n) { std::printf("time series size %d != grid size %d\n", x_seq.size(), n); return 1; }
        const double* u0 = u_seq.at(0);
        if (cyc % 10 == 0) std::printf("adaptive cycle %3d pose (%.3f %.3f %.3f) n %2d dt %.3f iters %d\n", cyc, pose.x, pose.y, pose.theta, n, ctl.lastDt(), ctl.lastIterations());
        pose.x += period * u0[0] * std::cos(pose.theta);
        pose.y += period * u0[0] * std::sin(pose.theta);
        pose.theta = normalize_theta(pose.theta + period * u0[1]);
        u_prev[0] = u0[0]; u_prev[1] = u0[1];
    }
    std::printf("adaptive: reached %d after %d cycles, failures %d, n first %d min %d max %d\n", (int)reached, cycles, failures, n_first, n_min_seen, n_max_seen);
    return (reached && failures <= 5 && n_first == 20 && n_min_seen <= 6) ? 0 : 1;
}

// third scenario: Controller::step with outer_ocp_iterations = 3 on the variable grid with adaptation, once with every outer iteration in ONE
// mpc_step_batch call (grid update on the device) and once with the facade's host loop (resample_trajectory / adaptation on the host, one mpc_solve_batch per
// outer iteration): the claim "the device grid update is bit for bit what the host code does" means identical time series, dt and grid size in every cycle
static int run_single_launch_vs_host_loop() {
    Controller a, b;
    Controller* cs[2] = {&a, &b};
    for (int i = 0; i < 2; ++i) {
        mpc_config c;
        mpc_config_defaults(&c);
        c.model = MPC_MODEL_SIMPLE_CAR; c.model_params[0] = 0.4;
        c.n = 30; c.dt_ref = 0.3; c.dt_free = 1; c.dt_lb = 0.0; c.dt_ub = 10.0;
        c.u_lb[0] = -0.2; c.u_ub[0] = 0.4; c.u_lb[1] = -1.4; c.u_ub[1] = 1.4;
        c.du_lb[0] = c.du_lb[1] = -0.5; c.du_ub[0] = c.du_ub[1] = 0.5;
        if (!cs[i]->configure(c, 0)) { std::printf("configure failed: %s\n", cs[i]->lastError().c_str()); return 2; }
        cs[i]->setGridAdaptation(true, 30, 0.1, 3);
        cs[i]->setNumOcpIterations(3);
        cs[i]->setSingleLaunchStep(i == 0);
    }
    PoseSE2 pose{0, 0, 0.3}, goal{2.5, 1.2, 0.8};
    Twist vel;
    const double period = 0.1;
    TimeSeries xa, ua, xb, ub;
    double u_prev[2] = {0, 0};
    int compared = 0, n_changes = 0, n_last = 0;
    for (int cyc = 0; cyc < 25; ++cyc) {
        a.setPreviousControlInput(u_prev, cyc == 0 ? 0.0 : period);
        b.setPreviousControlInput(u_prev, cyc == 0 ? 0.0 : period);
        const bool oka = a.step(pose, goal, vel, period, cyc * period, ua, xa);
        const bool okb = b.step(pose, goal, vel, period, cyc * period, ub, xb);
        if (oka != okb || a.gridSize() != b.gridSize() || a.lastDt() != b.lastDt() || a.lastIterations() != b.lastIterations()) {
            std::printf("single launch vs host loop: cycle %d differs: ok %d/%d n %d/%d dt %.17g/%.17g iters %d/%d\n", cyc, (int)oka, (int)okb, a.gridSize(), b.gridSize(), a.lastDt(), b.lastDt(),
                        a.lastIterations(), b.lastIterations());
            return 1;
        }
        if (!oka) { a.reset(); b.reset(); u_prev[0] = u_prev[1] = 0; continue; }
        const int n = a.gridSize();
        if (cyc > 0 && n != n_last) ++n_changes;
        n_last = n;
        for (int k = 0; k < n; ++k) {
            for (int i = 0; i < 3; ++i) if (xa.at(k)[i] != xb.at(k)[i]) { std::printf("single launch vs host loop: cycle %d x[%d][%d] %.17g != %.17g\n", cyc, k, i, xa.at(k)[i], xb.at(k)[i]); return 1; }
            for (int j = 0; j < 2; ++j) if (ua.at(k)[j] != ub.at(k)[j]) { std::printf("single launch vs host loop: cycle %d u[%d][%d] %.17g != %.17g\n", cyc, k, j, ua.at(k)[j], ub.at(k)[j]); return 1; }
        }
        ++compared;
        const double* u0 = ua.at(0);
        pose.x += period * u0[0] * std::cos(pose.theta);
        pose.y += period * u0[0] * std::sin(pose.theta);
        pose.theta = normalize_theta(pose.theta + period * u0[0] * std::tan(u0[1]) / 0.4);
        u_prev[0] = u0[0]; u_prev[1] = u0[1];
    }
    std::printf("single launch vs host loop: %d cycles identical (x, u, dt, grid size, iterations), grid size changed %d times, last n %d\n", compared, n_changes, n_last);
    return (compared >= 20 && n_changes >= 3) ? 0 : 1;
}

int main() {
    mpc_config c;
    mpc_config_defaults(&c);                 // unicycle, n=20, dt_ref=.3, variable grid, min-time, xf fixed
    c.u_lb[0] = -0.2; c.u_ub[0] = 0.4;       // max_vel_x_backwards / max_vel_x
    c.u_lb[1] = -0.3; c.u_ub[1] = 0.3;       // max_vel_theta
    c.min_obstacle_dist = 0.5; c.force_inclusion_dist = 0.5; c.cutoff_dist = 2.5;
    c.footprint_kind = MPC_FOOTPRINT_POINT;
    c.max_obstacles = 3; c.max_vertices = 1; c.max_obstacle_rows = 4;
    c.tol = 1e-6;
    Controller ctl;
    if (!ctl.configure(c, 0)) { std::printf("configure failed: %s\n", ctl.lastError().c_str()); return 2; }
    const int32_t n_obst[1] = {3};
    const int32_t n_vert[3] = {1, 1, 1};
    const double verts[6] = {-3.0, 1.0, 6.0, 2.0, 4.0, 0.1};
    mpc_obstacles ob = {n_obst, n_vert, verts, nullptr, nullptr};
    ctl.setObstacles(&ob);
    PoseSE2 pose{0, 0, 0}, goal{5, 2, 0};
    Twist vel;
    const double period = 0.05;
    TimeSeries x_seq, u_seq;
    double u_prev[2] = {0, 0};
    int failures = 0, cold_iters = 0, warm_iters = 0;
    double t_first = 0, t_last = 0, min_clear = 1e9;
    for (int cyc = 0; cyc < 40; ++cyc) {
        ctl.setPreviousControlInput(u_prev, cyc == 0 ? 0.0 : period);
        const bool ok = ctl.step(pose, goal, vel, period, cyc * period, u_seq, x_seq);
        if (!ok) { ++failures; ctl.reset(); std::printf("cycle %d: solve failed (%d iterations)\n", cyc, ctl.lastIterations()); continue; }
        const double T = ctl.lastDt() * (c.n - 1);
        if (cyc == 0) { t_first = T; cold_iters = ctl.lastIterations(); } else warm_iters += ctl.lastIterations();
        t_last = T;
        for (int k = 1; k < c.n - 1; ++k) {
            const double* x = x_seq.at(k);
            for (int o = 0; o < 3; ++o) min_clear = std::fmin(min_clear, std::hypot(x[0] - verts[2 * o], x[1] - verts[2 * o + 1]));
        }
        const double* u0 = u_seq.at(0);
        if (cyc % 8 == 0) std::printf("cycle %2d pose (%.3f %.3f %.3f) u0 (%.3f %.3f) T %.3f iters %d\n", cyc, pose.x, pose.y, pose.theta, u0[0], u0[1], T, ctl.lastIterations());
        // unicycle plant, explicit Euler over one control period
        pose.x += period * u0[0] * std::cos(pose.theta);
        pose.y += period * u0[0] * std::sin(pose.theta);
        pose.theta = normalize_theta(pose.theta + period * u0[1]);
        u_prev[0] = u0[0]; u_prev[1] = u0[1];
    }
    std::printf("failures %d  cold-start iterations %d  mean warm iterations %.1f  T first %.3f  T last %.3f  min clearance (associated or not) %.3f\n",
                failures, cold_iters, warm_iters / 39.0, t_first, t_last, min_clear);
    bool good = failures <= 2 && t_last < t_first && t_first > 5.385 / 0.4 - 1e-6 && pose.x > 0.5;
    const int ra = run_adaptive();
    good = good && ra == 0;
    const int rs = run_single_launch_vs_host_loop();
    good = good && rs == 0;
    std::printf(good ? "DEMO_OK\n" : "DEMO_FAILED\n");
    return good ? 0 : 1;
}

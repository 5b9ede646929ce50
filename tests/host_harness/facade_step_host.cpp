// tests only: drives the shipped Controller facade (include/mpc_controller.hpp) WITHOUT the GPU library.  The C-ABI entry points the facade calls are defined here
// as a recorder: mpc_solve_batch builds the vertex values the kernel would start from (what the ABI documents: x_init/u_init/dt_init with x_0 := x0 and the fixed
// goal components := xf, or the 2-pose cold start when they are NULL), shows them to a callback (the test's stand-in "solver", the same one that is plugged into the
// reference's Controller in oracle/ref_wrap_controller.cpp) and returns its answer.  tests/test_reference_pinned.py compares, step by step, what the reference's
// Controller::step and this facade hand to the solver and give back.
#include "../../include/mpc_controller.hpp"
#include <cstring>

typedef int (*solve_cb)(int n, double* x, double* u, double* dt, const double* u_prev, double u_prev_dt);

struct mpc_solver {
    mpc_config cfg;
    int n_grid;
    int resets = 0, solves = 0;
};
namespace {
solve_cb g_solver = nullptr;
std::vector<double> g_guess_x, g_guess_u; double g_guess_dt = 0; int g_guess_n = 0, g_guess_cold = 0;
std::string g_err;
std::vector<double> g_obst;          // per obstacle: n_vertices, radius, vx, vy, then 2 * V vertex coordinates
int g_obst_n = -1, g_obst_stride = 0;
}

extern "C" {
static mpc_config g_last_created;
int mpc_create(const mpc_config* cfg, int32_t, int32_t, mpc_solver** out) { g_last_created = *cfg; *out = new mpc_solver{*cfg, cfg->n}; return MPC_OK; }
// the configuration of the most recent mpc_create (what a binding built from its parameters)
void fs_last_created_config(mpc_config* out) { *out = g_last_created; }
void mpc_destroy(mpc_solver* s) { delete s; }
int mpc_reset(mpc_solver* s) { ++s->resets; return MPC_OK; }
const char* mpc_last_error(void) { return g_err.c_str(); }
int mpc_set_grid_sizes(mpc_solver* s, const int32_t* n_grid, int32_t) { s->n_grid = n_grid ? n_grid[0] : s->cfg.n; return MPC_OK; }
int mpc_set_via_points(mpc_solver*, int32_t, const int32_t*, const double*) { return MPC_OK; }
int mpc_last_rows_dropped(mpc_solver*, int32_t, int32_t* rows_dropped) { rows_dropped[0] = 0; return MPC_OK; }
int mpc_check_feasibility(mpc_solver*, int32_t, const double*, const uint8_t*, int32_t, int32_t, double, const double*, const double*, int32_t, double, double, int32_t, int32_t* ok) { *ok = 1; return MPC_OK; }
int mpc_solve_batch(mpc_solver* s, int32_t, const double* x0, const double* xf, const double* u_prev, const double* dt_prev, const double* x_init, const double* u_init,
                    const double* dt_init, const mpc_obstacles* ob, double* x_out, double* u_out, double* dt_out, int32_t* status, int32_t* iters) {
    using namespace mpc_local_planner_amd;
    g_obst_n = -1;
    if (ob) {
        const int V = s->cfg.max_vertices;
        g_obst_n = ob->n_obstacles[0]; g_obst_stride = 4 + 2 * V; g_obst.assign((size_t)g_obst_n * g_obst_stride, 0.0);
        for (int o = 0; o < g_obst_n; ++o) {
            double* r = &g_obst[(size_t)o * g_obst_stride];
            r[0] = ob->n_vertices[o]; r[1] = ob->radius ? ob->radius[o] : 0.0; r[2] = ob->velocity ? ob->velocity[2 * o] : 0.0; r[3] = ob->velocity ? ob->velocity[2 * o + 1] : 0.0;
            for (int i = 0; i < 2 * ob->n_vertices[o]; ++i) r[4 + i] = ob->vertices[(size_t)o * V * 2 + i];
        }
    }
    const int n = s->n_grid;
    std::vector<double> x((size_t)3 * n), u((size_t)2 * (n - 1), 0.0);
    double dt = s->cfg.dt_ref;
    g_guess_cold = x_init ? 0 : 1;
    if (x_init) {
        for (int i = 0; i < 3 * n; ++i) x[(size_t)i] = x_init[i];
        for (int i = 0; i < 2 * (n - 1); ++i) u[(size_t)i] = u_init[i];
        dt = *dt_init;
        for (int i = 0; i < 3; ++i) x[(size_t)i] = x0[i];                                             // x_0 := x0
        for (int i = 0; i < 3; ++i) if (s->cfg.xf_fixed[i]) x[(size_t)(3 * (n - 1) + i)] = xf[i];     // fixed goal components := xf
    } else {
        // the device-side cold start = the reference's 2-pose plan: linear x0 -> xf in time, shortest-arc heading, u = 0, dt = dt_ref (include/mpc_hip.h, mpc_solve_batch)
        std::vector<PoseSE2> plan(2);
        plan[0].x = x0[0]; plan[0].y = x0[1]; plan[0].theta = x0[2]; plan[1].x = xf[0]; plan[1].y = xf[1]; plan[1].theta = xf[2];
        initial_state_trajectory(plan, x0, xf, n, s->cfg.dt_ref, true, x.data());
    }
    g_guess_x = x; g_guess_u = u; g_guess_dt = dt; g_guess_n = n;
    int ok = 1;
    if (g_solver) ok = g_solver(n, x.data(), u.data(), &dt, u_prev, dt_prev ? *dt_prev : 0.0);
    for (int i = 0; i < 3 * n; ++i) x_out[i] = x[(size_t)i];
    for (int i = 0; i < 2 * (n - 1); ++i) u_out[i] = u[(size_t)i];
    u_out[2 * (n - 1)] = u[(size_t)(2 * (n - 2))]; u_out[2 * (n - 1) + 1] = u[(size_t)(2 * (n - 2) + 1)];      // the last control repeated
    *dt_out = dt;
    if (status) *status = ok ? MPC_CONVERGED : MPC_MAX_ITER;
    if (iters) *iters = 1;
    ++s->solves;
    return MPC_OK;
}

// ---- the facade under test
void* fs_create(const mpc_config* cfg, const double* opt /* the ControllerOptions of tests/host_harness/params_host.cpp, same order */, int estimate_orientation) {
    using namespace mpc_local_planner_amd;
    Controller* c = new Controller;
    c->setGridAdaptation(opt[0] != 0, (int)opt[1], opt[2], (int)opt[3]);
    c->setWarmStart(opt[5] != 0);
    c->setNumOcpIterations((int)opt[6]);
    c->setForceReinit((int)opt[10], opt[7], opt[8]);
    c->setPreferStateFeedback(opt[11] != 0);
    c->setInitialPlanEstimateOrientation(estimate_orientation != 0);
    if (!c->configure(*cfg)) { delete c; return nullptr; }
    return c;
}
void fs_destroy(void* p) { delete static_cast<mpc_local_planner_amd::Controller*>(p); }
void fs_set_solver(solve_cb cb) { g_solver = cb; }
void fs_set_previous_control(void* p, const double* u, double dt) { static_cast<mpc_local_planner_amd::Controller*>(p)->setPreviousControlInput(u, dt); }
void fs_state_feedback(void* p, const double* state, double stamp) { static_cast<mpc_local_planner_amd::Controller*>(p)->stateFeedbackCallback(state, stamp); }
void fs_reset(void* p) { static_cast<mpc_local_planner_amd::Controller*>(p)->reset(); }
int fs_step(void* p, int n_plan, const double* plan, const double* vel, double dt, double t, int cap, double* t_out, double* x_out, double* u_out, int* n_out) {
    using namespace mpc_local_planner_amd;
    std::vector<PoseSE2> poses((size_t)n_plan);
    for (int i = 0; i < n_plan; ++i) { poses[(size_t)i].x = plan[3 * i]; poses[(size_t)i].y = plan[3 * i + 1]; poses[(size_t)i].theta = plan[3 * i + 2]; }
    Twist tw; tw.linear_x = vel[0]; tw.linear_y = vel[1]; tw.angular_z = vel[2];
    TimeSeries xs, us;
    const bool ok = static_cast<Controller*>(p)->step(poses, tw, dt, t, us, xs);
    const int m = xs.size() < cap ? xs.size() : cap;
    for (int k = 0; k < m; ++k) { t_out[k] = xs.time[(size_t)k]; for (int i = 0; i < 3; ++i) x_out[3 * k + i] = xs.at(k)[i]; for (int j = 0; j < 2; ++j) u_out[2 * k + j] = us.at(k)[j]; }
    *n_out = xs.size();
    return ok ? 1 : 0;
}
void fs_forget_guess() { g_guess_n = 0; g_obst_n = -1; }
// the obstacles of the last mpc_solve_batch: rec [cap][4] = n_vertices, radius, vx, vy; verts [cap][cap_v][2]; returns their number (-1: none were handed over)
int fs_last_obstacles(int cap, int cap_v, double* rec, double* verts) {
    for (int o = 0; o < g_obst_n && o < cap; ++o) {
        const double* r = &g_obst[(size_t)o * g_obst_stride];
        for (int i = 0; i < 4; ++i) rec[4 * o + i] = r[i];
        for (int i = 0; i < 2 * (int)r[0] && i < 2 * cap_v; ++i) verts[(size_t)o * cap_v * 2 + i] = r[4 + i];
    }
    return g_obst_n;
}
int fs_last_guess(int cap, double* x, double* u, double* dt, int* cold) {
    const int n = g_guess_n < cap ? g_guess_n : cap;
    for (int i = 0; i < 3 * n; ++i) x[i] = g_guess_x[(size_t)i];
    for (int i = 0; i < 2 * (n - 1); ++i) u[i] = g_guess_u[(size_t)i];
    *dt = g_guess_dt; *cold = g_guess_cold;
    return g_guess_n;
}
// the message of the last step (Controller::publishOptimalControlResult): head as ref_ctl_result_msg of oracle/ref_wrap_controller.cpp
void fs_result_msg(void* p, int n, const double* t, const double* x, const double* u, double* head, double* time_states, double* states, double* time_controls, double* controls) {
    using namespace mpc_local_planner_amd;
    TimeSeries xs, us;
    for (int k = 0; k < n; ++k) { xs.add(t[k], x + 3 * k, 3); us.add(t[k], u + 2 * k, 2); }
    OptimalControlResult m;
    static_cast<Controller*>(p)->optimalControlResult(xs, us, m);
    head[0] = m.seq; head[1] = (double)m.dim_states; head[2] = (double)m.dim_controls; head[3] = m.optimal_solution_found; head[4] = m.cpu_time;
    head[5] = (double)m.time_states.size(); head[6] = (double)m.states.size(); head[7] = (double)m.time_controls.size(); head[8] = (double)m.controls.size();
    std::memcpy(time_states, m.time_states.data(), 8 * m.time_states.size()); std::memcpy(states, m.states.data(), 8 * m.states.size());
    std::memcpy(time_controls, m.time_controls.data(), 8 * m.time_controls.size()); std::memcpy(controls, m.controls.data(), 8 * m.controls.size());
}
}  // extern "C"

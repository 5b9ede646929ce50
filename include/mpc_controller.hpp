// mpc_controller.hpp -- ROS-free C++ host facade with the shape of mpc_local_planner's Controller
// (reference: mpc_local_planner/include/mpc_local_planner/controller.h:61-104,
//             mpc_local_planner/src/controller.cpp:102-179, 807-857).
//
// Header-only; all arithmetic of the solve happens behind the C ABI (include/mpc_hip.h, libmpc_hip.so).  What lives
// here is exactly the host logic the reference runs around PredictiveController::step:
//   * goal -> xf, start/odometry -> x0 (state == pose for every model, base_robot_se2.h),
//   * re-initialisation decision (every k steps / goal jumped), src/controller.cpp:152-158,
//   * initial state trajectory from the plan (time-equidistant poses, yaw from finite differences),
//     src/controller.cpp:807-857, sampled onto the grid with the SE2-aware linear interpolation of
//     src/utils/time_series_se2.cpp:34-111  (full_discretization_grid_base_se2.cpp:192-239),
//   * warm start: previous solution handed back with x_0 overwritten (fixed grid: shifted first, warm_start_shifting(); variable grid: no shifting,
//     finite_differences_variable_grid_se2.h:85),
//   * result time series as getStateAndControlTimeSeries (full_discretization_grid_base_se2.cpp:579-615).
// Types carry the reference's names without ROS: PoseSE2 (teb), Twist (geometry_msgs), TimeSeries (corbo).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

extern "C" {
#include "mpc_hip.h"

// mpc_step_batch (all outer OCP iterations of a control cycle enqueued at once) is used when the linked C ABI has it: libmpc_hip.so does; the recording
// stand-in ABI of the CPU tests (tests/host_harness/facade_step_host.cpp) does not, and the facade then iterates on the host as before
#ifndef MPC_FACADE_HOST_LOOP_ONLY      // (defined by builds on a stand-in ABI that live in one process with the real library)
extern "C" int mpc_step_batch(mpc_solver* s, int32_t B, const double* x0, const double* xf, const double* u_prev, const double* dt_prev, const double* x_init,
                              const double* u_init, const double* dt_init, const mpc_obstacles* obstacles, int32_t outer_iterations, int32_t adapt, int32_t n_min,
                              int32_t n_max, double dt_hyst_ratio, double* x_out, double* u_out, double* dt_out, int32_t* status, int32_t* iters, int32_t* n_grid_out)
    __attribute__((weak));
#define MPC_FACADE_HAS_STEP_BATCH (mpc_step_batch != nullptr)
#else
#define MPC_FACADE_HAS_STEP_BATCH false
#endif
}

namespace mpc_local_planner_amd {

struct PoseSE2 {
    double x = 0, y = 0, theta = 0;
};
struct Twist {
    double linear_x = 0, linear_y = 0, angular_z = 0;
};

// corbo::TimeSeries, reduced to what Controller::step fills: times + column-major values
struct TimeSeries {
    int dim = 0;
    std::vector<double> time;
    std::vector<double> values;   // values[dim * k + i]
    void clear() { time.clear(); values.clear(); }
    void add(double t, const double* v, int d) { dim = d; time.push_back(t); values.insert(values.end(), v, v + d); }
    int size() const { return (int)time.size(); }
    const double* at(int k) const { return &values[(size_t)dim * k]; }
};

// include/mpc_local_planner/utils/math_utils.h:81-103
inline double normalize_theta(double th) {
    const double pi = 3.14159265358979323846;
    if (th >= -pi && th < pi) return th;
    double m = std::floor(th / (2.0 * pi));
    th = th - m * 2.0 * pi;
    if (th >= pi) th -= 2.0 * pi;
    if (th < -pi) th += 2.0 * pi;
    return th;
}
inline double interpolate_angle(double a1, double a2, double f) { return normalize_theta(a1 + f * normalize_theta(a2 - a1)); }

// TimeSeriesSE2::getValuesInterpolate, linear + zero-order-hold extrapolation (src/utils/time_series_se2.cpp:34-111)
inline void interpolate_se2(const std::vector<double>& times, const std::vector<double>& vals, double t, double out[3], double tol = 1e-6) {
    const int n = (int)times.size();
    int idx = -1;
    for (int i = 0; i < n; ++i) if (times[i] >= t) { idx = i; break; }
    if (idx < 0) { for (int i = 0; i < 3; ++i) out[i] = vals[3 * (n - 1) + i]; return; }
    if (std::fabs(t - times[idx]) < tol || idx < 1) { for (int i = 0; i < 3; ++i) out[i] = vals[3 * idx + i]; return; }
    const double fr = (t - times[idx - 1]) / (times[idx] - times[idx - 1]);
    for (int i = 0; i < 2; ++i) out[i] = vals[3 * (idx - 1) + i] + fr * (vals[3 * idx + i] - vals[3 * (idx - 1) + i]);
    out[2] = interpolate_angle(vals[3 * (idx - 1) + 2], vals[3 * idx + 2], fr);
}

// Controller::generateInitialStateTrajectory (src/controller.cpp:807-857) followed by the sampling of
// initializeSequences(xinit) (full_discretization_grid_base_se2.cpp:192-239): fills x_init[n][3].
// NOTE: `backward` has no effect in the reference (the result of normalize_theta(yaw + pi) is discarded at :841).
// dt_sample: the spacing the trajectory is sampled at, dt_ref unless the caller reproduces the reference's re-initialisation (Controller::step below)
inline void initial_state_trajectory(const std::vector<PoseSE2>& plan, const double x0[3], const double xf[3], int n, double dt_ref,
                                     bool estimate_orientation, double* x_init, double dt_sample = -1.0) {
    if (dt_sample <= 0.0) dt_sample = dt_ref;
    const int np = (int)plan.size();
    std::vector<double> times, vals;
    times.push_back(0.0); vals.insert(vals.end(), x0, x0 + 3);
    const double tf = (n - 1) * dt_ref;
    const double dt_init = tf / double(np - 1);
    double t = dt_init;
    for (int i = 1; i < np - 1; ++i) {
        double yaw = plan[i].theta;
        if (estimate_orientation) yaw = std::atan2(plan[i + 1].y - plan[i].y, plan[i + 1].x - plan[i].x);
        const double v[3] = {plan[i].x, plan[i].y, yaw};
        times.push_back(t); vals.insert(vals.end(), v, v + 3);
        t += dt_init;
    }
    times.push_back(tf); vals.insert(vals.end(), xf, xf + 3);
    for (int i = 0; i < 3; ++i) x_init[i] = x0[i];
    for (int k = 1; k < n - 1; ++k) interpolate_se2(times, vals, k * dt_sample, &x_init[3 * k]);
    for (int i = 0; i < 3; ++i) x_init[3 * (n - 1) + i] = xf[i];
}

// FullDiscretizationGridBaseSE2::resampleTrajectory (src/optimal_control/full_discretization_grid_base_se2.cpp:440-524):
// linear (theta-aware) re-interpolation of the previous solution onto n_new points, controls held, dt rescaled so that
// the horizon length is unchanged.  x/u are [n][3] / [n][2] with the duplicated last control row; in place.
inline void resample_trajectory(std::vector<double>& x, std::vector<double>& u, double& dt, int n, int n_new, int stride_n) {
    if (n == n_new) return;
    std::vector<double> xo(x.begin(), x.begin() + 3 * n), uo(u.begin(), u.begin() + 2 * n);
    const double dt_old = dt, dt_new = dt_old * double(n - 1) / double(n_new - 1);
    int idx_old = 1;
    for (int idx_new = 1; idx_new < n_new - 1; ++idx_new) {
        const double t_new = dt_new * double(idx_new);
        while (t_new > double(idx_old) * dt_old && idx_old < n) ++idx_old;
        const double t_old_p1 = double(idx_old) * dt_old;
        const double* xp = &xo[3 * (idx_old - 1)];
        const double* xc = idx_old < n - 1 ? &xo[3 * idx_old] : &xo[3 * (n - 1)];
        const double fr = (t_new - (t_old_p1 - dt_old)) / dt_old;
        for (int i = 0; i < 2; ++i) x[3 * idx_new + i] = xp[i] + fr * (xc[i] - xp[i]);
        x[3 * idx_new + 2] = interpolate_angle(xp[2], xc[2], fr);
        for (int j = 0; j < 2; ++j) u[2 * idx_new + j] = uo[2 * (idx_old - 1) + j];
    }
    for (int i = 0; i < 3; ++i) x[3 * (n_new - 1) + i] = xo[3 * (n - 1) + i];
    for (int j = 0; j < 2; ++j) u[2 * (n_new - 1) + j] = u[2 * (n_new - 2) + j];
    (void)stride_n;
    dt = dt_new;
}

// findNearestState + warmStartShifting (src/optimal_control/full_discretization_grid_base_se2.cpp:241-339): moving-horizon warm
// start of the FIXED-dt grid.  The previous solution is shifted by the index of the state nearest to the new start (search
// stops at the first non-improving sample, look-ahead <= 20), the tail is extrapolated linearly (angles with
// interpolate_angle(.., 2.0)) and the last control is held.  x: [n][3], u: [n][2] (row n-1 of u is the duplicated control).
inline int find_nearest_state(const double* x, int n, const double x0[3]) {
    auto dist = [&](int i) { const double a = x0[0] - x[3 * i], b = x0[1] - x[3 * i + 1], c = x0[2] - x[3 * i + 2]; return std::sqrt(a * a + b * b + c * c); };
    const double first = dist(0);
    if (std::fabs(first) < 1e-12) return 0;
    const int look = std::min((n - 1) - 1, 20);
    int best = 0; double cache = first;
    for (int i = 1; i <= look; ++i) {
        const double d = dist(i);
        if (d < cache) { cache = d; best = i; } else break;
    }
    return best;
}
inline void warm_start_shifting(double* x, double* u, int n, const double x0[3]) {
    const int ns = find_nearest_state(x, n, x0);
    if (ns <= 0 || ns > n - 2) return;
    for (int i = 0; i < n - ns; ++i) {
        const int idx = i + ns;
        for (int c = 0; c < 3; ++c) x[3 * i + c] = x[3 * (idx == n - 1 ? n - 1 : idx) + c];
        if (idx != n - 1) { u[2 * i] = u[2 * idx]; u[2 * i + 1] = u[2 * idx + 1]; }
    }
    int idx = n - ns;
    for (int i = 0; i < ns; ++i, ++idx) {
        for (int c = 0; c < 2; ++c) x[3 * idx + c] = x[3 * (idx - 2) + c] + 2.0 * (x[3 * (idx - 1) + c] - x[3 * (idx - 2) + c]);
        const double a1 = x[3 * (idx - 2) + 2], a2 = x[3 * (idx - 1) + 2];
        x[3 * idx + 2] = normalize_theta(a1 + 2.0 * normalize_theta(a2 - a1));          // interpolate_angle(a1, a2, 2.0)
        u[2 * (idx - 1)] = u[2 * (idx - 2)]; u[2 * (idx - 1) + 1] = u[2 * (idx - 2) + 1];
    }
    u[2 * (n - 1)] = u[2 * (n - 2)]; u[2 * (n - 1) + 1] = u[2 * (n - 2) + 1];               // keep the duplicated last control consistent
}

// ---- what the reference's plugin prepares before Controller::step (src/mpc_local_planner_ros.cpp); each function restates that source, line ranges cited per function
// (the plugin source needs ROS / teb headers the build image lacks: not executed here)

// MpcLocalPlannerROS::updateViaPointsContainer (:619-635): walking along the transformed plan, a pose becomes a via-point when it is at least min_separation away
// (x / y) from the previously inserted one; the first pose counts as inserted but is no via-point.  min_separation <= 0: none.
inline std::vector<PoseSE2> via_points_from_plan(const std::vector<PoseSE2>& plan, double min_separation) {
    std::vector<PoseSE2> out;
    if (min_separation <= 0) return out;
    size_t prev = 0;
    for (size_t i = 1; i < plan.size(); ++i) {
        const double dx = plan[i].x - plan[prev].x, dy = plan[i].y - plan[prev].y;
        if (std::sqrt(dx * dx + dy * dy) < min_separation) continue;
        out.push_back(plan[i]);
        prev = i;
    }
    return out;
}

// MpcLocalPlannerROS::pruneGlobalPlan (:645-685): cuts off what lies behind the robot -- everything before the FIRST plan pose closer than dist_behind_robot to it.  The robot
// pose is given in the plan's frame.  Always true, as in the reference (with no pose that close the plan stays as it is: `erase_end` starts at begin(), :661-675).
inline bool prune_global_plan(std::vector<PoseSE2>& plan, const PoseSE2& robot_in_plan_frame, double dist_behind_robot = 1.0) {
    const double thr = dist_behind_robot * dist_behind_robot;
    for (size_t i = 0; i < plan.size(); ++i) {
        const double dx = robot_in_plan_frame.x - plan[i].x, dy = robot_in_plan_frame.y - plan[i].y;
        if (dx * dx + dy * dy < thr) { plan.erase(plan.begin(), plan.begin() + (long)i); return true; }
    }
    return true;
}
// MpcLocalPlannerROS::transformGlobalPlan (:687-805) for a plan already in the planning frame: the part handed to the controller -- from the plan pose closest to the robot
// (searched until the plan leaves 85 % of the local costmap's half size) onwards, while the poses stay inside that radius and the length along the plan stays within
// max_plan_length (<= 0: no limit).  An empty selection yields the global goal alone.  Returns the index of the last selected pose in the global plan.
inline int transform_global_plan(const std::vector<PoseSE2>& plan, const PoseSE2& robot, int costmap_size_x, int costmap_size_y, double resolution, double max_plan_length,
                                 std::vector<PoseSE2>& selected) {
    selected.clear();
    const int n = (int)plan.size();
    if (n == 0) return -1;
    double thr = std::max(costmap_size_x * resolution / 2.0, costmap_size_y * resolution / 2.0);
    thr *= 0.85;
    const double sq_thr = thr * thr;
    auto sq = [&](int j) { const double dx = robot.x - plan[(size_t)j].x, dy = robot.y - plan[(size_t)j].y; return dx * dx + dy * dy; };
    int i = 0;
    double sq_dist = 1e10;
    for (int j = 0; j < n; ++j) {
        const double d = sq(j);
        if (d > sq_thr) break;
        if (d < sq_dist) { sq_dist = d; i = j; }
    }
    double length = 0.0;
    while (i < n && sq_dist <= sq_thr && (max_plan_length <= 0 || length <= max_plan_length)) {
        selected.push_back(plan[(size_t)i]);
        sq_dist = sq(i);
        if (i > 0 && max_plan_length > 0) {
            const double dx = plan[(size_t)i].x - plan[(size_t)i - 1].x, dy = plan[(size_t)i].y - plan[(size_t)i - 1].y;
            length += std::sqrt(dx * dx + dy * dy);
        }
        ++i;
    }
    if (selected.empty()) { selected.push_back(plan.back()); return n - 1; }
    return i - 1;
}

// costmap_converter/ObstacleMsg reduced to what the plugin reads: polygon points (x, y), radius, planar velocity
struct ObstacleMessage {
    std::vector<double> points;      // x0, y0, x1, y1, ...   (geometry_msgs/Point32: single precision on the wire)
    double radius = 0.0;
    double vx = 0.0, vy = 0.0;
};
// The obstacles of ONE planner instance in the layout of struct mpc_obstacles (capacity = the handle's max_obstacles / max_vertices).  fromMessages restates
// updateObstacleContainerWithCostmapConverter (:501-541; converter = true, messages already in the planning frame) and updateObstacleContainerWithCustomObstacles
// (:543-617; converter = false, moved by the planar transform (yaw, tx, ty), a message without points is skipped): 1 point + radius > 0 = circle, 1 point = point,
// 2 = line, more = polygon; a velocity below 1 mm/s leaves the obstacle static (teb_local_planner Obstacle::setCentroidVelocity); as in the reference the velocity of a
// message goes to the LAST obstacle of the container (in the converter path a message without points therefore re-labels the obstacle before it).
class ObstacleSet {
 public:
    ObstacleSet(int max_obstacles, int max_vertices)
        : _O(max_obstacles), _V(max_vertices), _nv((size_t)max_obstacles, 0), _verts((size_t)max_obstacles * max_vertices * 2, 0.0), _radius((size_t)max_obstacles, 0.0),
          _vel((size_t)max_obstacles * 2, 0.0) {}
    void clear() { _n = 0; }
    int size() const { return _n; }
    // false when the capacity is exceeded (nothing is dropped silently: the caller decides)
    bool add(const double* xy, int n_vertices, double radius = 0.0) {
        if (_n >= _O || n_vertices > _V || n_vertices < 1) return false;
        _nv[(size_t)_n] = n_vertices; _radius[(size_t)_n] = radius; _vel[(size_t)2 * _n] = _vel[(size_t)2 * _n + 1] = 0.0;
        for (int i = 0; i < 2 * n_vertices; ++i) _verts[((size_t)_n * _V) * 2 + i] = xy[i];
        ++_n;
        return true;
    }
    bool fromMessages(const std::vector<ObstacleMessage>& msgs, bool converter = true, double yaw = 0.0, double tx = 0.0, double ty = 0.0, bool append = false) {
        if (!append) clear();
        const double c = std::cos(yaw), s = std::sin(yaw);
        for (const ObstacleMessage& m : msgs) {
            const int k = (int)m.points.size() / 2;
            if (k == 0 && !converter) continue;                                   // :592-596 "Invalid custom obstacle received ... Skipping"
            if (k > 0) {
                std::vector<double> xy((size_t)2 * k);
                for (int i = 0; i < k; ++i) {
                    const double x = (double)(float)m.points[(size_t)2 * i], y = (double)(float)m.points[(size_t)2 * i + 1];
                    xy[(size_t)2 * i] = converter ? x : tx + c * x - s * y;
                    xy[(size_t)2 * i + 1] = converter ? y : ty + s * x + c * y;
                }
                if (!add(xy.data(), k, (k == 1 && m.radius > 0) ? m.radius : 0.0)) return false;
            }
            if (_n > 0 && std::sqrt(m.vx * m.vx + m.vy * m.vy) >= 0.001) { _vel[(size_t)2 * (_n - 1)] = m.vx; _vel[(size_t)2 * (_n - 1) + 1] = m.vy; }
        }
        return true;
    }
    void setLastVelocity(double vx, double vy) { if (_n > 0) { _vel[(size_t)2 * (_n - 1)] = vx; _vel[(size_t)2 * (_n - 1) + 1] = vy; } }
    // borrowed view for Controller::setObstacles / mpc_solve_batch with B = 1 (valid until the set changes)
    const mpc_obstacles* view() {
        _count = _n;
        _view.n_obstacles = &_count; _view.n_vertices = _nv.data(); _view.vertices = _verts.data(); _view.radius = _radius.data(); _view.velocity = _vel.data();
        return &_view;
    }
    int nVertices(int o) const { return _nv[(size_t)o]; }
    const double* vertices(int o) const { return &_verts[((size_t)o * _V) * 2]; }
    double radius(int o) const { return _radius[(size_t)o]; }
    const double* velocity(int o) const { return &_vel[(size_t)2 * o]; }
 private:
    int _O, _V, _n = 0;
    int32_t _count = 0;
    std::vector<int32_t> _nv;
    std::vector<double> _verts, _radius, _vel;
    mpc_obstacles _view{};
};

// MpcLocalPlannerROS::estimateLocalGoalOrientation (:807-852): the local goal's heading with controller/global_plan_overwrite_orientation -- near the end of the global
// plan the goal's own heading (rotated into the planning frame), otherwise the circular mean of the directions between up to moving_average_length successive plan poses
// beyond the local goal.  global_plan in its own frame, (yaw, tx, ty) the planar transform plan -> planning frame, local_goal already in the planning frame.
inline double estimate_local_goal_orientation(const std::vector<PoseSE2>& global_plan, const PoseSE2& local_goal, int current_goal_idx, double yaw, double tx, double ty,
                                              int moving_average_length = 3) {
    const int n = (int)global_plan.size();
    if (current_goal_idx > n - moving_average_length - 2) {
        if (current_goal_idx >= n - 1) return local_goal.theta;
        const double a = yaw, b = global_plan.back().theta;                      // getYaw(rotation * orientation), both rotations about z
        const double az = std::sin(0.5 * a), aw = std::cos(0.5 * a), bz = std::sin(0.5 * b), bw = std::cos(0.5 * b);
        const double z = aw * bz + az * bw, w = aw * bw - az * bz;
        return std::atan2(2.0 * w * z, 1.0 - 2.0 * z * z);
    }
    moving_average_length = std::min(moving_average_length, n - current_goal_idx - 1);
    const double c = std::cos(yaw), s = std::sin(yaw);
    double px = local_goal.x, py = local_goal.y, sx = 0.0, sy = 0.0;
    const int end = current_goal_idx + moving_average_length;
    for (int i = current_goal_idx; i < end; ++i) {
        const PoseSE2& q = global_plan[(size_t)i + 1];
        const double nx = tx + c * q.x - s * q.y, ny = ty + s * q.x + c * q.y;
        const double a = std::atan2(ny - py, nx - px);
        sx += std::cos(a); sy += std::sin(a);
        if (i < end - 1) { px = nx; py = ny; }
    }
    return (sx == 0.0 && sy == 0.0) ? 0.0 : std::atan2(sy, sx);
}

// mpc_local_planner_msgs/OptimalControlResult (msg/OptimalControlResult.msg:1-12) without the ROS header: the wire layout that
// Controller::publishOptimalControlResult fills (src/controller.cpp:197-221).  states / controls are "Column Major" = corbo::TimeSeries'
// value matrix (dim x N, column-major), i.e. sample-major: states[dim_states * k + i] -- exactly one row of the ABI's x_out / u_out.
struct OptimalControlResult {
    uint32_t seq = 0;                       // header.seq = _ocp_seq (:202)
    int64_t dim_states = 0, dim_controls = 0;
    std::vector<double> time_states, states;
    std::vector<double> time_controls, controls;
    bool optimal_solution_found = false;
    double cpu_time = 0.0;                  // _statistics.step_time (:206): wall time of the last step(), seconds
};
// fills `msg` from the time series a step() returned (:208-218: empty series leave the arrays empty)
inline void fill_optimal_control_result(const TimeSeries& x_seq, const TimeSeries& u_seq, bool solution_found, double cpu_time, uint32_t seq,
                                        OptimalControlResult& msg) {
    msg.seq = seq;
    msg.dim_states = 3;                     // _dynamics->getStateDimension(): SE2 state for every model (systems/base_robot_se2.h)
    msg.dim_controls = 2;                   // getInputDimension()
    msg.optimal_solution_found = solution_found;
    msg.cpu_time = cpu_time;
    msg.time_states.clear(); msg.states.clear(); msg.time_controls.clear(); msg.controls.clear();
    if (x_seq.size() > 0) { msg.time_states = x_seq.time; msg.states = x_seq.values; }
    if (u_seq.size() > 0) { msg.time_controls = u_seq.time; msg.controls = u_seq.values; }
}

class Controller {
 public:
    Controller() = default;
    Controller(const Controller&) = delete;
    Controller& operator=(const Controller&) = delete;
    ~Controller() { if (_h) mpc_destroy(_h); }

    // replaces configure(ros::NodeHandle&, obstacles, footprint, via_points): the caller fills mpc_config from its
    // parameter source (INTEGRATION.md lists the key -> field mapping)
    bool configure(const mpc_config& cfg, int device = 0) {
        if (_h) { mpc_destroy(_h); _h = nullptr; }
        _cfg = cfg;
        _n_ref = cfg.n;                                   // grid/grid_size_ref
        _n = _grid_adapt && cfg.dt_free ? (cfg.n > _n_max ? cfg.n : _n_max) : cfg.n;   // capacity (stride) of the arrays
        _n_cur = _n_ref;
        _cfg.n = _n;
        if (mpc_create(&_cfg, 1, device, &_h) != MPC_OK) { _last_error = mpc_last_error(); return false; }
        _x.assign((size_t)3 * _n, 0.0); _u.assign((size_t)2 * _n, 0.0);
        _xi.assign((size_t)3 * _n, 0.0); _ui.assign((size_t)2 * _n, 0.0);
        reset();
        return true;
    }

    // parameters of src/controller.cpp:74-84
    void setForceReinit(int num_steps, double new_goal_dist, double new_goal_angular) {
        _force_reinit_num_steps = num_steps; _force_reinit_new_goal_dist = new_goal_dist; _force_reinit_new_goal_angular = new_goal_angular;
    }
    void setInitialPlanEstimateOrientation(bool e) { _initial_plan_estimate_orientation = e; }
    // grid/variable_grid/grid_adaptation/* (src/controller.cpp:248-261); call BEFORE configure().  The batched solver needs
    // at least 3 grid points, so min_grid_size is clamped to 3 (the reference allows 2).
    void setGridAdaptation(bool enable, int max_grid_size = 50, double dt_hyst_ratio = 0.1, int min_grid_size = 2) {
        _grid_adapt = enable; _n_max = max_grid_size; _dt_hyst = dt_hyst_ratio; _n_min = min_grid_size < 3 ? 3 : min_grid_size;
    }
    void setWarmStart(bool w) { _warm_start = w; }      // grid/warm_start (src/controller.cpp:294-296)
    void setNumOcpIterations(int n) { _num_ocp_iterations = n < 1 ? 1 : n; }      // controller/outer_ocp_iterations (src/controller.cpp:70-72)
    // all outer iterations of a step() in one mpc_step_batch call (no host round trip between them; same results).  Off by default: the CPU tests pin the
    // host loop against the executed reference; include/mpc_reference_binding.hpp switches it on
    void setSingleLaunchStep(bool on) { _single_launch_step = on; }
    // The reference samples the initial state trajectory at the grid's CURRENT dt (full_discretization_grid_base_se2.cpp:61-65: precompute(getDt(), ...)), and
    // clear() does not put dt back to dt_ref (:526-536).  So on the variable grid every re-initialisation AFTER a first solve (goal jump, reset(), force_reinit_num_steps)
    // samples the plan at the LAST OPTIMISED dt while the plan's time axis still spans (n_ref - 1) dt_ref: a guess that is compressed (dt < dt_ref) or runs into the
    // goal early (dt > dt_ref).  (Read off the cited lines; the reference's Controller does not build in this image.)  true (default): reproduce it, so that re-initialised
    // solves start where the reference's do; false: always sample at dt_ref.
    void setReferenceReinitSampling(bool on) { _reference_reinit_sampling = on; }
    int gridSize() const { return _n_cur; }

    // Controller::stateFeedbackCallback (src/controller.cpp:181-195) + controller/prefer_x_feedback (:82): a measured state that is younger than
    // two controller periods replaces the odometry pose when prefer_x_feedback is set; otherwise the odometry pose overwrites the whole state
    // (BaseRobotSE2::mergeStateFeedbackAndOdomFeedback, include/mpc_local_planner/systems/base_robot_se2.h:93-101), which also makes the
    // prediction from the previous state sequence (:139-142) irrelevant for every model of this package.
    void stateFeedbackCallback(const double state[3], double stamp) { _x_feedback[0] = state[0]; _x_feedback[1] = state[1]; _x_feedback[2] = state[2]; _x_feedback_time = stamp; _have_x_feedback = true; }
    void setPreferStateFeedback(bool p) { _prefer_x_feedback = p; }

    // ocp->setPreviousControlInput(u, dt)  (src/mpc_local_planner_ros.cpp:384)
    void setPreviousControlInput(const double u[2], double dt) { _u_prev[0] = u[0]; _u_prev[1] = u[1]; _dt_prev = dt; }

    // obstacles of the next step() calls (borrowed, like the reference's ObstContainer reference)
    void setObstacles(const mpc_obstacles* obst) { _obst = obst; }

    // via-points of the next step() calls (the reference's ViaPointContainer, refilled by the planner every cycle,
    // src/mpc_local_planner_ros.cpp:619-635); needs cfg.objective = MPC_OBJ_MIN_TIME_VIA_POINTS.  poses: n_via x (x, y, theta)
    bool setViaPoints(const double* poses, int n_via) {
        if (!_h || _cfg.max_via_points <= 0 || n_via > _cfg.max_via_points) return false;
        std::vector<double> buf((size_t)_cfg.max_via_points * 3, 0.0);
        for (int i = 0; i < 3 * n_via; ++i) buf[i] = poses[i];
        const int32_t nv = n_via;
        return mpc_set_via_points(_h, 1, &nv, buf.data()) == MPC_OK;
    }

    // Controller::step(start, goal, ...)  (src/controller.cpp:102-109)
    bool step(const PoseSE2& start, const PoseSE2& goal, const Twist& vel, double dt, double t, TimeSeries& u_seq, TimeSeries& x_seq) {
        std::vector<PoseSE2> plan(2);
        plan.front() = start; plan.back() = goal;
        return step(plan, vel, dt, t, u_seq, x_seq);
    }

    // Controller::step(initial_plan, ...)  (src/controller.cpp:111-179)
    bool step(const std::vector<PoseSE2>& plan, const Twist& /*vel*/, double dt, double t, TimeSeries& u_seq, TimeSeries& x_seq) {
        const auto t_step0 = std::chrono::steady_clock::now();
        _last_error.clear();
        if (!_h) { _last_error = "Controller must be configured before invoking step()."; return false; }
        if (plan.size() < 2) { _last_error = "Controller::step(): initial plan must contain at least two poses."; return false; }
        const PoseSE2& start = plan.front();
        const PoseSE2& goal = plan.back();
        const double xf[3] = {goal.x, goal.y, goal.theta};
        // state == pose for every model; :131-149: a fresh state measurement wins only with prefer_x_feedback, else the odometry pose overwrites the state
        const bool new_x = _have_x_feedback && (t - _x_feedback_time) < 2.0 * dt;
        const bool use_fb = new_x && _prefer_x_feedback;
        const double x0[3] = {use_fb ? _x_feedback[0] : start.x, use_fb ? _x_feedback[1] : start.y, use_fb ? _x_feedback[2] : start.theta};
        // re-initialisation decision, :152-158
        if (_force_reinit_num_steps > 0 && _ocp_seq % _force_reinit_num_steps == 0) _grid_empty = true;
        if (!_grid_empty) {
            const double dx = goal.x - _last_goal.x, dy = goal.y - _last_goal.y;
            if (std::sqrt(dx * dx + dy * dy) > _force_reinit_new_goal_dist ||
                std::fabs(normalize_theta(goal.theta - _last_goal.theta)) > _force_reinit_new_goal_angular)
                _grid_empty = true;
        }
        // PredictiveController::step repeats the OCP (grid update + solve) num_ocp_iterations times per control cycle
        // (controller/outer_ocp_iterations, src/controller.cpp:70-72); every repetition after the first starts from the solution just computed
        int32_t status = -1, iters = 0;
        const bool one_call = _single_launch_step && MPC_FACADE_HAS_STEP_BATCH && _num_ocp_iterations > 1;
        for (int outer = 0; outer < (one_call ? 1 : _num_ocp_iterations); ++outer) {
        const double* xi = nullptr; const double* ui = nullptr; const double* di = nullptr;
        if (_grid_empty) {
            _n_cur = _n_ref;
            const double dt_sample = (_reference_reinit_sampling && _have_solution && _cfg.dt_free && _dt_sol > 0.0) ? _dt_sol : _cfg.dt_ref;
            if (plan.size() > 2 || dt_sample != _cfg.dt_ref) {      // a 2-pose plan sampled at dt_ref is the device-side cold start
                initial_state_trajectory(plan, x0, xf, _n_cur, _cfg.dt_ref, _initial_plan_estimate_orientation, _xi.data(), dt_sample);
                std::fill(_ui.begin(), _ui.end(), 0.0);
                _dti = _cfg.dt_ref;
                xi = _xi.data(); ui = _ui.data(); di = &_dti;
            }
        } else {
            _xi = _x; _ui = _u; _dti = _dt_sol;      // previous solution = warm start (x_0 / fixed goal are overwritten by the solver)
            // fixed grid only, and only in the FIRST outer iteration of a control cycle: the reference shifts when `new_run` (…grid_base_se2.cpp:96-100); a later repetition starts
            // from the solution just computed as it is (a start heading outside [-pi, pi) is stored normalised: measured against it, x_0 would look 2 pi away -- ADVICE r03)
            if (_warm_start && !_cfg.dt_free && outer == 0) warm_start_shifting(_xi.data(), _ui.data(), _n_cur, x0);
            if (_grid_adapt && _cfg.dt_free) {
                // adaptGridTimeBasedSingleStep (src/optimal_control/finite_differences_variable_grid_se2.cpp:99-121)
                int n_new = _n_cur;
                if (_dt_sol > _cfg.dt_ref * (1.0 + _dt_hyst) && _n_cur < _n_max) n_new = _n_cur + 1;
                else if (_dt_sol < _cfg.dt_ref * (1.0 - _dt_hyst) && _n_cur > _n_min) n_new = _n_cur - 1;
                if (n_new != _n_cur) { resample_trajectory(_xi, _ui, _dti, _n_cur, n_new, _n); _n_cur = n_new; }
            }
            xi = _xi.data(); ui = _ui.data(); di = &_dti;
        }
        if (_n_cur != _n || _sizes_set) {
            const int32_t ng = _n_cur;
            if (mpc_set_grid_sizes(_h, &ng, 1) != MPC_OK) { _last_error = mpc_last_error(); return false; }
            _sizes_set = true;
        }
        int rc;
        if (one_call) {
            // every outer iteration in ONE call: the grid update between them (single-step adaptation + resampling on the variable grid; nothing on the fixed grid,
            // whose shift belongs to the first outer iteration only) runs on the device, bit for bit what the host code above does (tests/test_gpu_closed_loop.py)
            if (_grid_adapt && _cfg.dt_free && !_sizes_set) {
                const int32_t ng = _n_cur;
                if (mpc_set_grid_sizes(_h, &ng, 1) != MPC_OK) { _last_error = mpc_last_error(); return false; }
                _sizes_set = true;
            }
            int32_t n_after = _n_cur;
#ifndef MPC_FACADE_HOST_LOOP_ONLY
            rc = mpc_step_batch(_h, 1, x0, xf, _u_prev, &_dt_prev, xi, ui, di, _obst, _num_ocp_iterations, (_grid_adapt && _cfg.dt_free) ? 1 : 0, _n_min, _n_max, _dt_hyst,
                                _x.data(), _u.data(), &_dt_sol, &status, &iters, &n_after);
#else
            rc = MPC_EINVAL;
#endif
            if (rc == MPC_OK) _n_cur = n_after;
        } else
            rc = mpc_solve_batch(_h, 1, x0, xf, _u_prev, &_dt_prev, xi, ui, di, _obst, _x.data(), _u.data(), &_dt_sol, &status, &iters);
        if (rc != MPC_OK) { _last_error = mpc_last_error(); return false; }
        _grid_empty = false;
        _have_solution = true;
        }
        _last_iterations = iters;
        _ocp_successful = status == MPC_CONVERGED;
        x_seq.clear(); u_seq.clear();
        double t_k = 0.0;                        // getStateAndControlTimeSeries, …grid_base_se2.cpp:579-615: the stamps are ACCUMULATED (t += dt), not k dt
        for (int k = 0; k < _n_cur; ++k, t_k += _dt_sol) {
            x_seq.add(t_k, &_x[(size_t)3 * k], 3);
            u_seq.add(t_k, &_u[(size_t)2 * k], 2);
        }
        _grid_empty = false;
        ++_ocp_seq;
        _last_goal = goal;
        _last_step_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_step0).count();      // _statistics.step_time (:175)
        return _ocp_successful;
    }

    // Controller::publishOptimalControlResult (src/controller.cpp:197-221) without the publisher: the message of the last step()
    void optimalControlResult(const TimeSeries& x_seq, const TimeSeries& u_seq, OptimalControlResult& msg) const {
        // header.seq = _ocp_seq BEFORE step() increments it (:163 publishes, :165 ++_ocp_seq): the first step's message carries 0
        fill_optimal_control_result(x_seq, u_seq, _ocp_successful, _last_step_time, (uint32_t)(_ocp_seq > 0 ? _ocp_seq - 1 : 0), msg);
    }

    // Controller::isPoseTrajectoryFeasible (src/controller.cpp:859-917) of the trajectory the last step() planned, against the local
    // costmap (row-major cost[my * size_x + mx], origin = world position of cell (0, 0)'s lower-left corner).  footprint_spec: n_spec points
    // (x, y) in the robot frame (costmap_2d footprint).  Conventions of footprintCost: see mpc_check_feasibility in mpc_hip.h.
    bool isPoseTrajectoryFeasible(const uint8_t* cost, int size_x, int size_y, double resolution, const double origin[2], const double* footprint_spec,
                                  int n_spec, double inscribed_radius, double /*circumscribed_radius*/, double min_resolution_collision_check_angular,
                                  int look_ahead_idx) {
        if (!_h) { _last_error = "Controller must be configured before invoking step()."; return false; }
        if (_n_cur < 2) return false;
        int32_t ok = 0;
        if (mpc_check_feasibility(_h, 1, _x.data(), cost, size_x, size_y, resolution, origin, footprint_spec, n_spec, inscribed_radius,
                                  min_resolution_collision_check_angular, look_ahead_idx, &ok) != MPC_OK) { _last_error = mpc_last_error(); return false; }
        return ok != 0;
    }

    // Controller::reset (src/controller.cpp:223): the next step starts from a fresh initial guess
    void reset() { _grid_empty = true; if (_h) mpc_reset(_h); }

    int lastIterations() const { return _last_iterations; }
    // clearance rows of the last solve that did not fit into mpc_config.max_obstacle_rows per grid point (the reference has no cap: raise max_obstacle_rows when this is > 0)
    int lastRowsDropped() const { int32_t d = 0; return (_h && mpc_last_rows_dropped(_h, 1, &d) == MPC_OK) ? d : 0; }
    double lastStepTime() const { return _last_step_time; }       // _statistics.step_time (src/controller.cpp:175), seconds
    double lastDt() const { return _dt_sol; }
    const std::string& lastError() const { return _last_error; }

 private:
    mpc_solver* _h = nullptr;
    mpc_config _cfg{};
    int _n = 0, _n_ref = 0, _n_cur = 0;
    bool _grid_adapt = false, _sizes_set = false, _warm_start = true, _single_launch_step = false;
    int _n_max = 50, _n_min = 3;
    double _dt_hyst = 0.1;
    std::vector<double> _x, _u, _xi, _ui;
    double _dt_sol = 0, _dti = 0;
    double _u_prev[2] = {0, 0};
    double _dt_prev = 0;
    const mpc_obstacles* _obst = nullptr;
    bool _grid_empty = true;
    bool _have_solution = false, _reference_reinit_sampling = true;
    bool _ocp_successful = false;
    int _ocp_seq = 0;
    int _last_iterations = 0;
    double _last_step_time = 0.0;
    int _num_ocp_iterations = 1;
    PoseSE2 _last_goal;
    int _force_reinit_num_steps = 0;                     // src/controller.cpp:78
    double _force_reinit_new_goal_dist = 1.0;            // :74
    double _force_reinit_new_goal_angular = 1.5707963267948966;   // :76 (0.5*pi)
    bool _initial_plan_estimate_orientation = true;
    double _x_feedback[3] = {0, 0, 0};
    double _x_feedback_time = 0.0;
    bool _have_x_feedback = false, _prefer_x_feedback = false;      // controller/prefer_x_feedback (src/controller.cpp:82)
    std::string _last_error;
};

}  // namespace mpc_local_planner_amd

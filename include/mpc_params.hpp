// The reference's parameter set -> mpc_config + the options of the Controller facade.
//
// Controller::configure (src/controller.cpp:58-100) reads its parameters from the ROS parameter server while it builds the corbo objects:
// configureRobotDynamics (:344-378), configureGrid (:225-342), configureSolver (:380-481), configureOcp (:483-805); the plugin reads the
// footprint (src/mpc_local_planner_ros.cpp:890-1001).  config_from_params reads the SAME keys with the same in-code defaults, fix-ups and
// rejections from any ParamSource -- a ros::NodeHandle adapter is a dozen lines, MapParamSource below serves tests and
// ROS-free callers -- so a maintainer replaces the body of configure() by
//
//     mpc_config cfg; ControllerOptions opt; ParamReport rep;
//     if (config_from_params(NodeHandleSource(nh), cfg, opt, rep) != PARAMS_OK) { ROS_ERROR_STREAM(rep.error); return false; }
//     cfg.max_obstacles = ...;            // capacities of the handle: not parameters of the reference
//     opt.apply(controller); return controller.configure(cfg);
//
// Python twin with the same behaviour: mpc_local_planner_amd/params.py (tests/test_params.py compares the two field by field).
#pragma once

#include <cstdlib>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "mpc_controller.hpp"
#include "mpc_hip.h"

namespace mpc_local_planner_amd {

// what nh.param / nh.getParam offer, by type (a missing key or a value of another type returns false and leaves `v` alone)
struct ParamSource {
    virtual ~ParamSource() = default;
    virtual bool has(const std::string& key) const = 0;
    virtual bool get(const std::string& key, bool& v) const = 0;
    virtual bool get(const std::string& key, int& v) const = 0;
    virtual bool get(const std::string& key, double& v) const = 0;
    virtual bool get(const std::string& key, std::string& v) const = 0;
    virtual bool get(const std::string& key, std::vector<double>& v) const = 0;
    virtual bool get(const std::string& key, std::vector<bool>& v) const = 0;
    virtual bool get(const std::string& key, std::vector<std::vector<double>>& v) const = 0;      // footprint_model/vertices
    virtual bool get(const std::string& key, std::map<std::string, double>& v) const = 0;          // solver/ipopt/ipopt_numeric_options
    virtual bool get(const std::string& key, std::map<std::string, std::string>& v) const = 0;     // .../ipopt_string_options
    virtual bool get(const std::string& key, std::map<std::string, int>& v) const = 0;             // .../ipopt_integer_options
    template <class T> T param(const std::string& key, T def) const { get(key, def); return def; }     // nh.param(key, var, default)
};

// flat "a/b/c" -> value store
class MapParamSource : public ParamSource {
 public:
    struct Value {
        enum Kind { BOOL, INT, DOUBLE, STRING, DOUBLES, BOOLS, POINTS } kind = DOUBLE;
        bool b = false; int i = 0; double d = 0; std::string s; std::vector<double> dv; std::vector<bool> bv; std::vector<std::vector<double>> pv;
    };
    void set(const std::string& k, bool v) { Value x; x.kind = Value::BOOL; x.b = v; _m[k] = x; }
    void set(const std::string& k, int v) { Value x; x.kind = Value::INT; x.i = v; _m[k] = x; }
    void set(const std::string& k, double v) { Value x; x.kind = Value::DOUBLE; x.d = v; _m[k] = x; }
    void set(const std::string& k, const std::string& v) { Value x; x.kind = Value::STRING; x.s = v; _m[k] = x; }
    void set(const std::string& k, const char* v) { set(k, std::string(v)); }
    void set(const std::string& k, const std::vector<double>& v) { Value x; x.kind = Value::DOUBLES; x.dv = v; _m[k] = x; }
    void set(const std::string& k, const std::vector<bool>& v) { Value x; x.kind = Value::BOOLS; x.bv = v; _m[k] = x; }
    void set(const std::string& k, const std::vector<std::vector<double>>& v) { Value x; x.kind = Value::POINTS; x.pv = v; _m[k] = x; }

    bool has(const std::string& key) const override {
        if (_m.count(key)) return true;
        const std::string pre = key + "/";
        auto it = _m.lower_bound(pre);
        return it != _m.end() && it->first.compare(0, pre.size(), pre) == 0;
    }
    // roscpp (param.cpp): a double parameter takes an int, an int parameter takes a double rounded half up, a bool parameter takes a bool only
    bool get(const std::string& key, bool& v) const override { auto p = find(key); if (!p || p->kind != Value::BOOL) return false; v = p->b; return true; }
    bool get(const std::string& key, int& v) const override {
        auto p = find(key); if (!p) return false;
        if (p->kind == Value::INT) { v = p->i; return true; }
        if (p->kind == Value::DOUBLE) { const double d = p->d; v = (int)(std::fmod(d, 1.0) < 0.5 ? std::floor(d) : std::ceil(d)); return true; }
        return false;
    }
    bool get(const std::string& key, double& v) const override { auto p = find(key); if (!p) return false; if (p->kind == Value::DOUBLE) { v = p->d; return true; } if (p->kind == Value::INT) { v = p->i; return true; } return false; }
    bool get(const std::string& key, std::string& v) const override { auto p = find(key); if (!p || p->kind != Value::STRING) return false; v = p->s; return true; }
    bool get(const std::string& key, std::vector<double>& v) const override { auto p = find(key); if (!p || p->kind != Value::DOUBLES) return false; v = p->dv; return true; }
    bool get(const std::string& key, std::vector<bool>& v) const override { auto p = find(key); if (!p || p->kind != Value::BOOLS) return false; v = p->bv; return true; }
    bool get(const std::string& key, std::vector<std::vector<double>>& v) const override { auto p = find(key); if (!p || p->kind != Value::POINTS) return false; v = p->pv; return true; }
    bool get(const std::string& key, std::map<std::string, double>& v) const override {
        // a number written as text counts when it parses completely: `tol: 1e-4` is TEXT for a YAML 1.1 loader (rosparam's), and roscpp would then reject the
        // whole map, leaving Ipopt's defaults in force; the evident intent is honoured here (config_from_params notes it)
        return children(key, [&](const std::string& k, const Value& x) {
            if (x.kind == Value::DOUBLE) v[k] = x.d;
            else if (x.kind == Value::INT) v[k] = x.i;
            else if (x.kind == Value::STRING) { char* end = nullptr; const double d = std::strtod(x.s.c_str(), &end); if (end != x.s.c_str() && *end == 0) v[k] = d; }
        });
    }
    bool get(const std::string& key, std::map<std::string, std::string>& v) const override {
        return children(key, [&](const std::string& k, const Value& x) { if (x.kind == Value::STRING) v[k] = x.s; });
    }
    bool get(const std::string& key, std::map<std::string, int>& v) const override {
        return children(key, [&](const std::string& k, const Value& x) { if (x.kind == Value::INT) v[k] = x.i; });
    }

 private:
    const Value* find(const std::string& k) const { auto it = _m.find(k); return it == _m.end() ? nullptr : &it->second; }
    template <class F> bool children(const std::string& key, F&& f) const {
        const std::string pre = key + "/";
        bool any = false;
        for (auto it = _m.lower_bound(pre); it != _m.end() && it->first.compare(0, pre.size(), pre) == 0; ++it)
            if (it->first.find('/', pre.size()) == std::string::npos) { f(it->first.substr(pre.size()), it->second); any = true; }
        return any;
    }
    std::map<std::string, Value> _m;
};

// the parameters that live in the facade (src/controller.cpp:70-88, :248-261, :294-296; defaults include/mpc_local_planner/controller.h:124-143)
struct ControllerOptions {
    bool grid_adaptation = true; int max_grid_size = 50; double dt_hyst_ratio = 0.1; int min_grid_size = 2;
    int n_max = 50;                               // grid points the handle has to hold
    bool warm_start = true;
    int outer_ocp_iterations = 1;
    double force_reinit_new_goal_dist = 1.0, force_reinit_new_goal_angular = 1.5707963267948966;
    bool allow_init_with_backward_motion = true;
    int force_reinit_num_steps = 0;
    bool prefer_x_feedback = false, publish_ocp_results = false, print_cpu_time = false;
    void apply(Controller& c) const {             // call before Controller::configure
        c.setGridAdaptation(grid_adaptation, max_grid_size, dt_hyst_ratio, min_grid_size);
        c.setWarmStart(warm_start);
        c.setNumOcpIterations(outer_ocp_iterations);
        c.setForceReinit(force_reinit_num_steps, force_reinit_new_goal_dist, force_reinit_new_goal_angular);
        c.setPreferStateFeedback(prefer_x_feedback);
    }
};

// the parameters MpcLocalPlannerROS::initialize reads for ITSELF (src/mpc_local_planner_ros.cpp:96-125, :220; in-code defaults include/mpc_local_planner/mpc_local_planner_ros.h:369-391):
// what a binding needs around the solve -- goal tolerances, plan pruning / look-ahead, via-point separation (via_points_from_plan), the costmap scan
// (mpc_costmap_to_obstacles: include_costmap_obstacles, costmap_obstacles_behind_robot_dist), the feasibility check (mpc_check_feasibility: collision_check_*).
// controller_frequency is move_base's own parameter (the control period handed to step() is its inverse, :380): read from `move_base` when given.
struct PluginOptions {
    double xy_goal_tolerance = 0.2, yaw_goal_tolerance = 0.1;
    bool global_plan_overwrite_orientation = true;
    double global_plan_prune_distance = 1.0, max_global_plan_lookahead_dist = 1.5, global_plan_viapoint_sep = -1.0;
    std::string odom_topic = "odom";
    bool is_footprint_dynamic = false, include_costmap_obstacles = true;
    double costmap_obstacles_behind_robot_dist = 1.5;
    int collision_check_no_poses = -1;
    double collision_check_min_resolution_angular = 3.14159265358979323846;
    std::string costmap_converter_plugin;
    double costmap_converter_rate = 5.0;
    bool costmap_converter_spin_thread = true;
    double controller_frequency = 10.0;
};
inline PluginOptions plugin_options_from_params(const ParamSource& p, const ParamSource* move_base = nullptr) {
    PluginOptions o;
    o.xy_goal_tolerance = p.param("controller/xy_goal_tolerance", o.xy_goal_tolerance);
    o.yaw_goal_tolerance = p.param("controller/yaw_goal_tolerance", o.yaw_goal_tolerance);
    o.global_plan_overwrite_orientation = p.param("controller/global_plan_overwrite_orientation", o.global_plan_overwrite_orientation);
    o.global_plan_prune_distance = p.param("controller/global_plan_prune_distance", o.global_plan_prune_distance);
    o.max_global_plan_lookahead_dist = p.param("controller/max_global_plan_lookahead_dist", o.max_global_plan_lookahead_dist);
    o.global_plan_viapoint_sep = p.param("controller/global_plan_viapoint_sep", o.global_plan_viapoint_sep);
    o.odom_topic = p.param("odom_topic", o.odom_topic);
    o.is_footprint_dynamic = p.param("footprint_model/is_footprint_dynamic", o.is_footprint_dynamic);
    o.include_costmap_obstacles = p.param("collision_avoidance/include_costmap_obstacles", o.include_costmap_obstacles);
    o.costmap_obstacles_behind_robot_dist = p.param("collision_avoidance/costmap_obstacles_behind_robot_dist", o.costmap_obstacles_behind_robot_dist);
    o.collision_check_no_poses = p.param("collision_avoidance/collision_check_no_poses", o.collision_check_no_poses);
    o.collision_check_min_resolution_angular = p.param("collision_avoidance/collision_check_min_resolution_angular", o.collision_check_min_resolution_angular);
    o.costmap_converter_plugin = p.param("costmap_converter_plugin", o.costmap_converter_plugin);
    o.costmap_converter_rate = p.param("costmap_converter_rate", o.costmap_converter_rate);
    o.costmap_converter_spin_thread = p.param("costmap_converter_spin_thread", o.costmap_converter_spin_thread);
    if (move_base) o.controller_frequency = move_base->param("controller_frequency", o.controller_frequency);
    return o;
}

enum ParamStatus {
    PARAMS_OK = 0,
    PARAMS_REJECTED = 1,           // the reference's configure() returns false on this parameter set (error = its reason)
    PARAMS_NOT_IMPLEMENTED = 2     // valid for the reference, not built here: never a silent fall-back to a different NLP
};
struct ParamReport {
    std::string error;
    std::vector<std::string> notes;   // accepted but without effect, or mapped onto the nearest equivalent
};

namespace detail {
// length dim: diagonal; dim*dim: full matrix, column major (Eigen's default, src/controller.cpp:565-573); x'Mx only sees the symmetric part:
// diag[] gets the diagonal, off[] the symmetric off-diagonal terms (0,1)[, (0,2), (1,2)]
inline ParamStatus weights(const std::vector<double>& v, int dim, const char* reason, double* diag, double* off, ParamReport& rep) {
    const int noff = dim == 3 ? 3 : 1;
    for (int i = 0; i < noff; ++i) off[i] = 0.0;
    if ((int)v.size() == dim) { for (int i = 0; i < dim; ++i) diag[i] = v[i]; return PARAMS_OK; }
    if ((int)v.size() == dim * dim) {
        auto at = [&](int r, int c) { return 0.5 * (v[c * dim + r] + v[r * dim + c]); };
        for (int i = 0; i < dim; ++i) diag[i] = v[i * dim + i];
        off[0] = at(0, 1);
        if (dim == 3) { off[1] = at(0, 2); off[2] = at(1, 2); }
        return PARAMS_OK;
    }
    rep.error = reason;
    return PARAMS_REJECTED;
}
inline void footprint(const ParamSource& p, const std::vector<std::vector<double>>* costmap_footprint, mpc_config& c, ParamReport& rep, ParamStatus& st) {
    c.footprint_kind = MPC_FOOTPRINT_POINT;
    std::string kind;
    if (!p.get("footprint_model/type", kind)) return;                                                            // :894-898
    auto polygon = [&](const std::vector<std::vector<double>>& v) {
        c.footprint_kind = MPC_FOOTPRINT_POLYGON; c.footprint_n_vertices = (int)v.size();
        for (size_t i = 0; i < v.size(); ++i) { c.footprint_vertices[2 * i] = v[i][0]; c.footprint_vertices[2 * i + 1] = v[i][1]; }
    };
    if (kind == "costmap_2d") {
        if (!costmap_footprint || costmap_footprint->size() < 3 || costmap_footprint->size() > 16) {
            rep.notes.push_back("footprint_model/type costmap_2d without a costmap footprint: point model (as the reference without a costmap)"); return; }
        polygon(*costmap_footprint);
    } else if (kind == "point") {
    } else if (kind == "circular") {
        double r;
        if (!p.get("footprint_model/radius", r)) { rep.notes.push_back("Footprint model 'circular' cannot be loaded: footprint_model/radius does not exist. Using point-model instead."); return; }
        c.footprint_kind = MPC_FOOTPRINT_CIRCLE; c.footprint_radius = r;
    } else if (kind == "line") {
        std::vector<double> a, b;
        if (!p.get("footprint_model/line_start", a) || !p.get("footprint_model/line_end", b) || a.size() != 2 || b.size() != 2) {
            rep.notes.push_back("Footprint model 'line' cannot be loaded: line_start / line_end missing or not 2D. Using point-model instead."); return; }
        c.footprint_kind = MPC_FOOTPRINT_LINE;
        c.footprint_params[0] = a[0]; c.footprint_params[1] = a[1]; c.footprint_params[2] = b[0]; c.footprint_params[3] = b[1];
    } else if (kind == "two_circles") {
        const char* keys[4] = {"footprint_model/front_offset", "footprint_model/front_radius", "footprint_model/rear_offset", "footprint_model/rear_radius"};
        double v[4];
        for (int i = 0; i < 4; ++i)
            if (!p.get(keys[i], v[i])) { rep.notes.push_back("Footprint model 'two_circles' cannot be loaded: front_offset, front_radius, rear_offset and rear_radius are needed. Using point-model instead."); return; }
        c.footprint_kind = MPC_FOOTPRINT_TWO_CIRCLES;
        for (int i = 0; i < 4; ++i) c.footprint_params[i] = v[i];
    } else if (kind == "polygon") {
        std::vector<std::vector<double>> v;
        bool ok = p.get("footprint_model/vertices", v) && v.size() >= 3;
        for (size_t i = 0; ok && i < v.size(); ++i) ok = v[i].size() == 2;
        if (!ok) { rep.notes.push_back("Footprint model 'polygon' cannot be loaded: footprint_model/vertices must be a list of at least 3 [x, y] points. Using point-model instead."); return; }
        if (v.size() > 16) { rep.error = "footprint_model/vertices: more than 16 vertices (mpc_config.footprint_vertices)"; st = PARAMS_NOT_IMPLEMENTED; return; }
        polygon(v);
    } else {
        rep.notes.push_back("Footprint model '" + kind + "' unknown. Using point-model instead.");
    }
}
}  // namespace detail

// cfg: every field the reference has a parameter for; the capacities (max_obstacles, max_vertices, max_obstacle_rows, max_via_points) and the
// options the reference does not have (precision, candidates, dual_warm_start, ...) keep the values of mpc_config_defaults -- set them afterwards.
inline ParamStatus config_from_params(const ParamSource& p, mpc_config& c, ControllerOptions& o, ParamReport& rep,
                                      const std::vector<std::vector<double>>* costmap_footprint = nullptr) {
    mpc_config_defaults(&c);
    o = ControllerOptions();
    rep = ParamReport();
    auto reject = [&](const std::string& why) { rep.error = why; return PARAMS_REJECTED; };
    auto missing = [&](const std::string& what) { rep.error = what; return PARAMS_NOT_IMPLEMENTED; };
    const double inf = 1e30;

    // ---- robot (src/controller.cpp:344-378), control bounds (:494-548), control-rate rows (:731-797)
    const std::string robot = p.param<std::string>("robot/type", "unicycle");
    std::string ns, second, rate_key;
    double second_default;
    c.model_params[0] = c.model_params[1] = 0.0;
    if (robot == "unicycle") { c.model = MPC_MODEL_UNICYCLE; ns = "robot/unicycle"; second = "max_vel_theta"; second_default = 0.3; rate_key = "acc_lim_theta"; }
    else if (robot == "simple_car") {
        c.model_params[0] = p.param("robot/simple_car/wheelbase", 0.5);
        c.model = p.param("robot/simple_car/front_wheel_driving", false) ? MPC_MODEL_SIMPLE_CAR_FRONT : MPC_MODEL_SIMPLE_CAR;
        ns = "robot/simple_car"; second = "max_steering_angle"; second_default = 1.5; rate_key = "max_steering_rate";
    } else if (robot == "kinematic_bicycle_vel_input") {
        c.model = MPC_MODEL_KINEMATIC_BICYCLE;
        c.model_params[0] = p.param("robot/kinematic_bicycle_vel_input/length_rear", 1.0);
        c.model_params[1] = p.param("robot/kinematic_bicycle_vel_input/length_front", 1.0);
        ns = "robot/kinematic_bicycle_vel_input"; second = "max_steering_angle"; second_default = 1.5; rate_key = "max_steering_rate";
    } else return reject("Unknown robot type '" + robot + "' specified.");
    const double vmax = p.param(ns + "/max_vel_x", 0.4);
    double vback = p.param(ns + "/max_vel_x_backwards", 0.2);
    if (vback < 0) { rep.notes.push_back("max_vel_x_backwards must be >= 0 (sign flipped, as the reference does)"); vback = -vback; }
    const double wmax = p.param(ns + "/" + second, second_default);
    c.u_lb[0] = -vback; c.u_lb[1] = -wmax; c.u_ub[0] = vmax; c.u_ub[1] = wmax;
    double acc = p.param(ns + "/acc_lim_x", 0.0), dec = p.param(ns + "/dec_lim_x", 0.0);
    if (dec < 0) { rep.notes.push_back("dec_lim_x must be >= 0 (sign flipped, as the reference does)"); dec = -dec; }
    double rate = p.param(ns + "/" + rate_key, 0.0);
    if (acc <= 0) acc = inf;
    if (dec <= 0) dec = inf;
    if (rate <= 0) rate = inf;
    c.du_lb[0] = -dec; c.du_lb[1] = -rate; c.du_ub[0] = acc; c.du_ub[1] = rate;

    // ---- grid (:225-342)
    const std::string grid_type = p.param<std::string>("grid/type", "fd_grid");
    if (grid_type != "fd_grid") return reject("Unknown grid type '" + grid_type + "' specified.");
    c.dt_free = p.param("grid/variable_grid/enable", true) ? 1 : 0;
    if (c.dt_free) {
        c.dt_lb = p.param("grid/variable_grid/min_dt", 0.0);
        c.dt_ub = p.param("grid/variable_grid/max_dt", 10.0);
        o.grid_adaptation = p.param("grid/variable_grid/grid_adaptation/enable", true);
        if (o.grid_adaptation) {
            o.max_grid_size = p.param("grid/variable_grid/grid_adaptation/max_grid_size", 50);
            o.dt_hyst_ratio = p.param("grid/variable_grid/grid_adaptation/dt_hyst_ratio", 0.1);
            o.min_grid_size = p.param("grid/variable_grid/grid_adaptation/min_grid_size", 2);
        }
    } else o.grid_adaptation = false;
    c.n = p.param("grid/grid_size_ref", 20);
    c.dt_ref = p.param("grid/dt_ref", 0.3);
    std::vector<bool> xf_fixed = {true, true, true};
    p.get("grid/xf_fixed", xf_fixed);
    if (xf_fixed.size() != 3) return reject("Array size of `xf_fixed` does not match robot state dimension(): " + std::to_string(xf_fixed.size()) + " != 3");
    for (int i = 0; i < 3; ++i) c.xf_fixed[i] = xf_fixed[i] ? 1 : 0;
    o.warm_start = p.param("grid/warm_start", true);
    std::string colloc = p.param<std::string>("grid/collocation_method", "forward_differences");
    if (colloc == "forward_differences") c.collocation = MPC_COLLOC_FORWARD;
    else if (colloc == "midpoint_differences") c.collocation = MPC_COLLOC_MIDPOINT;
    else if (colloc == "crank_nicolson_differences") c.collocation = MPC_COLLOC_CRANK_NICOLSON;
    else {
        // :314: the reference logs "Falling back to default..." and goes on with the grid's initial rule, corbo's plain Crank-Nicolson differences
        // (full_discretization_grid_base_se2.h:206) -- a rule WITHOUT the SE(2) heading wrap that is not built here
        return missing("Unknown collocation method '" + colloc + "' specified: the reference falls back to corbo::CrankNicolsonDiffCollocation (no SE(2) heading wrap), not built here");
    }
    std::string integration = p.param<std::string>("grid/cost_integration_method", "left_sum");
    if (integration != "left_sum" && integration != "trapezoidal_rule") {
        rep.notes.push_back("Unknown cost integration method '" + integration + "' specified. Falling back to default..."); integration = "left_sum"; }
    o.n_max = o.grid_adaptation ? (c.n > o.max_grid_size ? c.n : o.max_grid_size) : c.n;

    // ---- solver (:380-481)
    const std::string solver = p.param<std::string>("solver/type", "ipopt");
    if (solver == "lsq_lm") return missing("solver/type lsq_lm: the Levenberg-Marquardt least-squares solver is outside this path (SURVEY.md section 8: out of scope)");
    if (solver != "ipopt") return reject("Unknown solver type '" + solver + "' specified.");
    c.max_iter = p.param("solver/ipopt/iterations", 100);
    {   // :395-397 -> mpc_config.max_time_us (per solve, on the device's clock)
        const double t = p.param("solver/ipopt/max_cpu_time", -1.0);
        // (a budget beyond the int32 range of microseconds -- ~2147 s, e.g. `max_cpu_time: 1e9` for "none" -- is no budget: 0)
        c.max_time_us = (t > 0 && t * 1e6 + 0.5 < 2147483647.0) ? (int32_t)(t * 1e6 + 0.5) : 0;
    }
    std::map<std::string, double> numeric; std::map<std::string, std::string> strings; std::map<std::string, int> integers;
    p.get("solver/ipopt/ipopt_numeric_options", numeric);
    {
        std::map<std::string, std::string> as_text;
        p.get("solver/ipopt/ipopt_numeric_options", as_text);
        for (const auto& kv : as_text)
            if (numeric.count(kv.first))
                rep.notes.push_back("ipopt numeric option " + kv.first + " is text ('" + kv.second + "' is not a YAML 1.1 float): roscpp rejects the whole map and the reference runs with Ipopt's defaults; the value is used here");
    }
    p.get("solver/ipopt/ipopt_string_options", strings);
    p.get("solver/ipopt/ipopt_integer_options", integers);
    c.tol = 1e-8;                                            // Ipopt's default tol
    for (const auto& kv : numeric) {
        if (kv.first == "tol") c.tol = kv.second;
        else if (kv.first == "mu_init") c.mu_init = kv.second;
        else if (kv.first == "acceptable_tol") c.acceptable_tol = kv.second > 0 ? kv.second : -1.0;
        else rep.notes.push_back("ipopt numeric option " + kv.first + ": no counterpart, ignored");
    }
    for (const auto& kv : strings) {
        if (kv.first == "hessian_approximation") {
            if (kv.second == "limited-memory") {
                c.hessian_mode = MPC_HESSIAN_CONVEXIFIED;
                rep.notes.push_back("hessian_approximation limited-memory -> MPC_HESSIAN_CONVEXIFIED (the positive-semidefinite part of the exact stage Hessians; no quasi-Newton update is built)");
            } else c.hessian_mode = MPC_HESSIAN_EXACT;
        } else if (kv.first == "mu_strategy") {
            if (kv.second == "monotone") c.mu_strategy = MPC_MU_MONOTONE;
            else if (kv.second == "adaptive") c.mu_strategy = MPC_MU_ADAPTIVE;
            else rep.notes.push_back("mu_strategy " + kv.second + ": unknown, the default (adaptive) is used");
        } else if (kv.first == "line_search_method") {      // Ipopt: filter (its default) | cg-penalty | penalty
            if (kv.second == "filter") c.line_search = MPC_LS_FILTER;
            else if (kv.second == "penalty" || kv.second == "cg-penalty") { c.line_search = MPC_LS_MERIT; rep.notes.push_back("line_search_method " + kv.second + " -> MPC_LS_MERIT (backtracking on the l1 merit function with Ipopt's penalty rule)"); }
            else rep.notes.push_back("line_search_method " + kv.second + ": unknown, the library's default is used");
        } else if (kv.first == "linear_solver") rep.notes.push_back("linear_solver " + kv.second + ": the KKT systems are solved by the stage-structured sweep of the kernel");
        else rep.notes.push_back("ipopt string option " + kv.first + ": no counterpart, ignored");
    }
    for (const auto& kv : integers) {
        if (kv.first == "max_iter") c.max_iter = kv.second;
        else if (kv.first == "acceptable_iter") c.acceptable_iter = kv.second > 0 ? kv.second : -1;      // Ipopt: 0 disables the heuristic
        else rep.notes.push_back("ipopt integer option " + kv.first + ": no counterpart, ignored");
    }
    if (c.hessian_mode == MPC_HESSIAN_CONVEXIFIED && c.tol < 1e-6) {
        // first-order curvature converges linearly close to the goal: with tol 1e-8 the convexified Hessian needs more iterations and fails one cycle of 54 of a closed-loop goal
        // approach of the shipped car-like file (measured in r04; before the acceptable-level stop existed it stalled in front of the goal), the exact Hessian none;
        // the file's tol 1e-4 keeps the convexified mode
        c.hessian_mode = MPC_HESSIAN_EXACT;
        rep.notes.push_back("hessian_approximation limited-memory with tol < 1e-6: the exact Hessian is used instead of MPC_HESSIAN_CONVEXIFIED (first-order curvature does not reach such a tolerance reliably close to the goal; the KKT points are the same)");
    }

    // ---- objective (:551-639), terminal cost (:641-672), terminal constraint (:674-713)
    const std::string objective = p.param<std::string>("planning/objective/type", "minimum_time");
    if (objective == "minimum_time") c.objective = MPC_OBJ_MIN_TIME;
    else if (objective == "quadratic_form") {
        c.objective = MPC_OBJ_QUADRATIC;
        std::vector<double> qw, rw;
        p.get("planning/objective/quadratic_form/state_weights", qw);
        p.get("planning/objective/quadratic_form/control_weights", rw);
        ParamStatus st = detail::weights(qw, 3, "State weights dimension invalid. Must be either 3 x 1 or 3 x 3.", c.Q, c.Q_offdiag, rep);
        if (st != PARAMS_OK) return st;
        st = detail::weights(rw, 2, "Control weights dimension invalid. Must be either 2 x 1 or 2 x 2.", c.R, &c.R_offdiag, rep);
        if (st != PARAMS_OK) return st;
        c.integral_form = p.param("planning/objective/quadratic_form/integral_form", false) ? 1 : 0;
        bool hybrid = p.param("planning/objective/quadratic_form/hybrid_cost_minimum_time", false);
        const bool q_zero = c.Q[0] == 0 && c.Q[1] == 0 && c.Q[2] == 0 && c.Q_offdiag[0] == 0 && c.Q_offdiag[1] == 0 && c.Q_offdiag[2] == 0;
        const bool r_zero = c.R[0] == 0 && c.R[1] == 0 && c.R_offdiag == 0;
        if (hybrid && !(q_zero && !r_zero)) {
            rep.notes.push_back("Hybrid minimum time and quadratic form cost is currently only supported for non-zero control weights only. Falling back to quadratic form.");
            hybrid = false;
        }
        c.hybrid_cost_minimum_time = hybrid ? 1 : 0;            // corbo::MinTimeQuadraticControls: (n - 1) dt + the control cost (:616-618)
        c.cost_integration = integration == "trapezoidal_rule" ? MPC_COST_TRAPEZOIDAL : MPC_COST_LEFT_SUM;      // integral-form terms only (:318-333)
    } else if (objective == "minimum_time_via_points") {
        c.objective = MPC_OBJ_MIN_TIME_VIA_POINTS;
        c.via_points_ordered = p.param("planning/objective/minimum_time_via_points/via_points_ordered", false) ? 1 : 0;
        c.vp_position_weight = p.param("planning/objective/minimum_time_via_points/position_weight", 1.0);
        c.vp_orientation_weight = p.param("planning/objective/minimum_time_via_points/orientation_weight", 0.0);
        c.max_via_points = 16;
    } else return reject("Unknown objective type '" + objective + "' specified ('planning/objective/type').");
    const std::string tcost = p.param<std::string>("planning/terminal_cost/type", "none");
    if (tcost == "quadratic") {
        std::vector<double> w;
        p.get("planning/terminal_cost/quadratic/final_state_weights", w);
        ParamStatus st = detail::weights(w, 3, "Final state weights dimension invalid. Must be either 3 x 1 or 3 x 3.", c.Qf, c.Qf_offdiag, rep);
        if (st != PARAMS_OK) return st;
        c.has_Qf = 1;
    } else if (tcost != "none") return reject("Unknown terminal_cost type '" + tcost + "' specified ('planning/terminal_cost/type').");
    const std::string tcon = p.param<std::string>("planning/terminal_constraint/type", "none");
    if (tcon == "l2_ball") {
        std::vector<double> w;
        p.get("planning/terminal_constraint/l2_ball/weight_matrix", w);
        ParamStatus st = detail::weights(w, 3, "l2-ball weight_matrix dimensions invalid. Must be either 3 x 1 or 3 x 3.", c.terminal_ball_S, c.terminal_ball_S_offdiag, rep);
        if (st != PARAMS_OK) return st;
        c.terminal_ball = 1;
        c.terminal_ball_gamma = p.param("planning/terminal_constraint/l2_ball/radius", 1.0);
    } else if (tcon != "none") return reject("Unknown terminal_constraint type '" + tcon + "' specified ('planning/terminal_constraint/type').");

    // ---- collision avoidance (:715-729), footprint (src/mpc_local_planner_ros.cpp:890-1001)
    c.min_obstacle_dist = p.param("collision_avoidance/min_obstacle_dist", 0.5);
    c.enable_dynamic_obstacles = p.param("collision_avoidance/enable_dynamic_obstacles", false) ? 1 : 0;
    c.force_inclusion_dist = p.param("collision_avoidance/force_inclusion_dist", 0.5);
    c.cutoff_dist = p.param("collision_avoidance/cutoff_dist", 2.0);
    ParamStatus fst = PARAMS_OK;
    detail::footprint(p, costmap_footprint, c, rep, fst);
    if (fst != PARAMS_OK) return fst;

    // ---- facade options (:70-88)
    o.outer_ocp_iterations = p.param("controller/outer_ocp_iterations", 1);
    o.force_reinit_new_goal_dist = p.param("controller/force_reinit_new_goal_dist", 1.0);
    o.force_reinit_new_goal_angular = p.param("controller/force_reinit_new_goal_angular", 1.5707963267948966);
    o.allow_init_with_backward_motion = p.param("controller/allow_init_with_backward_motion", true);
    o.force_reinit_num_steps = p.param("controller/force_reinit_num_steps", 0);
    o.prefer_x_feedback = p.param("controller/prefer_x_feedback", false);
    o.publish_ocp_results = p.param("controller/publish_ocp_results", false);
    o.print_cpu_time = p.param("controller/print_cpu_time", false);
    return PARAMS_OK;
}

// Controller::configure(nh, ...) in one call: parameters -> mpc_config + facade options -> a configured facade.  `caps` fills in the capacities the
// reference has no parameter for (max_obstacles, max_vertices, max_obstacle_rows, max_via_points; fields left at 0 keep their defaults).
struct HandleCapacities {
    int max_obstacles = 0, max_vertices = 0, max_obstacle_rows = 0, max_via_points = 0;
    // solver options the reference's parameter set has no key for (0 = the library's defaults): multipliers kept between solves (mpc_config.dual_warm_start: the grid update of a
    // control cycle and the outer OCP iterations then start from the previous multipliers, as Ipopt does with warm_start_init_point) and the barrier starts of such solves
    int dual_warm_start = 0;
    double mu_init_warm = 0.0, mu_init_dual = 0.0;
    int stage_data = 0;      // mpc_config.stage_data: MPC_STAGE_AUTO (0) / MPC_STAGE_LDS / MPC_STAGE_GLOBAL
    int two_wave_min_batch = 0;      // mpc_config.two_wave_min_batch (0: the library's default)
};
inline ParamStatus configure_from_params(Controller& controller, const ParamSource& p, ParamReport& rep, const HandleCapacities& caps = HandleCapacities(),
                                         int device = 0, const std::vector<std::vector<double>>* costmap_footprint = nullptr, mpc_config* cfg_out = nullptr,
                                         ControllerOptions* options_out = nullptr) {
    mpc_config cfg;
    ControllerOptions opt;
    const ParamStatus st = config_from_params(p, cfg, opt, rep, costmap_footprint);
    if (st != PARAMS_OK) return st;
    if (caps.max_obstacles > 0) { cfg.max_obstacles = caps.max_obstacles; cfg.max_vertices = caps.max_vertices > 0 ? caps.max_vertices : 1; }
    if (caps.max_obstacle_rows > 0) cfg.max_obstacle_rows = caps.max_obstacle_rows;
    if (caps.max_via_points > 0) cfg.max_via_points = caps.max_via_points;
    if (caps.dual_warm_start) cfg.dual_warm_start = 1;
    if (caps.mu_init_warm > 0) cfg.mu_init_warm = caps.mu_init_warm;
    if (caps.mu_init_dual > 0) cfg.mu_init_dual = caps.mu_init_dual;
    cfg.stage_data = caps.stage_data;
    cfg.two_wave_min_batch = caps.two_wave_min_batch;
    if (cfg_out) *cfg_out = cfg;
    if (options_out) *options_out = opt;
    opt.apply(controller);                       // grid adaptation etc. BEFORE configure(): it sizes the handle for the largest grid
    if (!controller.configure(cfg, device)) { rep.error = controller.lastError(); return PARAMS_REJECTED; }
    return PARAMS_OK;
}

}  // namespace mpc_local_planner_amd

// ORACLE (test infrastructure only): the reference's plugin source src/mpc_local_planner_ros.cpp built on THIS repository's binding (include/mpc_reference_binding.hpp takes
// the place of include/mpc_local_planner/controller.h: oracle/ref_binding_shadow/ comes first on the include path; src/controller.cpp is not compiled) and on the recording
// C ABI of tests/host_harness/facade_step_host.cpp (no GPU: mpc_solve_batch shows the vertex values to the test's stand-in solver).  Same entry points as
// oracle/ref_wrap_plugin.cpp, prefix amd_plugin_; tests/test_reference_pinned.py runs both plugins through the same cycles.
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <Eigen/Core>
#include <ros/ros.h>

#define private public
#define protected public
#include <mpc_local_planner/mpc_local_planner_ros.h>
#undef private
#undef protected
#include "ref_wrap_plugin_run.hpp"

namespace mpc_local_planner {          // RViz markers (src/utils/publisher.cpp) are not compiled
Publisher::Publisher(ros::NodeHandle&, RobotDynamicsInterface::Ptr, const std::string&) {}
void Publisher::initialize(ros::NodeHandle&, RobotDynamicsInterface::Ptr, const std::string&) {}
void Publisher::publishLocalPlan(const std::vector<geometry_msgs::PoseStamped>&) const {}
void Publisher::publishLocalPlan(const corbo::TimeSeries&) const {}
void Publisher::publishGlobalPlan(const std::vector<geometry_msgs::PoseStamped>&) const {}
void Publisher::publishRobotFootprintModel(const teb_local_planner::PoseSE2&, const teb_local_planner::BaseRobotFootprintModel&, const std::string&, const std_msgs::ColorRGBA&) {}
void Publisher::publishObstacles(const teb_local_planner::ObstContainer&) const {}
void Publisher::publishViaPoints(const std::vector<teb_local_planner::PoseSE2>&, const std::string&) const {}
std_msgs::ColorRGBA Publisher::toColorMsg(float a, float r, float g, float b) { std_msgs::ColorRGBA c; c.a = a; c.r = r; c.g = g; c.b = b; return c; }
}  // namespace mpc_local_planner

#ifdef PLUGIN_ON_HIP
// third build: linked against the PRODUCT library libmpc_hip.so -- the reference's plugin driving the MI355X solver (tests/test_gpu_reference_plugin.py); no stand-in
#define PLUGIN_ENTRY(name) hip_plugin_##name
struct SolverPort {
    void attach(mpc_local_planner::MpcLocalPlannerROS&) {}
    void set(plugin_run::solve_cb) {}
    void begin_cycle() {}
    int guess_n() const { return 0; }
    int last_guess(int, double*, double*, double*) const { return 0; }
};
#else
extern "C" {
void fs_set_solver(plugin_run::solve_cb cb);                                  // tests/host_harness/facade_step_host.cpp
int fs_last_guess(int cap, double* x, double* u, double* dt, int* cold);
void fs_forget_guess();
}
#define PLUGIN_ENTRY(name) amd_plugin_##name
// the stand-in solver sits behind the recording C ABI
struct SolverPort {
    void attach(mpc_local_planner::MpcLocalPlannerROS&) {}
    void set(plugin_run::solve_cb c) { fs_set_solver(c); }
    void begin_cycle() { fs_forget_guess(); }
    int guess_n() const { double x[3 * 256], u[2 * 256], dt; int cold; return fs_last_guess(0, x, u, &dt, &cold); }
    int last_guess(int cap, double* x, double* u, double* dt) const { int cold; return fs_last_guess(cap, x, u, dt, &cold); }
};
#endif
extern "C" {
#include "ref_wrap_plugin_cycle.inc"
#ifndef PLUGIN_ON_HIP
// the binding on its own: configure() from a parameter text, optionally after setCostmapFootprint (the one call a maintainer adds for footprint_model/type costmap_2d);
// returns configure()'s result; the mpc_config it created the handle with is read back with fs_last_created_config
int amd_binding_configure(const char* params_text, int n_fp, const double* fp) {
    ros::stub_log().lines.clear();
    ros::ParamStore store;
    plugin_run::parse_params_plugin(params_text, store);
    ros::NodeHandle nh; nh.store = &store;
    mpc_local_planner::Controller c;
    if (n_fp > 0) { std::vector<geometry_msgs::Point> pts((size_t)n_fp); for (int i = 0; i < n_fp; ++i) { pts[(size_t)i].x = fp[2 * i]; pts[(size_t)i].y = fp[2 * i + 1]; } c.setCostmapFootprint(pts); }
    teb_local_planner::ObstContainer obstacles; std::vector<teb_local_planner::PoseSE2> via;
    return c.configure(nh, obstacles, std::make_shared<teb_local_planner::PointRobotFootprint>(), via) ? 1 : 0;
}
int fs_last_obstacles(int cap, int cap_v, double* rec, double* verts);
// what the binding handed to mpc_solve_batch in the last cycle (same layout as amd_plugin_container)
int amd_plugin_abi_obstacles(void*, int cap, int cap_v, double* rec, double* verts) { return fs_last_obstacles(cap, cap_v, rec, verts); }
#endif
}

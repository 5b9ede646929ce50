// ORACLE (test infrastructure only): the reference's plugin source src/mpc_local_planner_ros.cpp, compiled from where it lies under /root/reference (the whole translation
// unit, against the stand-ins of oracle/ref_stubs/ for roscpp, tf2, costmap_2d, costmap_converter, pluginlib, dynamic_reconfigure, nav_core, mbf, boost) and EXECUTED for the
// functions that prepare the solver's inputs:
//   updateObstacleContainerWithCostmap (:474-499)            lethal costmap cells -> point obstacles
//   updateObstacleContainerWithCostmapConverter (:501-541),
//   updateObstacleContainerWithCustomObstacles (:543-617)    obstacle messages -> point / circle / line / polygon obstacles (+ velocities)
//   updateViaPointsContainer (:619-635)                      via-points from the transformed plan
//   getRobotFootprintFromParamServer (:890-1001) + makeFootprintFromXMLRPC / getNumberFromXMLRPC (:1046-1095)
//   estimateLocalGoalOrientation (:807-852)
// The member functions are protected / the members private: the class definition is read with those two keywords turned into `public` (this file only; the reference's
// source file itself is compiled unchanged).  Publisher (src/utils/publisher.cpp: RViz markers) is not compiled; its members are empty functions here.
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>
#include <Eigen/Core>
#include <ros/ros.h>

#define private public
#define protected public
#include <mpc_local_planner/mpc_local_planner_ros.h>
#undef private
#undef protected
#include "ref_wrap_controller_access.hpp"
#include "ref_wrap_plugin_run.hpp"

#define PLUGIN_ENTRY(name) ref_plugin_##name
// the stand-in solver goes into the reference's Controller (its optimal-control-problem stand-in)
struct SolverPort {
    ref_access::solve_cb cb = nullptr;
    ref_access::GuessRecord guess;
    void attach(mpc_local_planner::MpcLocalPlannerROS& p) { ref_access::install_solver(p._controller, &cb, &guess); }
    void set(plugin_run::solve_cb c) { cb = c; }
    void begin_cycle() { guess.n = 0; }
    int guess_n() const { return guess.n; }
    int last_guess(int cap, double* x, double* u, double* dt) const {
        const int n = guess.n < cap ? guess.n : cap;
        for (int i = 0; i < 3 * n; ++i) x[i] = guess.x[(size_t)i];
        for (int i = 0; i < 2 * (n - 1); ++i) u[i] = guess.u[(size_t)i];
        *dt = guess.dt;
        return guess.n;
    }
};

namespace mpc_local_planner {
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

namespace {
using mpc_local_planner::MpcLocalPlannerROS;
using plugin_run::parse_params_plugin;
using plugin_run::split;
// obstacles of the container -> flat records: kind (0 point, 1 circle, 2 line, 3 polygon), n_vertices, radius, dynamic, vx, vy; vertices [cap_v][2]
int dump_obstacles(const teb_local_planner::ObstContainer& obst, int cap, int cap_v, double* rec, double* verts) {
    int n = 0;
    for (const auto& o : obst) {
        if (n >= cap) break;
        double* r = rec + 6 * n; double* v = verts + (size_t)2 * cap_v * n;
        std::vector<Eigen::Vector2d> pts; double radius = 0; int kind = 0;
        if (auto* s = dynamic_cast<const teb_local_planner::ShapeObstacle*>(o.get())) {
            pts = s->pts; radius = s->radius_;
            kind = dynamic_cast<const teb_local_planner::CircularObstacle*>(s) ? 1 : dynamic_cast<const teb_local_planner::LineObstacle*>(s) ? 2 : 3;
        } else pts.push_back(o->getCentroid());
        r[0] = kind; r[1] = (double)pts.size(); r[2] = radius; r[3] = o->isDynamic(); r[4] = o->getCentroidVelocity().x(); r[5] = o->getCentroidVelocity().y();
        for (size_t i = 0; i < pts.size() && (int)i < cap_v; ++i) { v[2 * i] = pts[i].x(); v[2 * i + 1] = pts[i].y(); }
        ++n;
    }
    return (int)obst.size();
}
costmap_converter::ObstacleArrayMsg make_msgs(int n_msgs, const int* n_points, const double* points /* [sum][3] */, const double* radius, const double* vel /* [n][2] */) {
    costmap_converter::ObstacleArrayMsg arr;
    arr.header.frame_id = "sensor";
    size_t off = 0;
    for (int i = 0; i < n_msgs; ++i) {
        costmap_converter::ObstacleMsg m;
        for (int j = 0; j < n_points[i]; ++j, ++off) { geometry_msgs::Point32 q; q.x = (float)points[3 * off]; q.y = (float)points[3 * off + 1]; q.z = (float)points[3 * off + 2]; m.polygon.points.push_back(q); }
        m.radius = radius[i]; m.velocities.twist.linear.x = vel[2 * i]; m.velocities.twist.linear.y = vel[2 * i + 1];
        arr.obstacles.push_back(m);
    }
    return arr;
}
}  // namespace

extern "C" {
// updateObstacleContainerWithCostmap: cells [size_y][size_x] row-major (getCost(mx, my) = cells[my * size_x + mx]); out_xy [cap][2]; returns the number of obstacles
int ref_plugin_costmap_obstacles(int size_x, int size_y, const unsigned char* cells, double resolution, double origin_x, double origin_y, const double* robot_pose, double behind_robot_dist,
                                 int include_costmap_obstacles, int cap, double* out_xy) {
    MpcLocalPlannerROS p;
    costmap_2d::Costmap2D cm((unsigned)size_x, (unsigned)size_y, resolution, origin_x, origin_y);
    cm.cells.assign(cells, cells + (size_t)size_x * size_y);
    p._costmap = &cm;
    p._params.include_costmap_obstacles = include_costmap_obstacles != 0;
    p._params.costmap_obstacles_behind_robot_dist = behind_robot_dist;
    p._robot_pose = teb_local_planner::PoseSE2(robot_pose[0], robot_pose[1], robot_pose[2]);
    p.updateObstacleContainerWithCostmap();
    int n = 0;
    for (const auto& o : p._obstacles) { if (n < cap) { out_xy[2 * n] = o->getCentroid().x(); out_xy[2 * n + 1] = o->getCentroid().y(); } ++n; }
    return n;
}
// updateViaPointsContainer: plan [n][3] (x, y, yaw); out [cap][3]; returns the number of via-points
int ref_plugin_via_points(int n_plan, const double* plan, double min_separation, int cap, double* out) {
    MpcLocalPlannerROS p;
    std::vector<geometry_msgs::PoseStamped> poses((size_t)n_plan);
    for (int i = 0; i < n_plan; ++i) teb_local_planner::PoseSE2(plan[3 * i], plan[3 * i + 1], plan[3 * i + 2]).toPoseMsg(poses[(size_t)i].pose);
    p._via_points.emplace_back(9.0, 9.0, 9.0);            // stale content: must be cleared
    p.updateViaPointsContainer(poses, min_separation);
    int n = 0;
    for (const auto& v : p._via_points) { if (n < cap) { out[3 * n] = v.x(); out[3 * n + 1] = v.y(); out[3 * n + 2] = v.theta(); } ++n; }
    return n;
}
// obstacle messages -> the container.  converter != 0: updateObstacleContainerWithCostmapConverter (no transform); else updateObstacleContainerWithCustomObstacles with the
// planar transform (yaw, tx, ty) answered by the tf buffer.  rec [cap][6], verts [cap][cap_v][2] as dump_obstacles; returns the number of obstacles
int ref_plugin_obstacle_messages(int converter, int n_msgs, const int* n_points, const double* points, const double* radius, const double* vel, const double* transform, int cap, int cap_v,
                                 double* rec, double* verts) {
    MpcLocalPlannerROS p;
    ros::stub_log().lines.clear();
    if (converter) {
        auto conv = std::make_shared<costmap_converter::BaseCostmapToPolygons>();
        conv->obstacles = std::make_shared<costmap_converter::ObstacleArrayMsg>(make_msgs(n_msgs, n_points, points, radius, vel));
        p._costmap_converter = conv;
        p.updateObstacleContainerWithCostmapConverter();
    } else {
        tf2_ros::Buffer tf;
        tf.answer.transform.rotation.z = std::sin(0.5 * transform[0]); tf.answer.transform.rotation.w = std::cos(0.5 * transform[0]);
        tf.answer.transform.translation.x = transform[1]; tf.answer.transform.translation.y = transform[2];
        p._tf = &tf; p._global_frame = "odom";
        p._custom_obstacle_msg = make_msgs(n_msgs, n_points, points, radius, vel);
        p.updateObstacleContainerWithCustomObstacles();
    }
    return dump_obstacles(p._obstacles, cap, cap_v, rec, verts);
}
// getRobotFootprintFromParamServer: params as text (see parse_params_plugin); costmap footprint [n_cfp][2] or n_cfp < 0 for "no costmap".  kind: 0 point, 1 circular, 2 line,
// 3 two_circles, 4 polygon; args [4]; vertices [cap][2], *n_vertices; log = the console lines ("<level>|text")
int ref_plugin_footprint(const char* params_text, int n_cfp, const double* cfp, double* args, int cap, double* vertices, int* n_vertices, char* log, int log_cap) {
    ros::stub_log().lines.clear();
    ros::ParamStore store;
    parse_params_plugin(params_text, store);
    ros::NodeHandle nh; nh.store = &store;
    costmap_2d::Costmap2DROS cm;
    for (int i = 0; i < n_cfp; ++i) { geometry_msgs::Point q; q.x = cfp[2 * i]; q.y = cfp[2 * i + 1]; cm.footprint.push_back(q); }
    auto model = MpcLocalPlannerROS::getRobotFootprintFromParamServer(nh, n_cfp >= 0 ? &cm : nullptr);
    int kind = 0; *n_vertices = 0;
    if (auto* r = dynamic_cast<teb_local_planner::RecordedFootprint*>(model.get())) {
        kind = dynamic_cast<teb_local_planner::CircularRobotFootprint*>(r) ? 1 : dynamic_cast<teb_local_planner::LineRobotFootprint*>(r) ? 2
               : dynamic_cast<teb_local_planner::TwoCirclesRobotFootprint*>(r) ? 3 : 4;
        for (size_t i = 0; i < r->args.size() && i < 4; ++i) args[i] = r->args[i];
        *n_vertices = (int)r->vertices.size();
        for (size_t i = 0; i < r->vertices.size() && (int)i < cap; ++i) { vertices[2 * i] = r->vertices[i].x(); vertices[2 * i + 1] = r->vertices[i].y(); }
    }
    std::ostringstream o;
    for (const auto& l : ros::stub_log().lines) o << l.first << "|" << l.second << "\n";
    std::strncpy(log, o.str().c_str(), (size_t)log_cap - 1); log[log_cap - 1] = 0;
    return kind;
}
// estimateLocalGoalOrientation: global plan [n][3] (x, y, yaw) in the plan frame, local goal pose (already transformed), index of the current goal in the plan, the planar
// transform plan -> global (yaw, tx, ty)
double ref_plugin_goal_orientation(int n_plan, const double* plan, const double* local_goal, int current_goal_idx, const double* transform, int moving_average_length) {
    MpcLocalPlannerROS p;
    std::vector<geometry_msgs::PoseStamped> poses((size_t)n_plan);
    for (int i = 0; i < n_plan; ++i) teb_local_planner::PoseSE2(plan[3 * i], plan[3 * i + 1], plan[3 * i + 2]).toPoseMsg(poses[(size_t)i].pose);
    geometry_msgs::PoseStamped goal; teb_local_planner::PoseSE2(local_goal[0], local_goal[1], local_goal[2]).toPoseMsg(goal.pose);
    geometry_msgs::TransformStamped t;
    t.transform.rotation.z = std::sin(0.5 * transform[0]); t.transform.rotation.w = std::cos(0.5 * transform[0]); t.transform.translation.x = transform[1]; t.transform.translation.y = transform[2];
    return p.estimateLocalGoalOrientation(poses, goal, current_goal_idx, t, moving_average_length);
}

// pruneGlobalPlan (:645-685): plan [n][3] in its own frame ("map"), the robot pose in the global frame ("odom"), the planar transform plan -> global (yaw, tx, ty);
// out [n][3] = the pruned plan, *n_out its length; returns the function's result
int ref_plugin_prune_plan(int n, const double* plan, const double* robot_pose, const double* transform, double dist_behind_robot, double* out, int* n_out) {
    MpcLocalPlannerROS p;
    tf2_ros::Buffer tf;
    tf.answer.transform.rotation.z = std::sin(0.5 * transform[0]); tf.answer.transform.rotation.w = std::cos(0.5 * transform[0]);
    tf.answer.transform.translation.x = transform[1]; tf.answer.transform.translation.y = transform[2];
    std::vector<geometry_msgs::PoseStamped> poses((size_t)n);
    for (int i = 0; i < n; ++i) { teb_local_planner::PoseSE2(plan[3 * i], plan[3 * i + 1], plan[3 * i + 2]).toPoseMsg(poses[(size_t)i].pose); poses[(size_t)i].header.frame_id = "map"; }
    geometry_msgs::PoseStamped robot; teb_local_planner::PoseSE2(robot_pose[0], robot_pose[1], robot_pose[2]).toPoseMsg(robot.pose); robot.header.frame_id = "odom";
    const bool ok = p.pruneGlobalPlan(tf, robot, poses, dist_behind_robot);
    *n_out = (int)poses.size();
    for (size_t i = 0; i < poses.size(); ++i) { teb_local_planner::PoseSE2 q(poses[i].pose); out[3 * i] = q.x(); out[3 * i + 1] = q.y(); out[3 * i + 2] = q.theta(); }
    return ok ? 1 : 0;
}
// transformGlobalPlan (:687-805): as above plus the local costmap's size (cells) and resolution and max_plan_length; out [n][3] = the transformed plan (global frame),
// *m its length, *goal_idx the index of the current goal in the global plan; returns the function's result
int ref_plugin_transform_plan(int n, const double* plan, const double* robot_pose, int size_x, int size_y, double resolution, double max_plan_length, const double* transform, double* out,
                              int* m, int* goal_idx) {
    MpcLocalPlannerROS p;
    tf2_ros::Buffer tf;
    tf.answer.transform.rotation.z = std::sin(0.5 * transform[0]); tf.answer.transform.rotation.w = std::cos(0.5 * transform[0]);
    tf.answer.transform.translation.x = transform[1]; tf.answer.transform.translation.y = transform[2];
    std::vector<geometry_msgs::PoseStamped> poses((size_t)n), result;
    for (int i = 0; i < n; ++i) { teb_local_planner::PoseSE2(plan[3 * i], plan[3 * i + 1], plan[3 * i + 2]).toPoseMsg(poses[(size_t)i].pose); poses[(size_t)i].header.frame_id = "map"; }
    geometry_msgs::PoseStamped robot; teb_local_planner::PoseSE2(robot_pose[0], robot_pose[1], robot_pose[2]).toPoseMsg(robot.pose); robot.header.frame_id = "odom";
    costmap_2d::Costmap2D cm((unsigned)size_x, (unsigned)size_y, resolution, 0.0, 0.0);
    geometry_msgs::TransformStamped t;
    *goal_idx = -1;
    const bool ok = p.transformGlobalPlan(tf, poses, robot, cm, "odom", max_plan_length, result, goal_idx, &t);
    *m = (int)result.size();
    for (size_t i = 0; i < result.size(); ++i) { teb_local_planner::PoseSE2 q(result[i].pose); out[3 * i] = q.x(); out[3 * i + 1] = q.y(); out[3 * i + 2] = q.theta(); }
    return ok ? 1 : 0;
}
#include "ref_wrap_plugin_cycle.inc"
}  // extern "C"

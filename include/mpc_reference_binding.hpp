// mpc_reference_binding.hpp -- the reference-side binding: a class with the name, the public interface and the behaviour of the reference's
// mpc_local_planner::Controller (include/mpc_local_planner/controller.h:52-146, src/controller.cpp), implemented on the C ABI of this repository
// (include/mpc_hip.h through the facade include/mpc_controller.hpp and the parameter reader include/mpc_params.hpp) instead of control_box_rst + Ipopt.
//
// HOW A MAINTAINER OF THE REFERENCE USES IT: replace the body of include/mpc_local_planner/controller.h by `#include <mpc_reference_binding.hpp>`, drop
// src/controller.cpp from the library target, link libmpc_hip.so.  The plugin source src/mpc_local_planner_ros.cpp is meant to compile UNCHANGED on it.  NOT COMPILED IN THIS
// REPOSITORY'S TEST SUITE: the build image has none of the headers below, and writing stand-ins for them is not a reference build (INTEGRATION.md says what was checked instead).
//
// This header needs the reference's build environment (roscpp, teb_local_planner, corbo-core's TimeSeries, base_local_planner, Eigen, the reference's own
// robot-model headers); it is NOT part of libmpc_hip.so and nothing in this repository's product path includes it.
//
// Differences a maintainer should know (all reported, none silent):
//   * footprint_model/type costmap_2d: teb's PolygonRobotFootprint has no accessor for its vertices, so the model object handed to configure() cannot be read
//     back; call setCostmapFootprint(costmap_ros->getRobotFootprint()) before configure(), otherwise the point model is used and a warning is logged;
//   * (optional: mpc_hip/dual_warm_start, mpc_hip/mu_init_warm, mpc_hip/mu_init_dual -- multipliers kept between solves, barrier start of warm-started solves;
//      mpc_hip/stage_data = auto | lds | global -- where a solve keeps its factorisation data, mpc_config.stage_data)
//   * the handle's capacities come from extra parameters, mpc_hip/max_obstacles (default 256), mpc_hip/max_vertices (default 8) and mpc_hip/max_obstacle_rows (clearance rows
//     per grid point, default 4; the reference has no cap -- a warning says how many rows of a cycle did not fit); when a cycle has more obstacles than max_obstacles, the nearest
//     ones to the robot are kept and a warning is logged; a polygon with more vertices than max_vertices is an error (step fails);
//   * solver/type lsq_lm, an unknown collocation method and polygon footprints with more than 16 vertices are refused at configure() (the reference accepts them).
#pragma once
#include <algorithm>
#include <cmath>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <ros/ros.h>
#include <base_local_planner/costmap_model.h>
#include <corbo-core/time_series.h>
#include <geometry_msgs/PoseStamped.h>
#include <geometry_msgs/Twist.h>
#include <mpc_local_planner/systems/kinematic_bicycle_model.h>
#include <mpc_local_planner/systems/robot_dynamics_interface.h>
#include <mpc_local_planner/systems/simple_car.h>
#include <mpc_local_planner/systems/unicycle_robot.h>
#include <mpc_local_planner_msgs/OptimalControlResult.h>
#include <mpc_local_planner_msgs/StateFeedback.h>
#include <teb_local_planner/obstacles.h>
#include <teb_local_planner/pose_se2.h>
#include <teb_local_planner/robot_footprint_model.h>
#include <tf2/utils.h>

#include "mpc_controller.hpp"
#include "mpc_params.hpp"

namespace mpc_local_planner {

namespace binding_detail {
// ros::NodeHandle as the parameter source of include/mpc_params.hpp
class RosParamSource : public mpc_local_planner_amd::ParamSource {
 public:
    explicit RosParamSource(const ros::NodeHandle& nh) : _nh(nh) {}
    bool has(const std::string& key) const override { return _nh.hasParam(key); }
    bool get(const std::string& key, bool& v) const override { return _nh.getParam(key, v); }
    bool get(const std::string& key, int& v) const override { return _nh.getParam(key, v); }
    bool get(const std::string& key, double& v) const override { return _nh.getParam(key, v); }
    bool get(const std::string& key, std::string& v) const override { return _nh.getParam(key, v); }
    bool get(const std::string& key, std::vector<double>& v) const override { return _nh.getParam(key, v); }
    bool get(const std::string& key, std::vector<bool>& v) const override { return _nh.getParam(key, v); }
    bool get(const std::string& key, std::map<std::string, double>& v) const override { return _nh.getParam(key, v); }
    bool get(const std::string& key, std::map<std::string, std::string>& v) const override { return _nh.getParam(key, v); }
    bool get(const std::string& key, std::map<std::string, int>& v) const override { return _nh.getParam(key, v); }
    // footprint_model/vertices: a list of [x, y] lists (what makeFootprintFromXMLRPC accepts, src/mpc_local_planner_ros.cpp:1046-1095)
    bool get(const std::string& key, std::vector<std::vector<double>>& v) const override {
        XmlRpc::XmlRpcValue a;
        if (!_nh.getParam(key, a) || a.getType() != XmlRpc::XmlRpcValue::TypeArray) return false;
        v.clear();
        for (int i = 0; i < a.size(); ++i) {
            XmlRpc::XmlRpcValue p = a[i];
            if (p.getType() != XmlRpc::XmlRpcValue::TypeArray) return false;
            std::vector<double> q;
            for (int j = 0; j < p.size(); ++j) {
                XmlRpc::XmlRpcValue e = p[j];
                if (e.getType() == XmlRpc::XmlRpcValue::TypeInt) q.push_back((double)(int)e);
                else if (e.getType() == XmlRpc::XmlRpcValue::TypeDouble) q.push_back((double)e);
                else return false;
            }
            v.push_back(q);
        }
        return true;
    }
 private:
    const ros::NodeHandle& _nh;
};
}  // namespace binding_detail

// what the plugin asks the inequality constraint for (validateFootprints, src/mpc_local_planner_ros.cpp:209)
class StageInequalityView {
 public:
    using Ptr = std::shared_ptr<StageInequalityView>;
    explicit StageInequalityView(double min_dist) : _min_dist(min_dist) {}
    double getMinimumDistance() const { return _min_dist; }
 private:
    double _min_dist;
};
// what the plugin does with the optimal control problem: ocp->setPreviousControlInput(u, dt) (src/mpc_local_planner_ros.cpp:384)
class OptimalControlProblemView {
 public:
    using Ptr = std::shared_ptr<OptimalControlProblemView>;
    explicit OptimalControlProblemView(mpc_local_planner_amd::Controller* c) : _c(c) {}
    void setPreviousControlInput(const Eigen::Ref<const Eigen::VectorXd>& u_prev, double dt) { const double u[2] = {u_prev[0], u_prev[1]}; _c->setPreviousControlInput(u, dt); }
 private:
    mpc_local_planner_amd::Controller* _c;
};

class Controller {
 public:
    using Ptr     = std::shared_ptr<Controller>;
    using PoseSE2 = teb_local_planner::PoseSE2;

    Controller() = default;

    // call BEFORE configure() for footprint_model/type costmap_2d (see the note at the top)
    void setCostmapFootprint(const std::vector<geometry_msgs::Point>& footprint) {
        _costmap_footprint.clear();
        for (const auto& p : footprint) _costmap_footprint.push_back({p.x, p.y});
    }

    // Controller::configure (src/controller.cpp:58-100).  obstacles and via_points are BORROWED, as in the reference: the plugin refills them before every step
    bool configure(ros::NodeHandle& nh, const teb_local_planner::ObstContainer& obstacles, teb_local_planner::RobotFootprintModelPtr /*robot_model: see the note at the top*/,
                   const std::vector<teb_local_planner::PoseSE2>& via_points) {
        namespace amd = mpc_local_planner_amd;
        _obstacles = &obstacles; _via_points = &via_points;
        binding_detail::RosParamSource src(nh);
        amd::HandleCapacities caps;
        int max_obstacles = 256, max_vertices = 8, max_obstacle_rows = 0;
        nh.param("mpc_hip/max_obstacles", max_obstacles, max_obstacles);
        nh.param("mpc_hip/max_vertices", max_vertices, max_vertices);
        nh.param("mpc_hip/max_obstacle_rows", max_obstacle_rows, max_obstacle_rows);            // clearance rows per grid point; 0: the library's default (4)
        caps.max_obstacles = max_obstacles; caps.max_vertices = max_vertices; caps.max_obstacle_rows = max_obstacle_rows; caps.max_via_points = 16;
        // multipliers kept between the solves of this controller (default off: every solve starts its multipliers as the recorded CPU runs of the test suite do)
        { bool dual = false; nh.param("mpc_hip/dual_warm_start", dual, dual); caps.dual_warm_start = dual ? 1 : 0; }
        nh.param("mpc_hip/mu_init_warm", caps.mu_init_warm, caps.mu_init_warm);
        nh.param("mpc_hip/mu_init_dual", caps.mu_init_dual, caps.mu_init_dual);
        { std::string sd = "auto"; nh.param("mpc_hip/stage_data", sd, sd); caps.stage_data = sd == "lds" ? MPC_STAGE_LDS : (sd == "global" ? MPC_STAGE_GLOBAL : MPC_STAGE_AUTO); }
        nh.param("mpc_hip/two_wave_min_batch", caps.two_wave_min_batch, caps.two_wave_min_batch);
        amd::ParamReport report;
        _amd.setInitialPlanEstimateOrientation(_initial_plan_estimate_orientation);
        std::string type; if (nh.getParam("footprint_model/type", type) && type == "costmap_2d" && _costmap_footprint.empty())
            ROS_WARN("footprint_model/type costmap_2d: setCostmapFootprint() was not called before configure(); using the point model.");
        const amd::ParamStatus st = amd::configure_from_params(_amd, src, report, caps, 0, _costmap_footprint.empty() ? nullptr : &_costmap_footprint, &_cfg, &_options);
        for (const std::string& note : report.notes) ROS_WARN_STREAM("mpc_local_planner (hip): " << note);
        if (st != amd::PARAMS_OK) { ROS_ERROR_STREAM(report.error); return false; }
        { bool one_call = true; nh.param("mpc_hip/single_launch_step", one_call, one_call); _amd.setSingleLaunchStep(one_call); }      // all outer OCP iterations of a cycle in one mpc_step_batch call
        switch (_cfg.model) {
            case MPC_MODEL_UNICYCLE: _dynamics = std::make_shared<UnicycleModel>(); break;
            case MPC_MODEL_SIMPLE_CAR: _dynamics = std::make_shared<SimpleCarModel>(_cfg.model_params[0]); break;
            case MPC_MODEL_SIMPLE_CAR_FRONT: _dynamics = std::make_shared<SimpleCarFrontWheelDrivingModel>(_cfg.model_params[0]); break;
            default: _dynamics = std::make_shared<KinematicBicycleModelVelocityInput>(_cfg.model_params[0], _cfg.model_params[1]); break;
        }
        _inequality = std::make_shared<StageInequalityView>(_cfg.min_obstacle_dist);
        _ocp_view = std::make_shared<OptimalControlProblemView>(&_amd);
        _obstacle_set.reset(new amd::ObstacleSet(caps.max_obstacles, caps.max_vertices));
        _state_feedback_subscription = nh.subscribe("state_feedback", 1, &Controller::stateFeedbackCallback, this);
        _result_publisher = nh.advertise<mpc_local_planner_msgs::OptimalControlResult>("ocp_result", 100);
        ROS_INFO("OCP initialized.");
        return true;
    }

    bool step(const PoseSE2& start, const PoseSE2& goal, const geometry_msgs::Twist& vel, double dt, ros::Time t, corbo::TimeSeries::Ptr u_seq, corbo::TimeSeries::Ptr x_seq) {
        geometry_msgs::PoseStamped first, last;                       // a plan of two poses: start and goal (src/controller.cpp:102-109)
        start.toPoseMsg(first.pose);
        goal.toPoseMsg(last.pose);
        return step(std::vector<geometry_msgs::PoseStamped>{first, last}, vel, dt, t, u_seq, x_seq);
    }

    // Controller::step (src/controller.cpp:111-179)
    bool step(const std::vector<geometry_msgs::PoseStamped>& initial_plan, const geometry_msgs::Twist& vel, double dt, ros::Time t, corbo::TimeSeries::Ptr u_seq,
              corbo::TimeSeries::Ptr x_seq) {
        namespace amd = mpc_local_planner_amd;
        if (!_dynamics) { ROS_ERROR("Controller must be configured before invoking step()."); return false; }
        if (initial_plan.size() < 2) { ROS_ERROR("Controller::step(): initial plan must contain at least two poses."); return false; }
        std::vector<amd::PoseSE2> plan(initial_plan.size());
        for (size_t i = 0; i < initial_plan.size(); ++i) {
            plan[i].x = initial_plan[i].pose.position.x; plan[i].y = initial_plan[i].pose.position.y; plan[i].theta = tf2::getYaw(initial_plan[i].pose.orientation);
        }
        // the borrowed containers of this cycle
        if (_obstacles && _cfg.max_obstacles > 0) {
            if (!fillObstacles(plan.front())) return false;
            _amd.setObstacles(_obstacle_set->view());
        }
        if (_via_points && _cfg.objective == MPC_OBJ_MIN_TIME_VIA_POINTS) {
            std::vector<double> vp;
            const size_t n_via = std::min(_via_points->size(), (size_t)_cfg.max_via_points);
            if (n_via < _via_points->size()) ROS_WARN_STREAM("mpc_local_planner (hip): " << _via_points->size() << " via-points, the handle holds " << _cfg.max_via_points);
            for (size_t i = 0; i < n_via; ++i) { vp.push_back((*_via_points)[i].x()); vp.push_back((*_via_points)[i].y()); vp.push_back((*_via_points)[i].theta()); }
            _amd.setViaPoints(vp.data(), (int)n_via);
        }
        amd::Twist tw; tw.linear_x = vel.linear.x; tw.linear_y = vel.linear.y; tw.angular_z = vel.angular.z;
        amd::TimeSeries us, xs;
        const bool ok = _amd.step(plan, tw, dt, t.toSec(), us, xs);
        if (!ok && !_amd.lastError().empty()) ROS_ERROR_STREAM("mpc_local_planner (hip): " << _amd.lastError());
        else if (!ok) ROS_WARN_STREAM("mpc_local_planner (hip): the solve did not converge (" << _amd.lastIterations() << " iterations)");
        _x_last = xs;
        if (const int dropped = _amd.lastRowsDropped())
            ROS_WARN_STREAM("mpc_local_planner (hip): " << dropped << " clearance rows did not fit (max_obstacle_rows per grid point); the reference keeps all of them");
        if (x_seq) { x_seq->clear(); for (int k = 0; k < xs.size(); ++k) { Eigen::VectorXd v(3); for (int i = 0; i < 3; ++i) v[i] = xs.at(k)[i]; x_seq->add(xs.time[(size_t)k], v); } }
        if (u_seq) { u_seq->clear(); for (int k = 0; k < us.size(); ++k) { Eigen::VectorXd v(2); for (int j = 0; j < 2; ++j) v[j] = us.at(k)[j]; u_seq->add(us.time[(size_t)k], v); } }
        if (_options.publish_ocp_results) publishOptimalControlResult(xs, us);
        ROS_INFO_STREAM_COND(_options.print_cpu_time, "Cpu time: " << _amd.lastStepTime() * 1000.0 << " ms.");
        return ok;
    }

    RobotDynamicsInterface::Ptr getRobotDynamics() { return _dynamics; }
    StageInequalityView::Ptr getInequalityConstraint() { return _inequality; }
    OptimalControlProblemView::Ptr getOptimalControlProblem() { return _ocp_view; }

    void stateFeedbackCallback(const mpc_local_planner_msgs::StateFeedback::ConstPtr& msg) {
        if (!_dynamics) return;
        if ((int)msg->state.size() != 3) { ROS_ERROR_STREAM("stateFeedbackCallback(): state feedback dimension does not match robot state dimension: " << msg->state.size() << " != 3"); return; }
        std::lock_guard<std::mutex> lock(_feedback_guard);
        const double s[3] = {msg->state[0], msg->state[1], msg->state[2]};
        _amd.stateFeedbackCallback(s, msg->header.stamp.toSec());
    }

    void setInitialPlanEstimateOrientation(bool estimate) { _initial_plan_estimate_orientation = estimate; _amd.setInitialPlanEstimateOrientation(estimate); }

    // Controller::isPoseTrajectoryFeasible (src/controller.cpp:859-917) on the trajectory of the last step, asking the caller's costmap model exactly as the reference does
    bool isPoseTrajectoryFeasible(base_local_planner::CostmapModel* costmap_model, const std::vector<geometry_msgs::Point>& footprint_spec, double inscribed_radius = 0.0,
                                  double circumscribed_radius = 0.0, double min_resolution_collision_check_angular = M_PI, int look_ahead_idx = -1) {
        if (!_dynamics) { ROS_ERROR("Controller must be configured before invoking step()."); return false; }
        const int n = _x_last.size();
        if (n < 2) return false;
        if (look_ahead_idx < 0 || look_ahead_idx >= n) look_ahead_idx = n - 1;
        for (int i = 0; i <= look_ahead_idx; ++i) {
            const double* a = _x_last.at(i);
            if (costmap_model->footprintCost(a[0], a[1], a[2], footprint_spec, inscribed_radius, circumscribed_radius) == -1) return false;
            if (i < look_ahead_idx) {
                const double* b = _x_last.at(i + 1);
                const double delta_rot = mpc_local_planner_amd::normalize_theta(b[2] - a[2]);
                const double ddx = b[0] - a[0], ddy = b[1] - a[1], dist = std::sqrt(ddx * ddx + ddy * ddy);
                if (std::abs(delta_rot) > min_resolution_collision_check_angular || dist > inscribed_radius) {
                    const int n_add = (int)(std::max(std::ceil(std::abs(delta_rot) / min_resolution_collision_check_angular), std::ceil(dist / inscribed_radius)) - 1);
                    double px = a[0], py = a[1], pth = a[2];
                    for (int s = 0; s < n_add; ++s) {
                        px += ddx / (n_add + 1.0); py += ddy / (n_add + 1.0);
                        pth = mpc_local_planner_amd::normalize_theta(pth + delta_rot / (n_add + 1.0));
                        if (costmap_model->footprintCost(px, py, pth, footprint_spec, inscribed_radius, circumscribed_radius) == -1) return false;
                    }
                }
            }
        }
        return true;
    }

    void reset() { _amd.reset(); }

 private:
    // teb obstacle container -> the arrays of struct mpc_obstacles; more obstacles than the handle holds: the nearest to the robot are kept
    bool fillObstacles(const mpc_local_planner_amd::PoseSE2& robot) {
        struct Item { const teb_local_planner::Obstacle* o; double d; };
        std::vector<Item> items;
        for (const teb_local_planner::ObstaclePtr& o : *_obstacles) {
            const Eigen::Vector2d& c = o->getCentroid();
            items.push_back({o.get(), std::hypot(c.x() - robot.x, c.y() - robot.y)});
        }
        if ((int)items.size() > _cfg.max_obstacles) {
            ROS_WARN_STREAM("mpc_local_planner (hip): " << items.size() << " obstacles, the handle holds " << _cfg.max_obstacles << " (mpc_hip/max_obstacles): keeping the nearest");
            std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.d < b.d; });
            items.resize((size_t)_cfg.max_obstacles);
        }
        _obstacle_set->clear();
        std::vector<double> xy;
        for (const Item& it : items) {
            xy.clear();
            double radius = 0.0;
            if (auto* p = dynamic_cast<const teb_local_planner::PointObstacle*>(it.o)) { xy = {p->position().x(), p->position().y()}; }
            else if (auto* c = dynamic_cast<const teb_local_planner::CircularObstacle*>(it.o)) { xy = {c->position().x(), c->position().y()}; radius = c->radius(); }
            else if (auto* l = dynamic_cast<const teb_local_planner::LineObstacle*>(it.o)) { xy = {l->start().x(), l->start().y(), l->end().x(), l->end().y()}; }
            else if (auto* g = dynamic_cast<const teb_local_planner::PolygonObstacle*>(it.o)) { for (const Eigen::Vector2d& v : g->vertices()) { xy.push_back(v.x()); xy.push_back(v.y()); } }
            else { ROS_ERROR("mpc_local_planner (hip): obstacle of an unknown kind"); return false; }
            if (!_obstacle_set->add(xy.data(), (int)xy.size() / 2, radius)) { ROS_ERROR_STREAM("mpc_local_planner (hip): an obstacle with " << xy.size() / 2 << " vertices exceeds mpc_hip/max_vertices"); return false; }
            if (it.o->isDynamic()) _obstacle_set->setLastVelocity(it.o->getCentroidVelocity().x(), it.o->getCentroidVelocity().y());
        }
        return true;
    }
    // Controller::publishOptimalControlResult (src/controller.cpp:197-221)
    void publishOptimalControlResult(const mpc_local_planner_amd::TimeSeries& xs, const mpc_local_planner_amd::TimeSeries& us) {
        mpc_local_planner_amd::OptimalControlResult r;
        _amd.optimalControlResult(xs, us, r);
        mpc_local_planner_msgs::OptimalControlResult out;
        out.header.stamp = ros::Time::now(); out.header.seq = r.seq;
        out.dim_states = r.dim_states; out.dim_controls = r.dim_controls; out.optimal_solution_found = r.optimal_solution_found; out.cpu_time = r.cpu_time;
        out.time_states = r.time_states; out.states = r.states; out.time_controls = r.time_controls; out.controls = r.controls;
        _result_publisher.publish(out);
    }

    mpc_local_planner_amd::Controller _amd;
    mpc_config _cfg{};
    mpc_local_planner_amd::ControllerOptions _options;
    std::unique_ptr<mpc_local_planner_amd::ObstacleSet> _obstacle_set;
    std::vector<std::vector<double>> _costmap_footprint;
    const teb_local_planner::ObstContainer* _obstacles = nullptr;
    const std::vector<teb_local_planner::PoseSE2>* _via_points = nullptr;
    RobotDynamicsInterface::Ptr _dynamics;
    StageInequalityView::Ptr _inequality;
    OptimalControlProblemView::Ptr _ocp_view;
    mpc_local_planner_amd::TimeSeries _x_last;
    bool _initial_plan_estimate_orientation = true;
    ros::Subscriber _state_feedback_subscription;
    ros::Publisher _result_publisher;
    std::mutex _feedback_guard;
};

}  // namespace mpc_local_planner

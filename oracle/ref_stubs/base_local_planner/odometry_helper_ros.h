// ORACLE (test infrastructure only): base_local_planner::OdometryHelperRos handing out the velocity the test put in (as a pose: x, y, yaw = vx, vy, omega)
#pragma once
#include <geometry_msgs/Pose.h>
#include <string>
namespace base_local_planner {
class OdometryHelperRos {
 public:
    void setOdomTopic(const std::string&) {}
    void getRobotVel(geometry_msgs::PoseStamped& v) { v = robot_vel; }
    geometry_msgs::PoseStamped robot_vel;
};
}  // namespace base_local_planner

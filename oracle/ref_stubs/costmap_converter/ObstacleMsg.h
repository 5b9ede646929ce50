// ORACLE (test infrastructure only): fields of costmap_converter/ObstacleMsg and ObstacleArrayMsg
#pragma once
#include <geometry_msgs/Pose.h>
#include <memory>
namespace costmap_converter {
struct ObstacleMsg { geometry_msgs::Polygon polygon; double radius = 0; geometry_msgs::Quaternion orientation; geometry_msgs::TwistWithCovariance velocities; };
struct ObstacleArrayMsg { using ConstPtr = std::shared_ptr<const ObstacleArrayMsg>; geometry_msgs::Header header; std::vector<ObstacleMsg> obstacles; };
using ObstacleArrayConstPtr = std::shared_ptr<const ObstacleArrayMsg>;
}  // namespace costmap_converter

#pragma once
#include <geometry_msgs/Pose.h>
#include <ros/ros.h>
namespace geometry_msgs {
struct TwistStamped { Header header; Twist twist; };
}  // namespace geometry_msgs

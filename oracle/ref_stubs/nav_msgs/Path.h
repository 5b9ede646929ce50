// ORACLE (test infrastructure only): fields of nav_msgs/Path
#pragma once
#include <geometry_msgs/Pose.h>
#include <memory>
namespace nav_msgs {
struct Path { using ConstPtr = std::shared_ptr<const Path>; geometry_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; };
}

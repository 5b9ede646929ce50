// ORACLE (test infrastructure only).  NOT tf2: the yaw of a unit quaternion about z (tf2::getYaw, used by the reference for plan poses)
#pragma once
#include <cmath>
#include <geometry_msgs/Pose.h>
namespace tf2 {
inline double getYaw(const geometry_msgs::Quaternion& q) { return std::atan2(2.0 * (q.w * q.z + q.x * q.y), 1.0 - 2.0 * (q.y * q.y + q.z * q.z)); }
}

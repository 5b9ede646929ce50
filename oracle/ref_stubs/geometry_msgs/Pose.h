// ORACLE (test infrastructure only): the fields of the geometry_msgs messages the reference's sources touch
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <geometry_msgs/Twist.h>
#include <ros/ros.h>
namespace geometry_msgs {
struct StubStamp { double sec = 0; double toSec() const { return sec; } };
struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; };
struct Point { double x = 0, y = 0, z = 0; };
struct Point32 { float x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { Header header; Pose pose; };
struct Polygon { std::vector<Point32> points; };
struct TwistWithCovariance { Twist twist; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { Header header; std::string child_frame_id; Transform transform; };
}  // namespace geometry_msgs

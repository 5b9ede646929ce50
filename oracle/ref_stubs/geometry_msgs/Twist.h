// ORACLE (test infrastructure only): the fields of geometry_msgs/Twist that the reference's robot models touch (getTwistFromControl).
#pragma once
namespace geometry_msgs {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Twist { Vector3 linear, angular; };
}  // namespace geometry_msgs

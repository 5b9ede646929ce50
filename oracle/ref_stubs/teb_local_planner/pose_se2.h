// ORACLE (test infrastructure only): the accessors of teb_local_planner::PoseSE2 that the reference's sources use.
#pragma once
#include <Eigen/Core>
#include <cmath>
#include <vector>
#include <geometry_msgs/Pose.h>
namespace teb_local_planner {
class PoseSE2 {
 public:
    PoseSE2() = default;
    PoseSE2(double x, double y, double theta) : _position(x, y), _theta(theta) {}
    // teb: position from the message, theta = yaw of the quaternion; toPoseMsg writes the yaw back as a rotation about z
    explicit PoseSE2(const geometry_msgs::Pose& p)
        : _position(p.position.x, p.position.y),
          _theta(std::atan2(2.0 * (p.orientation.w * p.orientation.z + p.orientation.x * p.orientation.y), 1.0 - 2.0 * (p.orientation.y * p.orientation.y + p.orientation.z * p.orientation.z))) {}
    void toPoseMsg(geometry_msgs::Pose& p) const {
        p.position.x = _position.x(); p.position.y = _position.y(); p.position.z = 0;
        p.orientation.x = 0; p.orientation.y = 0; p.orientation.z = std::sin(0.5 * _theta); p.orientation.w = std::cos(0.5 * _theta);
    }
    Eigen::Vector2d& position() { return _position; }
    const Eigen::Vector2d& position() const { return _position; }
    double& x() { return _position.x(); }
    double& y() { return _position.y(); }
    double& theta() { return _theta; }
    const double& x() const { return _position.x(); }
    const double& y() const { return _position.y(); }
    const double& theta() const { return _theta; }
    Eigen::Vector2d orientationUnitVec() const { return Eigen::Vector2d(std::cos(_theta), std::sin(_theta)); }      // teb: pose_se2.h
 private:
    Eigen::Vector2d _position;
    double _theta = 0;
};
// teb: misc.h
inline double average_angles(const std::vector<double>& angles) { double x = 0, y = 0; for (double a : angles) { x += std::cos(a); y += std::sin(a); } return (x == 0 && y == 0) ? 0 : std::atan2(y, x); }
}  // namespace teb_local_planner
namespace g2o {      // teb's pose_se2.h pulls g2o/stuff/misc.h: the angle wrap to [-pi, pi) the reference calls once (src/controller.cpp:906)
inline double normalize_theta(double theta) {
    if (theta >= -M_PI && theta < M_PI) return theta;
    double multiplier = std::floor(theta / (2 * M_PI));
    theta = theta - multiplier * 2 * M_PI;
    if (theta >= M_PI) theta -= 2 * M_PI;
    if (theta < -M_PI) theta += 2 * M_PI;
    return theta;
}
}  // namespace g2o

// ORACLE (test infrastructure only): the accessors of teb_local_planner::PoseSE2 that the reference's sources use.
#pragma once
#include <Eigen/Core>
#include <cmath>
namespace teb_local_planner {
class PoseSE2 {
 public:
    PoseSE2() = default;
    PoseSE2(double x, double y, double theta) : _position(x, y), _theta(theta) {}
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
}  // namespace teb_local_planner

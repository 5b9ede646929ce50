// ORACLE (test infrastructure only): the accessors of teb_local_planner::PoseSE2 that the reference's robot-model interface uses.
#pragma once
namespace teb_local_planner {
class PoseSE2 {
 public:
    PoseSE2() = default;
    PoseSE2(double x, double y, double theta) : _x(x), _y(y), _theta(theta) {}
    double& x() { return _x; }
    double& y() { return _y; }
    double& theta() { return _theta; }
    const double& x() const { return _x; }
    const double& y() const { return _y; }
    const double& theta() const { return _theta; }
 private:
    double _x = 0, _y = 0, _theta = 0;
};
}  // namespace teb_local_planner

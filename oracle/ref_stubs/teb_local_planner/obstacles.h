// ORACLE (test infrastructure only): teb_local_planner's obstacle INTERFACE as the reference uses it, with ONE concrete kind: a point (optionally moving
// with a constant velocity).  Its distance semantics are unambiguous (Euclidean distance to the point), which is why the association rule of
// src/optimal_control/stage_inequality_se2.cpp can be executed here; lines / polygons / the turning footprints are teb's own algorithms and stay
// restated from its documentation (oracle/se2_nlp.py::footprint_distance).
#pragma once
#include <Eigen/Core>
#include <memory>
#include <vector>
namespace teb_local_planner {
class Obstacle {
 public:
    virtual ~Obstacle() = default;
    virtual const Eigen::Vector2d& getCentroid() const = 0;
    virtual bool isDynamic() const = 0;
    virtual const Eigen::Vector2d& getCentroidVelocity() const = 0;
};
class PointObstacle : public Obstacle {
 public:
    PointObstacle(double x, double y, double vx = 0, double vy = 0, bool dynamic = false) : _pos(x, y), _vel(vx, vy), _dynamic(dynamic) {}
    const Eigen::Vector2d& getCentroid() const override { return _pos; }
    bool isDynamic() const override { return _dynamic; }
    const Eigen::Vector2d& getCentroidVelocity() const override { return _vel; }
 private:
    Eigen::Vector2d _pos, _vel;
    bool _dynamic;
};
using ObstaclePtr = std::shared_ptr<Obstacle>;          // teb: boost::shared_ptr -- same use (bool test, get(), range-for over the container)
using ObstContainer = std::vector<ObstaclePtr>;
}  // namespace teb_local_planner

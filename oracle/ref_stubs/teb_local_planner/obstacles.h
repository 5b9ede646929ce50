// ORACLE (test infrastructure only): teb_local_planner's obstacle INTERFACE as the reference uses it, with ONE concrete kind: a point (optionally moving
// with a constant velocity).  Its distance semantics are unambiguous (Euclidean distance to the point), which is why the association rule of
// src/optimal_control/stage_inequality_se2.cpp can be executed here; lines / polygons / the turning footprints are teb's own algorithms and stay
// restated from its documentation (oracle/se2_nlp.py::footprint_distance).
#pragma once
#include <Eigen/Core>
#include <geometry_msgs/Pose.h>
#include <cmath>
#include <memory>
#include <vector>
namespace teb_local_planner {
using Point2dContainer = std::vector<Eigen::Vector2d>;
// teb: distance_calculations.h
template <class P1, class P2> inline double distance_points2d(const P1& a, const P2& b) { return std::sqrt(std::pow(b.x - a.x, 2) + std::pow(b.y - a.y, 2)); }
class Obstacle {
 public:
    virtual ~Obstacle() = default;
    virtual const Eigen::Vector2d& getCentroid() const = 0;
    virtual bool isDynamic() const = 0;
    virtual const Eigen::Vector2d& getCentroidVelocity() const = 0;
    virtual void setCentroidVelocity(const Eigen::Vector2d&) {}
    // teb (obstacles.h): the planar velocity of the message; below 1 mm/s the obstacle stays static; the orientation is not applied
    void setCentroidVelocity(const geometry_msgs::TwistWithCovariance& velocity, const geometry_msgs::Quaternion&) {
        Eigen::Vector2d vel(velocity.twist.linear.x, velocity.twist.linear.y);
        if (vel.norm() < 0.001) return;
        setCentroidVelocity(vel);
    }
};
class PointObstacle : public Obstacle {
 public:
    PointObstacle(double x, double y, double vx = 0, double vy = 0, bool dynamic = false) : _pos(x, y), _vel(vx, vy), _dynamic(dynamic) {}
    explicit PointObstacle(const Eigen::Vector2d& p) : _pos(p), _vel(0, 0), _dynamic(false) {}
    void setCentroidVelocity(const Eigen::Vector2d& v) override { _vel = v; _dynamic = true; }
    using Obstacle::setCentroidVelocity;
    const Eigen::Vector2d& getCentroid() const override { return _pos; }
    const Eigen::Vector2d& position() const { return _pos; }
    bool isDynamic() const override { return _dynamic; }
    const Eigen::Vector2d& getCentroidVelocity() const override { return _vel; }
 private:
    Eigen::Vector2d _pos, _vel;
    bool _dynamic;
};
// the other kinds: RECORDS of what they were constructed from (their distance functions are teb's and are not executed here)
class ShapeObstacle : public Obstacle {
 public:
    const Eigen::Vector2d& getCentroid() const override { return _centroid; }
    bool isDynamic() const override { return _dynamic; }
    const Eigen::Vector2d& getCentroidVelocity() const override { return _vel; }
    void setCentroidVelocity(const Eigen::Vector2d& v) override { _vel = v; _dynamic = true; }
    using Obstacle::setCentroidVelocity;
    std::vector<Eigen::Vector2d> pts;
    double radius_ = 0;
 protected:
    Eigen::Vector2d _centroid, _vel;
    bool _dynamic = false;
};
class CircularObstacle : public ShapeObstacle {
 public:
    CircularObstacle(double x, double y, double r) { pts.emplace_back(x, y); radius_ = r; _centroid = pts[0]; }
    CircularObstacle(const Eigen::Vector2d& p, double r) { pts.push_back(p); radius_ = r; _centroid = p; }
    const Eigen::Vector2d& position() const { return pts[0]; }
    double radius() const { return radius_; }
};
class LineObstacle : public ShapeObstacle {
 public:
    LineObstacle(double x1, double y1, double x2, double y2) { pts.emplace_back(x1, y1); pts.emplace_back(x2, y2); _centroid = 0.5 * (pts[0] + pts[1]); }
    LineObstacle(const Eigen::Vector2d& a, const Eigen::Vector2d& b) { pts.push_back(a); pts.push_back(b); _centroid = 0.5 * (a + b); }
    const Eigen::Vector2d& start() const { return pts[0]; }
    const Eigen::Vector2d& end() const { return pts[1]; }
};
class PolygonObstacle : public ShapeObstacle {
 public:
    void pushBackVertex(double x, double y) { pts.emplace_back(x, y); }
    void pushBackVertex(const Eigen::Vector2d& v) { pts.push_back(v); }
    // teb computes the AREA centroid; this stand-in takes the mean of the vertices (only used to rank obstacles by distance in the binding)
    void finalizePolygon() { finalized = true; double x = 0, y = 0; for (const auto& q : pts) { x += q.x(); y += q.y(); } if (!pts.empty()) _centroid = Eigen::Vector2d(x / pts.size(), y / pts.size()); }
    const Point2dContainer& vertices() const { return pts; }
    bool finalized = false;
};
using ObstaclePtr = std::shared_ptr<Obstacle>;          // teb: boost::shared_ptr -- same use (bool test, get(), range-for over the container)
using ObstContainer = std::vector<ObstaclePtr>;
}  // namespace teb_local_planner

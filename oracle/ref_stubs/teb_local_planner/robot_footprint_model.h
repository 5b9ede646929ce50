// ORACLE (test infrastructure only): teb_local_planner's footprint INTERFACE with the point footprint (distance of the pose's position to the
// obstacle; for a moving obstacle the position predicted t seconds ahead with its constant velocity -- teb: PointRobotFootprint).
#pragma once
#include <teb_local_planner/obstacles.h>
#include <teb_local_planner/pose_se2.h>
namespace teb_local_planner {
class BaseRobotFootprintModel {
 public:
    virtual ~BaseRobotFootprintModel() = default;
    virtual double calculateDistance(const PoseSE2& pose, const Obstacle* obstacle) const = 0;
    virtual double estimateSpatioTemporalDistance(const PoseSE2& pose, const Obstacle* obstacle, double t) const = 0;
    virtual double getInscribedRadius() { return 0.0; }
};
// the other models: RECORDS of their constructor arguments (what getRobotFootprintFromParamServer builds, src/mpc_local_planner_ros.cpp:890-1001); their distance
// functions are teb's and are not executed here
class RecordedFootprint : public BaseRobotFootprintModel {
 public:
    double calculateDistance(const PoseSE2&, const Obstacle*) const override { return 0.0; }
    double estimateSpatioTemporalDistance(const PoseSE2&, const Obstacle*, double) const override { return 0.0; }
    std::vector<double> args;
    std::vector<Eigen::Vector2d> vertices;
};
class CircularRobotFootprint : public RecordedFootprint { public: explicit CircularRobotFootprint(double radius) { args = {radius}; } };
class LineRobotFootprint : public RecordedFootprint { public: LineRobotFootprint(const Eigen::Vector2d& a, const Eigen::Vector2d& b) { args = {a.x(), a.y(), b.x(), b.y()}; } };
class TwoCirclesRobotFootprint : public RecordedFootprint { public: TwoCirclesRobotFootprint(double fo, double fr, double ro, double rr) { args = {fo, fr, ro, rr}; } };
class PolygonRobotFootprint : public RecordedFootprint { public: explicit PolygonRobotFootprint(const Point2dContainer& v) { vertices = v; } };
class PointRobotFootprint : public BaseRobotFootprintModel {
 public:
    double calculateDistance(const PoseSE2& pose, const Obstacle* obstacle) const override { return (pose.position() - obstacle->getCentroid()).norm(); }
    double estimateSpatioTemporalDistance(const PoseSE2& pose, const Obstacle* obstacle, double t) const override {
        return (pose.position() - (obstacle->getCentroid() + t * obstacle->getCentroidVelocity())).norm();
    }
};
using RobotFootprintModelPtr = std::shared_ptr<BaseRobotFootprintModel>;
}  // namespace teb_local_planner

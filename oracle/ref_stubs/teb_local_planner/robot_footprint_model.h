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
};
class PointRobotFootprint : public BaseRobotFootprintModel {
 public:
    double calculateDistance(const PoseSE2& pose, const Obstacle* obstacle) const override { return (pose.position() - obstacle->getCentroid()).norm(); }
    double estimateSpatioTemporalDistance(const PoseSE2& pose, const Obstacle* obstacle, double t) const override {
        return (pose.position() - (obstacle->getCentroid() + t * obstacle->getCentroidVelocity())).norm();
    }
};
using RobotFootprintModelPtr = std::shared_ptr<BaseRobotFootprintModel>;
}  // namespace teb_local_planner

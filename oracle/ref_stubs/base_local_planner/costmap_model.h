// ORACLE (test infrastructure only).  NOT base_local_planner: the one call the reference makes, footprintCost(x, y, theta, footprint_spec, inscribed, circumscribed),
// answered by a function the test provides; every call is recorded.
#pragma once
#include <functional>
#include <vector>
#include <geometry_msgs/Pose.h>
namespace costmap_2d { class Costmap2D; }
namespace base_local_planner {
class CostmapModel {
 public:
    struct Call { double x, y, theta; };
    std::vector<Call> calls;
    std::function<double(double, double, double)> answer;
    CostmapModel() = default;
    explicit CostmapModel(const costmap_2d::Costmap2D&) {}
    virtual ~CostmapModel() = default;
    virtual double footprintCost(double x, double y, double theta, const std::vector<geometry_msgs::Point>&, double = 0.0, double = 0.0) {
        calls.push_back(Call{x, y, theta});
        return answer ? answer(x, y, theta) : 0.0;
    }
};
}  // namespace base_local_planner

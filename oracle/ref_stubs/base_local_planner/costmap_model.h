// ORACLE (test infrastructure only).  NOT base_local_planner: the one call the reference makes, footprintCost(x, y, theta, footprint_spec, inscribed, circumscribed),
// answered by a function the test provides; every call is recorded.
#pragma once
#include <functional>
#include <vector>
#include <geometry_msgs/Pose.h>
#include <costmap_2d/costmap_2d_ros.h>
namespace base_local_planner {
class CostmapModel {
 public:
    struct Call { double x, y, theta; };
    std::vector<Call> calls;
    std::function<double(double, double, double)> answer;
    CostmapModel() = default;
    explicit CostmapModel(const costmap_2d::Costmap2D& c) : costmap(&c) {}
    const costmap_2d::Costmap2D* costmap = nullptr;
    virtual ~CostmapModel() = default;
    virtual double footprintCost(double x, double y, double theta, const std::vector<geometry_msgs::Point>&, double = 0.0, double = 0.0) {
        calls.push_back(Call{x, y, theta});
        if (answer) return answer(x, y, theta);
        if (!costmap) return 0.0;
        // without a test-provided answer: the cost of the CENTRE cell, as base_local_planner answers for a footprint of fewer than 3 points (outside the map -3,
        // unknown -2, lethal / inscribed -1); the edge ray-tracing of larger footprints is not restated in this stand-in
        const double mx = (x - costmap->_ox) / costmap->_res, my = (y - costmap->_oy) / costmap->_res;
        if (mx < 0 || my < 0 || mx >= costmap->_sx || my >= costmap->_sy) return -3.0;
        const unsigned char c = costmap->getCost((unsigned)mx, (unsigned)my);
        return c == costmap_2d::NO_INFORMATION ? -2.0 : (c == costmap_2d::LETHAL_OBSTACLE || c == costmap_2d::INSCRIBED_INFLATED_OBSTACLE) ? -1.0 : (double)c;
    }
};
}  // namespace base_local_planner

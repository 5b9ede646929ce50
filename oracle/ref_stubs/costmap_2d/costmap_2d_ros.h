// ORACLE (test infrastructure only).  NOT costmap_2d: an array-backed Costmap2D with the accessors the reference's plugin calls -- getCost(mx, my) = cells[my * size_x + mx],
// mapToWorld = origin + (m + 0.5) * resolution (costmap_2d's published definitions) -- and a Costmap2DROS that hands out what the test put in
#pragma once
#include <algorithm>
#include <cmath>
#include <geometry_msgs/Pose.h>
#include <string>
#include <vector>
namespace costmap_2d {
static const unsigned char NO_INFORMATION = 255, LETHAL_OBSTACLE = 254, INSCRIBED_INFLATED_OBSTACLE = 253, FREE_SPACE = 0;
class Costmap2D {
 public:
    Costmap2D() = default;
    Costmap2D(unsigned int sx, unsigned int sy, double res, double ox, double oy) : _sx(sx), _sy(sy), _res(res), _ox(ox), _oy(oy), cells((size_t)sx * sy, 0) {}
    unsigned int getSizeInCellsX() const { return _sx; }
    unsigned int getSizeInCellsY() const { return _sy; }
    double getResolution() const { return _res; }
    unsigned char getCost(unsigned int mx, unsigned int my) const { return cells[(size_t)my * _sx + mx]; }
    void mapToWorld(unsigned int mx, unsigned int my, double& wx, double& wy) const { wx = _ox + (mx + 0.5) * _res; wy = _oy + (my + 0.5) * _res; }
    unsigned int _sx = 0, _sy = 0;
    double _res = 1, _ox = 0, _oy = 0;
    std::vector<unsigned char> cells;
};
class Costmap2DROS {
 public:
    Costmap2D* getCostmap() { return &costmap; }
    std::string getGlobalFrameID() const { return "odom"; }
    std::string getBaseFrameID() const { return "base_link"; }
    bool getRobotPose(geometry_msgs::PoseStamped& p) const { p = robot_pose; return true; }
    std::vector<geometry_msgs::Point> getRobotFootprint() const { return footprint; }
    geometry_msgs::Polygon getRobotFootprintPolygon() const { geometry_msgs::Polygon g; for (const auto& q : footprint) { geometry_msgs::Point32 r; r.x = (float)q.x; r.y = (float)q.y; g.points.push_back(r); } return g; }
    Costmap2D costmap;
    geometry_msgs::PoseStamped robot_pose;
    std::vector<geometry_msgs::Point> footprint;
};
// costmap_2d/footprint.cpp: the smallest distance from the robot centre to an edge of the footprint polygon and the largest to a vertex (inscribed / circumscribed radius)
inline void calculateMinAndMaxDistances(const std::vector<geometry_msgs::Point>& fp, double& min_dist, double& max_dist) {
    min_dist = 1e300; max_dist = 0;
    if (fp.size() <= 2) { min_dist = 0; return; }
    auto to_segment = [](double ax, double ay, double bx, double by) {
        const double dx = bx - ax, dy = by - ay, l2 = dx * dx + dy * dy;
        double t = l2 > 0 ? -(ax * dx + ay * dy) / l2 : 0.0;
        t = t < 0 ? 0 : (t > 1 ? 1 : t);
        return std::sqrt((ax + t * dx) * (ax + t * dx) + (ay + t * dy) * (ay + t * dy));
    };
    for (size_t i = 0; i < fp.size(); ++i) {
        const auto& a = fp[i]; const auto& b = fp[(i + 1) % fp.size()];
        min_dist = std::min(min_dist, std::min(std::sqrt(a.x * a.x + a.y * a.y), to_segment(a.x, a.y, b.x, b.y)));
        max_dist = std::max(max_dist, std::sqrt(a.x * a.x + a.y * a.y));
    }
}
}  // namespace costmap_2d

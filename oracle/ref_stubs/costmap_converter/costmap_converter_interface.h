// ORACLE (test infrastructure only): the costmap_converter plugin interface as the reference drives it; getObstacles() hands out what the test put in.  pluginlib's loader
// never finds a plugin here (createInstance throws, as for a plugin name that does not exist)
#pragma once
#include <costmap_2d/costmap_2d_ros.h>
#include <costmap_converter/ObstacleMsg.h>
#include <ros/ros.h>
#include <stdexcept>
namespace costmap_converter {
class BaseCostmapToPolygons {
 public:
    virtual ~BaseCostmapToPolygons() = default;
    ObstacleArrayConstPtr getObstacles() { return obstacles; }
    void setOdomTopic(const std::string&) {}
    void initialize(ros::NodeHandle) {}
    void setCostmap2D(costmap_2d::Costmap2D*) {}
    void startWorker(ros::Rate, costmap_2d::Costmap2D*, bool) {}
    ObstacleArrayConstPtr obstacles;
};
}  // namespace costmap_converter
namespace pluginlib {
class PluginlibException : public std::runtime_error { public: using std::runtime_error::runtime_error; };
template <class T> class ClassLoader {
 public:
    ClassLoader(const std::string&, const std::string&) {}
    boost::shared_ptr<T> createInstance(const std::string& name) { throw PluginlibException("no plugin named " + name); }
    std::string getName(const std::string& name) { return name; }
};
}  // namespace pluginlib

// ORACLE (test infrastructure only): tf2_ros::Buffer answering every lookup with the transform the test put in (identity by default)
#pragma once
#include <ros/ros.h>
#include <tf2/utils.h>
namespace tf2_ros {
class Buffer {
 public:
    geometry_msgs::TransformStamped lookupTransform(const std::string& target, const std::string&, const ros::Time&) const { auto t = answer; t.header.frame_id = target; return t; }
    geometry_msgs::TransformStamped lookupTransform(const std::string& target, const ros::Time&, const std::string&, const ros::Time&, const std::string&, const ros::Duration&) const {
        auto t = answer; t.header.frame_id = target; return t;
    }
    void transform(const geometry_msgs::PoseStamped& in, geometry_msgs::PoseStamped& out, const std::string& target) const { auto t = answer; t.header.frame_id = target; tf2::doTransform(in, out, t); }
    geometry_msgs::TransformStamped answer;
};
}  // namespace tf2_ros

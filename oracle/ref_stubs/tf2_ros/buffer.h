// ORACLE (test infrastructure only): tf2_ros::Buffer with ONE planar transform the test puts in: `answer` maps the plan / sensor frame into `global_frame`; a lookup whose
// target is `global_frame` gets it, any other lookup its inverse (identity by default)
#pragma once
#include <ros/ros.h>
#include <tf2/utils.h>
namespace tf2_ros {
class Buffer {
 public:
    geometry_msgs::TransformStamped lookupTransform(const std::string& target, const std::string&, const ros::Time&) const { return towards(target); }
    geometry_msgs::TransformStamped lookupTransform(const std::string& target, const ros::Time&, const std::string&, const ros::Time&, const std::string&, const ros::Duration&) const {
        return towards(target);
    }
    void transform(const geometry_msgs::PoseStamped& in, geometry_msgs::PoseStamped& out, const std::string& target) const { tf2::doTransform(in, out, towards(target)); }
    geometry_msgs::TransformStamped answer;
    std::string global_frame = "odom";
 private:
    geometry_msgs::TransformStamped towards(const std::string& target) const {
        geometry_msgs::TransformStamped t = answer;
        if (target != global_frame) {
            const double yaw = tf2::getYaw(answer.transform.rotation), c = std::cos(yaw), s = std::sin(yaw);
            const double tx = answer.transform.translation.x, ty = answer.transform.translation.y;
            t.transform.rotation.z = std::sin(-0.5 * yaw); t.transform.rotation.w = std::cos(-0.5 * yaw);
            t.transform.translation.x = -(c * tx + s * ty); t.transform.translation.y = -(-s * tx + c * ty);
        }
        t.header.frame_id = target;
        return t;
    }
};
}  // namespace tf2_ros

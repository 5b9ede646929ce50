// ORACLE (test infrastructure only).  NOT tf2 / tf: yaw of a quaternion, quaternion product, a rigid transform applied to a pose (rotation about z and translation: the
// planar case the plugin deals with), the exception types it catches
#pragma once
#include <cmath>
#include <stdexcept>
#include <geometry_msgs/Pose.h>
#include <Eigen/Core>
namespace tf2 {
class Quaternion {
 public:
    Quaternion() = default;
    Quaternion(double x, double y, double z, double w) : _x(x), _y(y), _z(z), _w(w) {}
    void setRPY(double, double, double yaw) { _x = 0; _y = 0; _z = std::sin(0.5 * yaw); _w = std::cos(0.5 * yaw); }
    Quaternion operator*(const Quaternion& q) const {
        return Quaternion(_w * q._x + _x * q._w + _y * q._z - _z * q._y, _w * q._y + _y * q._w + _z * q._x - _x * q._z, _w * q._z + _z * q._w + _x * q._y - _y * q._x,
                          _w * q._w - _x * q._x - _y * q._y - _z * q._z);
    }
    double _x = 0, _y = 0, _z = 0, _w = 1;
};
inline double getYaw(const geometry_msgs::Quaternion& q) { return std::atan2(2.0 * (q.w * q.z + q.x * q.y), 1.0 - 2.0 * (q.y * q.y + q.z * q.z)); }
inline double getYaw(const Quaternion& q) { return std::atan2(2.0 * (q._w * q._z + q._x * q._y), 1.0 - 2.0 * (q._y * q._y + q._z * q._z)); }
inline void convert(const Quaternion& a, geometry_msgs::Quaternion& b) { b.x = a._x; b.y = a._y; b.z = a._z; b.w = a._w; }
inline void convert(const geometry_msgs::Quaternion& a, Quaternion& b) { b = Quaternion(a.x, a.y, a.z, a.w); }
inline void doTransform(const geometry_msgs::PoseStamped& in, geometry_msgs::PoseStamped& out, const geometry_msgs::TransformStamped& t) {
    const double yaw = getYaw(t.transform.rotation), c = std::cos(yaw), s = std::sin(yaw);
    out.header.frame_id = t.header.frame_id;
    out.pose.position.x = t.transform.translation.x + c * in.pose.position.x - s * in.pose.position.y;
    out.pose.position.y = t.transform.translation.y + s * in.pose.position.x + c * in.pose.position.y;
    out.pose.position.z = in.pose.position.z;
    Quaternion r, q; convert(t.transform.rotation, r); convert(in.pose.orientation, q);
    convert(r * q, out.pose.orientation);
}
inline Eigen::Affine3d transformToEigen(const geometry_msgs::TransformStamped& t) {
    Eigen::Affine3d a; a.setIdentity();
    const double yaw = getYaw(t.transform.rotation);
    a.c = std::cos(yaw); a.s = std::sin(yaw); a.tx = t.transform.translation.x; a.ty = t.transform.translation.y; a.tz = t.transform.translation.z;
    return a;
}
}  // namespace tf2
namespace tf {
class TransformException : public std::runtime_error { public: using std::runtime_error::runtime_error; };
class LookupException : public TransformException { public: using TransformException::TransformException; };
class ConnectivityException : public TransformException { public: using TransformException::TransformException; };
class ExtrapolationException : public TransformException { public: using TransformException::TransformException; };
struct Vector3 { double x = 0, y = 0, z = 0; double getX() const { return x; } double getY() const { return y; } };
using Quaternion = tf2::Quaternion;
struct Pose { Vector3 origin; Quaternion rotation; const Vector3& getOrigin() const { return origin; } Quaternion getRotation() const { return rotation; } };
inline double getYaw(const Quaternion& q) { return tf2::getYaw(q); }
}  // namespace tf

// ORACLE (test infrastructure only): dynamic_reconfigure::Server as an object that remembers its callback
#pragma once
#include <functional>
#include <ros/ros.h>
namespace dynamic_reconfigure {
template <class Config> class Server {
 public:
    using CallbackType = std::function<void(Config&, uint32_t)>;
    explicit Server(const ros::NodeHandle&) {}
    void setCallback(const CallbackType& cb) { callback = cb; }
    CallbackType callback;
};
}  // namespace dynamic_reconfigure

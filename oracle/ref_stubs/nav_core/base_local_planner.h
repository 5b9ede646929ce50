// ORACLE (test infrastructure only): nav_core::BaseLocalPlanner as a base to derive from
#pragma once
#include <geometry_msgs/TwistStamped.h>
namespace nav_core { class BaseLocalPlanner { public: virtual ~BaseLocalPlanner() = default; }; }

#pragma once
namespace nav_msgs { struct Odometry {}; }

#pragma once
#include <geometry_msgs/Pose.h>

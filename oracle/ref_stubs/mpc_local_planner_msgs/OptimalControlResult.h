// ORACLE (test infrastructure only): the fields of mpc_local_planner_msgs/OptimalControlResult (the reference's msg/OptimalControlResult.msg:1-12) as a message
// generator lays them out in C++
#pragma once
#include <ros/ros.h>
#include <cstdint>
namespace mpc_local_planner_msgs {
struct Header { ros::Time stamp; uint32_t seq = 0; };
struct OptimalControlResult {
    Header header;
    int64_t dim_states = 0, dim_controls = 0;
    std::vector<double> time_states, states, time_controls, controls;
    bool optimal_solution_found = false;
    double cpu_time = 0;
};
}  // namespace mpc_local_planner_msgs

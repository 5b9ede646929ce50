// ORACLE (test infrastructure only): the fields of mpc_local_planner_msgs/StateFeedback (msg/StateFeedback.msg)
#pragma once
#include <mpc_local_planner_msgs/OptimalControlResult.h>
namespace mpc_local_planner_msgs {
struct StateFeedback {
    using ConstPtr = std::shared_ptr<const StateFeedback>;
    Header header;
    std::vector<double> state;
};
}  // namespace mpc_local_planner_msgs

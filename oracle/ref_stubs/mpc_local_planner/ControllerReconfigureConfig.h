// ORACLE (test infrastructure only): the fields of the generated dynamic_reconfigure config (cfg/ControllerReconfigure.cfg) the plugin copies
#pragma once
namespace mpc_local_planner { struct ControllerReconfigureConfig { double xy_goal_tolerance = 0.2, yaw_goal_tolerance = 0.1; bool global_plan_overwrite_orientation = true;
    double global_plan_prune_distance = 1.0, max_global_plan_lookahead_dist = 1.5, global_plan_viapoint_sep = -1; }; }

#pragma once
namespace mpc_local_planner { struct CollisionReconfigureConfig { bool include_costmap_obstacles = true; double costmap_obstacles_behind_robot_dist = 1.5,
    collision_check_min_resolution_angular = 3.141592653589793; int collision_check_no_poses = -1; }; }

"""ORACLE (test infrastructure, not the product): costmap -> point obstacles, the step in front of the solve.

Restates MpcLocalPlannerROS::updateObstacleContainerWithCostmap (src/mpc_local_planner_ros.cpp:474-499).  Third-party pieces
(costmap_2d, ROS navigation -- absent from /root/reference, restated from its published interface): getCost(mx, my) =
costmap[my * size_x + mx]; LETHAL_OBSTACLE = 254; mapToWorld(mx, my): w = origin + (m + 0.5) * resolution.
Parity UNPINNED: restated from the cited lines (the plugin source needs ROS / costmap_2d / teb headers, which the image lacks); getCost / mapToWorld are costmap_2d's published
definitions (third-party).
"""
import math

import numpy as np

LETHAL_OBSTACLE = 254


def costmap_to_obstacles(cost, resolution, origin, robot_pose, behind_robot_dist=1.5):
    """cost: (size_y, size_x) uint8.  Returns the (P, 2) point obstacles in the reference's container order: for i in 0..size_x-2
    (outer), for j in 0..size_y-2 (inner) -- the last row and column are not visited (:481-483)."""
    cost = np.asarray(cost, np.uint8)
    size_y, size_x = cost.shape
    ox, oy = math.cos(robot_pose[2]), math.sin(robot_pose[2])       # PoseSE2::orientationUnitVec
    out = []
    ii, jj = np.nonzero(cost[:size_y - 1, :size_x - 1].T == LETHAL_OBSTACLE)      # transposed: i outer, j inner
    for i, j in zip(ii.tolist(), jj.tolist()):
        wx = origin[0] + (i + 0.5) * resolution
        wy = origin[1] + (j + 0.5) * resolution
        dx, dy = wx - robot_pose[0], wy - robot_pose[1]
        if dx * ox + dy * oy < 0 and math.sqrt(dx * dx + dy * dy) > behind_robot_dist:     # :491-492
            continue
        out.append((wx, wy))
    return np.array(out, float).reshape(-1, 2)

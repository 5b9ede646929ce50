"""What the reference's plugin prepares for a control cycle BEFORE Controller::step -- the data formats on the caller's side of the path
(src/mpc_local_planner_ros.cpp), for a binding that feeds BatchSolver / the C ABI instead of corbo:

    via_points_from_plan            updateViaPointsContainer (:619-635)
    obstacles_from_messages         updateObstacleContainerWithCostmapConverter (:501-541) / updateObstacleContainerWithCustomObstacles (:543-617)
    pack_obstacles                  the obstacle records -> the arrays of struct mpc_obstacles (include/mpc_hip.h) for one instance
    estimate_local_goal_orientation estimateLocalGoalOrientation (:807-852)
    prune_global_plan / transform_global_plan   pruneGlobalPlan (:645-685) / transformGlobalPlan (:687-805)

(the costmap -> point obstacle scan, :474-499, is the device kernel mpc_costmap_to_obstacles).  Each function is held to the reference's own source,
compiled and executed by the test suite (tests/test_reference_pinned.py)."""
from __future__ import annotations

import math
from typing import Iterable, Sequence

import numpy as np

DYNAMIC_VELOCITY_THRESHOLD = 0.001      # teb_local_planner Obstacle::setCentroidVelocity: below 1 mm/s an obstacle stays static


def via_points_from_plan(plan, min_separation: float) -> np.ndarray:
    """plan (n, 3) poses of the transformed global plan -> (P, 3) via-points: walking along the plan, a pose becomes a via-point when it is at least
    min_separation away (Euclidean, x/y) from the previously inserted one -- the first pose counts as inserted but is not a via-point itself.
    min_separation <= 0: none."""
    plan = np.asarray(plan, float).reshape(-1, 3)
    if min_separation <= 0:
        return np.zeros((0, 3))
    out, prev = [], 0
    for i in range(1, plan.shape[0]):
        if math.sqrt((plan[i, 0] - plan[prev, 0]) ** 2 + (plan[i, 1] - plan[prev, 1]) ** 2) < min_separation:
            continue
        out.append(plan[i])
        prev = i
    return np.array(out, float).reshape(-1, 3)


def _to_global(transform, x, y, yaw):
    c, s = math.cos(transform[0]), math.sin(transform[0])
    return transform[1] + c * x - s * y, transform[2] + s * x + c * y, _yaw_compose(transform[0], yaw)


def _to_plan_frame(transform, x, y):
    c, s = math.cos(transform[0]), math.sin(transform[0])
    dx, dy = x - transform[1], y - transform[2]
    return c * dx + s * dy, -s * dx + c * dy


def prune_global_plan(plan, robot_pose, transform: Sequence[float] = (0.0, 0.0, 0.0), dist_behind_robot: float = 1.0):
    """pruneGlobalPlan (:645-685): cuts off the part of the global plan the robot has passed -- everything before the FIRST pose closer than dist_behind_robot to the robot.
    plan (n, 3) in its own frame, robot_pose in the planning frame, transform = (yaw, tx, ty) plan frame -> planning frame.  Returns (True, plan'); with no pose that close the plan stays as it is -- and the
    result is still True: as coded (:661-675, `erase_end` starts at begin(), not end()) the reference never reports the failure its `return false` was written for."""
    plan = np.asarray(plan, float).reshape(-1, 3)
    if plan.shape[0] == 0:
        return True, plan
    rx, ry = _to_plan_frame(transform, float(robot_pose[0]), float(robot_pose[1]))
    for i in range(plan.shape[0]):
        if (rx - plan[i, 0]) ** 2 + (ry - plan[i, 1]) ** 2 < dist_behind_robot * dist_behind_robot:
            return True, plan[i:].copy()
    return True, plan.copy()


def transform_global_plan(plan, robot_pose, costmap_size_x: int, costmap_size_y: int, resolution: float, max_plan_length: float, transform: Sequence[float] = (0.0, 0.0, 0.0)):
    """transformGlobalPlan (:687-805): the part of the global plan that is handed to the controller -- from the plan pose closest to the robot (searched only until the plan
    leaves 85 % of the local costmap's half size) onwards, while the poses stay inside that radius and the length along the plan stays within max_plan_length (<= 0: no limit);
    poses moved into the planning frame.  An empty selection yields the global goal alone.  Returns (poses (m, 3), index of the last selected pose in the global plan)."""
    plan = np.asarray(plan, float).reshape(-1, 3)
    n = plan.shape[0]
    if n == 0:
        raise ValueError("Received plan with zero length")
    rx, ry = _to_plan_frame(transform, float(robot_pose[0]), float(robot_pose[1]))
    thr = 0.85 * max(costmap_size_x * resolution / 2.0, costmap_size_y * resolution / 2.0)
    sq_thr = thr * thr
    # the loops of the reference, evaluated with array operations (a fleet calls this once per robot and cycle)
    d2 = (rx - plan[:, 0]) ** 2 + (ry - plan[:, 1]) ** 2
    outside = np.nonzero(d2 > sq_thr)[0]
    stop = int(outside[0]) if outside.size else n                       # the search ends at the first pose beyond the radius (:724)
    if stop == 0:
        i, sq_dist = 0, 1e10
    else:
        i = int(np.argmin(d2[:stop]))                                   # the first of equal minima, like the strict `<` of the loop
        sq_dist = float(d2[i])
    if sq_dist > sq_thr:
        return _poses_to_global(transform, plan[-1:]), n - 1            # empty selection: the global goal alone (:766-774)
    # pose j (j >= i) is taken while the pose BEFORE it was inside the radius and the length accumulated BEFORE it is within the limit; the length counts the segment
    # (j-1, j) of every pose taken, the one in front of pose i included when i > 0 (:760-761)
    prev_inside = np.concatenate([[True], d2[i:n - 1] <= sq_thr])
    if max_plan_length > 0:
        seg = np.zeros(n - i)
        idx = np.arange(i, n)
        has_prev = idx > 0
        seg[has_prev] = np.sqrt((plan[idx[has_prev], 0] - plan[idx[has_prev] - 1, 0]) ** 2 + (plan[idx[has_prev], 1] - plan[idx[has_prev] - 1, 1]) ** 2)
        length_before = np.concatenate([[0.0], np.cumsum(seg)[:-1]])
        take = prev_inside & (length_before <= max_plan_length)
    else:
        take = prev_inside
    bad = np.nonzero(~take)[0]
    count = int(bad[0]) if bad.size else n - i
    return _poses_to_global(transform, plan[i:i + count]), i + count - 1


def _poses_to_global(transform, poses):
    poses = np.asarray(poses, float).reshape(-1, 3)
    c, s = math.cos(transform[0]), math.sin(transform[0])
    out = np.empty_like(poses)
    out[:, 0] = transform[1] + c * poses[:, 0] - s * poses[:, 1]
    out[:, 1] = transform[2] + s * poses[:, 0] + c * poses[:, 1]
    # heading through quaternions, as tf2::doTransform composes the rotations (getYaw(rotation * orientation))
    az, aw = math.sin(0.5 * transform[0]), math.cos(0.5 * transform[0])
    bz, bw = np.sin(0.5 * poses[:, 2]), np.cos(0.5 * poses[:, 2])
    z, w = aw * bz + az * bw, aw * bw - az * bz
    out[:, 2] = np.arctan2(2.0 * w * z, 1.0 - 2.0 * z * z)
    return out


def obstacles_from_messages(msgs: Iterable[dict], converter: bool = True, transform: Sequence[float] = (0.0, 0.0, 0.0)):
    """costmap_converter/ObstacleMsg-like records {points: [(x, y[, z]), ...], radius: r, velocity: (vx, vy)} -> obstacle records
    (vertices (k, 2), radius, velocity (2,)) in container order.  1 point + radius > 0: circle; 1 point: point; 2: line; more: polygon.
    converter=True: the messages of a costmap_converter plugin, already in the planning frame.  converter=False: custom obstacles, moved into the planning
    frame by the planar transform (yaw, tx, ty); a message without points is skipped (the reference warns).
    A velocity below 1 mm/s leaves the obstacle static.  As in the reference, the velocity of a message goes to the LAST obstacle in the container: in the
    converter path a message without points therefore re-labels the obstacle before it."""
    c, s = math.cos(transform[0]), math.sin(transform[0])
    out = []

    def moved(q):
        if converter:
            return (float(np.float32(q[0])), float(np.float32(q[1])))            # geometry_msgs/Point32: single precision on the wire
        x, y = float(np.float32(q[0])), float(np.float32(q[1]))
        return (transform[1] + c * x - s * y, transform[2] + s * x + c * y)
    for m in msgs:
        pts = list(m.get("points", ()))
        radius = float(m.get("radius", 0.0))
        if len(pts) == 1 and radius > 0:
            out.append([np.array([moved(pts[0])]), radius, np.zeros(2)])
        elif len(pts) == 1:
            out.append([np.array([moved(pts[0])]), 0.0, np.zeros(2)])
        elif len(pts) == 2:
            out.append([np.array([moved(pts[0]), moved(pts[1])]), 0.0, np.zeros(2)])
        elif len(pts) == 0:
            if not converter:
                continue                                                        # :592-596 "Invalid custom obstacle received ... Skipping"
        else:
            out.append([np.array([moved(q) for q in pts]), 0.0, np.zeros(2)])
        v = np.asarray(m.get("velocity", (0.0, 0.0)), float)
        if out and math.sqrt(v[0] * v[0] + v[1] * v[1]) >= DYNAMIC_VELOCITY_THRESHOLD:
            out[-1][2] = v.copy()
    return [(a, r, v) for a, r, v in out]


def pack_obstacles(records, max_obstacles: int, max_vertices: int):
    """obstacle records of ONE instance -> (n_obstacles, n_vertices (O,), vertices (O, V, 2), radius (O,), velocity (O, 2)); raises when the handle's capacity is
    exceeded (never drops silently)"""
    if len(records) > max_obstacles:
        raise ValueError(f"{len(records)} obstacles exceed max_obstacles = {max_obstacles}")
    nv = np.zeros(max_obstacles, np.int32); vv = np.zeros((max_obstacles, max_vertices, 2)); rr = np.zeros(max_obstacles); vel = np.zeros((max_obstacles, 2))
    for i, (verts, radius, v) in enumerate(records):
        k = len(verts)
        if k > max_vertices:
            raise ValueError(f"obstacle {i} has {k} vertices, max_vertices = {max_vertices}")
        nv[i] = k; vv[i, :k] = verts; rr[i] = radius; vel[i] = v
    return len(records), nv, vv, rr, vel


def _yaw_compose(a: float, b: float) -> float:
    """yaw of the product of two rotations about z given by their yaw angles, through quaternions as tf2 does (getYaw(rotation * orientation))"""
    az, aw, bz, bw = math.sin(0.5 * a), math.cos(0.5 * a), math.sin(0.5 * b), math.cos(0.5 * b)
    z, w = aw * bz + az * bw, aw * bw - az * bz
    return math.atan2(2.0 * w * z, 1.0 - 2.0 * z * z)


def estimate_local_goal_orientation(global_plan, local_goal, current_goal_idx: int, transform: Sequence[float] = (0.0, 0.0, 0.0), moving_average_length: int = 3) -> float:
    """heading of the local goal when controller/global_plan_overwrite_orientation is set: near the end of the global plan the goal's own heading, otherwise the
    circular mean of the directions between up to moving_average_length successive plan poses beyond the local goal (plan poses moved into the planning frame by
    the planar transform (yaw, tx, ty); the local goal is given in the planning frame)."""
    plan = np.asarray(global_plan, float).reshape(-1, 3)
    n = plan.shape[0]
    if current_goal_idx > n - moving_average_length - 2:
        if current_goal_idx >= n - 1:
            return float(local_goal[2])
        return _yaw_compose(transform[0], plan[-1, 2])
    moving_average_length = min(moving_average_length, n - current_goal_idx - 1)
    c, s = math.cos(transform[0]), math.sin(transform[0])
    prev = (float(local_goal[0]), float(local_goal[1]))
    sx = sy = 0.0
    end = current_goal_idx + moving_average_length
    for i in range(current_goal_idx, end):
        q = plan[i + 1]
        nxt = (transform[1] + c * q[0] - s * q[1], transform[2] + s * q[0] + c * q[1])
        a = math.atan2(nxt[1] - prev[1], nxt[0] - prev[0])
        sx += math.cos(a); sy += math.sin(a)
        if i < end - 1:
            prev = nxt
    return 0.0 if (sx == 0 and sy == 0) else math.atan2(sy, sx)

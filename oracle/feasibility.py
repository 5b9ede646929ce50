"""ORACLE (test infrastructure, not the product): post-solve costmap feasibility check of the planned pose trajectory.

Restates Controller::isPoseTrajectoryFeasible (src/controller.cpp:859-917): every grid point up to look_ahead_idx -- plus interpolated poses
where two neighbours are farther apart than the inscribed radius or turn by more than min_resolution_collision_check_angular (:889-912,
the intermediate pose is ACCUMULATED step by step, :903-904) -- is tested with base_local_planner::CostmapModel::footprintCost; the
trajectory is infeasible iff one of the calls returns -1.

Third-party pieces (ROS navigation, absent from /root/reference; PINNED CONVENTION = navigation 1.17 / noetic,
base_local_planner/src/costmap_model.cpp + line_iterator.h + costmap_2d::Costmap2D::worldToMap), restated from their published source:
  footprintCost(x, y, theta, spec, r_in, r_circ): spec rotated by (cos, sin) and shifted to the pose (world_model.h), then
    centre cell outside the map                              -> -3
    fewer than 3 footprint points: cost of the centre cell:     NO_INFORMATION (255) -> -2, LETHAL (254) or INSCRIBED (253) -> -1
    otherwise every edge i -> i+1 and last -> first, in this order: an endpoint outside the map -> -3; the cells of the ray-traced line
    (LineIterator = Bresenham, x0,y0 .. x1,y1) in order: NO_INFORMATION -> -2, LETHAL -> -1; the FIRST negative value is returned.
  worldToMap(wx, wy): false if wx < origin_x or wy < origin_y; m = (int)((w - origin) / resolution); false unless mx < size_x and my < size_y.
(Older navigation releases return -1 for all three cases; with that convention unknown / outside cells would also make a trajectory
infeasible.  The reference compares with -1 only.)  Parity UNPINNED: restated from src/controller.cpp:859-917 (that file needs corbo / ROS headers,
which the image lacks); footprintCost itself is third-party (base_local_planner) and restated from its published source.
"""
import math

import numpy as np

NO_INFORMATION, LETHAL_OBSTACLE, INSCRIBED = 255, 254, 253


def _normalize_theta(th):
    """g2o::normalize_theta / math_utils.h:81-91 (used at :891, :904)"""
    if -math.pi <= th < math.pi:
        return th
    m = th - math.floor(th / (2 * math.pi)) * 2 * math.pi
    if m >= math.pi:
        m -= 2 * math.pi
    if m < -math.pi:
        m += 2 * math.pi
    return m


def world_to_map(cost, resolution, origin, wx, wy):
    size_y, size_x = cost.shape
    if wx < origin[0] or wy < origin[1]:
        return None
    mx = int((wx - origin[0]) / resolution)
    my = int((wy - origin[1]) / resolution)
    if mx < size_x and my < size_y:
        return mx, my
    return None


def line_cells(x0, y0, x1, y1):
    """base_local_planner::LineIterator"""
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    xinc1 = xinc2 = 1 if x1 >= x0 else -1
    yinc1 = yinc2 = 1 if y1 >= y0 else -1
    if dx >= dy:
        xinc1 = 0; yinc2 = 0; den = dx; num = dx // 2; numadd = dy; numpixels = dx
    else:
        xinc2 = 0; yinc1 = 0; den = dy; num = dy // 2; numadd = dx; numpixels = dy
    x, y = x0, y0
    for _ in range(numpixels + 1):
        yield x, y
        num += numadd
        if num >= den:
            num -= den; x += xinc1; y += yinc1
        x += xinc2; y += yinc2


def footprint_cost(cost, resolution, origin, x, y, theta, spec):
    c = world_to_map(cost, resolution, origin, x, y)
    if c is None:
        return -3.0
    spec = np.asarray(spec, float).reshape(-1, 2)
    if len(spec) < 3:
        v = int(cost[c[1], c[0]])
        if v == NO_INFORMATION:
            return -2.0
        if v in (LETHAL_OBSTACLE, INSCRIBED):
            return -1.0
        return float(v)
    ct, st = math.cos(theta), math.sin(theta)
    pts = [(x + (px * ct - py * st), y + (px * st + py * ct)) for px, py in spec]
    worst = 0.0
    for i in range(len(pts)):
        a, b = pts[i], pts[(i + 1) % len(pts)]
        ca = world_to_map(cost, resolution, origin, a[0], a[1])
        if ca is None:
            return -3.0
        cb = world_to_map(cost, resolution, origin, b[0], b[1])
        if cb is None:
            return -3.0
        for mx, my in line_cells(ca[0], ca[1], cb[0], cb[1]):
            v = int(cost[my, mx])
            if v == NO_INFORMATION:
                return -2.0
            if v == LETHAL_OBSTACLE:
                return -1.0
            worst = max(worst, float(v))
    return worst


def is_pose_trajectory_feasible(cost, resolution, origin, x, spec, inscribed_radius, min_resolution_collision_check_angular, look_ahead_idx=-1, pose_cost=None):
    """cost (size_y, size_x) uint8, x (n, 3) planned states.  src/controller.cpp:859-917.  pose_cost(x, y, theta): replaces the costmap lookup (a test hook: the poses asked
    about, in order, and the early exit)"""
    if pose_cost is not None:
        footprint_cost = lambda _c, _r, _o, px, py, pth, _s: pose_cost(px, py, pth)      # noqa: E731
    else:
        footprint_cost = globals()["footprint_cost"]
    n = x.shape[0]
    if n < 2:
        return False
    if look_ahead_idx < 0 or look_ahead_idx >= n:
        look_ahead_idx = n - 1
    for i in range(look_ahead_idx + 1):
        if footprint_cost(cost, resolution, origin, x[i, 0], x[i, 1], x[i, 2], spec) == -1:
            return False
        if i < look_ahead_idx:
            delta_rot = _normalize_theta(x[i + 1, 2] - x[i, 2])
            ddx, ddy = x[i + 1, 0] - x[i, 0], x[i + 1, 1] - x[i, 1]
            dist = math.sqrt(ddx * ddx + ddy * ddy)
            if abs(delta_rot) > min_resolution_collision_check_angular or dist > inscribed_radius:
                n_add = int(max(math.ceil(abs(delta_rot) / min_resolution_collision_check_angular), math.ceil(dist / inscribed_radius))) - 1
                px, py, pth = x[i, 0], x[i, 1], x[i, 2]
                for _ in range(n_add):
                    px = px + ddx / (n_add + 1.0)
                    py = py + ddy / (n_add + 1.0)
                    pth = _normalize_theta(pth + delta_rot / (n_add + 1.0))
                    if footprint_cost(cost, resolution, origin, px, py, pth, spec) == -1:
                        return False
    return True

"""CPU tests of oracle/feasibility.py (the numpy restatement of Controller::isPoseTrajectoryFeasible + the pinned footprintCost convention)
on hand-worked maps; the -m gpu test compares the device kernel with it bit for bit."""
import numpy as np

from oracle import feasibility as F

SQUARE = [(0.2, 0.2), (-0.2, 0.2), (-0.2, -0.2), (0.2, -0.2)]


def test_line_iterator_is_bresenham_with_both_endpoints():
    assert list(F.line_cells(0, 0, 4, 2)) == [(0, 0), (1, 1), (2, 1), (3, 2), (4, 2)]      # num starts at dx/2: the ROS iterator steps early
    assert list(F.line_cells(3, 3, 3, 0)) == [(3, 3), (3, 2), (3, 1), (3, 0)]
    assert list(F.line_cells(2, 2, 2, 2)) == [(2, 2)]


def test_footprint_cost_codes_and_their_order():
    cost = np.zeros((40, 40), np.uint8)
    res, org = 0.1, (0.0, 0.0)
    assert F.footprint_cost(cost, res, org, 2.0, 2.0, 0.3, SQUARE) == 0.0
    cost[22, 18:23] = 254                                   # a lethal bar through the top edge (y index 22 = world 2.2..2.3)
    assert F.footprint_cost(cost, res, org, 2.0, 2.05, 0.0, SQUARE) == -1.0
    assert F.footprint_cost(cost, res, org, 2.0, 1.5, 0.0, SQUARE) == 0.0          # the footprint is an OUTLINE: free below the bar
    cost[:] = 0; cost[20, 20] = 254
    assert F.footprint_cost(cost, res, org, 2.05, 2.05, 0.0, SQUARE) == 0.0        # lethal cell strictly inside the outline is not seen (as in ROS)
    assert F.footprint_cost(cost, res, org, 2.05, 2.05, 0.0, []) == -1.0           # < 3 points: centre cell only
    cost[20, 20] = 253
    assert F.footprint_cost(cost, res, org, 2.05, 2.05, 0.0, []) == -1.0           # INSCRIBED counts there too
    assert F.footprint_cost(cost, res, org, 0.1, 2.0, 0.0, SQUARE) == -3.0         # an outline vertex leaves the map
    assert F.footprint_cost(cost, res, org, -0.5, 2.0, 0.0, SQUARE) == -3.0        # the centre does
    cost[:] = 0; cost[22, 18] = 255; cost[22, 21] = 254                            # first edge (0.2,0.2)->(-0.2,0.2) runs right to left: sees 254 before 255
    assert F.footprint_cost(cost, res, org, 2.0, 2.05, 0.0, SQUARE) == -1.0
    cost[22, 22] = 255                                                               # now a no-information cell comes first on that edge
    assert F.footprint_cost(cost, res, org, 2.0, 2.05, 0.0, SQUARE) == -2.0


def test_trajectory_check_interpolates_between_distant_poses():
    cost = np.zeros((60, 60), np.uint8)
    res, org = 0.1, (0.0, 0.0)
    x = np.array([[1.0, 3.0, 0.0], [2.0, 3.0, 0.0], [4.0, 3.0, 0.0], [5.0, 3.0, 0.0]])
    assert F.is_pose_trajectory_feasible(cost, res, org, x, SQUARE, 0.25, 0.3)
    cost[28:33, 30] = 254                                    # a wall at x = 3.0..3.1 between grid points 1 and 2 (2 m apart)
    assert not F.is_pose_trajectory_feasible(cost, res, org, x, SQUARE, 0.25, 0.3)      # found by an interpolated pose
    assert F.is_pose_trajectory_feasible(cost, res, org, x, SQUARE, 0.25, 0.3, look_ahead_idx=1)   # not looked at: only poses 0..1
    assert F.is_pose_trajectory_feasible(cost, res, org, x, SQUARE, 5.0, 0.3)            # huge inscribed radius: no interpolation, the grid points are free
    turn = np.array([[1.0, 1.0, 0.0], [1.0, 1.0, 3.0]])
    assert F.is_pose_trajectory_feasible(cost, res, org, turn, SQUARE, 0.25, 0.3)
    assert not F.is_pose_trajectory_feasible(cost, res, org, x[:1], SQUARE, 0.25, 0.3)   # fewer than 2 grid points (:869)

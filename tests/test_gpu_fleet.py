"""FleetPlanner on the MI355X: the script of tests/test_fleet.py with the real BatchSolver behind it -- every robot of the batch reproduces the recorded run of the reference's
plugin (reference plugin + reference Controller + the C oracle's solve, recorded on the CPU) within the north-star tolerance."""
import pytest

from test_fleet import LOOPS, replay


@pytest.mark.gpu
@pytest.mark.parametrize("loop", LOOPS)
def test_fleet_cycle_on_the_gpu_reproduces_the_recorded_runs_of_the_reference_plugin(loop):
    worst_cmd, worst_x = replay(loop, None, [0, 5, 0, None], 1e-4)
    print(f"{loop}: fleet cycle on the GPU against the recorded plugin runs: largest command difference {worst_cmd:.2e}, largest state difference {worst_x:.2e}")


@pytest.mark.gpu
def test_fleet_of_64_robots_in_closed_loop():
    """64 car-like robots, each with its own plan (rotated copies of one path) and costmap, 25 control cycles with a simple-car plant: every cycle is one costmap-scan launch,
    two batched solves (outer iterations) and one feasibility-check launch for the whole fleet"""
    import json, os, time
    import numpy as np
    from fleet import FleetPlanner      # examples/fleet.py
    here = os.path.dirname(os.path.abspath(__file__))
    prm = json.load(open(os.path.join(here, "golden", "ref_plugin_closed_loop_carlike_line_footprint.json")))
    B, res = 64, 0.1
    fleet = FleetPlanner(prm, batch=B, max_obstacles=32, max_vertices=4)
    base = np.stack([np.linspace(0, 9, 70), 1.2 * np.sin(np.linspace(0, 3, 70)), np.zeros(70)], 1)
    poses, origins = np.zeros((B, 3)), np.zeros((B, 2))
    cost = np.zeros((B, 100, 140), np.uint8)
    plans = []
    for b in range(B):
        a = 2 * np.pi * b / B
        c, s = np.cos(a), np.sin(a)
        plan = np.stack([c * base[:, 0] - s * base[:, 1], s * base[:, 0] + c * base[:, 1], np.zeros(70)], 1)
        plan[:-1, 2] = np.arctan2(np.diff(plan[:, 1]), np.diff(plan[:, 0])); plan[-1, 2] = plan[-2, 2]
        plans.append(plan)
        fleet.set_plan(b, plan)
        poses[b] = (0.0, 0.0, plan[0, 2])
        origins[b] = (-7.0, -5.0)
        k = 18
        cc = plan[k, :2] + 0.55 * np.array([-np.sin(plan[k, 2]), np.cos(plan[k, 2])])            # a block beside the path
        j, i = int((cc[0] - origins[b, 0]) / res), int((cc[1] - origins[b, 1]) / res)
        cost[b, i:i + 2, j:j + 2] = 254
    fp = np.array([(0.45, 0.15), (-0.05, 0.15), (-0.05, -0.15), (0.45, -0.15)])
    ok_cycles, wall = 0, []
    for cycle in range(25):
        t0 = time.perf_counter()
        out = fleet.step(poses, cost, res, origins, fp, inscribed_radius=0.15)
        wall.append(time.perf_counter() - t0)
        ok_cycles += int((out.code == 0).sum())
        v, phi = out.cmd[:, 0], out.cmd[:, 2]
        poses = poses + 0.1 * np.stack([v * np.cos(poses[:, 2]), v * np.sin(poses[:, 2]), v / 0.4 * np.tan(phi)], 1)
    travelled = np.hypot(poses[:, 0], poses[:, 1])
    track = np.array([np.hypot(plans[b][:, 0] - poses[b, 0], plans[b][:, 1] - poses[b, 1]).min() for b in range(B)])
    print(f"fleet of {B}: {ok_cycles} / {25 * B} robot-cycles SUCCESS, travelled {travelled.min():.2f}..{travelled.max():.2f} m, distance from the plan <= {track.max():.3f} m, "
          f"wall time per fleet cycle median {1e3 * np.median(wall):.1f} ms (host bookkeeping in Python included)")
    assert ok_cycles >= 0.97 * 25 * B and travelled.min() > 0.6 and track.max() < 0.3

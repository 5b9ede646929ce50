"""The reference's plugin source (src/mpc_local_planner_ros.cpp, compiled UNCHANGED) running on the MI355X solver: oracle/_ref/libmpc_plugin_on_hip.so is that source built on
include/mpc_reference_binding.hpp (which takes the place of the reference's controller.h) and linked against the product library libmpc_hip.so -- built in the container that
has /root/reference (`make -C oracle ref`), it travels to the GPU box as a prebuilt file; nothing here reads the reference tree at run time.  The ROS side (parameter server,
costmap, tf, odometry) is the stand-in set of oracle/ref_stubs/; the solver is the real one."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(ROOT, "oracle", "_ref", "libmpc_plugin_on_hip.so")
sys.path.insert(0, os.path.join(HERE, "golden"))


def _load():
    lib = C.CDLL(LIB)
    V, D, I = C.c_void_p, C.c_double, C.c_int
    lib.hip_plugin_create.restype = V; lib.hip_plugin_create.argtypes = [C.c_char_p, I, I, V, D, D, D, I, V]
    lib.hip_plugin_destroy.restype = None; lib.hip_plugin_destroy.argtypes = [V]
    lib.hip_plugin_initialized.restype = I; lib.hip_plugin_initialized.argtypes = [V]
    lib.hip_plugin_set_solver.restype = None; lib.hip_plugin_set_solver.argtypes = [V, V]
    lib.hip_plugin_set_plan.restype = I; lib.hip_plugin_set_plan.argtypes = [V, I, V]
    lib.hip_plugin_cycle.restype = C.c_uint; lib.hip_plugin_cycle.argtypes = [V, V, V, V, V, V, I, V]
    lib.hip_plugin_last_guess.restype = I; lib.hip_plugin_last_guess.argtypes = [V, I, V, V, V]
    lib.hip_plugin_set_custom_obstacles.restype = None; lib.hip_plugin_set_custom_obstacles.argtypes = [V, I, V, V, V, V]
    lib.hip_plugin_container.restype = I; lib.hip_plugin_container.argtypes = [V, I, I, V, V]
    return lib


@pytest.mark.skipif(not os.path.exists(LIB), reason="oracle/_ref/libmpc_plugin_on_hip.so is built where the reference tree is (make -C oracle ref)")
def test_plugin_on_hip_library_exports_the_entry_points():
    lib = _load()
    assert lib.hip_plugin_cycle and lib.hip_plugin_create


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(LIB), reason="oracle/_ref/libmpc_plugin_on_hip.so is built where the reference tree is (make -C oracle ref)")
@pytest.mark.parametrize("dual_warm_start", [False, True])
def test_reference_plugin_drives_the_robot_with_the_gpu_solver(dual_warm_start):
    """initialize() with the car-like example parameters, setPlan(), then 80 x computeVelocityCommands() in closed loop with a simple-car plant: every command comes from a
    solve on the GPU; the robot follows the plan around the costmap's obstacles, the commands respect the control bounds, the plugin reports SUCCESS.
    dual_warm_start: the binding's optional parameter mpc_hip/dual_warm_start -- the multipliers of the last converged solve are kept in the handle, so the second and third
    outer iteration of a cycle (and the first solve of the next cycle) start from them at a barrier of 1e-3 instead of from scratch at 0.1, as Ipopt does with
    warm_start_init_point: same closed-loop behaviour asserted, fewer iterations per cycle (the wall time per cycle is printed)."""
    import configure_cases
    from oracle import ref_lib as RL
    prm = configure_cases.base_carlike()
    if dual_warm_start:
        prm["mpc_hip"] = {"dual_warm_start": True}
    prm["footprint_model"] = {"type": "line", "line_start": [0.0, 0.0], "line_end": [0.4, 0.0]}
    prm["controller"]["outer_ocp_iterations"] = 3
    rng = np.random.default_rng(11)
    cost = np.zeros((100, 140), np.uint8)
    res, org = 0.1, (-2.0, -5.0)
    plan = np.stack([np.linspace(0, 9, 70), 1.2 * np.sin(np.linspace(0, 3, 70)), np.zeros(70)], 1)
    plan[:-1, 2] = np.arctan2(np.diff(plan[:, 1]), np.diff(plan[:, 0])); plan[-1, 2] = plan[-2, 2]
    blocks = []
    for k in (18, 33, 48):                                   # blocks beside the path, alternating sides
        c = plan[k, :2] + np.array([0.0, 0.55 if (k // 15) % 2 else -0.55])
        j, i = int((c[0] - org[0]) / res), int((c[1] - org[1]) / res)
        cost[i:i + 2, j:j + 2] = 254
        blocks.append(c)
    lethal = np.argwhere(cost == 254)
    centres = np.stack([org[0] + (lethal[:, 1] + 0.5) * res, org[1] + (lethal[:, 0] + 0.5) * res], 1)
    fp = [(0.45, 0.15), (-0.05, 0.15), (-0.05, -0.15), (0.45, -0.15)]
    run = RL.PluginRunner(prm, cost, res, org, footprint=fp, lib=_load(), prefix="hip_plugin_")
    assert run.initialized
    assert run.set_plan(plan)
    pose, vel = np.array([0.0, 0.0, 0.1]), np.zeros(3)
    L, dt = 0.4, 0.1
    codes, cmds, clearance, track = [], [], [], []
    import time
    wall = []
    for _ in range(80):
        t0 = time.perf_counter()
        o = run.cycle(pose, vel)
        wall.append(time.perf_counter() - t0)
        codes.append(o["code"]); cmds.append(o["cmd"].copy())
        v, phi = o["cmd"][0], o["cmd"][2]                     # simple car: twist.angular.z carries the steering angle (getTwistFromControl, systems/simple_car.h)
        pose = pose + dt * np.array([v * np.cos(pose[2]), v * np.sin(pose[2]), v / L * np.tan(phi)])
        vel = np.array([v, 0.0, phi])
        clearance.append(np.hypot(centres[:, 0] - pose[0], centres[:, 1] - pose[1]).min())
        track.append(np.hypot(plan[:, 0] - pose[0], plan[:, 1] - pose[1]).min())
        if o["goal_reached"]:
            break
    cmds = np.array(cmds)
    print(f"reference plugin on the GPU solver (mpc_hip/dual_warm_start {dual_warm_start}): {len(codes)} cycles, {codes.count(0)} SUCCESS, final pose {np.round(pose, 3)}, min clearance {min(clearance):.3f} m, "
          f"max distance from the plan {max(track):.3f} m, computeVelocityCommands wall time median {1e3 * np.median(wall):.2f} ms / max {1e3 * max(wall):.2f} ms "
          f"(3 outer iterations = 3 solves per cycle, 12 point obstacles, host side = the stand-ins)")
    assert codes.count(0) >= int(0.9 * len(codes)), codes
    assert pose[0] > 2.0, pose                                # 80 cycles at <= 0.4 m/s: at most 3.2 m
    assert cmds[:, 0].max() <= 0.4 + 1e-6 and cmds[:, 0].min() >= -0.2 - 1e-6 and np.abs(cmds[:, 2]).max() <= 1.4 + 1e-6
    assert min(clearance) > 0.2, min(clearance)               # collision_avoidance/min_obstacle_dist 0.27 at the grid points of every plan
    assert max(track) < 1.0
    run.close()


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(LIB), reason="oracle/_ref/libmpc_plugin_on_hip.so is built where the reference tree is (make -C oracle ref)")
@pytest.mark.parametrize("dual_warm_start", [False, True])
@pytest.mark.parametrize("variant", ["via_points_polygon_footprint", "diff_drive_quadratic_form", "moving_obstacle_messages", "moving_obstacle_messages_two_circles_footprint"])
def test_reference_plugin_on_the_gpu_solver_other_configurations(variant, dual_warm_start):
    """the same closed loop with (a) the via-point objective (via-points taken from the plan every 0.6 m) and a polygon footprint, (b) a differential-drive robot with the
    quadratic-form objective on the fixed grid and a free goal, (c) obstacle messages on the "obstacles" topic with collision_avoidance/enable_dynamic_obstacles: a moving circle, a moving line and a static polygon beside the path"""
    import configure_cases
    from oracle import ref_lib as RL
    prm = configure_cases.base_carlike()
    prm["controller"]["outer_ocp_iterations"] = 2
    if dual_warm_start:        # the binding's optional parameter: multipliers kept between the solves of the controller (same assertions on the closed loop)
        prm["mpc_hip"] = {"dual_warm_start": True}
    car, L = True, 0.4
    msgs = None
    if variant == "via_points_polygon_footprint":
        prm["controller"]["global_plan_viapoint_sep"] = 0.6
        prm["planning"]["objective"] = {"type": "minimum_time_via_points", "minimum_time_via_points": {"position_weight": 8.0, "via_points_ordered": True}}
        prm["footprint_model"] = {"type": "polygon", "vertices": [[0.45, 0.15], [-0.05, 0.15], [-0.05, -0.15], [0.45, -0.15]]}
    elif variant == "diff_drive_quadratic_form":
        car = False
        prm["robot"] = {"type": "unicycle", "unicycle": {"max_vel_x": 0.4, "max_vel_x_backwards": 0.2, "max_vel_theta": 0.3, "acc_lim_x": 0.2, "dec_lim_x": 0.2, "acc_lim_theta": 0.2}}
        prm["grid"]["variable_grid"]["enable"] = False; prm["grid"]["xf_fixed"] = [False, False, False]
        prm["planning"]["objective"] = {"type": "quadratic_form", "quadratic_form": {"state_weights": [2.0, 2.0, 0.25], "control_weights": [0.1, 0.05], "integral_form": False}}
        prm["planning"]["terminal_cost"] = {"type": "quadratic", "quadratic": {"final_state_weights": [10.0, 10.0, 0.5]}}
        prm["controller"]["max_global_plan_lookahead_dist"] = 1.0
        prm["footprint_model"] = {"type": "circular", "radius": 0.2}
    else:
        prm["collision_avoidance"]["enable_dynamic_obstacles"] = True
        prm["footprint_model"] = ({"type": "circular", "radius": 0.2} if variant == "moving_obstacle_messages"
                                  else {"type": "two_circles", "front_offset": 0.3, "front_radius": 0.2, "rear_offset": 0.0, "rear_radius": 0.2})
        # moving obstacles that come close to the path but never into the clearance zone of the (fixed) local goal within the horizon: a predicted obstacle ON the goal
        # makes the NLP infeasible, here as in the reference (seen while writing this test: a circle drifting onto the path stops the robot in both)
        msgs = [{"points": [(2.0, 2.3, 0)], "radius": 0.25, "velocity": (0.0, -0.06)}, {"points": [(3.0, -1.6, 0), (3.4, -1.6, 0)], "velocity": (0.0, 0.05)},
                {"points": [(4.5, 2.0, 0), (4.9, 2.0, 0), (4.9, 2.4, 0), (4.5, 2.4, 0)]}]
    cost = np.zeros((100, 140), np.uint8)
    res, org = 0.1, (-2.0, -5.0)
    plan = np.stack([np.linspace(0, 9, 70), 1.2 * np.sin(np.linspace(0, 3, 70)), np.zeros(70)], 1)
    plan[:-1, 2] = np.arctan2(np.diff(plan[:, 1]), np.diff(plan[:, 0])); plan[-1, 2] = plan[-2, 2]
    for k in (18, 33, 48):
        c = plan[k, :2] + np.array([0.0, 0.6 if (k // 15) % 2 else -0.6])
        j, i = int((c[0] - org[0]) / res), int((c[1] - org[1]) / res)
        cost[i:i + 2, j:j + 2] = 254
    run = RL.PluginRunner(prm, cost, res, org, footprint=[(0.45, 0.15), (-0.05, 0.15), (-0.05, -0.15), (0.45, -0.15)], lib=_load(), prefix="hip_plugin_")
    assert run.initialized and run.set_plan(plan)
    if msgs:
        run.set_custom_obstacles(msgs)
    pose, vel, dt = np.array([0.0, 0.0, 0.1]), np.zeros(3), 0.1
    codes, cmds, track, n_via = [], [], [], []
    for _ in range(80):
        o = run.cycle(pose, vel)
        codes.append(o["code"]); cmds.append(o["cmd"].copy()); n_via.append(o["n_via"])
        v, w = o["cmd"][0], o["cmd"][2]
        pose = pose + dt * np.array([v * np.cos(pose[2]), v * np.sin(pose[2]), (v / L * np.tan(w)) if car else w])
        vel = np.array([v, 0.0, w])
        track.append(np.hypot(plan[:, 0] - pose[0], plan[:, 1] - pose[1]).min())
    cmds = np.array(cmds)
    print(f"{variant} (mpc_hip/dual_warm_start {dual_warm_start}): {codes.count(0)} / {len(codes)} SUCCESS, final pose {np.round(pose, 3)}, max distance from the plan {max(track):.3f} m, via-points per cycle {min(n_via)}..{max(n_via)}")
    assert codes.count(0) >= int(0.85 * len(codes)), (codes, run.log()[-3:])
    assert pose[0] > 1.5 and max(track) < 1.0, (pose, max(track))
    assert cmds[:, 0].max() <= 0.4 + 1e-6 and cmds[:, 0].min() >= -0.2 - 1e-6 and np.abs(cmds[:, 2]).max() <= (1.4 if car else 0.3) + 1e-6
    if variant == "via_points_polygon_footprint":
        assert max(n_via) >= 2
    if msgs:
        assert run.container()[0] >= 3 + 12
    run.close()


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(LIB), reason="oracle/_ref/libmpc_plugin_on_hip.so is built where the reference tree is (make -C oracle ref)")
@pytest.mark.parametrize("loop", ["carlike_line_footprint", "via_points_polygon_footprint", "diff_drive_quadratic_form", "carlike_to_the_goal", "carlike_block_close_to_the_path"])
def test_plugin_on_the_gpu_solver_reproduces_the_plugin_on_the_cpu_oracle(loop):
    """tests/golden/ref_plugin_closed_loop_<loop>.npz: 60 control cycles of the reference's plugin WITH THE REFERENCE'S OWN Controller (oracle/_ref), the C oracle's
    interior-point solve plugged in as its solver, recorded on the CPU (generator: tests/golden/make_ref_vectors.py) -- car-like minimum time with a line footprint, the
    via-point objective with a polygon footprint, differential drive with the quadratic form on the fixed grid and a free goal, a run to the goal (shrinking grid, a failed solve, goal reached), a block close to
    the path (clearance rows at work, a failed solve).  Here the same robot poses are replayed on
    the plugin built on the binding and the MI355X solver: the same outcome codes, the velocity commands and the planned trajectories within the north-star tolerance of 1e-4
    at every cycle -- the reference's orchestration and the CPU restatement of the solve on one side, the binding and the GPU kernel on the other"""
    import json
    from oracle import ref_lib as RL
    rec = np.load(os.path.join(HERE, "golden", f"ref_plugin_closed_loop_{loop}.npz"))
    prm = json.load(open(os.path.join(HERE, "golden", f"ref_plugin_closed_loop_{loop}.json")))
    res, ox, oy = rec["par"]
    run = RL.PluginRunner(prm, rec["cost"], float(res), (float(ox), float(oy)), footprint=rec["footprint"], lib=_load(), prefix="hip_plugin_")
    assert run.initialized and run.set_plan(rec["plan"])
    worst_cmd = worst_x = 0.0
    for i in range(rec["pose"].shape[0]):
        o = run.cycle(rec["pose"][i], rec["vel"][i])
        assert o["code"] == rec["code"][i], (i, o["code"], run.log()[-2:])
        m = int(rec["n"][i])
        assert o["x_seq"].shape[0] == m and o["n_via"] == rec["n_via"][i], (i, o["x_seq"].shape, m)
        worst_cmd = max(worst_cmd, np.abs(o["cmd"] - rec["cmd"][i]).max())
        worst_x = max(worst_x, np.abs(o["x_seq"] - rec["x_seq"][i, :m]).max())
    print(f"{loop}: plugin on the GPU solver against the plugin on the CPU oracle, {rec['pose'].shape[0]} cycles: largest command difference {worst_cmd:.2e}, largest state difference {worst_x:.2e}")
    assert worst_cmd < 1e-4 and worst_x < 1e-4
    run.close()

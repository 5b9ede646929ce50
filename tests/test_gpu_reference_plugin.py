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
    return lib


@pytest.mark.skipif(not os.path.exists(LIB), reason="oracle/_ref/libmpc_plugin_on_hip.so is built where the reference tree is (make -C oracle ref)")
def test_plugin_on_hip_library_exports_the_entry_points():
    lib = _load()
    assert lib.hip_plugin_cycle and lib.hip_plugin_create


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(LIB), reason="oracle/_ref/libmpc_plugin_on_hip.so is built where the reference tree is (make -C oracle ref)")
def test_reference_plugin_drives_the_robot_with_the_gpu_solver():
    """initialize() with the car-like example parameters, setPlan(), then 80 x computeVelocityCommands() in closed loop with a simple-car plant: every command comes from a
    solve on the GPU; the robot follows the plan around the costmap's obstacles, the commands respect the control bounds, the plugin reports SUCCESS"""
    import configure_cases
    from oracle import ref_lib as RL
    prm = configure_cases.base_carlike()
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
    print(f"reference plugin on the GPU solver: {len(codes)} cycles, {codes.count(0)} SUCCESS, final pose {np.round(pose, 3)}, min clearance {min(clearance):.3f} m, "
          f"max distance from the plan {max(track):.3f} m, computeVelocityCommands wall time median {1e3 * np.median(wall):.2f} ms / max {1e3 * max(wall):.2f} ms "
          f"(3 outer iterations = 3 solves per cycle, 12 point obstacles, host side = the stand-ins)")
    assert codes.count(0) >= int(0.9 * len(codes)), codes
    assert pose[0] > 2.0, pose                                # 80 cycles at <= 0.4 m/s: at most 3.2 m
    assert cmds[:, 0].max() <= 0.4 + 1e-6 and cmds[:, 0].min() >= -0.2 - 1e-6 and np.abs(cmds[:, 2]).max() <= 1.4 + 1e-6
    assert min(clearance) > 0.2, min(clearance)               # collision_avoidance/min_obstacle_dist 0.27 at the grid points of every plan
    assert max(track) < 1.0
    run.close()

"""The reference's parameter set -> mpc_config (mpc_local_planner_amd/params.py mirrors Controller::configure*, src/controller.cpp:58-100,225-805,
and the footprint parsing of src/mpc_local_planner_ros.cpp:890-1001).  CPU only."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import mpc_local_planner_amd as m
from mpc_local_planner_amd import _abi as A, params as P

# a parameter file in the layout of the reference's examples (own values; the keys are the reference's)
CARLIKE_YAML = """
MpcLocalPlannerROS:
  odom_topic: odom
  robot:
    type: "simple_car"
    simple_car:
      wheelbase: 0.35
      front_wheel_driving: False
      max_vel_x: 0.6
      max_vel_x_backwards: -0.25     # the reference flips the sign with a warning
      max_steering_angle: 1.2
      acc_lim_x: 0.4
      dec_lim_x: 0.0                 # zero: no row
      max_steering_rate: 0.7
  footprint_model:
    type: "line"
    line_start: [0.0, 0.0]
    line_end: [0.35, 0.0]
    is_footprint_dynamic: False
  collision_avoidance:
    min_obstacle_dist: 0.3
    enable_dynamic_obstacles: True
    force_inclusion_dist: 0.6
    cutoff_dist: 3.0
  grid:
    type: "fd_grid"
    grid_size_ref: 24
    dt_ref: 0.25
    xf_fixed: [True, True, False]
    warm_start: False
    collocation_method: "crank_nicolson_differences"
    cost_integration_method: "left_sum"
    variable_grid:
      enable: True
      min_dt: 0.01
      max_dt: 5.0
      grid_adaptation:
        enable: True
        dt_hyst_ratio: 0.2
        min_grid_size: 4
        max_grid_size: 60
  planning:
    objective:
      type: "minimum_time"
    terminal_cost:
      type: "quadratic"
      quadratic:
        final_state_weights: [1.0, 2.0, 3.0]
    terminal_constraint:
      type: "l2_ball"
      l2_ball:
        weight_matrix: [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.5]     # 3 x 3, column major, diagonal
        radius: 0.2
  controller:
    outer_ocp_iterations: 3
    force_reinit_new_goal_dist: 2.0
    force_reinit_num_steps: 5
    prefer_x_feedback: True
    publish_ocp_results: True
  solver:
    type: "ipopt"
    ipopt:
      iterations: 80
      max_cpu_time: 0.1
      ipopt_numeric_options:
        tol: 1.0e-4
      ipopt_string_options:
        linear_solver: "mumps"
        hessian_approximation: "limited-memory"
      ipopt_integer_options:
        print_level: 2
"""


def test_yaml_in_the_reference_layout(tmp_path):
    f = tmp_path / "params.yaml"
    f.write_text(CARLIKE_YAML)
    cfg, ctrl, notes = P.config_from_yaml(str(f), max_obstacles=32, max_vertices=6)
    assert cfg.model == A.MODEL_SIMPLE_CAR and cfg.model_params[0] == 0.35
    assert list(cfg.u_lb) == [-0.25, -1.2] and list(cfg.u_ub) == [0.6, 1.2]          # sign flipped
    assert list(cfg.du_ub) == [0.4, 0.7] and cfg.du_lb[0] == -A.INF and cfg.du_lb[1] == -0.7      # dec_lim_x 0 -> no row
    assert cfg.footprint_kind == P.FOOTPRINT_LINE and list(cfg.footprint_params) == [0.0, 0.0, 0.35, 0.0]
    assert (cfg.min_obstacle_dist, cfg.force_inclusion_dist, cfg.cutoff_dist, cfg.enable_dynamic_obstacles) == (0.3, 0.6, 3.0, 1)
    assert (cfg.n, cfg.dt_ref, cfg.dt_free, cfg.dt_lb, cfg.dt_ub) == (24, 0.25, 1, 0.01, 5.0)
    assert list(cfg.xf_fixed) == [1, 1, 0] and cfg.collocation == A.COLLOC_CRANK_NICOLSON
    assert cfg.objective == A.OBJ_MIN_TIME and cfg.has_Qf == 1 and list(cfg.Qf) == [1.0, 2.0, 3.0]
    assert cfg.terminal_ball == 1 and list(cfg.terminal_ball_S) == [1.0, 1.0, 0.5] and cfg.terminal_ball_gamma == 0.2
    assert (cfg.max_iter, cfg.tol, cfg.hessian_mode) == (80, 1e-4, P.HESSIAN_CONVEXIFIED)
    assert (cfg.max_obstacles, cfg.max_vertices) == (32, 6)
    assert ctrl["grid_adaptation"] and (ctrl["max_grid_size"], ctrl["min_grid_size"], ctrl["dt_hyst_ratio"], ctrl["n_max"]) == (60, 4, 0.2, 60)
    assert ctrl["warm_start"] is False and ctrl["outer_ocp_iterations"] == 3 and ctrl["force_reinit_new_goal_dist"] == 2.0
    assert ctrl["force_reinit_new_goal_angular"] == 0.5 * math.pi and ctrl["force_reinit_num_steps"] == 5
    assert ctrl["prefer_x_feedback"] and ctrl["publish_ocp_results"] and not ctrl["print_cpu_time"] and ctrl["allow_init_with_backward_motion"]
    text = " | ".join(notes)
    for word in ("max_vel_x_backwards", "linear_solver", "limited-memory", "print_level"):
        assert word in text
    assert cfg.max_time_us == 100000 and "max_cpu_time" not in text          # solver/ipopt/max_cpu_time 0.1 s -> mpc_config.max_time_us (r04)
    # the library accepts what the loader produced (mpc_create validates without touching a GPU until it allocates; here: field ranges only)
    assert 3 <= cfg.n <= 4096 and cfg.dt_ref > 0


def test_code_defaults_equal_the_librarys_defaults():
    """an empty parameter server = the in-code defaults of Controller::configure* = mpc_config_defaults (include/mpc_hip.h)"""
    from mpc_local_planner_amd import _lib
    cfg, ctrl, notes = P.config_from_params({})
    d = A.MpcConfig()
    _lib.load().mpc_config_defaults(C.byref(d))
    for name in ("model", "n", "dt_ref", "dt_free", "dt_lb", "dt_ub", "collocation", "objective", "integral_form", "has_Qf", "max_iter", "tol", "mu_init", "precision",
                 "min_obstacle_dist", "force_inclusion_dist", "cutoff_dist", "footprint_kind", "max_obstacles", "terminal_ball", "enable_dynamic_obstacles", "hessian_mode"):
        assert getattr(cfg, name) == getattr(d, name), name
    for name in ("xf_fixed", "u_lb", "u_ub", "du_lb", "du_ub"):
        assert list(getattr(cfg, name)) == list(getattr(d, name)), name
    assert ctrl == dict(grid_adaptation=True, max_grid_size=50, dt_hyst_ratio=0.1, min_grid_size=2, n_max=50, warm_start=True, outer_ocp_iterations=1,
                        force_reinit_new_goal_dist=1.0, force_reinit_new_goal_angular=0.5 * math.pi, allow_init_with_backward_motion=True,
                        force_reinit_num_steps=0, prefer_x_feedback=False, publish_ocp_results=False, print_cpu_time=False)
    assert notes == []


def test_every_model_and_objective():
    c, _, _ = P.config_from_params({"robot": {"type": "kinematic_bicycle_vel_input", "kinematic_bicycle_vel_input": {"length_rear": 0.8, "length_front": 1.1, "max_steering_rate": 0.4}}})
    assert c.model == A.MODEL_KINEMATIC_BICYCLE and list(c.model_params)[:2] == [0.8, 1.1] and c.u_ub[1] == 1.5 and list(c.du_ub) == [A.INF, 0.4]
    c, _, _ = P.config_from_params({"robot": {"type": "simple_car", "simple_car": {"front_wheel_driving": True}}})
    assert c.model == A.MODEL_SIMPLE_CAR_FRONT and c.model_params[0] == 0.5
    q = {"planning": {"objective": {"type": "quadratic_form", "quadratic_form": {"state_weights": [2, 2, 0.25], "control_weights": [0.1, 0, 0, 0.05], "integral_form": True}}},
         "grid": {"variable_grid": {"enable": False}, "xf_fixed": [False, False, False]}}
    c, ctrl, _ = P.config_from_params(q)
    assert c.objective == A.OBJ_QUADRATIC and list(c.Q) == [2, 2, 0.25] and list(c.R) == [0.1, 0.05] and c.integral_form == 1 and c.dt_free == 0
    assert ctrl["grid_adaptation"] is False and ctrl["n_max"] == 20
    v = {"planning": {"objective": {"type": "minimum_time_via_points", "minimum_time_via_points": {"position_weight": 10.5, "orientation_weight": 0.1, "via_points_ordered": True}}}}
    c, _, _ = P.config_from_params(v, max_via_points=8)
    assert c.objective == A.OBJ_MIN_TIME_VIA_POINTS and (c.vp_position_weight, c.vp_orientation_weight, c.via_points_ordered, c.max_via_points) == (10.5, 0.1, 1, 8)


def test_what_the_reference_rejects_is_rejected():
    for params, word in (({"robot": {"type": "hovercraft"}}, "Unknown robot type"),
                         ({"grid": {"type": "shooting"}}, "Unknown grid type"),
                         ({"grid": {"xf_fixed": [True, True]}}, "xf_fixed"),
                         ({"solver": {"type": "sqp"}}, "Unknown solver type"),
                         ({"planning": {"objective": {"type": "shortest_path"}}}, "Unknown objective type"),
                         ({"planning": {"objective": {"type": "quadratic_form", "quadratic_form": {"state_weights": [1, 2], "control_weights": [1, 1]}}}}, "State weights dimension"),
                         ({"planning": {"objective": {"type": "quadratic_form", "quadratic_form": {"state_weights": [1, 2, 3], "control_weights": [1]}}}}, "Control weights dimension"),
                         ({"planning": {"terminal_cost": {"type": "quadratic", "quadratic": {"final_state_weights": [1] * 4}}}}, "Final state weights"),
                         ({"planning": {"terminal_cost": {"type": "cubic"}}}, "Unknown terminal_cost"),
                         ({"planning": {"terminal_constraint": {"type": "l2_ball", "l2_ball": {"weight_matrix": []}}}}, "l2-ball weight_matrix"),
                         ({"planning": {"terminal_constraint": {"type": "box"}}}, "Unknown terminal_constraint")):
        with pytest.raises(P.ParamError, match=word):
            P.config_from_params(params)


def test_what_is_not_built_says_so_and_the_cost_variants_pass_through():
    qf = lambda **kw: {"planning": {"objective": {"type": "quadratic_form", "quadratic_form": kw}}}
    with pytest.raises(P.ParamNotImplemented, match="lsq_lm"):
        P.config_from_params({"solver": {"type": "lsq_lm"}})
    with pytest.raises(P.ParamNotImplemented, match="16 vertices"):
        P.config_from_params({"footprint_model": {"type": "polygon", "vertices": [[math.cos(0.3 * i), math.sin(0.3 * i)] for i in range(20)]}})
    # full weight matrices (column major, src/controller.cpp:565-573): the symmetric part is what x'Qx sees
    c, _, _ = P.config_from_params(qf(state_weights=[1, 0.5, 0, 0.3, 2, 0, 0, 0.2, 3], control_weights=[1, 0.1, 0.3, 2]))
    assert list(c.Q) == [1, 2, 3] and list(c.Q_offdiag) == [0.4, 0.0, 0.1] and list(c.R) == [1, 2] and c.R_offdiag == 0.2
    c, _, _ = P.config_from_params(qf(state_weights=[1, 0.5, 0, -0.5, 1, 0, 0, 0, 1], control_weights=[1, 1]))       # antisymmetric off-diagonal part: no effect
    assert list(c.Q) == [1, 1, 1] and list(c.Q_offdiag) == [0, 0, 0]
    tc = {"planning": {"terminal_cost": {"type": "quadratic", "quadratic": {"final_state_weights": [5, 1, 0, 1, 6, 0.5, 0, 0.5, 7]}},
                       "terminal_constraint": {"type": "l2_ball", "l2_ball": {"weight_matrix": [1, 0.2, 0, 0.2, 1, 0, 0, 0, 0.5]}}}}
    c, _, _ = P.config_from_params(tc)
    assert list(c.Qf) == [5, 6, 7] and list(c.Qf_offdiag) == [1, 0, 0.5] and list(c.terminal_ball_S_offdiag) == [0.2, 0, 0]
    # hybrid cost: only with zero state weights and non-zero control weights; otherwise the reference itself falls back (src/controller.cpp:603-618)
    c, _, notes = P.config_from_params(qf(state_weights=[0, 0, 0], control_weights=[1, 1], hybrid_cost_minimum_time=True))
    assert c.objective == A.OBJ_QUADRATIC and c.hybrid_cost_minimum_time == 1 and not notes
    c, _, notes = P.config_from_params(qf(state_weights=[1, 1, 1], control_weights=[1, 1], hybrid_cost_minimum_time=True))
    assert c.hybrid_cost_minimum_time == 0 and any("Falling back to quadratic form" in s for s in notes)
    tr = qf(state_weights=[1, 1, 1], control_weights=[1, 1], integral_form=True)
    tr["grid"] = {"cost_integration_method": "trapezoidal_rule"}
    c, _, _ = P.config_from_params(tr)
    assert c.cost_integration == A.COST_TRAPEZOIDAL and c.integral_form == 1
    assert P.config_from_params(qf(state_weights=[1, 1, 1], control_weights=[1, 1], integral_form=True))[0].cost_integration == A.COST_LEFT_SUM


def test_footprint_models_and_their_fallbacks():
    fp = lambda **kw: P.config_from_params({"footprint_model": kw})
    assert fp(type="point")[0].footprint_kind == P.FOOTPRINT_POINT
    c, _, _ = fp(type="circular", radius=0.3)
    assert c.footprint_kind == P.FOOTPRINT_CIRCLE and c.footprint_radius == 0.3
    c, _, _ = fp(type="two_circles", front_offset=0.2, front_radius=0.25, rear_offset=0.1, rear_radius=0.2)
    assert c.footprint_kind == P.FOOTPRINT_TWO_CIRCLES and list(c.footprint_params) == [0.2, 0.25, 0.1, 0.2]
    c, _, _ = fp(type="polygon", vertices=[[0.3, 0.2], [-0.3, 0.2], [-0.3, -0.2], [0.3, -0.2]])
    assert c.footprint_kind == P.FOOTPRINT_POLYGON and c.footprint_n_vertices == 4 and list(c.footprint_vertices)[:4] == [0.3, 0.2, -0.3, 0.2]
    c, _, _ = P.config_from_params({"footprint_model": {"type": "costmap_2d"}}, costmap_footprint=[(0.2, 0.1), (-0.2, 0.1), (-0.2, -0.1), (0.2, -0.1)])
    assert c.footprint_kind == P.FOOTPRINT_POLYGON and c.footprint_n_vertices == 4
    # malformed models fall back to the point model with the reference's complaint (src/mpc_local_planner_ros.cpp:921-1010)
    for bad in (dict(type="circular"), dict(type="line", line_start=[0, 0]), dict(type="line", line_start=[0, 0, 0], line_end=[1, 0]),
                dict(type="two_circles", front_offset=0.2), dict(type="polygon", vertices=[[0, 0], [1, 0]]), dict(type="costmap_2d"), dict(type="blob")):
        c, _, notes = fp(**bad)
        assert c.footprint_kind == P.FOOTPRINT_POINT and notes, bad


def test_wrong_typed_values_keep_the_default_like_roscpp():
    c, _, _ = P.config_from_params({"grid": {"grid_size_ref": "many", "dt_ref": 1}, "solver": {"ipopt": {"iterations": 60.0}}})
    assert c.n == 20 and c.dt_ref == 1.0 and c.max_iter == 60


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_the_references_own_parameter_files_load():
    """the four parameter files the reference ships; BASELINE.json's configs 1 and 2 are the quadratic-form and the car-like file with n overridden"""
    ex = os.path.join(REF, "mpc_local_planner_examples", "cfg")
    car, _, _ = P.config_from_yaml(os.path.join(ex, "carlike", "mpc_local_planner_params.yaml"))
    ours = m.config_carlike_min_time(n=20)
    for name in ("model", "n", "dt_ref", "dt_free", "dt_lb", "dt_ub", "objective", "collocation"):
        assert getattr(car, name) == getattr(ours, name), name
    for name in ("u_lb", "u_ub", "du_lb", "du_ub", "xf_fixed"):
        assert list(getattr(car, name)) == list(getattr(ours, name)), name
    assert car.model_params[0] == ours.model_params[0] == 0.4
    assert car.footprint_kind == P.FOOTPRINT_LINE and list(car.footprint_params) == [0.0, 0.0, 0.4, 0.0] and car.min_obstacle_dist == 0.27
    assert car.tol == 1e-4 and car.hessian_mode == P.HESSIAN_CONVEXIFIED
    qf, ctrl, _ = P.config_from_yaml(os.path.join(ex, "diff_drive", "mpc_local_planner_params_quadratic_form.yaml"))
    ours = m.config_unicycle_quadratic(n=20)
    for name in ("model", "n", "dt_ref", "dt_free", "objective", "integral_form", "has_Qf", "min_obstacle_dist", "force_inclusion_dist", "cutoff_dist"):
        assert getattr(qf, name) == getattr(ours, name), name
    for name in ("Q", "R", "Qf", "u_lb", "u_ub", "du_lb", "du_ub", "xf_fixed"):
        assert list(getattr(qf, name)) == list(getattr(ours, name)), name
    assert ctrl["grid_adaptation"] is False
    mt, ctrl, _ = P.config_from_yaml(os.path.join(ex, "diff_drive", "mpc_local_planner_params_minimum_time.yaml"))
    assert mt.objective == A.OBJ_MIN_TIME and ctrl["outer_ocp_iterations"] == 5 and mt.hessian_mode == P.HESSIAN_EXACT
    node, ctrl, _ = P.config_from_yaml(os.path.join(REF, "mpc_local_planner", "cfg", "test_mpc_optim_node.yaml"), namespace=None)
    assert node.model == A.MODEL_UNICYCLE and node.n == 20 and node.has_Qf == 1 and list(node.xf_fixed) == [1, 1, 1] and ctrl["publish_ocp_results"]


# ---- the C++ twin (include/mpc_params.hpp) gives the same mpc_config, options and verdicts --------------------------------------------------
HERE = os.path.dirname(os.path.abspath(__file__))
OPT_NAMES = ["grid_adaptation", "max_grid_size", "dt_hyst_ratio", "min_grid_size", "n_max", "warm_start", "outer_ocp_iterations", "force_reinit_new_goal_dist",
             "force_reinit_new_goal_angular", "allow_init_with_backward_motion", "force_reinit_num_steps", "prefer_x_feedback", "publish_ocp_results", "print_cpu_time"]


@pytest.fixture(scope="module")
def cpp():
    import subprocess
    from mpc_local_planner_amd import _lib
    _lib.load()
    C.CDLL(_lib.LIB_PATH, mode=C.RTLD_GLOBAL)          # mpc_config_defaults for the lazily linked harness
    src = os.path.join(HERE, "host_harness", "params_host.cpp")
    out = os.path.join(HERE, "host_harness", "_build", "libctl_params.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wl,--unresolved-symbols=ignore-all", src, "-o", out], check=True)
    lib = C.CDLL(out)
    lib.ctl_config_from_params.restype = C.c_int
    lib.ctl_config_from_params.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(A.MpcConfig), C.c_void_p, C.c_char_p, C.c_int]
    return lib


def _flatten(tree, prefix=""):
    lines = []
    for k, v in tree.items():
        key = f"{prefix}/{k}" if prefix else str(k)
        if isinstance(v, dict):
            lines += _flatten(v, key)
        elif isinstance(v, bool):
            lines.append(f"{key}\tb\t{int(v)}")
        elif isinstance(v, int):
            lines.append(f"{key}\ti\t{v}")
        elif isinstance(v, float):
            lines.append(f"{key}\td\t{v!r}")
        elif isinstance(v, str):
            lines.append(f"{key}\ts\t{v}")
        elif isinstance(v, (list, tuple)) and v and isinstance(v[0], (list, tuple)):
            lines.append(f"{key}\tpv\t" + ";".join(",".join(repr(float(x)) for x in q) for q in v))
        elif isinstance(v, (list, tuple)) and v and isinstance(v[0], bool):
            lines.append(f"{key}\tbv\t" + ",".join(str(int(x)) for x in v))
        elif isinstance(v, (list, tuple)):
            lines.append(f"{key}\tdv\t" + ",".join(repr(float(x)) for x in v))
    return lines


def _cpp_config(cpp, tree, costmap_footprint=None):
    cfg = A.MpcConfig()
    opt = np.zeros(len(OPT_NAMES))
    rep = C.create_string_buffer(8192)
    fp = ";".join(f"{x!r},{y!r}" for x, y in costmap_footprint).encode() if costmap_footprint else b""
    st = cpp.ctl_config_from_params("\n".join(_flatten(tree)).encode(), fp, C.byref(cfg), opt.ctypes.data_as(C.c_void_p), rep, 8192)
    return st, cfg, dict(zip(OPT_NAMES, opt)), rep.value.decode()


SCALARS = ["model", "n", "dt_ref", "dt_free", "dt_lb", "dt_ub", "collocation", "objective", "integral_form", "has_Qf", "max_iter", "tol", "mu_init", "precision",
           "min_obstacle_dist", "force_inclusion_dist", "cutoff_dist", "footprint_kind", "footprint_radius", "footprint_n_vertices", "max_obstacles", "max_vertices",
           "max_obstacle_rows", "terminal_ball", "enable_dynamic_obstacles", "hessian_mode", "via_points_ordered", "n_candidates", "dual_warm_start", "hybrid_cost_minimum_time", "cost_integration", "R_offdiag", "acceptable_tol", "acceptable_iter", "mu_strategy", "max_time_us", "line_search"]
ARRAYS = ["model_params", "xf_fixed", "Q", "R", "Qf", "u_lb", "u_ub", "du_lb", "du_ub", "terminal_ball_S", "footprint_params", "footprint_vertices", "Q_offdiag", "Qf_offdiag",
          "terminal_ball_S_offdiag"]


def test_cpp_reader_agrees_with_the_python_reader(cpp, tmp_path):
    import yaml
    cases = [({}, None), (yaml.safe_load(CARLIKE_YAML)["MpcLocalPlannerROS"], None),
             ({"robot": {"type": "kinematic_bicycle_vel_input", "kinematic_bicycle_vel_input": {"length_rear": 0.8, "max_vel_x_backwards": -0.1, "dec_lim_x": -0.3}},
               "footprint_model": {"type": "polygon", "vertices": [[0.3, 0.2], [-0.3, 0.2], [-0.3, -0.2], [0.3, -0.2]]},
               "grid": {"collocation_method": "midpoint_differences", "variable_grid": {"grid_adaptation": {"enable": False}}}}, None),
             ({"planning": {"objective": {"type": "quadratic_form", "quadratic_form": {"state_weights": [2.0, 2.0, 0.25], "control_weights": [0.1, 0.0, 0.0, 0.05], "integral_form": True}},
                            "terminal_cost": {"type": "quadratic", "quadratic": {"final_state_weights": [10.0, 10.0, 0.5]}}},
               "grid": {"variable_grid": {"enable": False}, "xf_fixed": [False, False, True], "grid_size_ref": 33}}, None),
             ({"planning": {"objective": {"type": "minimum_time_via_points", "minimum_time_via_points": {"position_weight": 10.5, "via_points_ordered": True}}},
               "footprint_model": {"type": "two_circles", "front_offset": 0.2, "front_radius": 0.25, "rear_offset": 0.1, "rear_radius": 0.2},
               "solver": {"ipopt": {"iterations": 55, "ipopt_numeric_options": {"tol": 1e-5, "mu_init": 0.05, "acceptable_tol": 1e-3}, "ipopt_integer_options": {"max_iter": 70},
                                    "ipopt_string_options": {"line_search_method": "filter"}}}}, None),
             ({"solver": {"ipopt": {"ipopt_string_options": {"line_search_method": "cg-penalty", "mu_strategy": "monotone"}}}}, None),
             ({"planning": {"objective": {"type": "quadratic_form", "quadratic_form": {"state_weights": [1, 0.5, 0, 0.3, 2, 0, 0, 0.2, 3.0], "control_weights": [1, 0.1, 0.3, 2.0], "integral_form": True}},
                            "terminal_cost": {"type": "quadratic", "quadratic": {"final_state_weights": [5, 1, 0, 1, 6, 0.5, 0, 0.5, 7.0]}},
                            "terminal_constraint": {"type": "l2_ball", "l2_ball": {"weight_matrix": [1, 0.2, 0, 0.2, 1, 0, 0, 0, 0.5], "radius": 0.4}}},
               "grid": {"cost_integration_method": "trapezoidal_rule", "xf_fixed": [False, False, False]}}, None),
             ({"planning": {"objective": {"type": "quadratic_form", "quadratic_form": {"state_weights": [0.0, 0.0, 0.0], "control_weights": [1.0, 0.5], "hybrid_cost_minimum_time": True}}}}, None),
             ({"footprint_model": {"type": "costmap_2d"}}, [(0.2, 0.1), (-0.2, 0.1), (-0.2, -0.1), (0.2, -0.1)]),
             ({"footprint_model": {"type": "circular"}}, None), ({"footprint_model": {"type": "line", "line_start": [0.0, 0.0], "line_end": [0.4, 0.0]}}, None)]
    for tree, fp in cases:
        py_cfg, py_ctrl, py_notes = P.config_from_params(tree, costmap_footprint=fp)
        st, cfg, opt, rep = _cpp_config(cpp, tree, fp)
        assert st == 0, rep
        for name in SCALARS:
            assert getattr(cfg, name) == getattr(py_cfg, name), (name, tree)
        for name in ARRAYS:
            assert list(getattr(cfg, name)) == list(getattr(py_cfg, name)), (name, tree)
        if py_cfg.terminal_ball:
            assert cfg.terminal_ball_gamma == py_cfg.terminal_ball_gamma
        if py_cfg.objective == A.OBJ_MIN_TIME_VIA_POINTS:
            assert (cfg.vp_position_weight, cfg.vp_orientation_weight, cfg.max_via_points) == (py_cfg.vp_position_weight, py_cfg.vp_orientation_weight, py_cfg.max_via_points)
        for k, v in py_ctrl.items():
            assert opt[k] == float(v), (k, tree)
        assert len([s for s in rep.split("\n") if s]) == len(py_notes), (rep, py_notes)


def test_cpp_reader_gives_the_same_verdicts(cpp):
    qf = lambda **kw: {"planning": {"objective": {"type": "quadratic_form", "quadratic_form": kw}}}
    rejected = [{"robot": {"type": "hovercraft"}}, {"grid": {"type": "shooting"}}, {"grid": {"xf_fixed": [True, True]}}, {"solver": {"type": "sqp"}},
                {"planning": {"objective": {"type": "shortest_path"}}}, qf(state_weights=[1.0, 2.0], control_weights=[1.0, 1.0]),
                {"planning": {"terminal_cost": {"type": "cubic"}}}, {"planning": {"terminal_constraint": {"type": "box"}}}]
    missing = [{"solver": {"type": "lsq_lm"}}]
    for tree in rejected:
        with pytest.raises(P.ParamError) as e:
            P.config_from_params(tree)
        st, _, _, rep = _cpp_config(cpp, tree)
        assert st == 1 and rep.split("\n")[0] == str(e.value), (rep, str(e.value))
    for tree in missing:
        with pytest.raises(P.ParamNotImplemented):
            P.config_from_params(tree)
        assert _cpp_config(cpp, tree)[0] == 2

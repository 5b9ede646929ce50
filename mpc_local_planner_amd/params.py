"""The reference's parameter set -> struct mpc_config.

`Controller::configure` (src/controller.cpp:58-100) reads ~70 keys from the ROS parameter server (`nh.param(key, var, default)`) while it
builds the corbo objects: configureRobotDynamics (:344-378), configureGrid (:225-342), configureSolver (:380-481), configureOcp (:483-805);
the footprint is read by the plugin (src/mpc_local_planner_ros.cpp:890-1001).  A user of the reference owns YAML files with exactly those
keys (mpc_local_planner_examples/cfg/**).  This module reads the same keys with the same in-code defaults, the same fix-ups (negative
`max_vel_x_backwards` / `dec_lim_x` are flipped, a rate limit <= 0 means "none") and the same rejections, and returns

    cfg    the mpc_config for mpc_create (what shapes the NLP)
    ctrl   the options that live in the Controller facade, not in the solve (include/mpc_controller.hpp: grid adaptation, warm start,
           outer iterations, re-initialisation thresholds, state feedback, result publishing)
    notes  parameters that were accepted but have no effect here, or were mapped onto the nearest equivalent (one line each)

A configuration the reference rejects (`configure` returns false) raises ParamError with the reference's reason; one that the reference accepts
but this path does not implement raises ParamNotImplemented (it never falls back silently to a different NLP): today that is the
Levenberg-Marquardt solver (`solver/type lsq_lm`) and a polygon footprint with more than 16 vertices.

    cfg, ctrl, notes = config_from_yaml("mpc_local_planner_params.yaml", max_obstacles=64, max_vertices=8)
    solver = BatchSolver(cfg, max_batch=1024)

Sizing parameters that the reference does not have (obstacle / via-point capacity of a handle, arithmetic precision, candidates, ...) are
keyword arguments and go to make_config unchanged.
"""
from __future__ import annotations

import math
from typing import Any

from . import _abi as A

PLUGIN_NAMESPACE = "MpcLocalPlannerROS"      # move_base loads the plugin's parameters under this name (the example YAMLs' top-level key)
FOOTPRINT_POINT, FOOTPRINT_CIRCLE, FOOTPRINT_LINE, FOOTPRINT_TWO_CIRCLES, FOOTPRINT_POLYGON = range(5)   # include/mpc_hip.h, enum mpc_footprint_kind
HESSIAN_EXACT, HESSIAN_CONVEXIFIED = 0, 1
MU_ADAPTIVE, MU_MONOTONE = 0, 1
LS_DEFAULT, LS_MERIT, LS_FILTER = 0, 1, 2      # enum mpc_line_search


class ParamError(ValueError):
    """the reference's configure() fails on this parameter set"""


class ParamNotImplemented(NotImplementedError):
    """valid for the reference, not built here"""


class _Reader:
    """nh.param(key, var, default): nested dicts addressed by 'a/b/c'; records which keys were read"""

    def __init__(self, tree: dict):
        self.tree = tree or {}
        self.used: set[str] = set()

    def has(self, key: str) -> bool:
        node: Any = self.tree
        for part in key.split("/"):
            if not isinstance(node, dict) or part not in node:
                return False
            node = node[part]
        return True

    def get(self, key: str, default):
        node: Any = self.tree
        for part in key.split("/"):
            if not isinstance(node, dict) or part not in node:
                return default
            node = node[part]
        self.used.add(key)
        # roscpp (param.cpp): a double parameter takes an int, an int parameter takes a double rounded half up, a bool parameter takes a bool only;
        # a value of any other kind leaves the default in place
        if isinstance(default, bool):
            return node if isinstance(node, bool) else default
        if isinstance(default, int):
            if isinstance(node, float):
                return int(math.floor(node) if math.fmod(node, 1.0) < 0.5 else math.ceil(node))
            return node if isinstance(node, int) and not isinstance(node, bool) else default
        if isinstance(default, float):
            return float(node) if isinstance(node, (int, float)) and not isinstance(node, bool) else default
        if isinstance(default, str):
            return node if isinstance(node, str) else default
        return node

    def unused(self, prefix: str = "") -> list[str]:
        out = []

        def walk(node, path):
            if isinstance(node, dict) and node:
                for k, v in node.items():
                    walk(v, f"{path}/{k}" if path else str(k))
            elif path not in self.used and not any(u.startswith(path + "/") or path.startswith(u + "/") for u in self.used):
                out.append(path)
        walk(self.tree, prefix)
        return out


def _weights(values, dim: int, reason: str):
    """a weight list of length dim (diagonal) or dim*dim (full matrix, column major: Eigen's default, src/controller.cpp:565-573) -> what make_config
    takes: the diagonal, or the matrix as nested rows (only its symmetric part enters x' W x)"""
    v = [float(x) for x in (values or [])]
    if len(v) == dim:
        return tuple(v)
    if len(v) == dim * dim:
        m = [[v[c * dim + r] for c in range(dim)] for r in range(dim)]
        if all(m[r][c] + m[c][r] == 0.0 for r in range(dim) for c in range(dim) if r != c):
            return tuple(m[i][i] for i in range(dim))
        return tuple(tuple(row) for row in m)
    raise ParamError(reason)


def _is_zero(w) -> bool:
    return all((all(x == 0 for x in r) if hasattr(r, "__len__") else r == 0) for r in w)


def config_from_params(params: dict, costmap_footprint=None, **sizing):
    """params: the plugin's parameter namespace as nested dicts (what rosparam holds under /move_base/MpcLocalPlannerROS).
    costmap_footprint: vertices [(x, y), ...] for footprint_model/type == costmap_2d (the reference asks the costmap; without one it
    falls back to the point model, src/mpc_local_planner_ros.cpp:902-911).
    sizing: make_config keywords the reference has no parameter for (max_obstacles, max_vertices, max_obstacle_rows, max_via_points,
    precision, candidates, ...); they override what the parameters give."""
    p = _Reader(params)
    notes: list[str] = []
    kw: dict[str, Any] = {}

    # ---- configureRobotDynamics (src/controller.cpp:344-378) and the control bounds of configureOcp (:494-548)
    robot = p.get("robot/type", "unicycle")
    if robot == "unicycle":
        kw["model"], kw["model_params"] = A.MODEL_UNICYCLE, (0.0,)
        ns, second, second_default, rate_key = "robot/unicycle", "max_vel_theta", 0.3, "acc_lim_theta"
    elif robot == "simple_car":
        wheelbase = p.get("robot/simple_car/wheelbase", 0.5)
        front = p.get("robot/simple_car/front_wheel_driving", False)
        kw["model"], kw["model_params"] = (A.MODEL_SIMPLE_CAR_FRONT if front else A.MODEL_SIMPLE_CAR), (wheelbase,)
        ns, second, second_default, rate_key = "robot/simple_car", "max_steering_angle", 1.5, "max_steering_rate"
    elif robot == "kinematic_bicycle_vel_input":
        lr = p.get("robot/kinematic_bicycle_vel_input/length_rear", 1.0)
        lf = p.get("robot/kinematic_bicycle_vel_input/length_front", 1.0)
        kw["model"], kw["model_params"] = A.MODEL_KINEMATIC_BICYCLE, (lr, lf)
        ns, second, second_default, rate_key = "robot/kinematic_bicycle_vel_input", "max_steering_angle", 1.5, "max_steering_rate"
    else:
        raise ParamError(f"Unknown robot type '{robot}' specified.")                       # :373
    vmax = p.get(f"{ns}/max_vel_x", 0.4)
    vback = p.get(f"{ns}/max_vel_x_backwards", 0.2)
    if vback < 0:
        notes.append("max_vel_x_backwards must be >= 0 (sign flipped, as the reference does)")
        vback = -vback
    wmax = p.get(f"{ns}/{second}", second_default)
    kw["u_lb"], kw["u_ub"] = (-vback, -wmax), (vmax, wmax)
    # control-rate rows (:731-797): a limit <= 0 means "no row"
    acc = p.get(f"{ns}/acc_lim_x", 0.0)
    dec = p.get(f"{ns}/dec_lim_x", 0.0)
    if dec < 0:
        notes.append("dec_lim_x must be >= 0 (sign flipped, as the reference does)")
        dec = -dec
    rate = p.get(f"{ns}/{rate_key}", 0.0)
    inf = A.INF
    acc, dec, rate = (acc if acc > 0 else inf), (dec if dec > 0 else inf), (rate if rate > 0 else inf)
    kw["du_lb"], kw["du_ub"] = (-dec, -rate), (acc, rate)

    # ---- configureGrid (:225-342)
    grid_type = p.get("grid/type", "fd_grid")
    if grid_type != "fd_grid":
        raise ParamError(f"Unknown grid type '{grid_type}' specified.")                      # :337
    ctrl: dict[str, Any] = {}
    variable = p.get("grid/variable_grid/enable", True)
    kw["dt_free"] = variable
    if variable:
        kw["dt_lb"] = p.get("grid/variable_grid/min_dt", 0.0)
        kw["dt_ub"] = p.get("grid/variable_grid/max_dt", 10.0)
        adapt = p.get("grid/variable_grid/grid_adaptation/enable", True)
        ctrl["grid_adaptation"] = adapt
        if adapt:
            ctrl["max_grid_size"] = p.get("grid/variable_grid/grid_adaptation/max_grid_size", 50)
            ctrl["dt_hyst_ratio"] = p.get("grid/variable_grid/grid_adaptation/dt_hyst_ratio", 0.1)
            ctrl["min_grid_size"] = p.get("grid/variable_grid/grid_adaptation/min_grid_size", 2)
    else:
        ctrl["grid_adaptation"] = False
    kw["n"] = p.get("grid/grid_size_ref", 20)
    kw["dt_ref"] = p.get("grid/dt_ref", 0.3)
    xf_fixed = p.get("grid/xf_fixed", [True, True, True])
    if len(xf_fixed) != 3:
        raise ParamError(f"Array size of `xf_fixed` does not match robot state dimension(): {len(xf_fixed)} != 3")     # :285
    kw["xf_fixed"] = tuple(bool(v) for v in xf_fixed)
    ctrl["warm_start"] = p.get("grid/warm_start", True)
    colloc = p.get("grid/collocation_method", "forward_differences")
    colloc_ids = {"forward_differences": A.COLLOC_FORWARD, "midpoint_differences": A.COLLOC_MIDPOINT, "crank_nicolson_differences": A.COLLOC_CRANK_NICOLSON}
    if colloc not in colloc_ids:
        # :314: the reference logs "Falling back to default..." and goes on with the grid's initial rule, corbo's plain Crank-Nicolson differences
        # (full_discretization_grid_base_se2.h:206), a rule WITHOUT the SE(2) heading wrap that is not built here
        raise ParamNotImplemented(f"Unknown collocation method '{colloc}' specified: the reference falls back to corbo::CrankNicolsonDiffCollocation "
                                  "(no SE(2) heading wrap), not built here")
    kw["collocation"] = colloc_ids[colloc]
    integration = p.get("grid/cost_integration_method", "left_sum")
    if integration not in ("left_sum", "trapezoidal_rule"):
        notes.append(f"Unknown cost integration method '{integration}' specified. Falling back to default...")       # :331
        integration = "left_sum"
    # the grid with max_grid_size points must fit the handle: the solver is sized for the largest grid the adaptation may reach
    ctrl["n_max"] = max(kw["n"], ctrl.get("max_grid_size", 0)) if ctrl["grid_adaptation"] else kw["n"]

    # ---- configureSolver (:380-481)
    solver = p.get("solver/type", "ipopt")
    if solver == "lsq_lm":
        raise ParamNotImplemented("solver/type lsq_lm: the Levenberg-Marquardt least-squares solver is outside this path (SURVEY.md section 8: out of scope)")
    if solver != "ipopt":
        raise ParamError(f"Unknown solver type '{solver}' specified.")                        # :477
    kw["max_iter"] = p.get("solver/ipopt/iterations", 100)
    if p.get("solver/ipopt/max_cpu_time", -1.0) > 0:                                         # :395-397 -> mpc_config.max_time_us (per solve, on the device's clock)
        kw["max_cpu_time"] = float(p.get("solver/ipopt/max_cpu_time", -1.0))
    numeric = p.get("solver/ipopt/ipopt_numeric_options", {}) or {}
    strings = p.get("solver/ipopt/ipopt_string_options", {}) or {}
    integers = p.get("solver/ipopt/ipopt_integer_options", {}) or {}
    for k, v in numeric.items():
        if isinstance(v, str):
            # `tol: 1e-4` is TEXT for a YAML 1.1 loader (PyYAML, which rosparam uses, wants a dot in a float): roscpp then rejects the whole map
            # (param.cpp: every element must be castable) and the reference runs with Ipopt's defaults.  The evident intent is honoured here.
            try:
                v = float(v)
            except ValueError:
                notes.append(f"ipopt numeric option {k} = {v!r}: not a number, ignored")
                continue
            notes.append(f"ipopt numeric option {k} is text ('{numeric[k]}' is not a YAML 1.1 float): roscpp rejects the whole map and the reference runs with "
                         "Ipopt's defaults; the value is used here")
        if k == "tol":
            kw["tol"] = float(v)
        elif k == "mu_init":
            kw["mu_init"] = float(v)
        elif k == "acceptable_tol":
            kw["acceptable_tol"] = float(v) if float(v) > 0 else -1.0
        else:
            notes.append(f"ipopt numeric option {k} = {v}: no counterpart, ignored")
    for k, v in strings.items():
        if k == "hessian_approximation":
            if v == "limited-memory":
                kw["hessian_mode"] = HESSIAN_CONVEXIFIED
                notes.append("hessian_approximation limited-memory -> MPC_HESSIAN_CONVEXIFIED (the positive-semidefinite part of the exact stage Hessians; "
                             "no quasi-Newton update is built)")
            else:
                kw["hessian_mode"] = HESSIAN_EXACT
        elif k == "mu_strategy":
            if v in ("monotone", "adaptive"):
                kw["mu_strategy"] = MU_MONOTONE if v == "monotone" else MU_ADAPTIVE
            else:
                notes.append(f"mu_strategy {v}: unknown, the default (adaptive) is used")
        elif k == "line_search_method":      # Ipopt: filter (its default) | cg-penalty | penalty
            if v == "filter":
                kw["line_search"] = LS_FILTER
            elif v in ("penalty", "cg-penalty"):
                kw["line_search"] = LS_MERIT
                notes.append(f"line_search_method {v} -> MPC_LS_MERIT (backtracking on the l1 merit function with Ipopt's penalty rule)")
            else:
                notes.append(f"line_search_method {v}: unknown, the library's default is used")
        elif k == "linear_solver":
            notes.append(f"linear_solver {v}: the KKT systems are solved by the stage-structured sweep of the kernel")
        else:
            notes.append(f"ipopt string option {k} = {v}: no counterpart, ignored")
    for k, v in integers.items():
        if k == "max_iter":
            kw["max_iter"] = int(v)
        elif k == "acceptable_iter":
            kw["acceptable_iter"] = int(v) if int(v) > 0 else -1      # Ipopt: 0 disables the heuristic
        else:
            notes.append(f"ipopt integer option {k} = {v}: no counterpart, ignored")
    if "tol" not in kw:
        kw["tol"] = 1e-8       # Ipopt's default tol
    if kw.get("hessian_mode") == HESSIAN_CONVEXIFIED and kw["tol"] < 1e-6:
        # measured on a closed-loop goal approach of the shipped car-like parameter file (r04, C oracle as the solver): first-order curvature converges linearly close to
        # the goal -- with tol 1e-8 the convexified Hessian needs 23.3 iterations per solve and fails one cycle of 54 (at the iteration limit; before the acceptable-level stop existed
        # it stalled 0.27 m in front of the goal, 41 of 90 cycles failing), the exact Hessian 18.0 iterations and no failing cycle.  tol 1e-4 -- what the shipped car-like file asks
        # for -- keeps the convexified mode
        kw["hessian_mode"] = HESSIAN_EXACT
        notes.append("hessian_approximation limited-memory with tol < 1e-6: the exact Hessian is used instead of MPC_HESSIAN_CONVEXIFIED (first-order curvature does not reach "
                     "such a tolerance reliably close to the goal; the KKT points are the same)")

    # ---- configureOcp: objective (:551-639)
    objective = p.get("planning/objective/type", "minimum_time")
    if objective == "minimum_time":
        kw["objective"] = A.OBJ_MIN_TIME
    elif objective == "quadratic_form":
        kw["objective"] = A.OBJ_QUADRATIC
        Q = _weights(p.get("planning/objective/quadratic_form/state_weights", []), 3, "State weights dimension invalid. Must be either 3 x 1 or 3 x 3.")     # :575
        R = _weights(p.get("planning/objective/quadratic_form/control_weights", []), 2, "Control weights dimension invalid. Must be either 2 x 1 or 2 x 2.")   # :590
        integral = p.get("planning/objective/quadratic_form/integral_form", False)
        hybrid = p.get("planning/objective/quadratic_form/hybrid_cost_minimum_time", False)
        q_zero, r_zero = _is_zero(Q), _is_zero(R)
        if hybrid and not (q_zero and not r_zero):
            # :603-612: only the pure control cost has a hybrid variant
            notes.append("Hybrid minimum time and quadratic form cost is currently only supported for non-zero control weights only. Falling back to quadratic form.")
            hybrid = False
        kw["Q"], kw["R"], kw["integral_form"] = Q, R, integral
        kw["hybrid_cost_minimum_time"] = hybrid                  # corbo::MinTimeQuadraticControls: (n - 1) dt + the control cost (:616-618)
        kw["cost_integration"] = A.COST_TRAPEZOIDAL if integration == "trapezoidal_rule" else A.COST_LEFT_SUM      # integral-form terms only (:318-333)
    elif objective == "minimum_time_via_points":
        kw["objective"] = A.OBJ_MIN_TIME_VIA_POINTS
        kw["via_points_ordered"] = p.get("planning/objective/minimum_time_via_points/via_points_ordered", False)
        kw["vp_position_weight"] = p.get("planning/objective/minimum_time_via_points/position_weight", 1.0)
        kw["vp_orientation_weight"] = p.get("planning/objective/minimum_time_via_points/orientation_weight", 0.0)
        kw["max_via_points"] = 16
    else:
        raise ParamError(f"Unknown objective type '{objective}' specified ('planning/objective/type').")      # :636
    # terminal cost (:641-672) and terminal constraint (:674-713)
    tcost = p.get("planning/terminal_cost/type", "none")
    if tcost == "quadratic":
        kw["Qf"] = _weights(p.get("planning/terminal_cost/quadratic/final_state_weights", []), 3, "Final state weights dimension invalid. Must be either 3 x 1 or 3 x 3.")   # :664
    elif tcost != "none":
        raise ParamError(f"Unknown terminal_cost type '{tcost}' specified ('planning/terminal_cost/type').")     # :670
    tcon = p.get("planning/terminal_constraint/type", "none")
    if tcon == "l2_ball":
        kw["terminal_ball_S"] = _weights(p.get("planning/terminal_constraint/l2_ball/weight_matrix", []), 3, "l2-ball weight_matrix dimensions invalid. Must be either 3 x 1 or 3 x 3.")  # :699
        kw["terminal_ball_gamma"] = p.get("planning/terminal_constraint/l2_ball/radius", 1.0)
    elif tcon != "none":
        raise ParamError(f"Unknown terminal_constraint type '{tcon}' specified ('planning/terminal_constraint/type').")   # :711

    # ---- collision avoidance (:715-729)
    kw["min_obstacle_dist"] = p.get("collision_avoidance/min_obstacle_dist", 0.5)
    kw["enable_dynamic_obstacles"] = p.get("collision_avoidance/enable_dynamic_obstacles", False)
    kw["force_inclusion_dist"] = p.get("collision_avoidance/force_inclusion_dist", 0.5)
    kw["cutoff_dist"] = p.get("collision_avoidance/cutoff_dist", 2.0)

    # ---- footprint (src/mpc_local_planner_ros.cpp:890-1001): every malformed model falls back to the point model, as there
    kw.update(_footprint(p, costmap_footprint, notes))

    # ---- options of the Controller facade (src/controller.cpp:70-88; defaults: include/mpc_local_planner/controller.h:124-143)
    ctrl["outer_ocp_iterations"] = p.get("controller/outer_ocp_iterations", 1)
    ctrl["force_reinit_new_goal_dist"] = p.get("controller/force_reinit_new_goal_dist", 1.0)
    ctrl["force_reinit_new_goal_angular"] = p.get("controller/force_reinit_new_goal_angular", 0.5 * math.pi)
    ctrl["allow_init_with_backward_motion"] = p.get("controller/allow_init_with_backward_motion", True)
    ctrl["force_reinit_num_steps"] = p.get("controller/force_reinit_num_steps", 0)
    ctrl["prefer_x_feedback"] = p.get("controller/prefer_x_feedback", False)
    ctrl["publish_ocp_results"] = p.get("controller/publish_ocp_results", False)
    ctrl["print_cpu_time"] = p.get("controller/print_cpu_time", False)

    kw.update(sizing)
    if kw.get("max_obstacles", 0) <= 0:
        kw.pop("max_vertices", None)
    return A.make_config(**kw), ctrl, notes


def plugin_options_from_params(params: dict, move_base_params: dict | None = None) -> dict:
    """The parameters that MpcLocalPlannerROS::initialize reads for ITSELF (src/mpc_local_planner_ros.cpp:96-125, :220), with the in-code defaults of
    include/mpc_local_planner/mpc_local_planner_ros.h:369-391 -- what a binding needs around the solve: goal tolerances, plan pruning / look-ahead, via-point separation
    (plugin_inputs.via_points_from_plan), the costmap scan (BatchSolver.costmap_to_obstacles: include_costmap_obstacles, costmap_obstacles_behind_robot_dist), the
    feasibility check (BatchSolver.check_feasibility: collision_check_min_resolution_angular, collision_check_no_poses).  controller_frequency is move_base's own
    parameter (the control period handed to step() is its inverse, :380): taken from move_base_params."""
    p = _Reader(params)
    mb = _Reader(move_base_params or {})
    return {
        "xy_goal_tolerance": p.get("controller/xy_goal_tolerance", 0.2), "yaw_goal_tolerance": p.get("controller/yaw_goal_tolerance", 0.1),
        "global_plan_overwrite_orientation": p.get("controller/global_plan_overwrite_orientation", True),
        "global_plan_prune_distance": p.get("controller/global_plan_prune_distance", 1.0),
        "max_global_plan_lookahead_dist": p.get("controller/max_global_plan_lookahead_dist", 1.5),
        "global_plan_viapoint_sep": p.get("controller/global_plan_viapoint_sep", -1.0),
        "odom_topic": p.get("odom_topic", "odom"),
        "is_footprint_dynamic": p.get("footprint_model/is_footprint_dynamic", False),
        "include_costmap_obstacles": p.get("collision_avoidance/include_costmap_obstacles", True),
        "costmap_obstacles_behind_robot_dist": p.get("collision_avoidance/costmap_obstacles_behind_robot_dist", 1.5),
        "collision_check_no_poses": p.get("collision_avoidance/collision_check_no_poses", -1),
        "collision_check_min_resolution_angular": p.get("collision_avoidance/collision_check_min_resolution_angular", math.pi),
        "costmap_converter_plugin": p.get("costmap_converter_plugin", ""), "costmap_converter_rate": p.get("costmap_converter_rate", 5.0),
        "costmap_converter_spin_thread": p.get("costmap_converter_spin_thread", True),
        "controller_frequency": mb.get("controller_frequency", 10.0),
    }


def _is_number(v) -> bool:
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def _footprint(p: _Reader, costmap_footprint, notes: list[str]) -> dict:
    point = {"footprint_kind": FOOTPRINT_POINT}
    if not p.has("footprint_model/type"):
        return point                                           # :894-898
    kind = p.get("footprint_model/type", "point")
    if kind == "costmap_2d":
        if not costmap_footprint:
            notes.append("footprint_model/type costmap_2d without a costmap footprint: point model (as the reference without a costmap)")
            return point
        return {"footprint_kind": FOOTPRINT_POLYGON, "footprint_vertices": [tuple(map(float, v)) for v in costmap_footprint]}
    if kind == "point":
        return point
    if kind == "circular":
        if not p.has("footprint_model/radius"):
            notes.append("Footprint model 'circular' cannot be loaded: footprint_model/radius does not exist. Using point-model instead.")
            return point
        radius = p.get("footprint_model/radius", None)
        if not _is_number(radius):            # getParam(double) fails on anything that is not a number (:925)
            notes.append("Footprint model 'circular' cannot be loaded: footprint_model/radius does not exist. Using point-model instead.")
            return point
        return {"footprint_kind": FOOTPRINT_CIRCLE, "footprint_radius": float(radius)}
    if kind == "line":
        a, b = p.get("footprint_model/line_start", None), p.get("footprint_model/line_end", None)
        numeric = lambda v: isinstance(v, (list, tuple)) and all(_is_number(e) for e in v)      # noqa: E731    a list with text in it reads as empty (:944-946)
        if not numeric(a) or not numeric(b) or len(a) != 2 or len(b) != 2:
            notes.append("Footprint model 'line' cannot be loaded: line_start / line_end missing or not 2D. Using point-model instead.")
            return point
        return {"footprint_kind": FOOTPRINT_LINE, "footprint_params": (float(a[0]), float(a[1]), float(b[0]), float(b[1]))}
    if kind == "two_circles":
        keys = ("front_offset", "front_radius", "rear_offset", "rear_radius")
        if not all(p.has(f"footprint_model/{k}") for k in keys):
            notes.append("Footprint model 'two_circles' cannot be loaded: front_offset, front_radius, rear_offset and rear_radius are needed. Using point-model instead.")
            return point
        vals = [p.get(f"footprint_model/{k}", None) for k in keys]
        if not all(_is_number(v) for v in vals):
            # the reference only asks hasParam (:964-965) and would go on with uninitialised numbers: treated as "cannot be loaded" here
            notes.append("Footprint model 'two_circles' cannot be loaded: front_offset, front_radius, rear_offset and rear_radius must be numbers. Using point-model instead.")
            return point
        return {"footprint_kind": FOOTPRINT_TWO_CIRCLES, "footprint_params": tuple(float(v) for v in vals)}
    if kind == "polygon":
        v = p.get("footprint_model/vertices", None)
        ok = isinstance(v, (list, tuple)) and len(v) >= 3 and all(isinstance(q, (list, tuple)) and len(q) == 2 and all(_is_number(e) for e in q) for q in v)
        if not ok:
            # makeFootprintFromXMLRPC / getNumberFromXMLRPC (:1046-1095) throw for fewer than 3 points, points that are not [x, y], coordinates that are not numbers
            notes.append("Footprint model 'polygon' cannot be loaded: footprint_model/vertices must be a list of at least 3 [x, y] points. Using point-model instead.")
            return point
        if len(v) > 16:
            raise ParamNotImplemented("footprint_model/vertices: more than 16 vertices (mpc_config.footprint_vertices)")
        return {"footprint_kind": FOOTPRINT_POLYGON, "footprint_vertices": [(float(q[0]), float(q[1])) for q in v]}
    notes.append(f"Footprint model '{kind}' unknown. Using point-model instead.")
    return point


def config_from_yaml(path: str, namespace: str | None = PLUGIN_NAMESPACE, costmap_footprint=None, **sizing):
    """Reads a parameter file of the reference (the plugin's keys under `namespace`, as in mpc_local_planner_examples/cfg/**; namespace=None
    or a file without that top-level key: the keys start at the top, as in cfg/test_mpc_optim_node.yaml)."""
    import yaml
    with open(path) as f:
        tree = yaml.safe_load(f) or {}
    if namespace and isinstance(tree.get(namespace), dict):
        tree = tree[namespace]
    return config_from_params(tree, costmap_footprint=costmap_footprint, **sizing)

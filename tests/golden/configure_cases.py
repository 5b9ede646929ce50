"""Parameter dictionaries that drive the reference's Controller::configure through every branch (src/controller.cpp:58-100, :225-805): the three example files of
mpc_local_planner_examples restated key by key would need /root/reference at test time, so the cases are built here; tests/golden/make_ref_vectors.py runs the
REFERENCE's configure() on each (oracle/ref_wrap_controller.cpp) and records what it built in tests/golden/ref_configure.json."""
import copy
import math


def _set(tree, key, value):
    node = tree
    parts = key.split("/")
    for p in parts[:-1]:
        node = node.setdefault(p, {})
    node[parts[-1]] = value
    return tree


def base_carlike():
    """the values of mpc_local_planner_examples/cfg/carlike/mpc_local_planner_params.yaml that reach configure()"""
    t = {}
    for k, v in {
        "controller/outer_ocp_iterations": 5, "controller/xy_goal_tolerance": 0.2, "controller/publish_ocp_results": False, "controller/print_cpu_time": False,
        "controller/force_reinit_new_goal_dist": 1.0, "controller/force_reinit_new_goal_angular": 1.57, "controller/force_reinit_num_steps": 0,
        "controller/prefer_x_feedback": False, "controller/allow_init_with_backward_motion": True,
        "robot/type": "simple_car", "robot/simple_car/wheelbase": 0.4, "robot/simple_car/front_wheel_driving": False, "robot/simple_car/max_vel_x": 0.4,
        "robot/simple_car/max_vel_x_backwards": 0.2, "robot/simple_car/max_steering_angle": 1.4, "robot/simple_car/acc_lim_x": 0.5, "robot/simple_car/dec_lim_x": 0.5,
        "robot/simple_car/max_steering_rate": 0.5,
        "grid/type": "fd_grid", "grid/grid_size_ref": 20, "grid/dt_ref": 0.3, "grid/xf_fixed": [True, True, True], "grid/warm_start": True,
        "grid/collocation_method": "forward_differences", "grid/cost_integration_method": "left_sum", "grid/variable_grid/enable": True, "grid/variable_grid/min_dt": 0.0,
        "grid/variable_grid/max_dt": 10.0, "grid/variable_grid/grid_adaptation/enable": True, "grid/variable_grid/grid_adaptation/dt_hyst_ratio": 0.1,
        "grid/variable_grid/grid_adaptation/min_grid_size": 2, "grid/variable_grid/grid_adaptation/max_grid_size": 50,
        "planning/objective/type": "minimum_time", "planning/terminal_cost/type": "none", "planning/terminal_constraint/type": "none",
        "collision_avoidance/min_obstacle_dist": 0.27, "collision_avoidance/enable_dynamic_obstacles": False, "collision_avoidance/force_inclusion_dist": 0.5,
        "collision_avoidance/cutoff_dist": 2.5,
        "solver/type": "ipopt", "solver/ipopt/iterations": 100, "solver/ipopt/max_cpu_time": -1.0,
        "solver/ipopt/ipopt_string_options": {"linear_solver": "mumps", "hessian_approximation": "limited-memory"},
    }.items():
        _set(t, k, v)
    return t


def cases():
    out = {}

    def add(name, base, **changes):
        t = copy.deepcopy(base)
        for k, v in changes.items():
            _set(t, k.replace("__", "/"), v)
        out[name] = t
    car = base_carlike()
    add("defaults_only", {})
    add("carlike_example", car)
    add("carlike_front_wheel", car, robot__simple_car__front_wheel_driving=True)
    add("carlike_negative_backwards_and_dec", car, robot__simple_car__max_vel_x_backwards=-0.3, robot__simple_car__dec_lim_x=-0.7)
    add("carlike_no_rate_limits", car, robot__simple_car__acc_lim_x=0.0, robot__simple_car__dec_lim_x=-0.0, robot__simple_car__max_steering_rate=-1.0)
    add("carlike_int_for_double", car, robot__simple_car__max_vel_x=1, grid__dt_ref=1, collision_avoidance__cutoff_dist=3)
    add("carlike_double_for_int", car, grid__grid_size_ref=24.6, solver__ipopt__iterations=59.5)
    add("carlike_int_for_bool_keeps_default", car, grid__warm_start=0, grid__variable_grid__enable=0)
    add("carlike_numeric_options", car, solver__ipopt__ipopt_numeric_options={"tol": 1.0e-4, "mu_init": 0.05}, solver__ipopt__ipopt_integer_options={"max_iter": 77})
    add("carlike_numeric_option_as_text", car, solver__ipopt__ipopt_numeric_options={"tol": "1e-4"})      # what a YAML 1.1 loader makes of `tol: 1e-4`
    add("unicycle", {}, robot__type="unicycle", robot__unicycle__max_vel_x=0.5, robot__unicycle__max_vel_x_backwards=0.1, robot__unicycle__max_vel_theta=0.7,
        robot__unicycle__acc_lim_x=0.2, robot__unicycle__dec_lim_x=0.3, robot__unicycle__acc_lim_theta=0.4)
    add("unicycle_negative_fixups", {}, robot__unicycle__max_vel_x_backwards=-0.25, robot__unicycle__dec_lim_x=-0.3)
    add("bicycle", {}, robot__type="kinematic_bicycle_vel_input", robot__kinematic_bicycle_vel_input__length_rear=1.2, robot__kinematic_bicycle_vel_input__length_front=0.9,
        robot__kinematic_bicycle_vel_input__max_vel_x=2.0, robot__kinematic_bicycle_vel_input__max_steering_angle=0.6, robot__kinematic_bicycle_vel_input__max_steering_rate=0.3)
    add("bicycle_defaults", {}, robot__type="kinematic_bicycle_vel_input")
    add("unknown_robot", {}, robot__type="hovercraft")
    add("fixed_grid", car, grid__variable_grid__enable=False)
    add("variable_grid_no_adaptation", car, grid__variable_grid__grid_adaptation__enable=False, grid__variable_grid__min_dt=0.01, grid__variable_grid__max_dt=2.0)
    add("adaptation_settings", car, grid__variable_grid__grid_adaptation__max_grid_size=80, grid__variable_grid__grid_adaptation__min_grid_size=5,
        grid__variable_grid__grid_adaptation__dt_hyst_ratio=0.25, grid__grid_size_ref=30, grid__dt_ref=0.15)
    add("xf_partially_fixed_no_warm_start", car, grid__xf_fixed=[True, True, False], grid__warm_start=False)
    add("xf_fixed_wrong_size", car, grid__xf_fixed=[True, True])
    add("unknown_grid_type", car, grid__type="multiple_shooting")
    add("midpoint", car, grid__collocation_method="midpoint_differences")
    add("crank_nicolson_trapezoid", car, grid__collocation_method="crank_nicolson_differences", grid__cost_integration_method="trapezoidal_rule")
    add("unknown_collocation", car, grid__collocation_method="simpson")
    # solver/type unknown is NOT among the cases: configureSolver returns an empty pointer and configureOcp dereferences it (src/controller.cpp:552,
    # `_solver->isLsqSolver()`): the reference crashes instead of returning false.  The package rejects it with ParamError.
    add("lsq_lm_solver", car, solver__type="lsq_lm", solver__lsq_lm__iterations=7, solver__lsq_lm__weight_init_eq=3.0, solver__lsq_lm__weight_adapt_factor_ineq=1.5,
        solver__lsq_lm__weight_adapt_max_bounds=100.0)
    quad = copy.deepcopy(car)
    _set(quad, "planning/objective/type", "quadratic_form")
    _set(quad, "planning/objective/quadratic_form/state_weights", [2.0, 2.0, 0.25])
    _set(quad, "planning/objective/quadratic_form/control_weights", [0.1, 0.05])
    add("quadratic_diagonal", quad)
    add("quadratic_integral_trapezoid", quad, planning__objective__quadratic_form__integral_form=True, grid__cost_integration_method="trapezoidal_rule")
    add("quadratic_full_matrices", quad, planning__objective__quadratic_form__state_weights=[2.0, 0.1, 0.2, 0.3, 3.0, 0.4, 0.5, 0.6, 1.0],
        planning__objective__quadratic_form__control_weights=[0.5, 0.05, 0.07, 0.2])
    add("quadratic_int_weights", quad, planning__objective__quadratic_form__state_weights=[2, 2, 1], planning__objective__quadratic_form__control_weights=[1, 0.5])
    add("quadratic_state_only", quad, planning__objective__quadratic_form__control_weights=[0.0, 0.0])
    add("quadratic_controls_only", quad, planning__objective__quadratic_form__state_weights=[0.0, 0.0, 0.0])
    add("quadratic_controls_hybrid", quad, planning__objective__quadratic_form__state_weights=[0.0, 0.0, 0.0], planning__objective__quadratic_form__hybrid_cost_minimum_time=True)
    add("quadratic_hybrid_with_state_weights", quad, planning__objective__quadratic_form__hybrid_cost_minimum_time=True)
    add("quadratic_all_zero", quad, planning__objective__quadratic_form__state_weights=[0.0, 0.0, 0.0], planning__objective__quadratic_form__control_weights=[0.0, 0.0])
    add("quadratic_bad_state_weights", quad, planning__objective__quadratic_form__state_weights=[1.0, 2.0])
    add("quadratic_bad_control_weights", quad, planning__objective__quadratic_form__control_weights=[1.0, 2.0, 3.0])
    add("quadratic_missing_weights", car, planning__objective__type="quadratic_form")
    add("terminal_cost", quad, planning__terminal_cost__type="quadratic", planning__terminal_cost__quadratic__final_state_weights=[10.0, 10.0, 0.5], grid__xf_fixed=[False, False, False])
    add("terminal_cost_full", quad, planning__terminal_cost__type="quadratic", planning__terminal_cost__quadratic__final_state_weights=[10.0, 1.0, 0.0, 1.0, 10.0, 0.2, 0.0, 0.2, 0.5])
    add("terminal_cost_bad", quad, planning__terminal_cost__type="quadratic", planning__terminal_cost__quadratic__final_state_weights=[1.0])
    add("terminal_cost_unknown", quad, planning__terminal_cost__type="cubic")
    add("terminal_ball", quad, planning__terminal_constraint__type="l2_ball", planning__terminal_constraint__l2_ball__weight_matrix=[1.0, 1.0, 0.1],
        planning__terminal_constraint__l2_ball__radius=0.3, grid__xf_fixed=[False, False, False])
    add("terminal_ball_default_radius_full", quad, planning__terminal_constraint__type="l2_ball",
        planning__terminal_constraint__l2_ball__weight_matrix=[1.0, 0.1, 0.0, 0.1, 1.0, 0.0, 0.0, 0.0, 0.1])
    add("terminal_ball_bad", quad, planning__terminal_constraint__type="l2_ball", planning__terminal_constraint__l2_ball__weight_matrix=[1.0, 1.0])
    add("terminal_constraint_unknown", quad, planning__terminal_constraint__type="box")
    add("via_points", car, planning__objective__type="minimum_time_via_points", planning__objective__minimum_time_via_points__position_weight=10.5,
        planning__objective__minimum_time_via_points__orientation_weight=0.2, planning__objective__minimum_time_via_points__via_points_ordered=True)
    add("via_points_defaults", car, planning__objective__type="minimum_time_via_points")
    add("unknown_objective", car, planning__objective__type="maximum_comfort")
    add("collision_settings", car, collision_avoidance__min_obstacle_dist=0.4, collision_avoidance__enable_dynamic_obstacles=True, collision_avoidance__force_inclusion_dist=0.8,
        collision_avoidance__cutoff_dist=4.0)
    add("controller_settings", car, controller__outer_ocp_iterations=3, controller__force_reinit_new_goal_dist=2.5, controller__force_reinit_new_goal_angular=0.5 * math.pi,
        controller__allow_init_with_backward_motion=False, controller__force_reinit_num_steps=7, controller__prefer_x_feedback=True, controller__publish_ocp_results=True,
        controller__print_cpu_time=True)
    return out

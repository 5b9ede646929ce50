"""ctypes mirror of include/mpc_hip.h (struct mpc_config and the enum values).

Pure data definitions -- shared by the product binding (mpc_local_planner_amd/_lib.py)
and by tests.  No solver code here.
"""
from __future__ import annotations

import ctypes as C

MODEL_UNICYCLE = 0
MODEL_SIMPLE_CAR = 1
MODEL_SIMPLE_CAR_FRONT = 2
MODEL_KINEMATIC_BICYCLE = 3

COLLOC_FORWARD = 0
COLLOC_MIDPOINT = 1
COLLOC_CRANK_NICOLSON = 2

OBJ_MIN_TIME = 0
OBJ_QUADRATIC = 1
OBJ_MIN_TIME_VIA_POINTS = 2

FP64 = 0
FP32 = 1
MIXED = 2

STATUS_NAMES = {0: "converged", 1: "max_iter", 2: "linesearch_failed", 3: "linsolve_failed", 4: "numerical_error", 5: "time_limit"}

MPC_OK = 0
MPC_EINVAL = -1
MPC_ENODEV = -2
MPC_ENOMEM = -3
MPC_EHIP = -4
MPC_EBATCH = -5

INF = 1e30

COST_LEFT_SUM, COST_TRAPEZOIDAL = 0, 1
MU_ADAPTIVE, MU_MONOTONE = 0, 1          # enum mpc_mu_strategy
LS_DEFAULT, LS_MERIT, LS_FILTER = 0, 1, 2   # enum mpc_line_search
STAGE_AUTO, STAGE_LDS, STAGE_GLOBAL = 0, 1, 2      # enum mpc_stage_data: where a solve keeps its factorisation data

CAND_REFERENCE = 0
CAND_TRAVEL = 1
CAND_TRAVEL_REVERSE = 2
CAND_BLEND = 3
CAND_BLEND_REVERSE = 4
CAND_HERMITE_FF, CAND_HERMITE_RR, CAND_HERMITE_FR, CAND_HERMITE_RF = 5, 6, 7, 8
MAX_CANDIDATES = 4


class MpcConfig(C.Structure):
    """struct mpc_config (include/mpc_hip.h); field-for-field."""
    _fields_ = [
        ("model", C.c_int32),
        ("model_params", C.c_double * 4),
        ("n", C.c_int32),
        ("dt_ref", C.c_double),
        ("dt_free", C.c_int32),
        ("dt_lb", C.c_double),
        ("dt_ub", C.c_double),
        ("xf_fixed", C.c_int32 * 3),
        ("collocation", C.c_int32),
        ("objective", C.c_int32),
        ("Q", C.c_double * 3),
        ("R", C.c_double * 2),
        ("integral_form", C.c_int32),
        ("has_Qf", C.c_int32),
        ("Qf", C.c_double * 3),
        ("u_lb", C.c_double * 2),
        ("u_ub", C.c_double * 2),
        ("du_lb", C.c_double * 2),
        ("du_ub", C.c_double * 2),
        ("max_iter", C.c_int32),
        ("tol", C.c_double),
        ("mu_init", C.c_double),
        ("precision", C.c_int32),
        ("min_obstacle_dist", C.c_double),
        ("force_inclusion_dist", C.c_double),
        ("cutoff_dist", C.c_double),
        ("footprint_kind", C.c_int32),
        ("footprint_radius", C.c_double),
        ("max_obstacles", C.c_int32),
        ("max_vertices", C.c_int32),
        ("max_obstacle_rows", C.c_int32),
        ("mu_init_warm", C.c_double),
        ("terminal_ball", C.c_int32),
        ("terminal_ball_S", C.c_double * 3),
        ("terminal_ball_gamma", C.c_double),
        ("vp_position_weight", C.c_double),
        ("vp_orientation_weight", C.c_double),
        ("via_points_ordered", C.c_int32),
        ("max_via_points", C.c_int32),
        ("footprint_n_vertices", C.c_int32),
        ("footprint_vertices", C.c_double * 32),
        ("enable_dynamic_obstacles", C.c_int32),
        ("footprint_params", C.c_double * 4),
        ("n_candidates", C.c_int32),
        ("candidate_kind", C.c_int32 * 4),
        ("candidate_max_iter", C.c_int32 * 4),
        ("candidate_blend", C.c_int32),
        ("dual_warm_start", C.c_int32),
        ("mu_init_dual", C.c_double),
        ("candidate_param", C.c_double * 4),
        ("hessian_mode", C.c_int32),
        ("hybrid_cost_minimum_time", C.c_int32),
        ("cost_integration", C.c_int32),
        ("mu_strategy", C.c_int32), ("stage_data", C.c_int32), ("two_wave_min_batch", C.c_int32), ("line_search", C.c_int32),
        ("Q_offdiag", C.c_double * 3),
        ("R_offdiag", C.c_double),
        ("Qf_offdiag", C.c_double * 3),
        ("terminal_ball_S_offdiag", C.c_double * 3),
        ("acceptable_tol", C.c_double),
        ("acceptable_iter", C.c_int32),
        ("max_time_us", C.c_int32),
    ]


class MpcObstacles(C.Structure):
    """struct mpc_obstacles (include/mpc_hip.h): raw addresses (host or device)."""
    _fields_ = [
        ("n_obstacles", C.c_void_p),
        ("n_vertices", C.c_void_p),
        ("vertices", C.c_void_p),
        ("radius", C.c_void_p),
        ("velocity", C.c_void_p),
    ]


def _diag_offdiag(w, dim):
    """a weight given as its diagonal or as a full dim x dim matrix -> (diagonal, off-diagonal terms (0,1)[, (0,2), (1,2)] of the symmetric part)"""
    rows = [list(r) if hasattr(r, "__len__") else None for r in w]
    if rows and rows[0] is not None:
        m = [[0.5 * (float(rows[i][j]) + float(rows[j][i])) for j in range(dim)] for i in range(dim)]
        return [m[i][i] for i in range(dim)], ([m[0][1], m[0][2], m[1][2]] if dim == 3 else [m[0][1]])
    return [float(v) for v in w], [0.0] * (3 if dim == 3 else 1)


def make_config(model=MODEL_UNICYCLE, model_params=(0.5, 1.0), n=20, dt_ref=0.3, dt_free=True, dt_lb=0.0, dt_ub=10.0,
                xf_fixed=(True, True, True), objective=OBJ_MIN_TIME, Q=(0, 0, 0), R=(0, 0), integral_form=False, Qf=None,
                u_lb=(-0.2, -0.3), u_ub=(0.4, 0.3), du_lb=(-INF, -INF), du_ub=(INF, INF), max_iter=100, tol=1e-8,
                mu_init=0.1, precision=FP64, min_obstacle_dist=0.5, force_inclusion_dist=0.5, cutoff_dist=2.0,
                footprint_kind=0, footprint_radius=0.0, max_obstacles=0, max_vertices=1, max_obstacle_rows=4, mu_init_warm=0.0, collocation=COLLOC_FORWARD,
                terminal_ball_S=None, terminal_ball_gamma=1.0, vp_position_weight=1e-3, vp_orientation_weight=0.0,
                via_points_ordered=False, max_via_points=0, footprint_params=(0.0, 0.0, 0.0, 0.0),
                enable_dynamic_obstacles=False, footprint_vertices=(), candidates=(), candidate_max_iter=(), candidate_blend=0, dual_warm_start=False, mu_init_dual=0.0, candidate_param=(), hessian_mode=0,
                hybrid_cost_minimum_time=False, cost_integration=COST_LEFT_SUM, acceptable_tol=0.0, acceptable_iter=0, mu_strategy=0, max_cpu_time=0.0, stage_data=STAGE_AUTO, two_wave_min_batch=0, line_search=LS_DEFAULT) -> MpcConfig:
    """Q, R, Qf, terminal_ball_S: the diagonal (3 / 2 / 3 / 3 values) or the full matrix (nested 3 x 3 / 2 x 2; its symmetric part is used)."""
    c = MpcConfig()
    c.model = model
    mp = list(model_params) + [0.0] * 4
    for i in range(4):
        c.model_params[i] = mp[i]
    c.n = n
    c.dt_ref = dt_ref
    c.dt_free = int(bool(dt_free))
    c.dt_lb, c.dt_ub = dt_lb, dt_ub
    Qd, Qo = _diag_offdiag(Q, 3)
    Qfd, Qfo = _diag_offdiag(Qf if Qf is not None else (0, 0, 0), 3)
    Rd, Ro = _diag_offdiag(R, 2)
    for i in range(3):
        c.xf_fixed[i] = int(bool(xf_fixed[i]))
        c.Q[i], c.Q_offdiag[i] = Qd[i], Qo[i]
        c.Qf[i], c.Qf_offdiag[i] = Qfd[i], Qfo[i]
    c.R_offdiag = Ro[0]
    c.collocation = collocation
    c.objective = objective
    c.integral_form = int(bool(integral_form))
    c.has_Qf = int(Qf is not None)
    for j in range(2):
        c.R[j] = Rd[j]
        c.u_lb[j], c.u_ub[j] = u_lb[j], u_ub[j]
        c.du_lb[j], c.du_ub[j] = du_lb[j], du_ub[j]
    c.max_iter = max_iter
    c.tol = tol
    c.mu_init = mu_init
    c.precision = precision
    c.min_obstacle_dist, c.force_inclusion_dist, c.cutoff_dist = min_obstacle_dist, force_inclusion_dist, cutoff_dist
    c.footprint_kind, c.footprint_radius = footprint_kind, footprint_radius
    c.max_obstacles, c.max_vertices, c.max_obstacle_rows = max_obstacles, max_vertices, max_obstacle_rows
    c.mu_init_warm = mu_init_warm
    c.terminal_ball = int(terminal_ball_S is not None)
    Sd, So = _diag_offdiag(terminal_ball_S if terminal_ball_S is not None else (0, 0, 0), 3)
    for i in range(3):
        c.terminal_ball_S[i], c.terminal_ball_S_offdiag[i] = Sd[i], So[i]
    c.terminal_ball_gamma = terminal_ball_gamma
    c.vp_position_weight, c.vp_orientation_weight = vp_position_weight, vp_orientation_weight
    c.via_points_ordered, c.max_via_points = int(bool(via_points_ordered)), max_via_points
    for i in range(4):
        c.footprint_params[i] = footprint_params[i]
    c.enable_dynamic_obstacles = int(bool(enable_dynamic_obstacles))
    fv = []
    for row in footprint_vertices:          # flat (x0, y0, x1, ...) or nested ((x0, y0), ...)
        fv.extend([float(v) for v in row] if hasattr(row, "__len__") else [float(row)])
    c.footprint_n_vertices = len(fv) // 2
    for i, x in enumerate(fv[:32]):
        c.footprint_vertices[i] = x
    # candidate initial trajectories: kinds (CAND_*) in priority order, per-candidate iteration caps (0 / missing -> max_iter)
    c.n_candidates = len(candidates)
    for i, k in enumerate(candidates[:MAX_CANDIDATES]):
        c.candidate_kind[i] = int(k)
        c.candidate_max_iter[i] = int(candidate_max_iter[i]) if i < len(candidate_max_iter) else 0
        c.candidate_param[i] = float(candidate_param[i]) if i < len(candidate_param) else 0.0
    c.candidate_blend = int(candidate_blend)
    c.dual_warm_start = int(bool(dual_warm_start))
    c.mu_init_dual = float(mu_init_dual)
    c.hessian_mode = int(hessian_mode)
    c.mu_strategy = int(mu_strategy)          # MU_ADAPTIVE (0, default) | MU_MONOTONE
    c.stage_data = int(stage_data)            # STAGE_AUTO (0, default) | STAGE_LDS | STAGE_GLOBAL
    c.line_search = int(line_search)          # LS_DEFAULT (0) | LS_MERIT | LS_FILTER
    c.two_wave_min_batch = int(two_wave_min_batch)      # 0: the default threshold (4096 instances per launch), negative: never the two-waves-per-SIMD kernel
    c.hybrid_cost_minimum_time = int(bool(hybrid_cost_minimum_time))
    c.cost_integration = int(cost_integration)
    c.acceptable_tol, c.acceptable_iter = float(acceptable_tol), int(acceptable_iter)      # 0 = Ipopt's defaults (1e-6, 15), negative = off
    # solver/ipopt/max_cpu_time (seconds; <= 0 = none; beyond the int32 range of microseconds, ~2147 s, = none: same rule as include/mpc_params.hpp)
    c.max_time_us = int(max_cpu_time * 1e6 + 0.5) if max_cpu_time and 0 < max_cpu_time and max_cpu_time * 1e6 + 0.5 < 2147483647.0 else 0
    return c


def config_carlike_min_time(n=50, **kw) -> MpcConfig:
    """BASELINE.json config 2: mpc_local_planner_examples/cfg/carlike/mpc_local_planner_params.yaml:7-16,46-65."""
    d = dict(model=MODEL_SIMPLE_CAR, model_params=(0.4,), n=n, dt_ref=0.3, dt_free=True, dt_lb=0.0, dt_ub=10.0,
             xf_fixed=(True, True, True), objective=OBJ_MIN_TIME, u_lb=(-0.2, -1.4), u_ub=(0.4, 1.4),
             du_lb=(-0.5, -0.5), du_ub=(0.5, 0.5))
    d.update(kw)
    return make_config(**d)


def config_unicycle_quadratic(n=20, **kw) -> MpcConfig:
    """BASELINE.json config 1: .../cfg/diff_drive/mpc_local_planner_params_quadratic_form.yaml:7-14,33-41,53-61."""
    d = dict(model=MODEL_UNICYCLE, model_params=(0.0,), n=n, dt_ref=0.3, dt_free=False,
             xf_fixed=(False, False, False), objective=OBJ_QUADRATIC, Q=(2.0, 2.0, 0.25), R=(0.1, 0.05),
             Qf=(10.0, 10.0, 0.5), u_lb=(-0.2, -0.3), u_ub=(0.4, 0.3), du_lb=(-0.2, -0.2), du_ub=(0.2, 0.2),
             min_obstacle_dist=0.2, force_inclusion_dist=0.5, cutoff_dist=2.5)
    d.update(kw)
    return make_config(**d)


def config_bicycle_min_time(n=120, **kw) -> MpcConfig:
    """BASELINE.json config 5; lr = lf = 1.0 (src/controller.cpp:366-369)."""
    return make_config(model=MODEL_KINEMATIC_BICYCLE, model_params=(1.0, 1.0), n=n, dt_ref=0.3, dt_free=True,
                       xf_fixed=(True, True, True), objective=OBJ_MIN_TIME, u_lb=(-0.2, -1.5), u_ub=(0.4, 1.5),
                       du_lb=(-0.5, -0.5), du_ub=(0.5, 0.5), **kw)

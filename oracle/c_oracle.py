"""ORACLE (test infrastructure only).  ctypes wrapper of oracle/mpc_oracle.c."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "libmpc_oracle.so")


class OracleConfig(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("model_params", C.c_double * 4), ("n", C.c_int32), ("dt_ref", C.c_double),
        ("dt_free", C.c_int32), ("dt_lb", C.c_double), ("dt_ub", C.c_double), ("xf_fixed", C.c_int32 * 3),
        ("objective", C.c_int32), ("Q", C.c_double * 3), ("R", C.c_double * 2), ("has_Qf", C.c_int32),
        ("Qf", C.c_double * 3), ("u_lb", C.c_double * 2), ("u_ub", C.c_double * 2), ("du_lb", C.c_double * 2),
        ("du_ub", C.c_double * 2), ("max_iter", C.c_int32), ("tol", C.c_double), ("mu_init", C.c_double),
        ("collocation", C.c_int32),
        ("via", C.c_int32), ("vp_ordered", C.c_int32), ("vp_wp", C.c_double), ("vp_wo", C.c_double),
        ("ball", C.c_int32), ("ball_S", C.c_double * 3), ("ball_gamma", C.c_double),
        ("integral", C.c_int32),
        ("hessian_mode", C.c_int32),
        ("hybrid", C.c_int32), ("trapezoid", C.c_int32),
        ("Qo", C.c_double * 3), ("Ro", C.c_double), ("Qfo", C.c_double * 3), ("So", C.c_double * 3),
        ("acceptable_tol", C.c_double), ("acceptable_iter", C.c_int32), ("mu_strategy", C.c_int32), ("line_search", C.c_int32),
    ]


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(_HERE, "mpc_oracle.c")):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.oracle_solve_batch.restype = C.c_int
        _lib.oracle_solve_batch_obst.restype = C.c_int
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


LINE_SEARCH_DEFAULT = 1      # 0 l1 merit, 1 Ipopt's filter (what mpc_config.line_search = MPC_LS_DEFAULT resolves to in the library: mpc_problem.hpp)


def from_nlp_config(cfg, max_iter=100, tol=1e-8, mu_init=0.1, hessian_mode=0, acceptable_tol=0.0, acceptable_iter=0, mu_strategy=0, line_search=None) -> OracleConfig:
    """oracle.se2_nlp.OcpConfig -> OracleConfig.  acceptable_tol / acceptable_iter: Ipopt's acceptable-level stop (0 = its defaults 1e-6 / 15,
    negative = off), the same fields as mpc_config's."""
    o = OracleConfig()
    o.model = cfg.model
    mp = list(cfg.model_params) + [0.0] * 4
    for i in range(4):
        o.model_params[i] = mp[i]
    o.n, o.dt_ref, o.dt_free, o.dt_lb, o.dt_ub = cfg.n, cfg.dt_ref, int(cfg.dt_free), cfg.dt_lb, cfg.dt_ub
    from . import se2_nlp as _R
    Qm, Rm = _R.weight_matrix(cfg.Q), _R.weight_matrix(cfg.R)
    Qfm = _R.weight_matrix(cfg.Qf) if cfg.Qf is not None else np.zeros((3, 3))
    off = lambda m: (m[0, 1], m[0, 2], m[1, 2])
    for i in range(3):
        o.xf_fixed[i] = int(cfg.xf_fixed[i])
        o.Q[i] = Qm[i, i]
        o.Qf[i] = Qfm[i, i]
        o.Qo[i], o.Qfo[i] = off(Qm)[i], off(Qfm)[i]
    o.objective = cfg.objective
    o.has_Qf = int(cfg.Qf is not None)
    o.Ro = Rm[0, 1]
    o.hybrid = int(bool(getattr(cfg, "hybrid_min_time", False)) and cfg.objective == 1)
    o.trapezoid = int(getattr(cfg, "cost_integration", "left_sum") == "trapezoidal_rule")
    for j in range(2):
        o.R[j] = Rm[j, j]
        o.u_lb[j], o.u_ub[j] = cfg.u_lb[j], cfg.u_ub[j]
        o.du_lb[j], o.du_ub[j] = max(cfg.du_lb[j], -1e30), min(cfg.du_ub[j], 1e30)
    o.max_iter, o.tol, o.mu_init = max_iter, tol, mu_init
    o.hessian_mode = int(hessian_mode)
    o.acceptable_tol, o.acceptable_iter = float(acceptable_tol), int(acceptable_iter)
    o.mu_strategy = int(mu_strategy)      # 0 adaptive (default), 1 monotone
    o.line_search = int(LINE_SEARCH_DEFAULT if line_search is None else line_search)      # 0 l1 merit, 1 filter
    o.collocation = int(getattr(cfg, "collocation", 0))
    o.integral = int(bool(getattr(cfg, "integral_form", False)) and cfg.objective == 1)
    if getattr(cfg, "terminal_ball_S", None) is not None:
        o.ball, o.ball_gamma = 1, float(cfg.terminal_ball_gamma)
        Sm = _R.weight_matrix(cfg.terminal_ball_S)
        for i in range(3):
            o.ball_S[i] = Sm[i, i]
            o.So[i] = off(Sm)[i]
    if cfg.objective == 2:          # minimum_time_via_points = minimum time + via-point terms (set the points with set_via_points)
        o.objective, o.via = 0, 1
        o.vp_ordered, o.vp_wp, o.vp_wo = int(cfg.via_points_ordered), cfg.vp_position_weight, cfg.vp_orientation_weight
    return o


def dual_state(B: int, n: int):
    """zeroed per-instance multiplier state for solve_batch(..., dual_state=..., dual_mu0=...) (the product's dual_warm_start)"""
    lib = _load()
    lib.oracle_dual_words.restype = C.c_int
    return np.zeros((B, int(lib.oracle_dual_words(C.c_int(n)))))


def num_threads() -> int:
    return int(_load().oracle_num_threads())


class OracleObst(C.Structure):
    """struct oracle_obst (oracle/mpc_oracle.c): obstacle handling of a batch, same meaning as the mpc_config fields."""
    _fields_ = [("max_obstacles", C.c_int32), ("max_vertices", C.c_int32), ("max_rows", C.c_int32),
                ("min_obstacle_dist", C.c_double), ("force_inclusion_dist", C.c_double), ("cutoff_dist", C.c_double),
                ("footprint_radius", C.c_double),
                ("footprint_kind", C.c_int32), ("footprint_params", C.c_double * 4), ("footprint_nv", C.c_int32), ("footprint_poly", C.c_double * 32),
                ("dynamic", C.c_int32)]


def obst_from_nlp_config(cfg, max_obstacles: int, max_vertices: int, max_rows: int) -> OracleObst:
    kind = int(getattr(cfg, "footprint_kind", 0))
    fr = cfg.footprint_params[0] if kind == 1 and cfg.footprint_params else 0.0
    o = OracleObst(max_obstacles, max_vertices, max_rows, cfg.min_obstacle_dist, cfg.force_inclusion_dist, cfg.cutoff_dist, fr)
    o.footprint_kind = kind
    if kind in (2, 3):
        for i in range(4):
            o.footprint_params[i] = float(cfg.footprint_params[i])
    if kind == 4:
        o.footprint_nv = len(cfg.footprint_params) // 2
        for i, v in enumerate(cfg.footprint_params[:32]):
            o.footprint_poly[i] = float(v)
    o.dynamic = int(bool(getattr(cfg, "enable_dynamic_obstacles", False)))
    return o


def footprint_row(obst: "OracleObst", pose, vertices, radius: float = 0.0):
    """TEST HOOK (oracle_footprint_row): distance footprint(pose) <-> one obstacle and the analytic derivatives of the row g = d_min - dist as the
    solver uses them.  Returns (dist, a[3] = grad g, hk, h3[3] = hess g [x theta, y theta, theta theta]); hess g (x, y) = -hk (I - a_xy a_xy')."""
    lib = _load()
    pose = np.ascontiguousarray(pose, float)
    v = np.zeros((max(obst.max_vertices, 1), 2)); vv = np.asarray(vertices, float).reshape(-1, 2); v[: len(vv)] = vv
    out = np.zeros(8)
    lib.oracle_footprint_row(C.byref(obst), pose.ctypes.data_as(C.c_void_p), C.c_int(len(vv)), v.ctypes.data_as(C.c_void_p), C.c_double(radius), out.ctypes.data_as(C.c_void_p))
    return out[0], out[1:4].copy(), out[4], out[5:8].copy()


def associate_at(ocfg: OracleConfig, obst: "OracleObst", states, vertices, n_vertices, radius=None, velocity=None):
    """TEST HOOK (oracle_associate_at): the solver's obstacle association on GIVEN grid states (n, 3) for one instance's obstacles (vertices (O, V, 2),
    n_vertices (O,)).  Returns (oi (n, max_rows) obstacle indices, -1 = empty; moving obstacles first, then static ones in the reference's order, dropped)."""
    lib = _load()
    x = np.ascontiguousarray(states, float)
    O, V = obst.max_obstacles, obst.max_vertices
    nv = np.zeros(O, np.int32); vt = np.zeros((O, V, 2)); rd = np.zeros(O); ve = np.zeros((O, 2))
    k = len(n_vertices)
    nv[:k] = n_vertices; vt[:k, :np.asarray(vertices).shape[1]] = vertices
    if radius is not None:
        rd[:k] = radius
    if velocity is not None:
        ve[:k] = velocity
    oi = np.full((ocfg.n, obst.max_rows), -1, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.oracle_associate_at.restype = C.c_int
    dropped = lib.oracle_associate_at(C.byref(ocfg), C.byref(obst), p(x), C.c_int(k), p(nv), p(vt), p(rd), p(ve), p(oi))
    return oi, int(dropped)


def solve_batch(ocfg: OracleConfig, x0, xf, u_prev=None, dt_prev=None, init=None, nthreads=0, obstacles=None, obst: "OracleObst" = None, via=None, rows_dropped=None, dual_state=None, dual_mu0=1e-3):
    """rows_dropped: optional int32 array (B,) that receives the number of clearance rows that did not fit into obst.max_rows.
    obstacles = (n_obstacles (B,), n_vertices (B,O), vertices (B,O,V,2)[, radius (B,O)]) together with obst (OracleObst).
    via = (n_via (B,), via (B,VP,3)) for a config made from objective minimum_time_via_points."""
    lib = _load()
    if dual_state is not None:
        lib.oracle_set_dual_state(dual_state.ctypes.data_as(C.c_void_p), C.c_int(dual_state.shape[1]), C.c_double(dual_mu0))
        try:
            return solve_batch(ocfg, x0, xf, u_prev, dt_prev, init, nthreads, obstacles, obst, via, rows_dropped)
        finally:
            lib.oracle_set_dual_state(None, C.c_int(0), C.c_double(0.0))
    if via is not None:
        nvia = np.ascontiguousarray(via[0], np.int32); vps = np.ascontiguousarray(via[1], float)
        lib.oracle_set_via_points(nvia.ctypes.data_as(C.c_void_p), vps.ctypes.data_as(C.c_void_p), C.c_int(vps.shape[1]))
        try:
            return solve_batch(ocfg, x0, xf, u_prev, dt_prev, init, nthreads, obstacles, obst, rows_dropped=rows_dropped)
        finally:
            lib.oracle_set_via_points(None, None, C.c_int(0))
    x0 = np.ascontiguousarray(x0, float)
    xf = np.ascontiguousarray(xf, float)
    B, n = x0.shape[0], ocfg.n
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    up = np.ascontiguousarray(u_prev, float) if u_prev is not None else None
    dp = np.ascontiguousarray(dt_prev, float) if dt_prev is not None else None
    xi = ui = di = None
    if init is not None:
        xi, ui, di = (np.ascontiguousarray(a, float) for a in init)
    xo = np.empty((B, n, 3)); uo = np.empty((B, n, 2)); do = np.empty(B)
    st = np.empty(B, np.int32); it = np.empty(B, np.int32)
    if obstacles is not None and obst is not None:
        no = np.ascontiguousarray(obstacles[0], np.int32)
        nv = np.ascontiguousarray(obstacles[1], np.int32)
        vv = np.ascontiguousarray(obstacles[2], float)
        rr = np.ascontiguousarray(obstacles[3], float) if len(obstacles) > 3 and obstacles[3] is not None else None
        vel = np.ascontiguousarray(obstacles[4], float) if len(obstacles) > 4 and obstacles[4] is not None else None
        lib.oracle_set_obstacle_velocities(p(vel))
        lib.oracle_set_rows_dropped_out(p(rows_dropped) if rows_dropped is not None else None)
        assert nv.shape == (B, obst.max_obstacles) and vv.shape == (B, obst.max_obstacles, obst.max_vertices, 2)
        lib.oracle_solve_batch_obst(C.byref(ocfg), C.c_int(B), p(x0), p(xf), p(up), p(dp), p(xi), p(ui), p(di), C.byref(obst), p(no), p(nv), p(vv),
                                    p(rr), p(xo), p(uo), p(do), p(st), p(it), C.c_int(nthreads))
        lib.oracle_set_obstacle_velocities(None)
        lib.oracle_set_rows_dropped_out(None)
        return xo, uo, do, st, it
    lib.oracle_solve_batch(C.byref(ocfg), C.c_int(B), p(x0), p(xf), p(up), p(dp), p(xi), p(ui), p(di), p(xo), p(uo), p(do), p(st), p(it),
                           C.c_int(nthreads))
    return xo, uo, do, st, it

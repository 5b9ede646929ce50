"""struct mpc_config (what the parameter readers produce) -> the oracle's OcpConfig, for tests that run the CPU oracles on the problem a parameter set describes"""
import numpy as np

from oracle import se2_nlp as R


def _weights(diag, off, dim):
    d = np.array(list(diag)[:dim], float)
    off = np.atleast_1d(np.array(off, float))
    if not off.any():
        return d
    m = np.diag(d)
    if dim == 3:
        m[0, 1] = m[1, 0] = off[0]; m[0, 2] = m[2, 0] = off[1]; m[1, 2] = m[2, 1] = off[2]
    else:
        m[0, 1] = m[1, 0] = off[0]
    return m


def ocp_config(cfg, n=None) -> R.OcpConfig:
    n = int(n if n is not None else cfg.n)
    fp_kind = int(cfg.footprint_kind)
    if fp_kind == R.FOOTPRINT_CIRCLE:
        fp_params = (float(cfg.footprint_radius),)
    elif fp_kind in (R.FOOTPRINT_LINE, R.FOOTPRINT_TWO_CIRCLES):
        fp_params = tuple(float(v) for v in cfg.footprint_params)
    elif fp_kind == R.FOOTPRINT_POLYGON:
        fp_params = tuple(float(v) for v in list(cfg.footprint_vertices)[:2 * int(cfg.footprint_n_vertices)])
    else:
        fp_params = ()
    inf = lambda v, s: np.array([s * R.INF if abs(x) >= 1e29 else x for x in v])
    nparams = {0: 0, 1: 1, 2: 1, 3: 2}[int(cfg.model)]
    return R.OcpConfig(
        model=int(cfg.model), model_params=tuple(list(cfg.model_params)[:nparams]), n=n, dt_ref=float(cfg.dt_ref), dt_free=bool(cfg.dt_free), dt_lb=float(cfg.dt_lb),
        dt_ub=float(cfg.dt_ub), xf_fixed=tuple(bool(f) for f in cfg.xf_fixed), collocation=int(cfg.collocation), objective=int(cfg.objective),
        Q=_weights(cfg.Q, cfg.Q_offdiag, 3), R=_weights(cfg.R, [cfg.R_offdiag], 2), integral_form=bool(cfg.integral_form),
        cost_integration="trapezoidal_rule" if cfg.cost_integration else "left_sum", hybrid_min_time=bool(cfg.hybrid_cost_minimum_time),
        Qf=_weights(cfg.Qf, cfg.Qf_offdiag, 3) if cfg.has_Qf else None, vp_position_weight=float(cfg.vp_position_weight), vp_orientation_weight=float(cfg.vp_orientation_weight),
        via_points_ordered=bool(cfg.via_points_ordered), terminal_ball_S=_weights(cfg.terminal_ball_S, cfg.terminal_ball_S_offdiag, 3) if cfg.terminal_ball else None,
        terminal_ball_gamma=float(cfg.terminal_ball_gamma), u_lb=np.array(list(cfg.u_lb)), u_ub=np.array(list(cfg.u_ub)), du_lb=inf(list(cfg.du_lb), -1.0), du_ub=inf(list(cfg.du_ub), 1.0),
        min_obstacle_dist=float(cfg.min_obstacle_dist), force_inclusion_dist=float(cfg.force_inclusion_dist), cutoff_dist=float(cfg.cutoff_dist),
        enable_dynamic_obstacles=bool(cfg.enable_dynamic_obstacles), footprint_kind=fp_kind, footprint_params=fp_params)

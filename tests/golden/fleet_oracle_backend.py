"""A stand-in for BatchSolver built on the CPU oracles (tests only): the methods FleetPlanner uses -- solve, set_grid_sizes, set_via_points, costmap_to_obstacles,
check_feasibility -- answered by oracle/mpc_oracle.c, oracle/costmap.py and oracle/feasibility.py, so that the fleet cycle can be tested without a GPU.  The GPU twin of
the test runs the same script on the real BatchSolver."""
from dataclasses import dataclass

import numpy as np

import oracle_from_config
from oracle import c_oracle as CO, costmap as CM, feasibility as FE


@dataclass
class Result:
    x: np.ndarray
    u: np.ndarray
    dt: np.ndarray
    status: np.ndarray
    iters: np.ndarray


class OracleBackend:
    def __init__(self, cfg, max_batch):
        CO.build()
        self.cfg, self.n, self.max_batch = cfg, int(cfg.n), max_batch
        self._n_grid = None
        self._via = None

    def set_grid_sizes(self, n_grid=None):
        self._n_grid = None if n_grid is None else np.asarray(n_grid, np.int32).copy()

    def set_via_points(self, n_via=None, via=None):
        self._via = None if n_via is None else (np.asarray(n_via, np.int32).copy(), np.asarray(via, float).copy())

    def solve(self, x0, xf, u_prev=None, dt_prev=None, init=None, obstacles=None):
        cfg = self.cfg
        B, N = x0.shape[0], self.n
        out = Result(np.zeros((B, N, 3)), np.zeros((B, N, 2)), np.zeros(B), np.zeros(B, np.int32), np.zeros(B, np.int32))
        O, V = int(cfg.max_obstacles), max(1, int(cfg.max_vertices))
        for b in range(B):
            n = int(self._n_grid[b]) if self._n_grid is not None else N
            ocfg = oracle_from_config.ocp_config(cfg, n)
            ini = None if init is None else (init[0][b:b + 1, :n], init[1][b:b + 1, :n], init[2][b:b + 1])
            ob = None if obstacles is None else tuple(a[b:b + 1] for a in obstacles)
            via = None
            if cfg.objective == 2 and self._via is not None:
                via = (self._via[0][b:b + 1], self._via[1][b:b + 1])
            xo, uo, do, st, it = CO.solve_batch(CO.from_nlp_config(ocfg, max_iter=int(cfg.max_iter), tol=float(cfg.tol), mu_init=float(cfg.mu_init), hessian_mode=int(cfg.hessian_mode),
                                                                    acceptable_tol=float(cfg.acceptable_tol), acceptable_iter=int(cfg.acceptable_iter), mu_strategy=int(cfg.mu_strategy)),
                                                x0[b:b + 1], xf[b:b + 1], u_prev[b:b + 1], dt_prev[b:b + 1], init=ini, obstacles=ob,
                                                obst=CO.obst_from_nlp_config(ocfg, O, V, int(cfg.max_obstacle_rows)) if ob is not None else None, via=via)
            out.x[b, :n], out.u[b, :n], out.dt[b], out.status[b], out.iters[b] = xo[0], uo[0], do[0], st[0], it[0]
            out.x[b, n:] = xo[0, -1]; out.u[b, n:] = uo[0, -1]
        return out

    def costmap_to_obstacles(self, cost, resolution, origin, robot_pose, behind_robot_dist=1.5):
        B = cost.shape[0]
        O, V = int(self.cfg.max_obstacles), max(1, int(self.cfg.max_vertices))
        no = np.zeros(B, np.int32); nv = np.zeros((B, O), np.int32); vt = np.zeros((B, O, V, 2)); dr = np.zeros(B, np.int32)
        for b in range(B):
            pts = CM.costmap_to_obstacles(cost[b], resolution, origin[b], robot_pose[b], behind_robot_dist)
            k = min(len(pts), O)
            no[b] = k; nv[b, :k] = 1; vt[b, :k, 0] = pts[:k]; dr[b] = len(pts) - k
        return no, nv, vt, dr

    def check_feasibility(self, x, cost, resolution, origin, footprint_spec, inscribed_radius, min_resolution_collision_check_angular=0.3, look_ahead_idx=-1):
        return np.array([int(FE.is_pose_trajectory_feasible(cost[b], resolution, origin[b], x[b], footprint_spec, inscribed_radius, min_resolution_collision_check_angular, look_ahead_idx))
                         for b in range(x.shape[0])], np.int32)

    def close(self):
        pass

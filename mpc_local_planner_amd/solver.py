"""BatchSolver -- thin Python host wrapper over the C ABI (include/mpc_hip.h).

Mirrors the lifecycle of the reference's Controller (configure -> step ... -> reset,
include/mpc_local_planner/controller.h:61-104) for B independent planner instances.
All arithmetic happens in the HIP library; this file only marshals pointers.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from ._abi import MpcConfig, MpcObstacles, MPC_OK


class MpcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mpc_hip error {code}: {msg}")
        self.code = code


@dataclass
class BatchResult:
    x: np.ndarray        # (B, n, 3)
    u: np.ndarray        # (B, n, 2)  last row duplicates u_{n-2}
    dt: np.ndarray       # (B,)
    status: np.ndarray   # (B,) int32, 0 = converged
    iters: np.ndarray    # (B,) int32


def _addr(a):
    if a is None:
        return None
    return C.c_void_p(a.ctypes.data)


def _as_f64(a, shape):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


class BatchSolver:
    def __init__(self, cfg: MpcConfig, max_batch: int, device: int = 0):
        self._lib = _lib.load()
        self.cfg = cfg
        self.n = int(cfg.n)
        self.max_batch = int(max_batch)
        self.device = int(device)
        h = C.c_void_p()
        rc = self._lib.mpc_create(C.byref(cfg), self.max_batch, self.device, C.byref(h))
        if rc != MPC_OK:
            raise MpcError(rc, self._lib.mpc_last_error().decode())
        self._h = h

    @classmethod
    def from_yaml(cls, path: str, max_batch: int, device: int = 0, namespace="MpcLocalPlannerROS", costmap_footprint=None, **sizing):
        """A solver configured from a parameter file of the reference (its keys, its defaults: mpc_local_planner_amd/params.py).  The grid
        is sized for the largest size the reference's grid adaptation may reach.  The facade options and the notes of the reader are kept
        as `controller_options` / `param_notes`."""
        from . import params
        cfg, ctrl, notes = params.config_from_yaml(path, namespace=namespace, costmap_footprint=costmap_footprint, **sizing)
        n_ref = int(cfg.n)
        cfg.n = max(n_ref, int(ctrl.get("n_max", n_ref)))          # capacity: grid/variable_grid/grid_adaptation/max_grid_size
        s = cls(cfg, max_batch, device)
        if cfg.n > n_ref:
            s.set_grid_sizes([n_ref] * max_batch)                   # every instance starts at grid/grid_size_ref, as the reference's grid does
        s.controller_options, s.param_notes, s.n_ref = ctrl, notes, n_ref
        return s

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != MPC_OK:
            raise MpcError(rc, self._lib.mpc_last_error().decode())

    def reset(self):
        self._check(self._lib.mpc_reset(self._h))

    # ---- host buffers (numpy) ------------------------------------------------
    def _pack_obstacles(self, obstacles, B):
        O, V = int(self.cfg.max_obstacles), int(self.cfg.max_vertices)
        no = np.ascontiguousarray(obstacles[0], dtype=np.int32)
        nv = np.ascontiguousarray(obstacles[1], dtype=np.int32)
        vv = _as_f64(obstacles[2], (B, O, V, 2))
        rr = _as_f64(obstacles[3], (B, O)) if len(obstacles) > 3 and obstacles[3] is not None else None
        vel = _as_f64(obstacles[4], (B, O, 2)) if len(obstacles) > 4 and obstacles[4] is not None else None
        if no.shape != (B,) or nv.shape != (B, O):
            raise ValueError("obstacle arrays have the wrong shape")
        keep = (no, nv, vv, rr, vel)       # the arrays must outlive the call
        ob = MpcObstacles(no.ctypes.data, nv.ctypes.data, vv.ctypes.data, rr.ctypes.data if rr is not None else None,
                          vel.ctypes.data if vel is not None else None)
        return ob, keep

    def solve(self, x0, xf, u_prev=None, dt_prev=None, init=None, obstacles=None) -> BatchResult:
        """One control cycle for B instances (Controller::step, src/controller.cpp:111-179).
        init = (x_init (B,n,3), u_init (B,n,2), dt_init (B,)) or None for the reference cold start.
        obstacles = (n_obstacles (B,), n_vertices (B,O), vertices (B,O,V,2)[, radius (B,O)[, velocity (B,O,2)]]) when the solver was
        created with max_obstacles > 0."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        B = x0.shape[0]
        n = self.n
        x0 = _as_f64(x0, (B, 3))
        xf = _as_f64(xf, (B, 3))
        u_prev = _as_f64(u_prev, (B, 2))
        dt_prev = _as_f64(dt_prev, (B,))
        xi = ui = di = None
        if init is not None:
            xi, ui, di = _as_f64(init[0], (B, n, 3)), _as_f64(init[1], (B, n, 2)), _as_f64(init[2], (B,))
        xo = np.empty((B, n, 3))
        uo = np.empty((B, n, 2))
        do = np.empty(B)
        st = np.empty(B, dtype=np.int32)
        it = np.empty(B, dtype=np.int32)
        ob, keep = self._pack_obstacles(obstacles, B) if obstacles is not None else (None, None)
        rc = self._lib.mpc_solve_batch(self._h, B, _addr(x0), _addr(xf), _addr(u_prev), _addr(dt_prev), _addr(xi), _addr(ui),
                                       _addr(di), C.byref(ob) if ob is not None else None, _addr(xo), _addr(uo), _addr(do),
                                       _addr(st), _addr(it))
        self._check(rc)
        return BatchResult(xo, uo, do, st, it)

    def step(self, x0, xf, u_prev=None, dt_prev=None, init=None, obstacles=None, outer_iterations=1, adapt=False, n_min=3, n_max=0, dt_hyst_ratio=0.1):
        """One control cycle = `outer_iterations` x (grid update -> solve) in ONE call (mpc_step_batch: PredictiveController::step's outer OCP iterations,
        src/controller.cpp:70-72,172, without a host round trip between them).  Returns (BatchResult of the last solve, grid sizes after the call)."""
        B = int(np.asarray(x0).shape[0])
        n = self.n
        x0 = _as_f64(x0, (B, 3)); xf = _as_f64(xf, (B, 3))
        u_prev = _as_f64(u_prev, (B, 2)) if u_prev is not None else None
        dt_prev = _as_f64(dt_prev, (B,)) if dt_prev is not None else None
        xi = ui = di = None
        if init is not None:
            xi, ui, di = _as_f64(init[0], (B, n, 3)), _as_f64(init[1], (B, n, 2)), _as_f64(init[2], (B,))
        ob, keep = self._pack_obstacles(obstacles, B) if obstacles is not None else (None, None)
        xo = np.zeros((B, n, 3)); uo = np.zeros((B, n, 2)); do = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); ng = np.zeros(B, np.int32)
        rc = self._lib.mpc_step_batch(self._h, B, _addr(x0), _addr(xf), _addr(u_prev), _addr(dt_prev), _addr(xi), _addr(ui), _addr(di),
                                      C.byref(ob) if ob is not None else None, int(outer_iterations), int(bool(adapt)), int(n_min), int(n_max if n_max > 0 else n),
                                      float(dt_hyst_ratio), _addr(xo), _addr(uo), _addr(do), _addr(st), _addr(it), _addr(ng))
        self._check(rc)
        return BatchResult(xo, uo, do, st, it), ng

    # ---- device buffers (raw HBM addresses, e.g. torch tensors' data_ptr()) -----
    def solve_device(self, B: int, x0: int, xf: int, u_prev: Optional[int], dt_prev: Optional[int], x_init: Optional[int],
                     u_init: Optional[int], dt_init: Optional[int], x_out: int, u_out: int, dt_out: int,
                     status: Optional[int], iters: Optional[int], obstacles=None) -> None:
        """Asynchronous solve on the solver's stream; all arguments are device addresses (ints)."""
        v = lambda p: C.c_void_p(p) if p else None
        ob = MpcObstacles(*(tuple(obstacles) + (None,) * (5 - len(obstacles)))) if obstacles is not None else None     # up to 5 device addresses
        rc = self._lib.mpc_solve_batch_device(self._h, B, v(x0), v(xf), v(u_prev), v(dt_prev), v(x_init), v(u_init), v(dt_init),
                                              C.byref(ob) if ob is not None else None, v(x_out), v(u_out), v(dt_out), v(status),
                                              v(iters))
        self._check(rc)

    def set_grid_sizes(self, n_grid=None):
        """Per-instance grid sizes n_i <= cfg.n for the following solves (grid adaptation); None = uniform cfg.n."""
        if n_grid is None:
            self._check(self._lib.mpc_set_grid_sizes(self._h, None, 0))
            return
        a = np.ascontiguousarray(n_grid, dtype=np.int32)
        self._check(self._lib.mpc_set_grid_sizes(self._h, C.c_void_p(a.ctypes.data), int(a.shape[0])))

    def set_via_points(self, n_via=None, via=None):
        """Via-points of the minimum_time_via_points objective for the following solves: n_via[B], via[B][cfg.max_via_points][3]
        (x, y, theta); None clears them."""
        if n_via is None:
            self._check(self._lib.mpc_set_via_points(self._h, 0, None, None))
            return
        nv = np.ascontiguousarray(n_via, dtype=np.int32)
        vp = np.ascontiguousarray(via, dtype=np.float64)
        assert vp.shape == (nv.shape[0], self.cfg.max_via_points, 3), vp.shape
        self._check(self._lib.mpc_set_via_points(self._h, int(nv.shape[0]), C.c_void_p(nv.ctypes.data), C.c_void_p(vp.ctypes.data)))

    def costmap_to_obstacles(self, cost, resolution, origin, robot_pose, behind_robot_dist=1.5):
        """Lethal costmap cells -> point obstacles in this solver's obstacle layout (mpc_costmap_to_obstacles).
        cost: uint8 [B][size_y][size_x]; origin [B][2]; robot_pose [B][3].  Returns (n_obstacles[B], n_vertices[B][O], vertices[B][O][V][2], dropped[B])."""
        cost = np.ascontiguousarray(cost, dtype=np.uint8)
        B, sy, sx = cost.shape
        org = np.ascontiguousarray(origin, dtype=np.float64).reshape(B, 2)
        pose = np.ascontiguousarray(robot_pose, dtype=np.float64).reshape(B, 3)
        O, V = self.cfg.max_obstacles, max(1, self.cfg.max_vertices)
        no = np.zeros(B, np.int32); nv = np.zeros((B, O), np.int32); vt = np.zeros((B, O, V, 2)); dr = np.zeros(B, np.int32)
        p = lambda a: C.c_void_p(a.ctypes.data)
        self._check(self._lib.mpc_costmap_to_obstacles(self._h, B, p(cost), sx, sy, float(resolution), p(org), p(pose), float(behind_robot_dist),
                                                       p(no), p(nv), p(vt), p(dr)))
        return no, nv, vt, dr

    def last_candidates(self, B: int):
        """(winner[B], iters_total[B]) of the most recent solve (mpc_last_candidates): index of the candidate initial trajectory that
        supplied each instance's result (-1: none converged) and the iterations spent over all candidates of the instance."""
        win = np.empty(B, np.int32); tot = np.empty(B, np.int32)
        self._check(self._lib.mpc_last_candidates(self._h, int(B), C.c_void_p(win.ctypes.data), C.c_void_p(tot.ctypes.data)))
        return win, tot

    def check_feasibility(self, x, cost, resolution, origin, footprint_spec, inscribed_radius, min_resolution_collision_check_angular=0.3, look_ahead_idx=-1):
        """Controller::isPoseTrajectoryFeasible for B planned trajectories (mpc_check_feasibility): x (B, n, 3), cost uint8 (B, size_y, size_x),
        origin (B, 2), footprint_spec (F, 2) in the robot frame.  Returns int32 (B,): 1 feasible, 0 not."""
        x = _as_f64(x, (np.asarray(x).shape[0], self.n, 3))
        B = x.shape[0]
        cost = np.ascontiguousarray(cost, dtype=np.uint8)
        _, sy, sx = cost.shape
        org = np.ascontiguousarray(origin, dtype=np.float64).reshape(B, 2)
        spec = np.ascontiguousarray(footprint_spec, dtype=np.float64).reshape(-1, 2)
        out = np.zeros(B, np.int32)
        p = lambda a: C.c_void_p(a.ctypes.data)
        self._check(self._lib.mpc_check_feasibility(self._h, B, p(x), p(cost), sx, sy, float(resolution), p(org), p(spec), int(spec.shape[0]),
                                                    float(inscribed_radius), float(min_resolution_collision_check_angular), int(look_ahead_idx), p(out)))
        return out

    def grid_update_device(self, B: int, x0_new: Optional[int], x: int, u: int, dt: int, adapt: bool = False, n_min: int = 3, n_max: int = 0, dt_hyst_ratio: float = 0.1):
        """mpc_grid_update_device: warm-start shift (fixed grid) / single-step grid adaptation + resampling (variable grid) of a whole batch, in place
        on device arrays (addresses)."""
        v = lambda p: C.c_void_p(p) if p else None
        self._check(self._lib.mpc_grid_update_device(self._h, int(B), v(x0_new), v(x), v(u), v(dt), int(bool(adapt)), int(n_min), int(n_max or self.n), float(dt_hyst_ratio)))

    def grid_sizes(self, B: int):
        out = np.zeros(B, np.int32)
        self._check(self._lib.mpc_get_grid_sizes(self._h, int(B), C.c_void_p(out.ctypes.data)))
        return out

    def last_rows_dropped(self, B: int):
        """per instance: clearance rows that did not fit into max_obstacle_rows in the most recent solve (mpc_last_rows_dropped)"""
        out = np.zeros(B, np.int32)
        self._check(self._lib.mpc_last_rows_dropped(self._h, int(B), C.c_void_p(out.ctypes.data)))
        return out

    def synchronize(self):
        self._check(self._lib.mpc_synchronize(self._h))

    def lds_bytes(self) -> int:
        """dynamic LDS of one workgroup of the solve kernel = the working set of one instance; 163840 // lds_bytes() workgroups are resident per compute unit"""
        b = C.c_int64(0)
        self._check(self._lib.mpc_lds_bytes(self._h, C.byref(b)))
        return int(b.value)

    def occupancy(self, B: int):
        """(resident workgroups per compute unit, dynamic LDS bytes of one) of the kernel a launch of B instances selects, from the runtime's occupancy calculation"""
        w, b = C.c_int32(0), C.c_int64(0)
        self._check(self._lib.mpc_occupancy(self._h, int(B), C.byref(w), C.byref(b)))
        return int(w.value), int(b.value)

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0)
        self._check(self._lib.mpc_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

"""Synthetic planner inputs for the BASELINE.json configurations.

Pure numpy; shared by bench.py and the tests so that the GPU path, the oracle and the CPU
baseline all see identical inputs for a given (config, seed, batch)."""
from __future__ import annotations

import numpy as np

SEED_CONFIG2 = 20260924
SEED_CONFIG3 = 20260925
SEED_CONFIG5 = 20260926


def carlike_min_time_inputs(batch: int, seed: int = SEED_CONFIG2, goal_range=(1.0, 6.0)):
    """config 2/4: x0=(0,0,th0), th0~U[-pi,pi); goal range r~U[1,6] m, bearing and yaw ~U[-pi,pi);
    u_prev=(v,phi), v~U[-.2,.4], phi~U[-.5,.5]; dt_prev=.2 (controller at 5 Hz,
    mpc_local_planner_examples/launch/carlike_minimum_time.launch:37)."""
    rng = np.random.default_rng(seed)
    th0 = rng.uniform(-np.pi, np.pi, batch)
    r = rng.uniform(goal_range[0], goal_range[1], batch)
    bearing = rng.uniform(-np.pi, np.pi, batch)
    yaw = rng.uniform(-np.pi, np.pi, batch)
    v = rng.uniform(-0.2, 0.4, batch)
    phi = rng.uniform(-0.5, 0.5, batch)
    x0 = np.stack([np.zeros(batch), np.zeros(batch), th0], axis=1)
    xf = np.stack([r * np.cos(bearing), r * np.sin(bearing), yaw], axis=1)
    u_prev = np.stack([v, phi], axis=1)
    dt_prev = np.full(batch, 0.2)
    return x0, xf, u_prev, dt_prev


def unicycle_quadratic_inputs(batch: int, seed: int = SEED_CONFIG3, goal_range=(0.5, 1.5)):
    """config 1 family: goal inside the look-ahead window of the quadratic-form example
    (mpc_local_planner_params_quadratic_form.yaml:76), u_prev = 0."""
    rng = np.random.default_rng(seed)
    th0 = rng.uniform(-np.pi, np.pi, batch)
    r = rng.uniform(goal_range[0], goal_range[1], batch)
    bearing = th0 + rng.uniform(-0.8, 0.8, batch)
    yaw = bearing + rng.uniform(-0.5, 0.5, batch)
    x0 = np.stack([np.zeros(batch), np.zeros(batch), th0], axis=1)
    xf = np.stack([r * np.cos(bearing), r * np.sin(bearing), yaw], axis=1)
    return x0, xf, np.zeros((batch, 2)), np.full(batch, 0.2)


def bicycle_min_time_inputs(batch: int, seed: int = SEED_CONFIG5, goal_range=(5.0, 40.0)):
    """config 5: kinematic bicycle, long horizon; goal range scaled to n=120."""
    return carlike_min_time_inputs(batch, seed, goal_range)


def unicycle_obstacle_inputs(batch: int, seed: int = SEED_CONFIG3, n_obst: int = 16, max_vertices: int = 6, goal_range=(3.0, 10.0),
                             clearance: float = 0.5, lateral=(0.3, 1.5)):
    """config 3: unicycle quadratic-form instances with `n_obst` convex polygons (4..max_vertices vertices, circum-radius
    U[.2,.6]) placed beside the straight line from start to goal (lateral offset U[r+lateral[0], r+lateral[1]] either side), at least
    `clearance` away from x0 and xf.  With the default `lateral` the corridor is wider than min_obstacle_dist (= 0.2 in config 3), so the
    clearance rows are associated and carried through the solve but rarely bind; `lateral=(0.02, 0.6)` puts about a third of the
    obstacles inside the clearance band, i.e. the rows start violated and are active at the solution.
    Returns (x0, xf, u_prev, dt_prev, (n_obstacles, n_vertices, vertices))."""
    rng = np.random.default_rng(seed)
    th0 = rng.uniform(-np.pi, np.pi, batch)
    r = rng.uniform(goal_range[0], goal_range[1], batch)
    bearing = th0 + rng.uniform(-0.5, 0.5, batch)
    yaw = bearing + rng.uniform(-0.5, 0.5, batch)
    x0 = np.stack([np.zeros(batch), np.zeros(batch), th0], axis=1)
    xf = np.stack([r * np.cos(bearing), r * np.sin(bearing), yaw], axis=1)
    nv = rng.integers(4, max_vertices + 1, size=(batch, n_obst)).astype(np.int32)
    verts = np.zeros((batch, n_obst, max_vertices, 2))
    for b in range(batch):
        d = xf[b, :2] - x0[b, :2]
        L = np.linalg.norm(d)
        e = d / L
        nrm = np.array([-e[1], e[0]])
        for o in range(n_obst):
            rad = rng.uniform(0.2, 0.6)
            # beside the straight start->goal line (the reference's initial plan is collision free): the polygon never
            # contains a point of the initial guess (teb's distance is 0 inside a polygon => zero gradient)
            for _ in range(100):
                off = rng.choice([-1.0, 1.0]) * rng.uniform(rad + lateral[0], rad + lateral[1])
                c = x0[b, :2] + rng.uniform(0.0, 1.0) * d + off * nrm
                if np.linalg.norm(c - x0[b, :2]) > rad + clearance and np.linalg.norm(c - xf[b, :2]) > rad + clearance:
                    break
            k = nv[b, o]
            ang = np.sort(rng.uniform(0, 2 * np.pi, k))
            ang = ang[0] + np.arange(k) * 2 * np.pi / k + rng.uniform(-0.2, 0.2, k) * (2 * np.pi / k)   # convex by construction
            verts[b, o, :k, 0] = c[0] + rad * np.cos(ang)
            verts[b, o, :k, 1] = c[1] + rad * np.sin(ang)
    no = np.full(batch, n_obst, dtype=np.int32)
    return x0, xf, np.zeros((batch, 2)), np.full(batch, 0.2), (no, nv, verts)


def carlike_moving_obstacle_inputs(batch: int, seed: int = 951, goal_range=(3.0, 8.0), n_obst: int = 3):
    """car-like minimum time with clearance rows of a footprint that turns with the pose: `n_obst` point obstacles 0.6 .. 1.1 m beside the start-goal line, the first of them a
    circle of radius 0.15 m that crosses the path at 0.12 m/s (stage_inequality_se2.cpp:177-189: the row at grid point k is evaluated against the obstacle moved by k dt v).
    Returns x0, xf, u_prev, dt_prev, (n_obstacles, n_vertices, vertices, radius, velocity)."""
    x0, xf, up, dtp = carlike_min_time_inputs(batch, seed=seed, goal_range=goal_range)
    rng = np.random.default_rng(seed + 1)
    d = xf[:, None, :2] - x0[:, None, :2]
    nrm = np.stack([-d[..., 1], d[..., 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    pts = x0[:, None, :2] + rng.uniform(0.2, 0.8, (batch, n_obst, 1)) * d + rng.uniform(0.6, 1.1, (batch, n_obst, 1)) * rng.choice([-1.0, 1.0], (batch, n_obst, 1)) * nrm
    vt = pts.reshape(batch, n_obst, 1, 2)
    rad = np.zeros((batch, n_obst)); vel = np.zeros((batch, n_obst, 2))
    vt[:, 0, 0] = x0[:, :2] + 0.5 * d[:, 0] + 1.2 * nrm[:, 0]; rad[:, 0] = 0.15; vel[:, 0] = -0.12 * nrm[:, 0]
    return x0, xf, up, dtp, (np.full(batch, n_obst, np.int32), np.ones((batch, n_obst), np.int32), vt, rad, vel)

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

"""The C oracle's interior-point solve (oracle/mpc_oracle.c) as the "solver" behind a plugin build driven by oracle/ref_lib.py::PluginRunner: obstacles, goal and via-points of
the cycle are read from the plugin, the problem comes from the parameter set (oracle_from_config).  Used by tests/golden/make_ref_vectors.py (reference plugin + reference
Controller) and by tests/test_reference_pinned.py (the same plugin source on the binding with the recording C ABI)."""
import numpy as np

import oracle_from_config
from oracle import c_oracle as CO

MAX_OBSTACLES, MAX_VERTICES = 32, 4


def make(runner, cfg, iterations_out=None):
    CO.build()

    def solve(x, u, dt, u_prev, dt_prev):
        n = x.shape[0]
        ocfg = oracle_from_config.ocp_config(cfg, n)
        count, cont = runner.container()
        goal, via = runner.goal_and_via_points()
        assert count <= MAX_OBSTACLES
        nv = np.zeros((1, MAX_OBSTACLES), np.int32); vt = np.zeros((1, MAX_OBSTACLES, MAX_VERTICES, 2)); rad = np.zeros((1, MAX_OBSTACLES)); vel = np.zeros((1, MAX_OBSTACLES, 2))
        for i, (v, r, ve) in enumerate(cont):
            nv[0, i] = len(v); vt[0, i, :len(v)] = v; rad[0, i] = r; vel[0, i] = ve
        ui = np.vstack([u, u[-1:]])[None]
        viap = None
        if cfg.objective == 2:
            vp = np.zeros((1, 16, 3)); vp[0, :len(via)] = via
            viap = (np.array([len(via)], np.int32), vp)
        xo, uo, do, st, it = CO.solve_batch(CO.from_nlp_config(ocfg, max_iter=int(cfg.max_iter), tol=float(cfg.tol), mu_init=float(cfg.mu_init), hessian_mode=int(cfg.hessian_mode),
                                                                acceptable_tol=float(cfg.acceptable_tol), acceptable_iter=int(cfg.acceptable_iter), mu_strategy=int(cfg.mu_strategy)),
                                            x[None, 0], goal[None], u_prev[None], np.array([dt_prev]), init=(x[None], ui, np.array([dt])),
                                            obstacles=(np.array([count], np.int32), nv, vt, rad, vel),
                                            obst=CO.obst_from_nlp_config(ocfg, MAX_OBSTACLES, MAX_VERTICES, int(cfg.max_obstacle_rows)), via=viap)
        if iterations_out is not None:
            iterations_out.append(int(it[0]))
        return xo[0], uo[0, :n - 1], float(do[0]), st[0] == 0
    return solve

"""developer script (GPU): the instances of the turning-footprint workloads of tests/test_gpu_ext_rows.py where device and C oracle end with different statuses or iteration counts.
The oracle's results are computed on the CPU beforehand (`python tests/tools/dev/footprint_mismatch.py cpu`) and kept next to this file."""
import os, sys
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import mpc_local_planner_amd as m
F = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_footprint_mismatch_oracle.npz")
POLY = (0.25, -0.05, 0.18, -0.05, 0.18, -0.18, -0.19, -0.18, -0.25, 0.0, -0.19, 0.18, 0.18, 0.18, 0.18, 0.05, 0.25, 0.05)
FOOTPRINTS = {"line": (2, (0.0, 0.0, 0.4, 0.0), 0.27), "polygon": (4, POLY, 0.15), "two_circles": (3, (0.2, 0.15, 0.2, 0.15), 0.1)}
def point_obstacles(x0, xf, seed, n_obst=4, lo=0.3, hi=0.9):
    rng = np.random.default_rng(seed)
    B = x0.shape[0]
    d = xf[:, None, :2] - x0[:, None, :2]
    nrm = np.stack([-d[..., 1], d[..., 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    pts = x0[:, None, :2] + rng.uniform(0.2, 0.8, (B, n_obst, 1)) * d + rng.uniform(lo, hi, (B, n_obst, 1)) * rng.choice([-1.0, 1.0], (B, n_obst, 1)) * nrm
    return np.full(B, n_obst, np.int32), np.ones((B, n_obst), np.int32), pts.reshape(B, n_obst, 1, 2)
def static_case(name):
    B, n = 192, 50
    kind, params, dmin = FOOTPRINTS[name]
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=901, goal_range=(2.0, 5.0))
    obs = point_obstacles(x0, xf, 902)
    kw = dict(footprint_kind=kind, min_obstacle_dist=dmin, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=4, max_vertices=1, max_obstacle_rows=4)
    kw.update(dict(footprint_vertices=params) if kind == 4 else dict(footprint_params=params))
    return B, n, (x0, xf, up, dtp), obs, kw, (kind, params, dmin, False, 4)
def dynamic_case(name):
    B, n = 128, 50
    kind, params, dmin = FOOTPRINTS[name]
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=931, goal_range=(2.0, 5.0))
    no, nv, vt = point_obstacles(x0, xf, 932, n_obst=3, lo=0.6, hi=1.1)
    rad = np.zeros((B, 3)); vel = np.zeros((B, 3, 2))
    d = xf[:, :2] - x0[:, :2]
    nrm = np.stack([-d[:, 1], d[:, 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    vt[:, 0, 0] = x0[:, :2] + 0.5 * d + 1.2 * nrm; rad[:, 0] = 0.15; vel[:, 0] = -0.12 * nrm
    kw = dict(footprint_kind=kind, footprint_params=params, enable_dynamic_obstacles=True, min_obstacle_dist=dmin, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=3, max_vertices=1, max_obstacle_rows=4)
    return B, n, (x0, xf, up, dtp), (no, nv, vt, rad, vel), kw, (kind, params, dmin, True, 3)
CASES = [("static " + k, static_case, k) for k in FOOTPRINTS] + [("dynamic " + k, dynamic_case, k) for k in ("line", "two_circles")]
def main():
    if sys.argv[1:] == ["cpu"]:
        from oracle import c_oracle, se2_nlp as R
        out = {}
        for label, fn, k in CASES:
            B, n, inp, obs, kw, (kind, params, dmin, dyn, O) = fn(k)
            ocfg = R.config_carlike_min_time(n)
            ocfg.footprint_kind, ocfg.footprint_params, ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = kind, params, dmin, 0.5, 2.5
            if dyn: ocfg.enable_dynamic_obstacles = True
            ref = c_oracle.solve_batch(c_oracle.from_nlp_config(ocfg), *inp, obstacles=obs, obst=c_oracle.obst_from_nlp_config(ocfg, O, 1, 4))
            out[label + "/status"] = ref[3]; out[label + "/iters"] = ref[4]; out[label + "/x"] = ref[0]
            print(label, np.bincount(ref[3], minlength=4))
        np.savez(F, **out); return
    o = np.load(F)
    for label, fn, k in CASES:
        B, n, inp, obs, kw, _ = fn(k)
        s = m.BatchSolver(m.config_carlike_min_time(n, **kw), max_batch=B)
        r = s.solve(*inp, obstacles=obs); s.close()
        st, it = o[label + "/status"], o[label + "/iters"]
        d = np.nonzero((r.status != st))[0]
        di = np.nonzero((r.status == st) & (r.iters != it))[0]
        err = np.abs(r.x - o[label + "/x"]).reshape(B, -1).max(1)
        print(f"{label}: device {np.bincount(r.status, minlength=4)} oracle {np.bincount(st, minlength=4)}; status differs {[(int(i), int(r.status[i]), int(r.iters[i]), int(st[i]), int(it[i])) for i in d]}; "
              f"same status, other iteration count: {[(int(i), int(r.iters[i]), int(it[i]), float('%.1e' % err[i])) for i in di]}", flush=True)


if __name__ == "__main__":
    main()

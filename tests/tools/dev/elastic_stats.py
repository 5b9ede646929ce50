"""C oracle, reference path alone, on the clearance-row workloads -- with the experiment switches of oracle_set_algo (elastic=<rho> etrig=<k>)
usage: python tests/tools/dev/elastic_stats.py [elastic=1000] [etrig=3] [B=128]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import c_oracle as CO, se2_nlp as R
from mpc_local_planner_amd import workloads as W
KEYS = dict(elastic=10, etrig=11, eap=12, eprog=13)
POLY = (0.25, -0.05, 0.18, -0.05, 0.18, -0.18, -0.19, -0.18, -0.25, 0.0, -0.19, 0.18, 0.18, 0.18, 0.18, 0.05, 0.25, 0.05)
FOOTPRINTS = {"point": (0, (0.0, 0.0, 0.0, 0.0), 0.2), "line": (2, (0.0, 0.0, 0.4, 0.0), 0.27), "polygon": (4, POLY, 0.15), "two_circles": (3, (0.2, 0.15, 0.2, 0.15), 0.1)}

def point_obstacles(x0, xf, seed, n_obst=4, lo=0.3, hi=0.9):
    rng = np.random.default_rng(seed)
    B = x0.shape[0]
    d = xf[:, None, :2] - x0[:, None, :2]
    nrm = np.stack([-d[..., 1], d[..., 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    pts = x0[:, None, :2] + rng.uniform(0.2, 0.8, (B, n_obst, 1)) * d + rng.uniform(lo, hi, (B, n_obst, 1)) * rng.choice([-1.0, 1.0], (B, n_obst, 1)) * nrm
    return np.full(B, n_obst, np.int32), np.ones((B, n_obst), np.int32), pts.reshape(B, n_obst, 1, 2)

def report(name, out, n):
    xo, uo, do, st, it = out[:5]
    ok = st == 0
    print(f"{name:58s}: converged {ok.mean()*100:6.2f}%  iters mean(all) {it.mean():6.2f} mean(conv) {it[ok].mean() if ok.any() else 0:6.2f}  status hist {np.bincount(st, minlength=5).tolist()}  T={do[ok].mean()*(n-1) if ok.any() else 0:.4f}", flush=True)
    return ok

if __name__ == "__main__":
    CO.build()
    lib = CO._load()
    B = 128
    CAP = 100
    for a in sys.argv[1:]:
        k, v = a.split("=")
        if k in KEYS: lib.oracle_set_algo(C.c_int(KEYS[k]), C.c_double(float(v)))
        if k == "B": B = int(v)
        if k == "cap": CAP = int(v)
    n = 50
    for name in ("point", "line", "two_circles", "polygon"):
        kind, params, dmin = FOOTPRINTS[name]
        x0, xf, up, dtp = W.carlike_min_time_inputs(B, seed=931, goal_range=(2.0, 5.0))
        no, nv, vt = point_obstacles(x0, xf, 932, n_obst=3, lo=0.6, hi=1.1)
        rad = np.zeros((B, 3)); vel = np.zeros((B, 3, 2))
        d = xf[:, :2] - x0[:, :2]
        nrm = np.stack([-d[:, 1], d[:, 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
        vt[:, 0, 0] = x0[:, :2] + 0.5 * d + 1.2 * nrm; rad[:, 0] = 0.15; vel[:, 0] = -0.12 * nrm
        ocfg = R.config_carlike_min_time(n)
        ocfg.footprint_kind, ocfg.footprint_params = kind, params
        ocfg.enable_dynamic_obstacles, ocfg.min_obstacle_dist, ocfg.force_inclusion_dist, ocfg.cutoff_dist = True, dmin, 0.5, 2.5
        report(f"dynamic obstacle + {name} footprint (car-like n50)", CO.solve_batch(CO.from_nlp_config(ocfg, max_iter=CAP), x0, xf, up, dtp, obstacles=(no, nv, vt, rad, vel), obst=CO.obst_from_nlp_config(ocfg, 3, 1, 4)), n)
        # static obstacles inside the clearance band
        no, nv, vt = point_obstacles(x0, xf, 77, n_obst=4, lo=0.05, hi=0.6)
        ocfg.enable_dynamic_obstacles = False
        report(f"static points 0.05..0.6 m beside the path + {name} footprint", CO.solve_batch(CO.from_nlp_config(ocfg, max_iter=CAP), x0, xf, up, dtp, obstacles=(no, nv, vt), obst=CO.obst_from_nlp_config(ocfg, 4, 1, 4)), n)
    c3 = R.config_unicycle_quadratic(80)
    for lat in ((0.15, 0.8), (0.02, 0.6)):
        inp = W.unicycle_obstacle_inputs(2 * B, n_obst=16, max_vertices=6, lateral=lat)
        report(f"config 3 n80 16 polygons lateral {lat}", CO.solve_batch(CO.from_nlp_config(c3, max_iter=CAP), *inp[:4], obstacles=inp[4], obst=CO.obst_from_nlp_config(c3, 16, 6, 4)), 80)

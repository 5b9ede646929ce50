import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpc_local_planner_amd as m
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in ("carlike_via_points_n30", "carlike_via_points_ordered_n30"):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    B, VP = g["x0"].shape[0], g["via"].shape[1]
    cfg = m.config_carlike_min_time(30, objective=m.OBJ_MIN_TIME_VIA_POINTS, vp_position_weight=float(g["wp"]), vp_orientation_weight=float(g["wo"]),
                                    via_points_ordered=bool(g["ordered"]), max_via_points=VP)
    s = m.BatchSolver(cfg, max_batch=B)
    s.set_via_points(g["n_via"], g["via"])
    r = s.solve(g["x0"], g["xf"], g["u_prev"], g["dt_prev"])
    ex = np.abs(r.x - g["x"]).reshape(B, -1).max(1); eu = np.abs(r.u - g["u"]).reshape(B, -1).max(1)
    print(name, "status", r.status, "iters", r.iters, g["iters"])
    print("  ex", ex, "\n  eu", eu, "\n  edt", np.abs(r.dt - g["dt"]))
    for i in range(B):
        k = np.abs(r.x[i] - g["x"][i]).max(1).argmax()
        print("   inst", i, "worst k", k, "idx", g["idx"][i], "dx", (r.x[i, k] - g["x"][i, k]))
    s.close()

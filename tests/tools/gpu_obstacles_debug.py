import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpc_local_planner_amd as m
from oracle import se2_nlp as R, ipm_dense as I
n, B, O, V, M = 30, 8, 6, 6, 4
x0, xf, up, dtp, (no, nv, verts) = m.workloads.unicycle_obstacle_inputs(B, seed=77, n_obst=O, max_vertices=V, goal_range=(2.0, 4.0))
cfg = m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M)
s = m.BatchSolver(cfg, max_batch=B)
r = s.solve(x0, xf, up, dtp, obstacles=(no, nv, verts))
print("gpu status", r.status, "iters", r.iters)
ocfg = R.config_unicycle_quadratic(n)
for i in range(B):
    obs = [R.Obstacle(R.OBST_POLYGON, verts[i, o, :nv[i, o]]) for o in range(no[i])]
    inp = R.CycleInputs(x0=x0[i], xf=xf[i], u_prev=up[i], dt_prev=float(dtp[i]), obstacles=obs)
    init = R.cold_start(ocfg, x0[i], xf[i])
    rel, _ = R.associate_obstacles(ocfg, init, obs, max_rows=M)
    t = time.time()
    ref = I.solve(ocfg, inp, init, relevant=rel, opt=I.IpmOptions(max_iter=100))
    err = max(np.abs(ref.traj.x - r.x[i]).max(), np.abs(ref.traj.u - r.u[i, :-1]).max())
    nrows = sum(len(q) for q in rel[1:n-1])
    dmin = min((R.footprint_distance(0, (), r.x[i, k], obs[j]) for k in range(1, n - 1) for j in rel[k]), default=9)
    print(i, "oracle", ref.status, ref.iters, "gpu", r.status[i], r.iters[i], "rows", nrows, "err %.2e" % err, "min clearance %.4f" % dmin, "%.1fs" % (time.time() - t))

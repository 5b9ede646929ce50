import sys, os, numpy as np, torch
torch.zeros(1, device="cuda")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import mpc_local_planner_amd as m
n, B, O, V, M = 40, 1024, 8, 4, 4
x0, xf, up, dtp, obs = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V)
# EXT workloads: line footprint with polygons; terminal ball
s = m.BatchSolver(m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M, footprint_kind=2, footprint_params=(0.0, 0.0, 0.3, 0.0)), max_batch=B)
for i in range(3): r = s.solve(x0, xf, up, dtp, obstacles=obs)
print("line footprint: kernel ms %.3f iters %.2f conv %.4f" % (s.last_kernel_ms(), r.iters.mean(), (r.status == 0).mean())); s.close()
s = m.BatchSolver(m.config_unicycle_quadratic(n, terminal_ball_S=(1.0, 1.0, 0.5), terminal_ball_gamma=0.2), max_batch=B)
for i in range(3): r = s.solve(x0, xf, up, dtp)
print("terminal ball: kernel ms %.3f iters %.2f conv %.4f" % (s.last_kernel_ms(), r.iters.mean(), (r.status == 0).mean())); s.close()
if os.environ.get("NEW"):
    from oracle import se2_nlp as R, c_oracle as CO, kkt_check as KC
    from test_oracle_solver import cost_variant, COST_VARIANTS
    import dataclasses
    FQ = [[2.0, 0.3, -0.1], [0.3, 1.5, 0.2], [-0.1, 0.2, 0.4]]; FR = [[0.1, 0.02], [0.02, 0.05]]
    FQF = [[8.0, 1.0, 0.0], [1.0, 9.0, 0.5], [0.0, 0.5, 0.6]]; FS = [[1.0, 0.2, 0.0], [0.2, 1.0, 0.1], [0.0, 0.1, 0.5]]
    free = dict(dt_free=True, dt_lb=0.05, dt_ub=1.0)
    hyb = dict(Q=(0, 0, 0), Qf=None, hybrid_cost_minimum_time=True, dt_free=True, xf_fixed=(True, True, True), R=(1.0, 0.5))
    dev = {
        "full_weights": dict(Q=FQ, R=FR, Qf=FQF),
        "trapezoid_fixed_dt": dict(integral_form=True, cost_integration=1),
        "trapezoid_free_dt": dict(integral_form=True, cost_integration=1, **free),
        "trapezoid_xf_fixed_free_dt": dict(integral_form=True, cost_integration=1, xf_fixed=(True, True, True), **free),
        "hybrid": hyb, "hybrid_integral": dict(integral_form=True, **hyb),
        "all": dict(Q=FQ, R=FR, Qf=FQF, integral_form=True, cost_integration=1, terminal_ball_S=FS, terminal_ball_gamma=0.3, **free),
    }
    Bq = 64
    x0, xf, up, dtp = m.workloads.unicycle_quadratic_inputs(Bq, seed=11)
    for name in COST_VARIANTS:
        ocfg = cost_variant(name)
        s = m.BatchSolver(m.config_unicycle_quadratic(16, **dev[name]), max_batch=Bq)
        r = s.solve(x0, xf, up, dtp)
        o = CO.solve_batch(CO.from_nlp_config(ocfg), x0, xf, up, dtp)
        both = (r.status == 0) & (o[3] == 0)
        err = np.maximum(np.abs(r.x - o[0]).reshape(Bq, -1).max(1), np.abs(r.dt - o[2]))
        print(name, "device conv %d oracle conv %d iters dev %.1f orc %.1f  max err(both) %.2e median %.2e" % ((r.status == 0).sum(), (o[3] == 0).sum(), r.iters.mean(), o[4].mean(), err[both].max() if both.any() else -1, np.median(err[both]) if both.any() else -1), flush=True)
        s.close()

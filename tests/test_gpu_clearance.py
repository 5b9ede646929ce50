"""-m gpu: what a returned trajectory keeps clear of -- measured with no solver quantity at all -- and what non-finite inputs do.

The clearance rows of ONE solve are the rows associated on the trajectory that solve STARTS from: StageInequalitySE2::update runs in the grid update, before the
solve (stage_inequality_se2.cpp:50-162 via full_discretization_grid_base_se2.cpp:38-134), in the reference as in the product.  A solve can therefore end closer than
min_obstacle_dist to an obstacle it carried no row for; the next outer OCP iteration (controller/outer_ocp_iterations, src/controller.cpp:70-72) or the next control
cycle re-associates on the solution and repairs it.  These tests assert both halves on the device: every returned trajectory keeps min_obstacle_dist to every
obstacle it had a row for (that is primal feasibility, checked by the KKT accounting elsewhere) and -- here -- to EVERY obstacle of its instance once the association
has been renewed on the solution (mpc_step_batch, 3 outer iterations) on BASELINE configs[2]; on a harder car-like workload the share is measured over
1 / 3 / 6 outer iterations, reference path and hedges alike."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("these tests need the MI355X (no HIP device here)")
    torch.zeros(1, device="cuda")
    import mpc_local_planner_amd as pkg
    return pkg


def clearance_to_polygons(x, no, nv, verts):
    """min over grid points 1..n-2 and ALL polygons of the instance of the distance point -> closed edge loop (teb's PolygonObstacle::getMinimumDistance for a
    point: no inside test); plain numpy, nothing of the solver."""
    B = x.shape[0]
    out = np.full(B, np.inf)
    for b in range(B):
        p = x[b, 1:-1, :2]
        for o in range(int(no[b])):
            k = int(nv[b, o]); a = verts[b, o, :k]; c = np.roll(a, -1, axis=0); ab = c - a
            t = np.clip(((p[:, None, :] - a[None]) * ab[None]).sum(-1) / (ab * ab).sum(-1)[None], 0.0, 1.0)
            q = a[None] + t[..., None] * ab[None]
            out[b] = min(out[b], float(np.sqrt(((p[:, None, :] - q) ** 2).sum(-1)).min()))
    return out


def test_config3_clearance_to_every_polygon_after_the_association_is_renewed(m):
    """BASELINE configs[2] on the binding placement (polygons 0.15 .. 0.8 m beside the start-goal line, d_min 0.2), 1024 instances.  One solve: the share of converged
    trajectories that keep d_min to all 16 polygons is what the frozen association gives (about 4 of 5; printed).  Three outer iterations (mpc_step_batch: solve, then
    twice re-associate on the solution + solve): every converged trajectory keeps d_min - 1e-6 to every polygon of its instance."""
    B, n, O, V, M, dmin = 1024, 80, 16, 6, 4, 0.2
    x0, xf, up, dtp, (no, nv, verts) = m.workloads.unicycle_obstacle_inputs(B, n_obst=O, max_vertices=V, lateral=(0.15, 0.8))
    s = m.BatchSolver(m.config_unicycle_quadratic(n, max_obstacles=O, max_vertices=V, max_obstacle_rows=M), max_batch=B)
    r1 = s.solve(x0, xf, up, dtp, obstacles=(no, nv, verts))
    r3, _ = s.step(x0, xf, up, dtp, obstacles=(no, nv, verts), outer_iterations=3)
    s.close()
    c1, c3 = clearance_to_polygons(r1.x, no, nv, verts), clearance_to_polygons(r3.x, no, nv, verts)
    ok1, ok3 = r1.status == 0, r3.status == 0
    print(f"[clearance to every polygon, config 3 binding placement, B={B}] one solve: converged {ok1.mean():.4f}, of those clear of ALL polygons {np.mean(c1[ok1] >= dmin - 1e-6):.4f} "
          f"(min {c1[ok1].min():.4f}); three outer iterations: converged {ok3.mean():.4f}, clear {np.mean(c3[ok3] >= dmin - 1e-6):.4f} (min {c3[ok3].min():.6f})")
    assert ok1.mean() >= 0.95 and ok3.mean() >= 0.97
    assert np.mean(c1[ok1] >= dmin - 1e-6) >= 0.7          # the single solve: rows of the cold start only
    assert (c3[ok3] >= dmin - 1e-6).all()                  # association renewed twice: clear of every obstacle
    assert np.mean(c3[ok3] <= dmin + 1e-4) >= 0.3          # and the rows do bind on this placement


def test_hedged_answers_and_the_clearance_to_every_point_obstacle(m):
    """car-like minimum time, point footprint, three point obstacles 0.1 .. 0.6 m beside the path (d_min 0.3: most start inside the band), reference path alone and with
    three Hermite hedges.  The association is PER GRID POINT (an obstacle is kept at pose k when it is the nearest one on its side of pose k, or within
    force_inclusion_dist of it): a solve can bring a pose k' close to an obstacle that pose had no row for, and the renewed association then constrains k' but frees
    another pose -- the share of converged answers that keep d_min to every obstacle grows with the outer iterations (printed: 1, 3 and 6) without having to reach 1.
    Asserted: it does not fall, it is >= 0.97 after three and >= 0.985 after six outer iterations, and the hedged solver's answers are no worse than the reference
    path's -- whichever candidate supplied them."""
    B, n, dmin = 512, 50, 0.3
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=961, goal_range=(2.0, 5.0))
    rng = np.random.default_rng(962)
    d = xf[:, None, :2] - x0[:, None, :2]
    nrm = np.stack([-d[..., 1], d[..., 0]], -1) / np.linalg.norm(d, axis=-1, keepdims=True)
    pts = x0[:, None, :2] + rng.uniform(0.2, 0.8, (B, 3, 1)) * d + rng.uniform(0.1, 0.6, (B, 3, 1)) * rng.choice([-1.0, 1.0], (B, 3, 1)) * nrm
    no, nv, vt = np.full(B, 3, np.int32), np.ones((B, 3), np.int32), pts.reshape(B, 3, 1, 2)
    kw = dict(min_obstacle_dist=dmin, force_inclusion_dist=0.5, cutoff_dist=2.5, max_obstacles=3, max_vertices=1, max_obstacle_rows=4)

    def clearance(x):
        return np.sqrt(((x[:, 1:-1, None, :2] - pts[:, None, :, :]) ** 2).sum(-1)).min((1, 2))
    out = {}
    for tag, ckw in (("reference path", {}), ("with hedges", dict(candidates=(0, 5, 5, 6), candidate_max_iter=(100, 100, 100, 100), candidate_param=(0.0, 2.0, 1.0, 2.0)))):
        s = m.BatchSolver(m.config_carlike_min_time(n, **kw, **ckw), max_batch=B)
        share, conv = [], []
        for outer in (1, 3, 6):
            r, _ = s.step(x0, xf, up, dtp, obstacles=(no, nv, vt), outer_iterations=outer)
            ok = r.status == 0
            c = clearance(r.x)
            share.append(float(np.mean(c[ok] >= dmin - 1e-6))); conv.append(float(ok.mean()))
        win, _ = s.last_candidates(B)
        s.close()
        print(f"[clearance to every point obstacle, car-like, {tag}, B={B}] outer iterations 1 / 3 / 6: converged {conv[0]:.4f} / {conv[1]:.4f} / {conv[2]:.4f}, "
              f"clear of ALL obstacles {share[0]:.4f} / {share[1]:.4f} / {share[2]:.4f}; answered by a hedge in the last solve {np.mean(win > 0):.3f}")
        assert share[1] >= share[0] - 0.01 and share[2] >= share[1] - 0.01 and share[1] >= 0.97 and share[2] >= 0.985, share
        out[tag] = (conv, share)
    assert out["with hedges"][0][0] >= out["reference path"][0][0] and out["with hedges"][0][0] >= 0.9
    assert out["with hedges"][1][2] >= out["reference path"][1][2] - 0.01


def test_non_finite_inputs_end_with_a_failure_status_and_leave_their_neighbours_alone(m):
    """NaN / Inf in x0, xf or u_prev: t_max / t_min drop a NaN operand, so a non-finite residual is caught through the sums of the KKT pass; such an
    instance must not report MPC_CONVERGED, and the other instances of the batch are bit for bit what they are without it."""
    B, n = 64, 50
    x0, xf, up, dtp = m.workloads.carlike_min_time_inputs(B, seed=971)
    for ckw in ({}, dict(candidates=(0, 5, 5, 7), candidate_max_iter=(60, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5))):
        s = m.BatchSolver(m.config_carlike_min_time(n, **ckw), max_batch=B)
        clean = s.solve(x0, xf, up, dtp)
        bx0, bxf, bup = x0.copy(), xf.copy(), up.copy()
        bad = [3, 10, 17, 24, 31, 38]
        bx0[3, 2] = np.nan; bx0[10, 0] = np.inf; bxf[17, 1] = np.nan; bxf[24, 2] = -np.inf; bup[31, 0] = np.nan; bup[38, 1] = np.inf
        r = s.solve(bx0, bxf, bup, dtp)
        s.close()
        assert (r.status[bad] != 0).all(), r.status[bad]
        keep = np.setdiff1d(np.arange(B), bad)
        assert np.array_equal(r.status[keep], clean.status[keep]) and np.array_equal(r.iters[keep], clean.iters[keep])
        assert np.array_equal(r.x[keep], clean.x[keep]) and np.array_equal(r.u[keep], clean.u[keep]) and np.array_equal(r.dt[keep], clean.dt[keep])

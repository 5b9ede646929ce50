"""ORACLE (test infrastructure only -- never imported by the product path).

Dense primal-dual interior-point solve of the mpc_local_planner NLP.

The reference hands its NLP to Ipopt (src/controller.cpp:388-421, UPSTREAM, not
vendored, version unpinned).  This file restates Ipopt's *published* algorithm
(Waechter & Biegler, Math. Prog. 106(1), 2006: log-barrier, primal-dual Newton
on the perturbed KKT system, fraction-to-boundary rule, monotone
Fiacco-McCormick barrier update, Hessian regularisation by a multiple of the
identity until the KKT matrix has the inertia (n, m, 0) -- read off LAPACK's
Bunch-Kaufman factorisation here, off the pivots of the Riccati sweep in the
product; r01-r03 used the inertia-free curvature test of Chiang & Zavala 2016,
which lets the iteration converge to saddle points: IpmOptions.inertia_test) in a
reduced form -- the filter line search without second-order corrections (r06;
IpmOptions.globalization = "merit": the l1-merit backtracking of rounds 1-5), no restoration
phase (a refused line search empties the filter and takes the shortest trial step) -- with DENSE linear algebra (numpy.linalg) on the full KKT matrix.  The HIP product solves the same Newton systems with a
stage-structured Riccati sweep; agreement of the two is the parity test.

PARITY UNPINNED for the solve: no Ipopt here and no golden outputs in the reference; the
solution is cross-checked against scipy (SLSQP / trust-constr) on the
reference-form NLP of oracle/se2_nlp.py, and by KKT residuals.  The NLP it solves is the
restatement of oracle/se2_nlp.py (its header says what of it is pinned: the angle helpers).

"Solver form" of the rows (same feasible set / same primal KKT points as the
reference form, rows rescaled by positive factors):
  equality   c_k = dt * F_k - [x_{k+1}-x_k, wrap(th_{k+1}-th_k)]  (= +dt * reference defect,
             include/.../fd_collocation_se2.h:54-69)
  rate rows  (u_k-u_{k-1}) - du_ub*dtp <= 0,  du_lb*dtp - (u_k-u_{k-1}) <= 0
             (= dtp * reference rows, src/optimal_control/stage_inequality_se2.cpp:207-221;
              dtp = dt for k>=1, the controller period for k=0, rows dropped for k=0 if dtp==0)
  clearance  d_min - dist <= 0 (unchanged)
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import se2_nlp as R


# --------------------------------------------------------------------------
# model derivatives w.r.t. q = (theta, v, w)   (w = 2nd input)
# --------------------------------------------------------------------------

def model_derivs(model, p, th, v, w):
    """Returns f (3,), G (3,3) = df_i/dq_j, H (3,3,3) = d2 f_i / dq_j dq_l."""
    f = np.zeros(3)
    G = np.zeros((3, 3))
    H = np.zeros((3, 3, 3))
    if model == R.MODEL_KINEMATIC_BICYCLE:
        lr, lf = p[0], p[1]
        kap = lr / (lf + lr)
        t = math.tan(w)
        tp = 1.0 + t * t
        tpp = 2.0 * t * tp
        den = 1.0 + kap * kap * t * t
        beta = math.atan(kap * t)
        bp = kap * tp / den
        bpp = kap * (tpp * den - tp * 2.0 * kap * kap * t * tp) / (den * den)
        c, s = math.cos(th + beta), math.sin(th + beta)
        sb, cb = math.sin(beta), math.cos(beta)
        f[:] = [v * c, v * s, v * sb / lr]
        # f0 = v cos(th+beta)
        G[0] = [-v * s, c, -v * s * bp]
        G[1] = [v * c, s, v * c * bp]
        G[2] = [0.0, sb / lr, v * cb * bp / lr]
        H[0, 0, 0] = -v * c; H[0, 0, 1] = H[0, 1, 0] = -s
        H[0, 0, 2] = H[0, 2, 0] = -v * c * bp
        H[0, 1, 2] = H[0, 2, 1] = -s * bp
        H[0, 2, 2] = -v * c * bp * bp - v * s * bpp
        H[1, 0, 0] = -v * s; H[1, 0, 1] = H[1, 1, 0] = c
        H[1, 0, 2] = H[1, 2, 0] = -v * s * bp
        H[1, 1, 2] = H[1, 2, 1] = c * bp
        H[1, 2, 2] = -v * s * bp * bp + v * c * bpp
        H[2, 1, 2] = H[2, 2, 1] = cb * bp / lr
        H[2, 2, 2] = v * (-sb * bp * bp + cb * bpp) / lr
        return f, G, H
    c, s = math.cos(th), math.sin(th)
    f[0], f[1] = v * c, v * s
    G[0, 0], G[0, 1] = -v * s, c
    G[1, 0], G[1, 1] = v * c, s
    H[0, 0, 0] = -v * c; H[0, 0, 1] = H[0, 1, 0] = -s
    H[1, 0, 0] = -v * s; H[1, 0, 1] = H[1, 1, 0] = c
    if model == R.MODEL_UNICYCLE:
        f[2] = w
        G[2, 2] = 1.0
    elif model == R.MODEL_SIMPLE_CAR:
        L = p[0]
        t = math.tan(w)
        tp = 1.0 + t * t
        f[2] = v * t / L
        G[2, 1] = t / L
        G[2, 2] = v * tp / L
        H[2, 1, 2] = H[2, 2, 1] = tp / L
        H[2, 2, 2] = v * 2.0 * t * tp / L
    elif model == R.MODEL_SIMPLE_CAR_FRONT:
        L = p[0]
        f[2] = v * math.sin(w) / L
        G[2, 1] = math.sin(w) / L
        G[2, 2] = v * math.cos(w) / L
        H[2, 1, 2] = H[2, 2, 1] = math.cos(w) / L
        H[2, 2, 2] = -v * math.sin(w) / L
    else:
        raise ValueError("model")
    return f, G, H


def colloc_points(method):
    """Evaluation points (weight, c) of the collocation increment D = dt sum_e weight_e f(theta_k + c_e dt f_2(u_k), u_k)."""
    if method == R.COLLOC_FORWARD:
        return [(1.0, 0.0)]
    if method == R.COLLOC_MIDPOINT:
        return [(1.0, 0.5)]
    if method == R.COLLOC_CRANK_NICOLSON:
        # the reference's code evaluates to 1.5 f(x_{k+1}) + 0.5 f(x_k) - quot (fd_collocation_se2.h:139-141, `error` aliased on the
        # right-hand side); restated literally: the heading row then reads theta_{k+1} = theta_k + 2 dt f_2(u_k)
        return [(0.5, 0.0), (1.5, 2.0)]
    raise NotImplementedError("collocation method")


def stage_map_derivs(cfg: R.OcpConfig, th, v, w, dt, lam=None):
    """Increment of the collocation row in solver form:  c_k = x_k + D(theta_k, u_k, dt) - x_{k+1}  (theta row wrapped).
      forward differences (fd_collocation_se2.h:54-69)    D = dt f(theta_k, u_k)
      midpoint differences (fd_collocation_se2.h:91-108)  D = dt f(theta_m, u_k),  theta_m = theta_k + dt f_2(u_k) / 2
      Crank-Nicolson (fd_collocation_se2.h:130-147, literal) D = dt (0.5 f(theta_k, u_k) + 1.5 f(theta_k + 2 dt f_2(u_k), u_k))
    The reference evaluates the dynamics at interpolate_angle(theta_k, theta_{k+1}, 0.5) resp. at theta_{k+1}; on the constraint
    manifold theta_{k+1} is an explicit function of (theta_k, u_k, dt) because the heading rate f_2 of every model is independent of
    the pose, so these are the SAME points: same feasible set and KKT points, and the rows stay explicit in x_{k+1} (stage structure).
    Returns val (3,), Jq (3,3) = dD/d(theta,v,w), Jdt (3,), and with lam: Hqq (3,3), Hqd (3,), Hdd of lam^T D."""
    pts = colloc_points(cfg.collocation)
    f0, G0, H0 = model_derivs(cfg.model, cfg.model_params, th, v, w)      # heading-rate row (independent of theta)
    f2, f2u, f2uu = f0[2], G0[2, 1:], H0[2, 1:, 1:]
    val = np.zeros(3); Jq = np.zeros((3, 3)); Jdt = np.zeros(3)
    L = np.zeros((4, 4))
    for (wt, ce) in pts:
        f, G, H = (f0, G0, H0) if ce == 0.0 else model_derivs(cfg.model, cfg.model_params, th + ce * dt * f2, v, w)
        m = np.array([1.0, ce * dt * f2u[0], ce * dt * f2u[1], ce * f2])        # d theta_e / d(theta, v, w, dt)
        dg = np.outer(G[:, 0], m)                                                 # g(theta,u,dt) = f(theta_e, u)
        dg[:, 1:3] += G[:, 1:3]
        val += wt * dt * f
        Jq += wt * dt * dg[:, :3]
        Jdt += wt * (f + dt * dg[:, 3])
        if lam is not None:
            gq = lam @ G                                  # phi_m, phi_v, phi_w
            Hl = np.einsum("a,ajl->jl", lam, H)           # second derivatives of phi = lam^T f wrt (theta_e, v, w)
            mab = np.zeros((4, 4))
            mab[1:3, 1:3] = ce * dt * f2uu
            mab[1:3, 3] = mab[3, 1:3] = ce * f2u
            Dphi = gq[0] * m
            Dphi[1:3] += gq[1:3]
            D2 = Hl[0, 0] * np.outer(m, m) + gq[0] * mab
            for j in (1, 2):
                D2[j, :] += Hl[0, j] * m
                D2[:, j] += Hl[0, j] * m
            D2[1:3, 1:3] += Hl[1:3, 1:3]
            Le = dt * D2
            Le[3, :] += Dphi
            Le[:, 3] += Dphi
            L += wt * Le
    out = dict(val=val, Jq=Jq, Jdt=Jdt)
    if lam is not None:
        out.update(Hqq=L[:3, :3], Hqd=L[:3, 3], Hdd=L[3, 3])
    return out


# --------------------------------------------------------------------------
# point-footprint distance derivatives (analytic); other footprints: numeric
# --------------------------------------------------------------------------

def _closest_point_on_obstacle(pt, ob: R.Obstacle):
    """closest point of the obstacle boundary/body to pt, and whether pt is inside (polygon)."""
    v = np.asarray(ob.vertices, float)
    if ob.kind in (R.OBST_POINT, R.OBST_CIRCLE) or len(v) == 1:
        return v[0], False
    if ob.kind == R.OBST_LINE or len(v) == 2:
        segs = [(v[0], v[1])]
        inside = False
    else:
        inside = False          # teb: distance to the boundary also for a point inside the polygon (no inside test)
        segs = [(v[i], v[(i + 1) % len(v)]) for i in range(len(v))]
    best, bd = None, float("inf")
    for a, b in segs:
        ab = b - a
        sq = float(ab @ ab)
        t = 0.0 if sq == 0 else min(1.0, max(0.0, float((pt - a) @ ab) / sq))
        q = a + t * ab
        d = float(np.linalg.norm(pt - q))
        if d < bd:
            bd, best = d, (q, t, ab)
    return best, inside


def _point_row(pt, ob: R.Obstacle):
    """distance of the point pt to the obstacle (teb semantics, radius subtracted), unit normal from the closest point to pt and
    hk = 1/|pt - closest| if the closest feature is a vertex (point / circle obstacle), 0 on an edge interior or inside a polygon."""
    cp, inside = _closest_point_on_obstacle(pt, ob)
    if inside:
        return 0.0, np.zeros(2), 0.0
    if isinstance(cp, tuple):
        q, t, ab = cp
        interior = 0.0 < t < 1.0
    else:
        q, interior = cp, False
    dvec = pt - q
    d = float(np.linalg.norm(dvec))
    rad = ob.radius if ob.kind == R.OBST_CIRCLE else 0.0
    if d > 0:
        return d - rad, dvec / d, (0.0 if interior else 1.0 / d)
    return d - rad, np.zeros(2), 0.0


def clearance_row(cfg: R.OcpConfig, xk, ob: R.Obstacle, want_hess=True):
    """value, gradient (3,), Hessian (3,3) of  d_min - dist(footprint(x_k), ob).
    Point / circular footprint, two-circle footprint, line footprint against point / circular obstacles: analytic.
    Others: central differences (like corbo's edges)."""
    if cfg.footprint_kind == R.FOOTPRINT_TWO_CIRCLES:
        # teb TwoCirclesRobotFootprint::calculateDistance: min(dist(front centre) - r_front, dist(rear centre) - r_rear), the centres sit
        # at +front_offset / -rear_offset along the heading: point rows at c(theta) = p + o (cos, sin), chain rule through theta
        fo, fr, ro, rr = cfg.footprint_params
        th = float(xk[2]); c, s = math.cos(th), math.sin(th)
        best = None
        for o, r in ((fo, fr), (-ro, rr)):
            ctr = np.asarray(xk[:2], float) + o * np.array([c, s])
            d, nrm, hk = _point_row(ctr, ob)
            if best is None or d - r < best[0]:
                best = (d - r, nrm, hk, o)
        dist, nrm, hk, o = best
        w = o * np.array([-s, c])                                         # d c / d theta
        Jc = np.array([[1.0, 0.0, w[0]], [0.0, 1.0, w[1]]])
        gd = Jc.T @ nrm
        Hd = Jc.T @ (hk * (np.eye(2) - np.outer(nrm, nrm))) @ Jc
        Hd[2, 2] += float(nrm @ (-o * np.array([c, s])))                  # n' d2c/dtheta2
        return cfg.min_obstacle_dist - dist, -gd, -Hd
    if cfg.footprint_kind in (R.FOOTPRINT_POINT, R.FOOTPRINT_CIRCLE):
        pt = np.asarray(xk[:2], float)
        off = cfg.footprint_params[0] if cfg.footprint_kind == R.FOOTPRINT_CIRCLE else 0.0
        cp, inside = _closest_point_on_obstacle(pt, ob)
        g = np.zeros(3)
        Hm = np.zeros((3, 3))
        if inside:
            return cfg.min_obstacle_dist + off, g, Hm
        if isinstance(cp, tuple):
            q, t, ab = cp
            interior = 0.0 < t < 1.0
        else:
            q, interior = cp, False
        dvec = pt - q
        d = float(np.linalg.norm(dvec))
        rad = ob.radius if ob.kind == R.OBST_CIRCLE else 0.0
        val = cfg.min_obstacle_dist - (d - rad - off)
        if d > 0:
            nrm = dvec / d
            g[:2] = -nrm
            if not interior:
                Hm[:2, :2] = -(np.eye(2) - np.outer(nrm, nrm)) / d
        return val, g, Hm
    if cfg.footprint_kind in (R.FOOTPRINT_LINE, R.FOOTPRINT_POLYGON) and (ob.kind in (R.OBST_POINT, R.OBST_CIRCLE) or len(np.asarray(ob.vertices)) == 1):
        # teb Line / PolygonRobotFootprint::calculateDistance with a point / circular obstacle = distance of the obstacle centre to the
        # footprint segment / closed edge loop; evaluated in the ROBOT frame, q = R(-theta)(p_o - p), where the footprint is fixed
        fv = np.asarray(cfg.footprint_params, float).reshape(-1, 2)
        if cfg.footprint_kind == R.FOOTPRINT_LINE or len(fv) == 2:
            edges = [(fv[0], fv[1])]
        elif len(fv) == 1:
            edges = [(fv[0], fv[0])]
        else:
            edges = [(fv[i], fv[(i + 1) % len(fv)]) for i in range(len(fv))]
        th = float(xk[2]); c, s = math.cos(th), math.sin(th)
        v = np.asarray(ob.vertices, float).reshape(-1, 2)[0] - np.asarray(xk[:2], float)
        q = np.array([c * v[0] + s * v[1], -s * v[0] + c * v[1]])
        d, dvec, t = float("inf"), None, 0.0
        for (a, b2) in edges:                      # first closest edge wins
            ab = b2 - a
            sq = float(ab @ ab)
            te = float((q - a) @ ab) / sq if sq > 0 else 0.0
            te = min(1.0, max(0.0, te))
            dv = q - (a + te * ab)
            de = float(np.linalg.norm(dv))
            if de < d:
                d, dvec, t = de, dv, te
        rad = ob.radius if ob.kind == R.OBST_CIRCLE else 0.0
        val = cfg.min_obstacle_dist - (d - rad)
        g = np.zeros(3); Hm = np.zeros((3, 3))
        if d > 0:
            nrm = dvec / d
            hk = 0.0 if 0.0 < t < 1.0 else 1.0 / d
            Jq = np.array([[-c, -s, q[1]], [s, -c, -q[0]]])                 # d q / d (x, y, theta)
            gd = Jq.T @ nrm
            Hd = Jq.T @ (hk * (np.eye(2) - np.outer(nrm, nrm))) @ Jq
            Hd[0, 2] += nrm[0] * s + nrm[1] * c; Hd[2, 0] = Hd[0, 2]        # n' d2q/dx dtheta
            Hd[1, 2] += -nrm[0] * c + nrm[1] * s; Hd[2, 1] = Hd[1, 2]
            Hd[2, 2] += -(nrm @ q)
            g, Hm = -gd, -Hd
        return val, g, Hm
    # numeric (two-circle footprints, line footprint with line / polygon obstacles)
    def fun(x):
        return cfg.min_obstacle_dist - R.footprint_distance(cfg.footprint_kind, cfg.footprint_params, x, ob)
    h = 1e-6
    x = np.asarray(xk, float)
    val = fun(x)
    g = np.zeros(3)
    for i in range(3):
        e = np.zeros(3); e[i] = h
        g[i] = (fun(x + e) - fun(x - e)) / (2 * h)
    Hm = np.zeros((3, 3))
    if want_hess:
        h2 = 1e-4
        for i in range(3):
            for j in range(i, 3):
                ei = np.zeros(3); ei[i] = h2
                ej = np.zeros(3); ej[j] = h2
                Hm[i, j] = Hm[j, i] = (fun(x + ei + ej) - fun(x + ei - ej) - fun(x - ei + ej) + fun(x - ei - ej)) / (4 * h2 * h2)
    return val, g, Hm


# --------------------------------------------------------------------------
# solver-form NLP with analytic first/second derivatives (forward differences)
# --------------------------------------------------------------------------
class SolverNlp:
    """Variables kept as full arrays X (n,3), U (n-1,2), dt; the free ones are
    indexed into a flat vector: [x_1 .. x_{n-2}, xf(free comps), u_0 .. u_{n-2}, dt(if free)]."""

    def __init__(self, cfg: R.OcpConfig, inp: R.CycleInputs, relevant=None, via_idx=None, relevant_dyn=None):
        colloc_points(cfg.collocation)      # raises for an unknown method
        self.cfg, self.inp = cfg, inp
        self.via_idx = via_idx if via_idx is not None else []
        n = self.n = cfg.n
        self.relevant = relevant if relevant is not None else [[] for _ in range(n)]
        # index maps
        self.ix = -np.ones((n, 3), int)
        p = 0
        for k in range(1, n - 1):
            self.ix[k] = [p, p + 1, p + 2]
            p += 3
        for i in range(3):
            if not cfg.xf_fixed[i]:
                self.ix[n - 1, i] = p
                p += 1
        self.iu = np.zeros((n - 1, 2), int)
        for k in range(n - 1):
            self.iu[k] = [p, p + 1]
            p += 2
        self.idt = -1
        if cfg.dt_free:
            self.idt = p
            p += 1
        self.nv = p
        self.theta_idx = [self.ix[k, 2] for k in range(n) if self.ix[k, 2] >= 0]
        # bounds
        self.lb = np.full(p, -R.INF)
        self.ub = np.full(p, R.INF)
        for k in range(n - 1):
            self.lb[self.iu[k]] = cfg.u_lb
            self.ub[self.iu[k]] = cfg.u_ub
        if cfg.dt_free:
            self.lb[self.idt], self.ub[self.idt] = cfg.dt_lb, cfg.dt_ub
        # rate rows: list of (k, i, sign) with sign=+1 for upper row, -1 for lower row
        self.rate_rows = []
        ks = list(range(n))          # k = n-1 is the final row against u_ref = 0
        for k in ks:
            if k == 0 and inp.dt_prev == 0:
                continue
            for i in range(2):
                if cfg.du_lb[i] > -R.INF:
                    self.rate_rows.append((k, i, -1))
            for i in range(2):
                if cfg.du_ub[i] < R.INF:
                    self.rate_rows.append((k, i, +1))
        self.obst_rows = [(k, j) for k in range(1, n - 1) for j in self.relevant[k]]
        # dynamic obstacles (stage_inequality_se2.cpp:99-106,177-189): a row per grid point and moving obstacle, evaluated against the
        # obstacle predicted at t = k dt -- the row depends on dt
        self.relevant_dyn = relevant_dyn if relevant_dyn is not None else [[] for _ in range(n)]
        self.dyn_rows = [(k, j) for k in range(1, n - 1) for j in self.relevant_dyn[k]]
        # terminal l2-ball row on the free final state (final_state_conditions_se2.cpp:54-64; edge only when xf is not fixed)
        self.ball_row = cfg.terminal_ball_S is not None and any(not f for f in cfg.xf_fixed)
        self.mg = len(self.rate_rows) + len(self.obst_rows) + len(self.dyn_rows) + (1 if self.ball_row else 0)
        self.mc = 3 * (n - 1)

    # ---- packing -----------------------------------------------------
    def to_vec(self, t: R.Trajectory):
        v = np.zeros(self.nv)
        m = self.ix >= 0
        v[self.ix[m]] = t.x[m]
        v[self.iu.ravel()] = t.u.ravel()
        if self.idt >= 0:
            v[self.idt] = t.dt
        return v

    def to_traj(self, v) -> R.Trajectory:
        n = self.n
        x = np.zeros((n, 3))
        x[0] = self.inp.x0
        x[n - 1] = self.inp.xf
        m = self.ix >= 0
        x[m] = v[self.ix[m]]
        u = v[self.iu.ravel()].reshape(n - 1, 2)
        dt = v[self.idt] if self.idt >= 0 else self.cfg.dt_ref
        return R.Trajectory(x, u, float(dt))

    def retract(self, v, dv):
        out = v + dv
        out[self.theta_idx] = R.normalize_theta(out[self.theta_idx])
        return out

    # ---- evaluation --------------------------------------------------
    def eval(self, v, lam=None, y=None, want_hess=False):
        """returns dict(f, gf, c, Jc, g, Jg[, W])."""
        cfg = self.cfg
        n = self.n
        t = self.to_traj(v)
        X, U, dt = t.x, t.u, t.dt
        nv = self.nv
        gf = np.zeros(nv)
        W = np.zeros((nv, nv)) if want_hess else None
        xf = np.asarray(self.inp.xf, float)
        # objective
        if cfg.objective in (R.OBJ_MIN_TIME, R.OBJ_MIN_TIME_VIA_POINTS):
            f = (n - 1) * dt
            if self.idt >= 0:
                gf[self.idt] = n - 1
            if cfg.objective == R.OBJ_MIN_TIME_VIA_POINTS:
                # min_time_via_points_cost.cpp:130-145; the orientation term is linear as coded (gradient -w_o, no curvature)
                wp, wo = cfg.vp_position_weight, cfg.vp_orientation_weight
                for vi, k in enumerate(self.via_idx):
                    if k < 0:
                        continue
                    vp = self.inp.via_points[vi]
                    f += wp * float((vp[0] - X[k, 0]) ** 2 + (vp[1] - X[k, 1]) ** 2)
                    for i in range(2):
                        gf[self.ix[k, i]] += 2 * wp * (X[k, i] - vp[i])
                        if want_hess:
                            W[self.ix[k, i], self.ix[k, i]] += 2 * wp
                    if wo > 0:
                        f += wo * float(R.normalize_theta(vp[2] - X[k, 2]))
                        gf[self.ix[k, 2]] -= wo
        else:
            # quadratic form x' Q x + u' R u per grid point (quadratic_cost_se2.cpp:31-83), Q / R diagonal or full; integral form: times dt by the
            # left sum or the trapezoidal rule (finite_differences_grid_se2.cpp:61-75); hybrid: + (n - 1) dt (corbo::MinTimeQuadraticControls)
            Qm, Rm = R.weight_matrix(cfg.Q), R.weight_matrix(cfg.R)
            integral = cfg.integral_form
            trapezoid = integral and cfg.cost_integration == "trapezoidal_rule"
            f = 0.0
            if cfg.hybrid_min_time:
                f += (n - 1) * dt
                if self.idt >= 0:
                    gf[self.idt] += n - 1
            # weight of the state term of grid point k (in units of dt when integral): left sum 1 for k < n-1; trapezoid 1/2 at both ends
            for k in range(n):
                ws = (0.5 if k in (0, n - 1) else 1.0) if trapezoid else (1.0 if k < n - 1 else 0.0)
                if ws == 0.0:
                    continue
                xd = X[k] - xf
                xd[2] = R.normalize_theta(xd[2])
                Qx = Qm @ xd
                sc = float(xd @ Qx) * ws
                w8 = dt if integral else 1.0
                f += sc * w8
                if integral and self.idt >= 0:
                    gf[self.idt] += sc
                for i in range(3):
                    if self.ix[k, i] < 0:
                        continue
                    gf[self.ix[k, i]] += 2 * Qx[i] * ws * w8
                    if want_hess:
                        for j in range(3):
                            if self.ix[k, j] >= 0:
                                W[self.ix[k, i], self.ix[k, j]] += 2 * Qm[i, j] * ws * w8
                        if integral and self.idt >= 0:
                            W[self.ix[k, i], self.idt] += 2 * Qx[i] * ws
                            W[self.idt, self.ix[k, i]] += 2 * Qx[i] * ws
            for k in range(n - 1):
                Ru = Rm @ U[k]
                sc = float(U[k] @ Ru)
                w8 = dt if integral else 1.0
                f += sc * w8
                if integral and self.idt >= 0:
                    gf[self.idt] += sc
                for i in range(2):
                    gf[self.iu[k, i]] += 2 * Ru[i] * w8
                    if want_hess:
                        for j in range(2):
                            W[self.iu[k, i], self.iu[k, j]] += 2 * Rm[i, j] * w8
                        if integral and self.idt >= 0:
                            W[self.iu[k, i], self.idt] += 2 * Ru[i]
                            W[self.idt, self.iu[k, i]] += 2 * Ru[i]
        if cfg.Qf is not None:         # terminal cost: independent of the stage cost's type (src/controller.cpp:641-672)
            Qfm = R.weight_matrix(cfg.Qf)
            xd = X[n - 1] - xf
            xd[2] = R.normalize_theta(xd[2])
            Qx = Qfm @ xd
            free = [i for i in range(3) if self.ix[n - 1, i] >= 0]
            if free:
                f += float(xd @ Qx)
            for i in free:
                gf[self.ix[n - 1, i]] += 2 * Qx[i]
                if want_hess:
                    for j in free:
                        W[self.ix[n - 1, i], self.ix[n - 1, j]] += 2 * Qfm[i, j]
        # equalities
        c = np.zeros(self.mc)
        Jc = np.zeros((self.mc, nv))
        for k in range(n - 1):
            lk = lam[3 * k:3 * k + 3] if (want_hess and lam is not None) else None
            sd = stage_map_derivs(cfg, X[k, 2], U[k, 0], U[k, 1], dt, lk)
            r = slice(3 * k, 3 * k + 3)
            d = X[k + 1] - X[k]
            d[2] = R.normalize_theta(d[2])
            c[r] = sd["val"] - d
            qidx = [self.ix[k, 2], self.iu[k, 0], self.iu[k, 1]]
            for a in range(3):
                row = 3 * k + a
                if self.ix[k, a] >= 0:
                    Jc[row, self.ix[k, a]] += 1.0
                if self.ix[k + 1, a] >= 0:
                    Jc[row, self.ix[k + 1, a]] -= 1.0
                for j, qi in enumerate(qidx):
                    if qi >= 0:
                        Jc[row, qi] += sd["Jq"][a, j]
                if self.idt >= 0:
                    Jc[row, self.idt] += sd["Jdt"][a]
            if lk is not None:
                for j, qj in enumerate(qidx):
                    if qj < 0:
                        continue
                    for l, ql in enumerate(qidx):
                        if ql >= 0:
                            W[qj, ql] += sd["Hqq"][j, l]
                    if self.idt >= 0:
                        W[qj, self.idt] += sd["Hqd"][j]
                        W[self.idt, qj] += sd["Hqd"][j]
                if self.idt >= 0:
                    W[self.idt, self.idt] += sd["Hdd"]
        # inequalities
        g = np.zeros(self.mg)
        Jg = np.zeros((self.mg, nv))
        r = 0
        for (k, i, sg) in self.rate_rows:
            uk = U[k, i] if k < n - 1 else 0.0
            up = U[k - 1, i] if k > 0 else self.inp.u_prev[i]
            dtp = dt if k > 0 else self.inp.dt_prev
            lim = cfg.du_ub[i] if sg > 0 else cfg.du_lb[i]
            g[r] = sg * ((uk - up) - lim * dtp)
            if k < n - 1:
                Jg[r, self.iu[k, i]] += sg
            if k > 0:
                Jg[r, self.iu[k - 1, i]] -= sg
                if self.idt >= 0:
                    Jg[r, self.idt] -= sg * lim
            r += 1
        for (k, j) in self.obst_rows:
            val, gr, Hm = clearance_row(cfg, X[k], self.inp.obstacles[j], want_hess)
            g[r] = val
            Jg[r, self.ix[k]] = gr
            if want_hess and y is not None:
                W[np.ix_(self.ix[k], self.ix[k])] += y[r] * Hm
            r += 1
        for (k, j) in self.dyn_rows:
            # point / circular footprint against the obstacle moved by k dt v = the static row at the point p - k dt v (chain rule in dt)
            ob = self.inp.obstacles[j]
            vel = np.asarray(ob.velocity, float)
            shifted = np.array([X[k, 0] - k * dt * vel[0], X[k, 1] - k * dt * vel[1], X[k, 2]])
            val, gr, Hm = clearance_row(cfg, shifted, ob, want_hess)
            g[r] = val
            Jg[r, self.ix[k]] = gr
            ad = float(gr[:2] @ (-k * vel))
            if self.idt >= 0:
                Jg[r, self.idt] = ad
            if want_hess and y is not None:
                W[np.ix_(self.ix[k], self.ix[k])] += y[r] * Hm
                if self.idt >= 0:
                    hxd = Hm[:2, :2] @ (-k * vel)
                    W[self.ix[k][:2], self.idt] += y[r] * hxd
                    W[self.idt, self.ix[k][:2]] += y[r] * hxd
                    W[self.idt, self.idt] += y[r] * float((-k * vel) @ hxd)
            r += 1
        if self.ball_row:
            S = R.weight_matrix(cfg.terminal_ball_S)
            xd = X[n - 1] - xf
            xd[2] = R.normalize_theta(xd[2])
            Sx = S @ xd
            g[r] = float(xd @ Sx) - cfg.terminal_ball_gamma
            for i in range(3):
                if self.ix[n - 1, i] >= 0:
                    Jg[r, self.ix[n - 1, i]] = 2 * Sx[i]
                    if want_hess and y is not None:
                        for j in range(3):
                            if self.ix[n - 1, j] >= 0:
                                W[self.ix[n - 1, i], self.ix[n - 1, j]] += y[r] * 2 * S[i, j]
            r += 1
        out = dict(f=f, gf=gf, c=c, Jc=Jc, g=g, Jg=Jg)
        if want_hess:
            out["W"] = W
        return out


# --------------------------------------------------------------------------
# interior-point method
# --------------------------------------------------------------------------
@dataclass
class IpmOptions:
    tol: float = 1e-8
    max_iter: int = 100                    # the reference's solver/ipopt/iterations (src/controller.cpp:390)
    mu_init: float = 0.1
    kappa_eps: float = 10.0
    kappa_mu: float = 0.2
    theta_mu: float = 1.5
    tau_min: float = 0.99
    bound_push: float = 1e-2
    slack_push: float = 1e-2
    clearance_slack_push: float = 0.5      # initial slack of a clearance row (static or moving obstacle): max(-g, 0.5), see Algo<T>::clearance_slack_push in csrc/mpc_core.hpp
    eta_armijo: float = 1e-4
    rho_frac: float = 0.1
    delta_first: float = 1e-4
    delta_min: float = 1e-20
    delta_max: float = 1e20
    kappa_plus: float = 8.0
    kappa_plus_first: float = 100.0
    kappa_minus: float = 1.0 / 3.0
    curv_kappa: float = 1e-10
    inertia_test: str = "inertia"        # "inertia": Ipopt's test of a factorisation (nv positive, mc negative eigenvalues; r04); "curvature": the inertia-free test of r01-r03
    s_max: float = 100.0
    max_ls: int = 30
    delta_c: float = 1e-8
    init_controls: bool = True
    globalization: str = "filter"          # "filter" = Ipopt's filter line search (Waechter & Biegler 2006, Algorithm A, no second-order correction; the default of the product, of oracle/mpc_oracle.c and of Ipopt) | "merit" = l1-merit backtracking (mpc_config.line_search = MPC_LS_MERIT, the globalisation of rounds 1-5)
    filter_cap: int = 16
    mu_strategy: str = "adaptive"          # "adaptive" (the default, mpc_config.mu_strategy = MPC_MU_ADAPTIVE) | "monotone" (Fiacco-McCormick, Ipopt's own default) | "loqo" (experiment)
    sigma_min: float = 0.05                # adaptive: sigma = clamp((1 - min(alpha, alpha_dual))^3, sigma_min, 1) from the LAST iteration's step lengths
    mu_err_floor: float = 3e-2             # adaptive: mu never falls below min(mu, mu_err_floor * E_0)
    mu_max_fact: float = 1e3               # adaptive: mu <= mu_max_fact * mu_init (Ipopt mu_max_fact)
    rate_seed_frac: float = 0.9            # seeded controls keep their increments inside this fraction of the control-rate limits
    kappa_c: float = 0.25
    # Ipopt's "solved to acceptable level", which the reference's wrapper counts as success (src/controller.cpp:388-421 configures SolverIpopt; corbo
    # reports success for Converged and EarlyTerminated alike).  Two halves, both at the level `acceptable_tol` (Ipopt default 1e-6; <= 0 = off):
    #   * `acceptable_iter` iterations in a row (Ipopt default 15; <= 0 = off) with an error of at most the level end the solve with status 0;
    #   * when the line search refuses every trial step, or accepts only one shorter than 1e-6 of the fraction-to-boundary step, at a point at that
    #     level, the solve ends THERE (neither the point nor the multipliers move) with status 0 instead of taking the shortest trial step.
    # Same rule, same defaults in oracle/mpc_oracle.c (oracle_config.acceptable_tol / _iter) and in the kernel (mpc_config.acceptable_tol / _iter).
    acceptable_tol: float = 1e-6
    acceptable_iter: int = 15
    # Restoration for clearance rows that jam (r05; same rule and constants in oracle/mpc_oracle.c::solve_one and in the kernel, mpc_wave.hpp::solve): after
    # `elastic_trigger` iterations in a row whose fraction-to-boundary limit on the primal step is below `elastic_ap` while the infeasibility is still at least
    # `elastic_prog` x its value at the start of the streak, the clearance rows become elastic for the rest of the solve: g + s - e = 0, e >= 0, + elastic_rho e in
    # the objective (the exact l1 penalty of the row's violation).  e counts as primal infeasibility, so the solve can only end with e <= tol.  elastic_rho = 0: off.
    elastic_rho: float = 1000.0
    elastic_ap: float = 5e-2
    elastic_prog: float = 0.8
    elastic_trigger: int = 5
    verbose: bool = False


@dataclass
class IpmResult:
    traj: R.Trajectory
    status: int          # 0 converged, 1 max-iter, 2 line-search failure, 3 linear-solve failure
    iters: int
    kkt_error: float
    objective: float
    lam: np.ndarray
    y: np.ndarray
    history: list
    piL: Optional[np.ndarray] = None     # multipliers of the lower / upper variable bounds (for dual_start of the next cycle)
    piU: Optional[np.ndarray] = None


def kkt_inertia(K: np.ndarray):
    """(positive, negative) eigenvalue counts of the symmetric KKT matrix from LAPACK's Bunch-Kaufman factorisation (scipy.linalg.ldl: K = L D L' with 1 x 1 and
    2 x 2 diagonal blocks; Sylvester's law).  A zero pivot counts for neither: the caller then sees a wrong inertia and regularises, as Ipopt does."""
    from scipy.linalg import ldl
    _, d, _ = ldl(K, lower=True, hermitian=True, overwrite_a=False, check_finite=False)
    n = d.shape[0]
    pos = neg = 0
    i = 0
    while i < n:
        if i + 1 < n and d[i + 1, i] != 0.0:
            a, b, c = d[i, i], d[i + 1, i], d[i + 1, i + 1]
            det, tr = a * c - b * b, a + c
            if det < 0: pos += 1; neg += 1
            elif det > 0: pos, neg = (pos + 2, neg) if tr > 0 else (pos, neg + 2)
            i += 2
        else:
            if d[i, i] > 0: pos += 1
            elif d[i, i] < 0: neg += 1
            i += 1
    return pos, neg


def controls_from_states(cfg: R.OcpConfig, init: R.Trajectory) -> R.Trajectory:
    """Solver-side preprocessing of the reference's cold start (u = 0,
    ...grid_base_se2.cpp:218-224): u = 0 is a point where the car-like models lose
    rank (v = 0 => no steering authority), so when ALL controls are zero the solver
    seeds them from the state guess: v_k = forward-difference speed projected on the
    heading, w_k from the heading rate, both clipped to the box."""
    if np.any(init.u != 0.0):
        return init
    t = init.copy()
    n = t.x.shape[0]
    for k in range(n - 1):
        d = t.x[k + 1] - t.x[k]
        dth = float(R.normalize_theta(d[2]))
        th = t.x[k, 2]
        v = (d[0] * math.cos(th) + d[1] * math.sin(th)) / t.dt
        v = min(max(v, cfg.u_lb[0]), cfg.u_ub[0])
        rate = dth / t.dt
        if cfg.model == R.MODEL_UNICYCLE:
            w = rate
        else:
            vv = v if abs(v) > 1e-3 else (1e-3 if v >= 0 else -1e-3)
            if cfg.model == R.MODEL_SIMPLE_CAR:
                w = math.atan(cfg.model_params[0] * rate / vv)
            elif cfg.model == R.MODEL_SIMPLE_CAR_FRONT:
                w = math.asin(min(1.0, max(-1.0, cfg.model_params[0] * rate / vv)))
            else:
                lr, lf = cfg.model_params
                sb = min(1.0, max(-1.0, lr * rate / vv))
                w = math.atan(math.tan(math.asin(sb)) * (lf + lr) / lr)
        w = min(max(w, cfg.u_lb[1]), cfg.u_ub[1])
        t.u[k] = [v, w]
    return t


def rate_feasible_controls(cfg: R.OcpConfig, inp: R.CycleInputs, t: R.Trajectory, frac: float) -> R.Trajectory:
    """... and keeps the seeded controls inside the control-rate rows, as the reference's u = 0 start is (every row but the first): increments
    clamped to `frac` x the rate limits forward from u_prev, then backward from the final row (against u_ref = 0).  A seed that jumps violates the
    rows it crosses: their slacks start at the 1e-2 floor with a residual, and the fraction-to-boundary rule pins the first iterations."""
    t = t.copy()
    n = t.x.shape[0]
    for j in range(2):
        if not (cfg.du_lb[j] > -1e29 and cfg.du_ub[j] < 1e29):
            continue
        lo, hi = cfg.du_lb[j] * t.dt * frac, cfg.du_ub[j] * t.dt * frac
        if inp.dt_prev != 0.0:
            t.u[0, j] = min(max(t.u[0, j], inp.u_prev[j] + cfg.du_lb[j] * inp.dt_prev * frac), inp.u_prev[j] + cfg.du_ub[j] * inp.dt_prev * frac)
        for k in range(1, n - 1):
            t.u[k, j] = min(max(t.u[k, j], t.u[k - 1, j] + lo), t.u[k - 1, j] + hi)
        nxt = 0.0
        for k in range(n - 2, -1, -1):
            t.u[k, j] = min(max(t.u[k, j], nxt - hi), nxt - lo)
            nxt = t.u[k, j]
    return t


def solve(cfg: R.OcpConfig, inp: R.CycleInputs, init: R.Trajectory, relevant=None,
          opt: Optional[IpmOptions] = None, dual_start: Optional[IpmResult] = None, relevant_dyn=None) -> IpmResult:
    """dual_start: result of the previous control cycle of the same problem structure.  Its multipliers are carried over
    (moving-horizon warm start of the duals): every inequality / bound multiplier is max(previous value, mu0 / slack) so
    that no complementarity product starts below the barrier parameter; slacks are re-derived from the new point; the
    collocation multipliers are taken as they are.  Ignored when the structure (row / variable counts) differs."""
    opt = opt or IpmOptions()
    # via-point association from the vertex values the solve starts from (MinTimeViaPointsCost::update runs in the grid update, before the solve)
    via_idx = R.associate_via_points(cfg, init.x, inp.via_points) if cfg.objective == R.OBJ_MIN_TIME_VIA_POINTS else None
    nlp = SolverNlp(cfg, inp, relevant, via_idx, relevant_dyn)
    nv, mc, mg = nlp.nv, nlp.mc, nlp.mg
    lb, ub = nlp.lb, nlp.ub
    hasL = lb > -R.INF
    hasU = ub < R.INF

    if opt.init_controls:
        seeded = not np.any(init.u != 0.0)
        init = controls_from_states(cfg, init)
        if seeded:
            init = rate_feasible_controls(cfg, inp, init, opt.rate_seed_frac)
    v = nlp.to_vec(init)
    # push into the interior of the box (Ipopt bound_push / bound_frac, sec. 3.6)
    for i in range(nv):
        if hasL[i] and hasU[i]:
            pl = min(opt.bound_push * max(1.0, abs(lb[i])), opt.bound_push * (ub[i] - lb[i]))
            pu = min(opt.bound_push * max(1.0, abs(ub[i])), opt.bound_push * (ub[i] - lb[i]))
            v[i] = min(max(v[i], lb[i] + pl), ub[i] - pu)
    mu = mu0 = opt.mu_init
    last_alpha = last_ad = 0.0
    endgame = False
    ev = nlp.eval(v)
    push = np.full(mg, opt.slack_push)
    push[len(nlp.rate_rows):len(nlp.rate_rows) + len(nlp.obst_rows) + len(nlp.dyn_rows)] = opt.clearance_slack_push
    s = np.maximum(-ev["g"], push)
    y = mu / s
    lam = np.zeros(mc)
    piL = np.where(hasL, mu / np.maximum(v - lb, 1e-300), 0.0)
    piU = np.where(hasU, mu / np.maximum(ub - v, 1e-300), 0.0)
    ds = dual_start
    if ds is not None and ds.piL is not None and ds.lam.shape == lam.shape and ds.y.shape == y.shape and ds.piL.shape == piL.shape:
        lam = ds.lam.copy()
        y = np.maximum(ds.y, y)
        piL = np.where(hasL, np.maximum(ds.piL, piL), 0.0)
        piU = np.where(hasU, np.maximum(ds.piU, piU), 0.0)
    delta_last = 0.0
    rho = 0.0
    nfix_term = sum(cfg.xf_fixed)
    fail0 = False
    theta0 = None
    filt = []
    mu_filter = -1.0
    nrest = 0
    history = []
    status = 1
    it = 0
    n_acceptable = 0
    # elastic clearance rows (restoration): e is zero outside the mode and on every other row
    cl = np.zeros(mg, dtype=bool)
    cl[len(nlp.rate_rows):len(nlp.rate_rows) + len(nlp.obst_rows) + len(nlp.dyn_rows)] = True
    el = np.zeros(mg)
    erho = 0.0
    jam_streak, jam_theta0 = 0, 0.0

    def kkt_err(ev, v, s, lam, y, piL, piU, mu_t):
        rd = ev["gf"] + ev["Jc"].T @ lam + ev["Jg"].T @ y - piL + piU
        rp = max(np.abs(ev["c"]).max(initial=0.0), np.abs(ev["g"] + s - el).max(initial=0.0), el.max(initial=0.0))
        comp = 0.0
        if mg:
            comp = max(comp, np.abs(s * y - mu_t).max())
        if erho > 0 and cl.any():
            comp = max(comp, np.abs(el[cl] * (erho - y[cl]) - mu_t).max())
        if hasL.any():
            comp = max(comp, np.abs((v - lb)[hasL] * piL[hasL] - mu_t).max())
        if hasU.any():
            comp = max(comp, np.abs((ub - v)[hasU] * piU[hasU] - mu_t).max())
        ne = int(cl.sum()) if erho > 0 else 0
        we = float((erho - y[cl]).sum()) if erho > 0 else 0.0          # the elastic variables' own multipliers, rho - y
        nm = mc + mg + hasL.sum() + hasU.sum() + ne
        sd = max(opt.s_max, (np.abs(lam).sum() + np.abs(y).sum() + piL.sum() + piU.sum() + we) / max(nm, 1)) / opt.s_max
        nz = mg + hasL.sum() + hasU.sum() + ne
        sc = max(opt.s_max, (np.abs(y).sum() + piL.sum() + piU.sum() + we) / max(nz, 1)) / opt.s_max
        return max(np.abs(rd).max() / sd, rp, comp / sc)

    def barrier_obj(f, v, s, mu, e_=None):
        val = f
        if mg:
            val -= mu * np.log(s).sum()
        if erho > 0 and cl.any():
            ee = el if e_ is None else e_
            val += erho * ee[cl].sum() - mu * np.log(ee[cl]).sum()
        val -= mu * np.log((v - lb)[hasL]).sum()
        val -= mu * np.log((ub - v)[hasU]).sum()
        return val

    while it < opt.max_iter:
        ev = nlp.eval(v, lam, y, want_hess=True)
        if erho == 0 and opt.elastic_rho > 0 and opt.elastic_trigger > 0 and cl.any() and jam_streak >= opt.elastic_trigger and \
                np.abs(ev["c"]).sum() + np.abs(ev["g"] + s).sum() >= opt.elastic_prog * jam_theta0:
            # enter the restoration mode at this point: every clearance row is satisfied again (e takes up the violation, the slack goes back to its start rule)
            erho = opt.elastic_rho
            gcl = ev["g"][cl]
            s[cl] = np.maximum(np.maximum(-gcl, opt.clearance_slack_push), s[cl])
            el[cl] = np.maximum(gcl + s[cl], mu / erho)
            y[cl] = np.maximum(np.minimum(y[cl], 0.5 * erho), mu / s[cl])
            rho = 0.0
            filt.clear()          # the objective changed: the filter's pairs are of another barrier function
            jam_streak = 0
            ev = nlp.eval(v, lam, y, want_hess=True)
        e0 = kkt_err(ev, v, s, lam, y, piL, piU, 0.0)
        if e0 <= opt.tol:
            status = 0
            break
        if opt.acceptable_iter > 0 and opt.acceptable_tol > 0:
            n_acceptable = n_acceptable + 1 if e0 <= opt.acceptable_tol else 0
            if n_acceptable >= opt.acceptable_iter:
                status = 0
                break
        # barrier update
        if opt.mu_strategy == "monotone" or endgame:
            while True:
                emu = kkt_err(ev, v, s, lam, y, piL, piU, mu)
                if emu <= opt.kappa_eps * mu and mu > opt.tol / 10.0:
                    mu = max(opt.tol / 10.0, min(opt.kappa_mu * mu, mu ** opt.theta_mu))
                    rho = 0.0
                    filt.clear()
                else:
                    break
        elif it > 0:                # adaptive; the first iteration keeps the start value (mu_init, or the warm start's)
            comp = []
            if mg:
                comp.append(s * y)
            if erho > 0 and cl.any():
                comp.append(el[cl] * (erho - y[cl]))
            comp.append(((v - lb) * piL)[hasL])
            comp.append(((ub - v) * piU)[hasU])
            comp = np.concatenate(comp)
            avg = comp.mean()
            if opt.mu_strategy == "loqo":      # LOQO rule (Vanderbei & Shanno 1999), as in Ipopt's mu_oracle=loqo (experiment)
                xi = comp.min() / avg
                sig = 0.1 * min(0.05 * (1 - xi) / xi, 2.0) ** 3
            else:
                # adaptive (see oracle/mpc_oracle.c solve_one): Mehrotra's sigma = (mu_aff / mu)^3 read off the step the LAST iteration actually took
                a_ = 1.0 - min(last_alpha, last_ad)
                sig = min(max(a_ * a_ * a_, opt.sigma_min), 1.0)
            mu_new = min(max(sig * avg, opt.tol / 10.0), opt.mu_max_fact * mu0)
            mu_new = max(mu_new, min(mu, opt.mu_err_floor * e0))
            if mu_new <= opt.tol:        # end game: from mu = tol on the monotone rule takes over (tol -> tol / 10 once the barrier problem is solved to kappa_eps mu)
                mu_new, endgame = opt.tol, True
            if mu_new != mu:
                mu = mu_new
                rho = 0.0
                filt.clear()
        tau = max(opt.tau_min, 1.0 - mu)
        W, Jc, Jg, c, g, gf = ev["W"], ev["Jc"], ev["Jg"], ev["c"], ev["g"], ev["gf"]
        dL = np.where(hasL, v - lb, 1.0)
        dU = np.where(hasU, ub - v, 1.0)
        SigZ = np.where(hasL, piL / dL, 0.0) + np.where(hasU, piU / dU, 0.0)
        SigS = y / s
        # gradient of the barrier function wrt z (no multipliers of c, g)
        gphi = gf - np.where(hasL, mu / dL, 0.0) + np.where(hasU, mu / dU, 0.0)
        rg = g + s - el
        # condensed rhs: y+ = mu/s + SigS*(g+s) + SigS*Jg dz
        ybar = mu / s + SigS * rg
        if erho > 0 and cl.any():      # elastic rows, (s, e) condensed together: sigma = 1 / (s / y + e / (rho - y)), ybar = y + sigma (res + mu / y - s - mu / (rho - y) + e)
            wv = erho - y[cl]
            SigS = SigS.copy()
            SigS[cl] = 1.0 / (s[cl] / y[cl] + el[cl] / wv)
            ybar[cl] = y[cl] + SigS[cl] * (rg[cl] + mu / y[cl] - s[cl] - mu / wv + el[cl])
        Hc = W + np.diag(SigZ) + Jg.T @ (SigS[:, None] * Jg)
        rhs1 = -(gphi + Jg.T @ ybar)
        rhs = np.concatenate([rhs1, -c])
        # regularisation loop: delta_w until the test of the factorisation holds (opt.inertia_test)
        # first trial: delta = 0, unless the previous iteration's delta = 0 attempt already failed (then continue from the
        # decayed previous regularisation; it decays by kappa_minus per iteration, so it fades out on its own)
        delta = 0.0
        if fail0 and delta_last > 0.0:
            delta = max(opt.delta_min, opt.kappa_minus * delta_last)
        started_zero = delta == 0.0
        ok = False
        ntry = 0
        while True:
            K = np.zeros((nv + mc, nv + mc))
            K[:nv, :nv] = Hc + delta * np.eye(nv)
            K[:nv, nv:] = Jc.T
            K[nv:, :nv] = Jc
            if opt.delta_c > 0 and nfix_term:
                dc = opt.delta_c * mu ** opt.kappa_c
                for a in range(3):
                    if cfg.xf_fixed[a]:
                        K[nv + mc - 3 + a, nv + mc - 3 + a] = -dc
            try:
                sol = np.linalg.solve(K, rhs)
                good = np.all(np.isfinite(sol))
            except np.linalg.LinAlgError:
                good = False
            if good:
                dz = sol[:nv]
                curv = dz @ ((Hc + delta * np.eye(nv)) @ dz)
                if opt.inertia_test == "inertia":
                    right = kkt_inertia(K) == (nv, mc)
                else:
                    right = curv >= opt.curv_kappa * (dz @ dz)
                if right:
                    ok = True
                    break
            # increase delta
            if delta == 0.0:
                delta = opt.delta_first if delta_last == 0.0 else max(opt.delta_min, opt.kappa_minus * delta_last)
            else:
                delta *= opt.kappa_plus_first if delta_last == 0.0 else opt.kappa_plus
            ntry += 1
            if delta > opt.delta_max or ntry > 40:
                break
        if not ok:
            status = 3
            break
        if delta > 0:
            delta_last = delta
        if started_zero:
            fail0 = delta > 0
        lam_new = sol[nv:]
        ds = -rg - Jg @ dz
        y_new = ybar + SigS * (Jg @ dz)
        dy = y_new - y
        de = np.zeros(mg)
        if erho > 0 and cl.any():
            wv = erho - y[cl]
            ds[cl] = mu / y[cl] - s[cl] - (s[cl] / y[cl]) * dy[cl]
            de[cl] = mu / wv - el[cl] + (el[cl] / wv) * dy[cl]
        dpiL = np.where(hasL, mu / dL - piL - (piL / dL) * dz, 0.0)
        dpiU = np.where(hasU, mu / dU - piU + (piU / dU) * dz, 0.0)

        # fraction to the boundary
        def max_step(val, dval, tau):
            neg = dval < 0
            if not neg.any():
                return 1.0
            return min(1.0, float((-tau * val[neg] / dval[neg]).min()))
        a_p = 1.0
        if hasL.any():
            a_p = min(a_p, max_step(dL[hasL], dz[hasL], tau))
        if hasU.any():
            a_p = min(a_p, max_step(dU[hasU], -dz[hasU], tau))
        if mg:
            a_p = min(a_p, max_step(s, ds, tau))
        a_d = 1.0
        if mg:
            a_d = min(a_d, max_step(y, dy, tau))
        if erho > 0 and cl.any():
            a_p = min(a_p, max_step(el[cl], de[cl], tau))
            a_d = min(a_d, max_step(erho - y[cl], -dy[cl], tau))
        if hasL.any():
            a_d = min(a_d, max_step(piL[hasL], dpiL[hasL], tau))
        if hasU.any():
            a_d = min(a_d, max_step(piU[hasU], dpiU[hasU], tau))

        theta = np.abs(c).sum() + np.abs(rg).sum()
        dphi = gphi @ dz - (mu / s) @ ds if mg else gphi @ dz
        if erho > 0 and cl.any():
            dphi += (erho - mu / el[cl]) @ de[cl]
        phi_cur = barrier_obj(ev["f"], v, s, mu)
        if erho == 0 and opt.elastic_rho > 0 and opt.elastic_trigger > 0 and cl.any():      # the restoration trigger's streak (see IpmOptions.elastic_*)
            rp_now = max(np.abs(c).max(initial=0.0), np.abs(rg).max(initial=0.0))
            if a_p < opt.elastic_ap and rp_now > 1e-3:
                if jam_streak == 0:
                    jam_theta0 = theta
                jam_streak += 1
            else:
                jam_streak = 0
        if theta0 is None:
            theta0 = theta
            theta_max = 1e4 * max(1.0, theta0)
            theta_min = 1e-4 * max(1.0, theta0)
        alpha = a_p
        accepted = False
        if opt.globalization == "merit":
            curv_full = dz @ (Hc @ dz) + delta * (dz @ dz)
            if theta > 0:
                sigma = 1.0 if curv_full > 0 else 0.0
                rho_trial = (dphi + 0.5 * sigma * curv_full) / ((1.0 - opt.rho_frac) * theta)
                if rho < rho_trial:
                    rho = rho_trial + 1.0
            phi0 = phi_cur + rho * theta
            D = dphi - rho * theta
            for ls in range(opt.max_ls):
                if ls > 0:
                    alpha *= 0.5
                vt = nlp.retract(v, alpha * dz)
                st = s + alpha * ds
                et = el + alpha * de
                evt = nlp.eval(vt)
                tht = np.abs(evt["c"]).sum() + np.abs(evt["g"] + st - et).sum()
                phit = barrier_obj(evt["f"], vt, st, mu, et) + rho * tht
                if np.isfinite(phit) and phit - phi0 - 10 * 2.220446049250313e-16 * abs(phi0) <= opt.eta_armijo * alpha * D:
                    accepted = True
                    break
        else:
            # Ipopt's filter line search (Waechter & Biegler 2006, Algorithm A, steps A-5.1 .. A-5.10) with Ipopt's default constants; no second-order correction, no
            # restoration phase: when every trial step is refused the filter is emptied and the shortest trial step taken.  Identical, statement by statement, in
            # oracle/mpc_oracle.c (solve_one), the kernel (mpc_wave_solve.inc) and tests/host_harness/ipm_serial.hpp.  The filter holds the (theta, phi) of the iterates
            # whose step was accepted by the sufficient-decrease rule (not by the switching / Armijo rule); its margins are applied when a trial is checked.
            g_th, g_ph, s_ph, s_th, eta_ph, dlt, g_al = 1e-5, 1e-8, 2.3, 1.1, 1e-8, 1.0, 0.05
            if mu != mu_filter:
                filt.clear()
                mu_filter = mu
            p_ph = (-dphi) ** s_ph if dphi < 0 else 0.0
            p_th = theta ** s_th
            a_min = g_th
            if dphi < 0:
                a_min = min(a_min, g_ph * theta / (-dphi))
                if theta <= theta_min:
                    a_min = min(a_min, dlt * p_th / p_ph)
            a_min *= g_al
            sw_arm = False
            evaluated = False
            for ls in range(opt.max_ls):
                if ls > 0:
                    alpha *= 0.5
                if ls > 0 and alpha < a_min * a_p:
                    evaluated = False
                    break
                vt = nlp.retract(v, alpha * dz)
                st = s + alpha * ds
                et = el + alpha * de
                evt = nlp.eval(vt)
                evaluated = True
                tht = np.abs(evt["c"]).sum() + np.abs(evt["g"] + st - et).sum()
                phit = barrier_obj(evt["f"], vt, st, mu, et)
                ok_t = np.isfinite(phit) and tht <= theta_max
                if ok_t:
                    for (fth, fph) in filt:
                        if not (tht <= (1 - g_th) * fth or phit <= fph - g_ph * fth):
                            ok_t = False
                            break
                switching = dphi < 0 and alpha * p_ph > dlt * p_th
                armijo = phit - phi_cur - 10 * 2.220446049250313e-16 * abs(phi_cur) <= eta_ph * alpha * dphi
                if ok_t:
                    if theta <= theta_min and switching:
                        if armijo:
                            accepted = True
                            sw_arm = True
                    elif tht <= (1 - g_th) * theta or phit <= phi_cur - g_ph * theta:
                        accepted = True
                if accepted:
                    break
            if accepted and not sw_arm:
                if len(filt) == opt.filter_cap:
                    filt.pop(0)
                filt.append((theta, phi_cur))
            if not accepted:
                # no restoration phase: empty the filter and take the last (shortest) trial step
                filt.clear()
                nrest += 1
                if not evaluated:
                    vt = nlp.retract(v, alpha * dz)
                    st = s + alpha * ds
                    et = el + alpha * de
                    evt = nlp.eval(vt)
                    phit = barrier_obj(evt["f"], vt, st, mu, et)
                if np.isfinite(phit):
                    accepted = True
        if opt.acceptable_tol > 0 and (not accepted or alpha < 1e-6 * a_p) and e0 <= opt.acceptable_tol:
            # nothing is moved (neither the point nor the multipliers): the next iteration would compute the same step and refuse it again.
            # Tested before the line-search failure: Ipopt answers a failed line search at an acceptable point with success.
            status = 0
            break
        if not accepted:
            if alpha * np.abs(dz).max() < 1e-14:
                status = 2
                break
        v, s = vt, st
        if erho > 0:
            el = el + alpha * de
        last_alpha, last_ad = alpha, a_d
        lam = lam + alpha * (lam_new - lam)
        y = y + a_d * dy
        piL = piL + a_d * dpiL
        piU = piU + a_d * dpiU
        # keep bound multipliers in the Ipopt safeguard band (eq. 16)
        kS = 1e10
        if mg:
            y = np.minimum(np.maximum(y, mu / (kS * s)), kS * mu / s)
        if erho > 0 and cl.any():      # the same safeguards for e and its multiplier rho - y
            y[cl] = np.minimum(np.maximum(y[cl], erho - kS * mu / el[cl]), erho - mu / (kS * el[cl]))
        dLn = np.where(hasL, v - lb, 1.0)
        dUn = np.where(hasU, ub - v, 1.0)
        piL = np.where(hasL, np.minimum(np.maximum(piL, mu / (kS * dLn)), kS * mu / dLn), 0.0)
        piU = np.where(hasU, np.minimum(np.maximum(piU, mu / (kS * dUn)), kS * mu / dUn), 0.0)
        it += 1
        history.append(dict(it=it, mu=mu, e0=e0, theta=theta, alpha=alpha, a_d=a_d, delta=delta, rho=rho, ls=ls, f=ev["f"]))
        if opt.verbose:
            print(f"{it:3d} f={ev['f']:.6f} e0={e0:.2e} mu={mu:.1e} th={theta:.2e} a={alpha:.3f} ad={a_d:.3f} dl={delta:.1e} ls={ls} rho={rho:.2e}")

    ev = nlp.eval(v, lam, y)
    e0 = kkt_err(ev, v, s, lam, y, piL, piU, 0.0)
    return IpmResult(nlp.to_traj(v), status, it, e0, ev["f"], lam, y, history, piL, piU)

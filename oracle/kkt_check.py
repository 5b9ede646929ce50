"""ORACLE (test infrastructure only): is a trajectory a KKT point of the REFERENCE-FORM NLP?

Pinning: the NLP this checker evaluates (oracle/se2_nlp.py::ReferenceNlp) is a restatement of the reference's edge classes from their sources (cited there, function by
function); of it only the angle helpers are held to executed reference code (tests/test_reference_math.py) -- nothing else of the reference builds in this image.  Also
unpinned: Ipopt's iterates (this file checks optimality conditions, not iterates) and teb's distance functions for lines / polygons (restated, checked against sampled
outlines in tests/test_footprint_bruteforce.py).


Used by the parity tests to classify solver results that do not coincide with the oracle's iterate sequence (a line-search tie that
flips, another candidate initial trajectory): such a result is acceptable iff it is feasible and stationary for the NLP exactly as the
reference poses it (oracle/se2_nlp.py::ReferenceNlp: rows of FiniteDifferencesGridSE2::createEdges, src/optimal_control/
finite_differences_grid_se2.cpp:36-154, derivatives by central differences through the vertex retraction as corbo's edges take them).

    feasibility   max(|c(z)|, g(z)^+, bound violations)
    stationarity  min over multipliers (lambda free; nu >= 0 on the nearly active rows; pi_l, pi_u >= 0 on the nearly active bounds) of
                  || grad f + J_c' lambda + J_g,act' nu - pi_l + pi_u ||_inf      (bounded least squares, scipy.optimize.lsq_linear)
    complementarity  max_i multiplier_i * slack_i over those rows / bounds
An interior-point result at barrier mu ~ 1e-9 keeps slack s = mu / y on a row with multiplier y, so "nearly active" has to be generous
(act_tol, default 1e-1: rows left out carry barrier multipliers below ~1e-8); what keeps a far-away row from being used is the complementarity number.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import lsq_linear

from . import se2_nlp as R


def _active_set_and_multipliers(nlp, z, act_tol, fd_step):
    """the bounded least-squares problem both checkers share: numeric gradients of the reference-form NLP at z, the nearly active rows / bounds, and the multipliers
    (lambda free; nu, pi_l, pi_u >= 0) that minimise the stationarity residual with the complementarity products as extra rows"""
    lb, ub = nlp.bounds()
    c = nlp.equalities(z)
    g = nlp.inequalities(z)
    gradf = nlp.numeric_jacobian(lambda v: np.array([nlp.objective(v)]), z, fd_step)[0]
    Jc = nlp.numeric_jacobian(nlp.equalities, z, fd_step)
    act = np.where(g > -act_tol)[0]
    Jg = nlp.numeric_jacobian(nlp.inequalities, z, fd_step)[act] if act.size else np.zeros((0, z.size))
    al = np.where(z - lb < act_tol)[0]
    au = np.where(ub - z < act_tol)[0]
    El = np.zeros((al.size, z.size)); El[np.arange(al.size), al] = -1.0
    Eu = np.zeros((au.size, z.size)); Eu[np.arange(au.size), au] = 1.0
    A = np.concatenate([Jc, Jg, El, Eu], axis=0).T            # columns = multipliers
    slack = np.concatenate([-g[act], (z - lb)[al], (ub - z)[au]])
    lo = np.concatenate([np.full(Jc.shape[0], -np.inf), np.zeros(slack.size)])
    # multipliers are not unique (degenerate rows): the complementarity products enter the least-squares problem as extra rows
    # slack_i * multiplier_i = 0, so that a row that is not active only gets the (tiny) multiplier the stationarity residual really needs
    W = np.zeros((slack.size, lo.size)); W[np.arange(slack.size), Jc.shape[0] + np.arange(slack.size)] = np.abs(slack)
    sol = lsq_linear(np.concatenate([A, W], axis=0), np.concatenate([-gradf, np.zeros(slack.size)]), bounds=(lo, np.full(lo.size, np.inf)),
                     tol=1e-15, max_iter=4000, method='bvls')
    return dict(lb=lb, ub=ub, c=c, g=g, gradf=gradf, Jc=Jc, act=act, al=al, au=au, A=A, slack=slack, mult=sol.x)


def kkt_residuals(ocfg, x0, xf, u_prev, dt_prev, x, u, dt, act_tol: float = 1e-1, nlp_kwargs=None, inp_kwargs=None, fd_step: float = 1e-5):
    """x (n,3), u (n,2) or (n-1,2) (a duplicated last control is dropped), dt.  Returns dict(feas, stat, objective, n_active)."""
    inp = R.CycleInputs(x0=np.asarray(x0, float), xf=np.asarray(xf, float), u_prev=np.asarray(u_prev, float), dt_prev=float(dt_prev), **(inp_kwargs or {}))
    nlp = R.ReferenceNlp(ocfg, inp, **(nlp_kwargs or {}))
    n = ocfg.n
    u = np.asarray(u, float)[: n - 1]
    z = nlp.pack(R.Trajectory(np.asarray(x, float), u, float(dt)))
    q = _active_set_and_multipliers(nlp, z, act_tol, fd_step)
    lb, ub, c, g, A, slack, me = q["lb"], q["ub"], q["c"], q["g"], q["A"], q["slack"], q["Jc"].shape[0]
    feas = max(float(np.abs(c).max(initial=0.0)), float(g.max(initial=0.0)), float((lb - z).max(initial=0.0)), float((z - ub).max(initial=0.0)))
    stat = float(np.abs(A @ q["mult"] + q["gradf"]).max())
    comp = float((q["mult"][me:] * np.abs(slack)).max(initial=0.0))
    act, al, au = q["act"], q["al"], q["au"]
    return {"feas": feas, "stat": stat, "comp": comp, "objective": float(nlp.objective(z)), "n_active": int(act.size + al.size + au.size)}


def second_order(ocfg, x0, xf, u_prev, dt_prev, x, u, dt, act_tol: float = 1e-1, strong_tol: float = 1e-5, slack_tol: float = 1e-5, nlp_kwargs=None, inp_kwargs=None,
                 fd_step: float = 1e-5, hess_step: float = 1e-4):
    """Is a KKT point a local MINIMUM?  The reduced Hessian of the Lagrangian Z' (grad^2 L) Z on the tangent space of the active rows, by differences of the reference-form
    NLP's functions alone (no solver quantity): multipliers from the same bounded least-squares problem as kkt_residuals; `strong` rows / bounds = multiplier > strong_tol,
    `active` = slack < slack_tol (an interior-point result keeps slack = mu / multiplier).  Z_s = null space of the equality rows + the strong rows (the subspace of the
    second-order SUFFICIENT condition when no row is weakly active), Z_a = null space with every active row (second-order NECESSARY condition).  grad^2 L Z by central
    differences (hess_step) of the numeric gradient of L (fd_step) along the columns of Z.  Returns dict(dim_s, min_eig_s, dim_a, min_eig_a, n_weak): a strict local minimum
    has min_eig_s > 0 (dim_s = 0: a vertex of the active set -- bang-bang minimum-time solutions -- min_eig = +inf)."""
    from scipy.linalg import null_space
    inp = R.CycleInputs(x0=np.asarray(x0, float), xf=np.asarray(xf, float), u_prev=np.asarray(u_prev, float), dt_prev=float(dt_prev), **(inp_kwargs or {}))
    nlp = R.ReferenceNlp(ocfg, inp, **(nlp_kwargs or {}))
    n = ocfg.n
    z = nlp.pack(R.Trajectory(np.asarray(x, float), np.asarray(u, float)[: n - 1], float(dt)))
    q = _active_set_and_multipliers(nlp, z, act_tol, fd_step)
    lb, ub, Jc, act, al, au, A, slack = q["lb"], q["ub"], q["Jc"], q["act"], q["al"], q["au"], q["A"], q["slack"]
    me = Jc.shape[0]
    lam, nu = q["mult"][:me], q["mult"][me:me + act.size]
    mult = q["mult"][me:]
    rows = A.T[me:]
    strong = mult > strong_tol
    active = np.abs(slack) < slack_tol
    # variables a fixed bound pins (start pose, a fixed goal) have lb == ub: they are in al AND au with slack 0 -- active whatever their multiplier
    fixed = np.concatenate([np.zeros(act.size, bool), (ub - lb)[al] <= 0, (ub - lb)[au] <= 0])

    def lagr(v):
        gi = nlp.inequalities(v)
        return np.array([nlp.objective(v) + lam @ nlp.equalities(v) + (nu @ gi[act] if act.size else 0.0)])

    def grad_l(v):
        return nlp.numeric_jacobian(lagr, v, fd_step)[0]

    full = {}

    def full_hessian():
        """the whole Hessian of L in 16 gradient differences: a variable of grid point k meets only variables of k - 1 .. k + 1 (collocation and rate rows) and dt, so
        variables at the same place of grid points three apart share a difference (z = [u_0 | x_k u_k, k = 1 .. n-2 | free x_{n-1} | dt], se2_nlp.ReferenceNlp.pack)"""
        if "H" in full:
            return full["H"]
        nz = z.size
        idt = nz - 1 if ocfg.dt_free else -1
        stage = np.zeros(nz, int); place = np.zeros(nz, int)
        for i in range(nz):
            if i == idt: stage[i], place[i] = -10, 0
            elif i < 2: stage[i], place[i] = 0, 3 + i
            elif i < 2 + 5 * (n - 2): stage[i], place[i] = (i - 2) // 5 + 1, (i - 2) % 5
            else: stage[i], place[i] = n - 1, i - (2 + 5 * (n - 2))
        H = np.zeros((nz, nz))
        for pl in range(5):
            for md in range(3):
                cols = [i for i in range(nz) if i != idt and place[i] == pl and stage[i] % 3 == md]
                if not cols:
                    continue
                e = np.zeros(nz); e[cols] = hess_step
                d = (grad_l(nlp.plus(z, e)) - grad_l(nlp.plus(z, -e))) / (2 * hess_step)
                for c in cols:
                    r = np.nonzero((np.abs(stage - stage[c]) <= 1) & (np.arange(nz) != idt))[0]
                    H[r, c] = d[r]
        if idt >= 0:
            e = np.zeros(nz); e[idt] = hess_step
            d = (grad_l(nlp.plus(z, e)) - grad_l(nlp.plus(z, -e))) / (2 * hess_step)
            H[:, idt] = d; H[idt, :] = d
        full["H"] = 0.5 * (H + H.T)
        return full["H"]

    def reduced(sel):
        M = np.concatenate([Jc, rows[sel | fixed]], axis=0)
        Z = null_space(M, rcond=1e-9)
        if Z.shape[1] == 0:
            return 0, np.inf
        if Z.shape[1] > 16 and not nlp_kwargs:          # (clearance rows couple nothing new, but keep the plain differences for them)
            HZ = full_hessian() @ Z
        else:
            HZ = np.zeros_like(Z)
            for i in range(Z.shape[1]):
                HZ[:, i] = (grad_l(nlp.plus(z, hess_step * Z[:, i])) - grad_l(nlp.plus(z, -hess_step * Z[:, i]))) / (2 * hess_step)
        Hr = Z.T @ HZ
        return Z.shape[1], float(np.linalg.eigvalsh(0.5 * (Hr + Hr.T)).min())
    ds, es = reduced(strong)
    weak = int((active & ~strong & ~fixed).sum())
    da, ea = (ds, es) if weak == 0 else reduced(active | strong)
    return {"dim_s": ds, "min_eig_s": es, "dim_a": da, "min_eig_a": ea, "n_weak": weak, "n_strong": int((strong & ~fixed).sum())}


def is_kkt_point(res, feas_tol: float = 1e-6, stat_tol: float = 1e-6, comp_tol: float = 1e-6) -> bool:
    return res["feas"] <= feas_tol and res["stat"] <= stat_tol and res["comp"] <= comp_tol


def obstacle_list(n_obstacles, n_vertices, vertices, radius=None, velocity=None):
    """one instance's slice of the ABI's mpc_obstacles arrays -> [se2_nlp.Obstacle]"""
    out = []
    for j in range(int(n_obstacles)):
        nv = int(n_vertices[j])
        r = float(radius[j]) if radius is not None else 0.0
        kind = (R.OBST_CIRCLE if r > 0 else R.OBST_POINT) if nv <= 1 else (R.OBST_LINE if nv == 2 else R.OBST_POLYGON)
        vel = np.asarray(velocity[j], float) if velocity is not None else None
        out.append(R.Obstacle(kind, np.asarray(vertices[j][:max(nv, 1)], float), r, vel))
    return out


def _refined(fn):
    """kkt_residuals at the standard difference step, and -- only where the stationarity number fails there -- at steps of 1e-6 and 1e-7: the rows of a turning footprint are
    C1 but not C2 where the closest feature changes (segment interior <-> end point), and an answer that sits ON such a transition shows a central-difference error
    of (h / 4) x (jump of the curvature) x multiplier in the gradient -- linear in h, 8e-4 / 8e-5 / 8e-6 / 8e-7 at h = 1e-4 .. 1e-7 on the instance that brought it up
    (r05, dynamic obstacle + line footprint).  The smallest stationarity number counts; rounding noise at h = 1e-7 is ~1e-7."""
    res = fn(1e-5)
    for h in (1e-6, 1e-7):
        if res["stat"] <= 1e-6:
            break
        r2 = fn(h)
        if r2["stat"] < res["stat"]:
            res = dict(r2, fd_step=h)
    return res


def _one(args):
    ocfg, x0, xf, up, dtp, x, u, dt, obst, max_rows = args[:10]
    start_x = args[10] if len(args) > 10 else None
    if obst is None:
        return _refined(lambda h: kkt_residuals(ocfg, x0, xf, up, dtp, x, u, dt, fd_step=h))
    # clearance rows: the association is frozen on the trajectory the solve STARTS from, as in the reference (StageInequalitySE2::update runs in the grid update, before the solve)
    # and in the product: the reference's cold start, or -- for an answer that a candidate initial trajectory supplied -- that candidate's seed (start_x, (n, 3))
    obs = obstacle_list(*obst)
    start = R.cold_start(ocfg, x0, xf) if start_x is None else R.Trajectory(np.asarray(start_x, float), np.zeros((ocfg.n - 1, 2)), float(ocfg.dt_ref))
    rel, rel_dyn = R.associate_obstacles(ocfg, start, obs, max_rows)
    return _refined(lambda h: kkt_residuals(ocfg, x0, xf, up, dtp, x, u, dt, nlp_kwargs=dict(relevant=rel, relevant_dyn=rel_dyn), inp_kwargs=dict(obstacles=obs), fd_step=h))


def kkt_many(ocfg, x0, xf, u_prev, dt_prev, x, u, dt, idx, workers: int = 0, obstacles=None, max_rows=None, start_x=None):
    """kkt_residuals for the instances `idx` of a batch, spread over worker processes (spawned: the caller may hold a GPU context).
    obstacles = the ABI arrays (n_obstacles (B,), n_vertices (B,O), vertices (B,O,V,2)[, radius (B,O)[, velocity (B,O,2)]]).
    start_x (B, n, 3) or None: the state trajectories the solves started from, where that is not the reference's cold start (candidate initial
    trajectories): the clearance rows of an instance are the ones associated on that trajectory."""
    import multiprocessing as mp
    import os
    idx = [int(i) for i in idx]
    if not idx:
        return {}
    def ob(i):
        if obstacles is None:
            return None
        return tuple(None if (k >= len(obstacles) or obstacles[k] is None) else obstacles[k][i] for k in range(5))
    jobs = [(ocfg, x0[i], xf[i], u_prev[i], float(dt_prev[i]), x[i], u[i], float(dt[i]), ob(i), max_rows, None if start_x is None else start_x[i]) for i in idx]
    workers = workers or max(1, min(len(jobs), (os.cpu_count() or 2) // 2, 32))
    if workers == 1 or len(jobs) < 3:
        return dict(zip(idx, map(_one, jobs)))
    # one BLAS / OpenMP thread per worker: the workers inherit the environment at spawn; 32 workers with a BLAS pool of one thread per core each (256 on the GPU boxes) spend
    # their time fighting for the cores (measured: 170 - 220 s for 60 instances instead of ~10 s)
    keys = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")
    saved = {k: os.environ.get(k) for k in keys}
    try:
        for k in keys:
            os.environ[k] = "1"
        pool = mp.get_context("spawn").Pool(workers)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    with pool:
        return dict(zip(idx, pool.map(_one, jobs, chunksize=1)))


def _one_so(args):
    ocfg, x0, xf, up, dtp, x, u, dt = args
    return second_order(ocfg, x0, xf, up, dtp, x, u, dt)


def second_order_many(ocfg, x0, xf, u_prev, dt_prev, x, u, dt, idx, workers: int = 0):
    """second_order for the instances `idx` of a batch (no clearance rows), spread over spawned worker processes like kkt_many"""
    import multiprocessing as mp
    import os
    idx = [int(i) for i in idx]
    if not idx:
        return {}
    jobs = [(ocfg, x0[i], xf[i], u_prev[i], float(dt_prev[i]), x[i], u[i], float(dt[i])) for i in idx]
    workers = workers or max(1, min(len(jobs), (os.cpu_count() or 2) // 2, 32))
    if workers == 1 or len(jobs) < 3:
        return dict(zip(idx, map(_one_so, jobs)))
    keys = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")
    saved = {k: os.environ.get(k) for k in keys}
    try:
        for k in keys:
            os.environ[k] = "1"
        pool = mp.get_context("spawn").Pool(workers)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    with pool:
        return dict(zip(idx, pool.map(_one_so, jobs, chunksize=1)))

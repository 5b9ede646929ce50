/*
 * ORACLE (test infrastructure only -- never linked, loaded or called by the product path).
 *
 * Plain-C restatement of the receding-horizon NLP solve of rst-tu-dortmund/mpc_local_planner,
 * used (a) as the fast CPU checker for the HIP path and (b) as the timed CPU baseline
 * ("cpu_baseline.kind = port" in bench.py).  It follows oracle/ipm_dense.py line by line
 * (which restates Ipopt's published primal-dual interior-point algorithm, the solver the
 * reference calls at mpc_local_planner/src/controller.cpp:388-421) but replaces numpy's dense
 * solve by a general BANDED LU with partial pivoting on the stage-interleaved KKT matrix plus a
 * bordered (Schur) step for the global dt column -- i.e. linear algebra that shares nothing
 * with the product's Riccati sweep.
 * A factorisation is accepted on its INERTIA (Ipopt's test, r04): the LU carries none, so a second pass counts the negative eigenvalues of the same assembled matrix by a
 * symmetric block elimination (kkt_negative_eigenvalues); oracle/ipm_dense.py reads the same count off LAPACK's Bunch-Kaufman factorisation, the product off its sweeps' pivots.
 *
 * PARITY UNPINNED (except the angle wrap): the reference ships no tests or golden outputs, Ipopt / corbo / teb are not vendored, and of the reference's sources only
 * utils/math_utils.h compiles in this image (everything else includes Eigen / corbo / ROS / teb headers, which are absent; no stand-ins are written).  This file follows
 * oracle/se2_nlp.py (same status: restated from the cited sources; normalize_theta / interpolate_angle held bit for bit to the executed header, tests/test_reference_math.py)
 * and is held to it and to independent scipy solves of the same NLP by the CPU tests.  The SOLVE (the iterates, the point a non-convex problem converges to) has no reference
 * to compare with: "local minimum of the restated NLP", never "what Ipopt returns".
 *
 * NLP pieces and where they come from (paths under /root/reference/mpc_local_planner/):
 *   dynamics          include/mpc_local_planner/systems/{unicycle_robot.h:59-68,simple_car.h:68-77,131-141,
 *                     kinematic_bicycle_model.h:65-77}
 *   collocation       include/mpc_local_planner/optimal_control/fd_collocation_se2.h:54-69 (forward differences), :91-108 midpoint, :130-147 crank-nicolson (stage_map)
 *   wrap              include/mpc_local_planner/utils/math_utils.h:81-91
 *   objective         (n-1)*dt (src/optimal_control/min_time_via_points_cost.cpp:52-56,120-124) |
 *                     quadratic form (src/optimal_control/quadratic_cost_se2.cpp:31-52) + terminal
 *                     (src/optimal_control/final_state_conditions_se2.cpp:30-52)
 *                     via-points (src/optimal_control/min_time_via_points_cost.cpp:39-145, findClosestPose
 *                     src/optimal_control/full_discretization_grid_base_se2.cpp:364-388) | integral form
 *                     (src/optimal_control/quadratic_cost_se2.cpp:54-83, left sum src/optimal_control/finite_differences_grid_se2.cpp:61-75)
 *   terminal ball     src/optimal_control/final_state_conditions_se2.cpp:54-64 (edge only with a free goal, finite_differences_grid_se2.cpp:128-143)
 *   clearance rows    src/optimal_control/stage_inequality_se2.cpp:50-189 (association :50-162, static rows :164-175, dynamic obstacles :99-106,177-189);
 *                     teb_local_planner distances and footprint models (point, circular, line, two circles, polygon) restated from upstream semantics
 *   rate rows         src/optimal_control/stage_inequality_se2.cpp:191-222
 *   boxes             src/controller.cpp:511,527,543 ; dt: src/optimal_control/finite_differences_variable_grid_se2.cpp:36-40
 *   cold start        src/controller.cpp:807-857 + src/optimal_control/full_discretization_grid_base_se2.cpp:192-239
 *   output            src/optimal_control/full_discretization_grid_base_se2.cpp:579-615
 * Rows are used in "solver form" (see oracle/ipm_dense.py header): c_k = dt*f - delta (= dt * reference
 * defect), rate rows multiplied by dt_prev.
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct oracle_config {
    int32_t model;
    double model_params[4];
    int32_t n;
    double dt_ref;
    int32_t dt_free;
    double dt_lb, dt_ub;
    int32_t xf_fixed[3];
    int32_t objective;          /* 0 min-time, 1 quadratic (non-integral) */
    double Q[3], R[2];
    int32_t has_Qf;
    double Qf[3];
    double u_lb[2], u_ub[2];
    double du_lb[2], du_ub[2];  /* +-1e30 = inf */
    int32_t max_iter;
    double tol;
    double mu_init;
    int32_t collocation;        /* 0 forward, 1 midpoint, 2 crank-nicolson (literal) -- same codes as include/mpc_hip.h */
    int32_t via;                /* minimum_time_via_points: objective 0 plus the via-point terms (min_time_via_points_cost.cpp:120-145) */
    int32_t vp_ordered;
    double vp_wp, vp_wo;
    int32_t ball;               /* terminal l2-ball row xd' S xd - gamma <= 0 on the free final state (final_state_conditions_se2.cpp:54-64) */
    double ball_S[3], ball_gamma;
    int32_t integral;           /* quadratic objective in integral form: stage cost x dt (left sum; quadratic_cost_se2.cpp:54-83, finite_differences_grid_se2.cpp:61-75) */
    int32_t hessian_mode;       /* (2: stage-structured quasi-Newton blocks, an experiment of this file: see bfgs_update) 0 exact Lagrangian Hessian; 1 convexified: the stage block [Hqq Hqd; Hqd' Hdd] of lam' D is replaced by its
                                 * positive semidefinite part (the product's MPC_HESSIAN_CONVEXIFIED, the "reference-like" mode) */
    int32_t hybrid;             /* quadratic_form/hybrid_cost_minimum_time: + (n-1) dt (corbo::MinTimeQuadraticControls, src/controller.cpp:616-618) */
    int32_t trapezoid;          /* grid/cost_integration_method trapezoidal_rule (integral-form terms only; finite_differences_grid_se2.cpp:63-68):
                                 * 0.5 dt (l(x_k, u_k) + l(x_{k+1}, u_k)) per interval */
    double Qo[3], Ro, Qfo[3], So[3];   /* off-diagonal terms (01, 02, 12) of the symmetric parts of the full weight matrices (src/controller.cpp:565-573,
                                        * 580-588, 656-664, 690-698); Q / R / Qf / ball_S hold the diagonals */
    double acceptable_tol;      /* Ipopt's "solved to acceptable level", which the reference's wrapper counts as success (src/controller.cpp:388-421 configures
                                 * SolverIpopt; corbo: success iff Converged or EarlyTerminated): 0 -> Ipopt's default 1e-6, < 0 -> rule off.  Same meaning
                                 * as mpc_config.acceptable_tol (include/mpc_hip.h) */
    int32_t acceptable_iter;    /* iterations in a row at that level that end the solve with status 0: 0 -> Ipopt's default 15, < 0 -> off */
    int32_t mu_strategy;        /* 0 adaptive (the default; see solve_one), 1 monotone Fiacco-McCormick -- mpc_config.mu_strategy */
    int32_t line_search;        /* mpc_config.line_search: 0 l1-merit backtracking (the globalisation of rounds 1-5), 1 Ipopt's filter line search (Waechter & Biegler
                                 * 2006, Algorithm A; no second-order correction, no restoration phase: when every trial step is refused the filter is emptied and the shortest trial step taken) */
} oracle_config;
static inline double acc_tol_of(const oracle_config* c) { return c->acceptable_tol > 0 ? c->acceptable_tol : (c->acceptable_tol < 0 ? 0.0 : 1e-6); }
static inline int acc_iter_of(const oracle_config* c) { return c->acceptable_iter > 0 ? c->acceptable_iter : (c->acceptable_iter < 0 ? 0 : 15); }

#define MINTIME(c) ((c)->objective == 0 || (c)->hybrid)
/* y = W x for the symmetric 3 x 3 matrix with diagonal d and off-diagonal terms o = (01, 02, 12) */
static inline void sym3_mul(const double d[3], const double o[3], const double x[3], double y[3]) {
    y[0] = d[0] * x[0] + o[0] * x[1] + o[1] * x[2];
    y[1] = o[0] * x[0] + d[1] * x[1] + o[2] * x[2];
    y[2] = o[1] * x[0] + o[2] * x[1] + d[2] * x[2];
}
static inline double sym3_at(const double d[3], const double o[3], int i, int j) { return i == j ? d[i] : o[i + j - 1]; }
/* weight of the state term of grid point k of the quadratic form (in units of dt when integral): one term per grid point 0..n-2 (sum or
 * left sum); trapezoidal rule: every interval gives half to either end, so the two end points keep 1/2 and the final state gets a term */
static inline double state_weight(const oracle_config* c, int n, int k) {
    if (c->objective != 1) return 0.0;
    if (c->integral && c->trapezoid) return (k == 0 || k == n - 1) ? 0.5 : 1.0;
    return k < n - 1 ? 1.0 : 0.0;
}

#define PI 3.14159265358979323846
#define KL 8      /* half bandwidth of the stage-interleaved KKT matrix */
#define KU 8
#define LDAB (2 * KL + KU + 1)

static double wrap(double th) {            /* math_utils.h:81-91 */
    if (th >= -PI && th < PI) return th;
    double m = floor(th / (2.0 * PI));
    th = th - m * 2.0 * PI;
    if (th >= PI) th -= 2.0 * PI;
    if (th < -PI) th += 2.0 * PI;
    return th;
}

/* f, G[a][j] = df_a/dq_j (q = theta, v, w), H[a][j][l] second derivatives */
static void model_derivs(const oracle_config* c, double th, double v, double w, double f[3], double G[3][3], double H[3][3][3]) {
    memset(G, 0, 9 * sizeof(double));
    memset(H, 0, 27 * sizeof(double));
    if (c->model == 3) {
        double lr = c->model_params[0], lf = c->model_params[1];
        double kap = lr / (lf + lr), t = tan(w), tp = 1.0 + t * t, tpp = 2.0 * t * tp;
        double den = 1.0 + kap * kap * t * t, beta = atan(kap * t);
        double bp = kap * tp / den, bpp = kap * (tpp * den - tp * 2.0 * kap * kap * t * tp) / (den * den);
        double cs = cos(th + beta), sn = sin(th + beta), sb = sin(beta), cb = cos(beta);
        f[0] = v * cs; f[1] = v * sn; f[2] = v * sb / lr;
        G[0][0] = -v * sn; G[0][1] = cs; G[0][2] = -v * sn * bp;
        G[1][0] = v * cs;  G[1][1] = sn; G[1][2] = v * cs * bp;
        G[2][1] = sb / lr; G[2][2] = v * cb * bp / lr;
        H[0][0][0] = -v * cs; H[0][0][1] = H[0][1][0] = -sn; H[0][0][2] = H[0][2][0] = -v * cs * bp;
        H[0][1][2] = H[0][2][1] = -sn * bp; H[0][2][2] = -v * cs * bp * bp - v * sn * bpp;
        H[1][0][0] = -v * sn; H[1][0][1] = H[1][1][0] = cs; H[1][0][2] = H[1][2][0] = -v * sn * bp;
        H[1][1][2] = H[1][2][1] = cs * bp; H[1][2][2] = -v * sn * bp * bp + v * cs * bpp;
        H[2][1][2] = H[2][2][1] = cb * bp / lr; H[2][2][2] = v * (-sb * bp * bp + cb * bpp) / lr;
        return;
    }
    double cs = cos(th), sn = sin(th);
    f[0] = v * cs; f[1] = v * sn;
    G[0][0] = -v * sn; G[0][1] = cs; G[1][0] = v * cs; G[1][1] = sn;
    H[0][0][0] = -v * cs; H[0][0][1] = H[0][1][0] = -sn;
    H[1][0][0] = -v * sn; H[1][0][1] = H[1][1][0] = cs;
    if (c->model == 0) { f[2] = w; G[2][2] = 1.0; }
    else if (c->model == 1) {
        double L = c->model_params[0], t = tan(w), tp = 1.0 + t * t;
        f[2] = v * t / L; G[2][1] = t / L; G[2][2] = v * tp / L;
        H[2][1][2] = H[2][2][1] = tp / L; H[2][2][2] = v * 2.0 * t * tp / L;
    } else {
        double L = c->model_params[0];
        f[2] = v * sin(w) / L; G[2][1] = sin(w) / L; G[2][2] = v * cos(w) / L;
        H[2][1][2] = H[2][2][1] = cos(w) / L; H[2][2][2] = -v * sin(w) / L;
    }
}

/* Increment of a collocation row in solver form, c_k = x_k + Dk(theta_k, u_k, dt) - x_{k+1}, for the three finite-difference rules
 * (fd_collocation_se2.h:54-69 forward, :91-108 midpoint, :130-147 crank-nicolson as coded: 0.5 f(x_k) + 1.5 f(x_{k+1})).  The
 * reference evaluates f at theta_k + ce dt f_2(u_k) on the constraint manifold (f_2 = heading rate, pose independent), so
 * Dk = dt sum_e wt_e f(theta_k + ce_e dt f_2, u_k).  q = (theta, v, w).  With lam: second derivatives of lam^T Dk. */
typedef struct stage_map_t { double val[3], Jq[3][3], Jdt[3], Hqq[3][3], Hqd[3], Hdd; } stage_map_t;
static void stage_map(const oracle_config* c, double th, double v, double w, double dt, const double* lam, stage_map_t* o) {
    static const double PT[3][2][2] = {{{1.0, 0.0}, {0, 0}}, {{1.0, 0.5}, {0, 0}}, {{0.5, 0.0}, {1.5, 2.0}}};
    const int npt = c->collocation == 2 ? 2 : 1;
    double f0[3], G0[3][3], H0[3][3][3];
    model_derivs(c, th, v, w, f0, G0, H0);
    const double f2 = f0[2], f2u[2] = {G0[2][1], G0[2][2]};
    memset(o, 0, sizeof(*o));
    double L[4][4]; memset(L, 0, sizeof(L));
    for (int e = 0; e < npt; ++e) {
        const double wt = PT[c->collocation][e][0], ce = PT[c->collocation][e][1];
        double f[3], G[3][3], H[3][3][3];
        model_derivs(c, th + ce * dt * f2, v, w, f, G, H);
        const double m[4] = {1.0, ce * dt * f2u[0], ce * dt * f2u[1], ce * f2};      /* d theta_e / d(theta, v, w, dt) */
        for (int a = 0; a < 3; ++a) {
            double dg[4];
            for (int j = 0; j < 4; ++j) dg[j] = G[a][0] * m[j];
            dg[1] += G[a][1]; dg[2] += G[a][2];
            o->val[a] += wt * dt * f[a];
            for (int j = 0; j < 3; ++j) o->Jq[a][j] += wt * dt * dg[j];
            o->Jdt[a] += wt * (f[a] + dt * dg[3]);
        }
        if (!lam) continue;
        double gq[3], Hl[3][3];
        for (int j = 0; j < 3; ++j) {
            gq[j] = lam[0] * G[0][j] + lam[1] * G[1][j] + lam[2] * G[2][j];
            for (int l = 0; l < 3; ++l) Hl[j][l] = lam[0] * H[0][j][l] + lam[1] * H[1][j][l] + lam[2] * H[2][j][l];
        }
        double mab[4][4]; memset(mab, 0, sizeof(mab));                               /* second derivatives of theta_e */
        for (int j = 1; j < 3; ++j) {
            for (int l = 1; l < 3; ++l) mab[j][l] = ce * dt * H0[2][j][l];
            mab[j][3] = mab[3][j] = ce * f2u[j - 1];
        }
        double Dphi[4], D2[4][4];
        for (int j = 0; j < 4; ++j) Dphi[j] = gq[0] * m[j];
        Dphi[1] += gq[1]; Dphi[2] += gq[2];
        for (int j = 0; j < 4; ++j) for (int l = 0; l < 4; ++l) D2[j][l] = Hl[0][0] * m[j] * m[l] + gq[0] * mab[j][l];
        for (int j = 1; j < 3; ++j) for (int l = 0; l < 4; ++l) { D2[j][l] += Hl[0][j] * m[l]; D2[l][j] += Hl[0][j] * m[l]; }
        for (int j = 1; j < 3; ++j) for (int l = 1; l < 3; ++l) D2[j][l] += Hl[j][l];
        for (int j = 0; j < 4; ++j) for (int l = 0; l < 4; ++l) {
            double le = dt * D2[j][l];
            if (j == 3) le += Dphi[l];
            if (l == 3) le += Dphi[j];
            L[j][l] += wt * le;
        }
    }
    if (lam) {
        for (int j = 0; j < 3; ++j) { for (int l = 0; l < 3; ++l) o->Hqq[j][l] = L[j][l]; o->Hqd[j] = L[j][3]; }
        o->Hdd = L[3][3];
    }
}

/* obstacle handling of a batch: same meaning as the fields of mpc_config (include/mpc_hip.h); point or circular footprint */
typedef struct oracle_obst {
    int32_t max_obstacles, max_vertices, max_rows;
    double min_obstacle_dist, force_inclusion_dist, cutoff_dist, footprint_radius;
    int32_t footprint_kind;         /* 0 point, 1 circle, 2 line, 3 two circles, 4 polygon (as include/mpc_hip.h) */
    double footprint_params[4];     /* line: start, end; two circles: front offset, front radius, rear offset, rear radius */
    int32_t footprint_nv;           /* polygon footprint */
    double footprint_poly[32];
    int32_t dynamic;                /* enable_dynamic_obstacles: velocities set with oracle_set_obstacle_velocities */
} oracle_obst;

/* ---------------------------------------------------------------- per-instance work area */
typedef struct {
    const oracle_config* c;
    int n, N;                 /* N = 8*(n-1): banded unknowns [u_k(2) lam_k(3) x_{k+1}(3)] per interval */
    double x0[3], xf[3], uprev[2], dtprev;
    double *X, *U, D;         /* X n*3, U (n-1)*2 */
    double *Xt, *Ut, Dt;
    double *lam, *lamn;       /* (n-1)*3 */
    double *s, *y;            /* rate rows n*4 */
    int *ron;                 /* row active flags n*4 */
    double *pl, *pu;          /* (n-1)*2 */
    double pdl, pdu;
    double *AB, *AB0;         /* band storage LDAB x N */
    int *ipiv;
    double *rhs, *bcol, *brow;/* rhs, border column (K part), work */
    double *dz_u, *dz_x;      /* steps */
    double ddt;
    double mu, rho, delta_last;
    /* clearance rows (stage_inequality_se2.cpp:50-175): obstacles of this instance + per grid point up to M associated rows */
    const struct oracle_obst* ob;
    int n_obst; const int32_t* n_vert; const double* verts; const double* radius;
    double* cent;             /* O*2 centroids */
    int* oi;                  /* n*M obstacle index or -1 */
    double *os, *oy, *ost, *ods, *ody;   /* n*M slack, multiplier, trial slack, steps */
    double *oe, *oet, *ode;              /* n*M elastic variable of the clearance rows (g + s - e = 0, e >= 0, cost rho e), its trial value and step; all zero unless elastic */
    double erho;                         /* > 0: the clearance rows are elastic with this penalty */
    double *og, *oax, *oay, *ohk;        /* n*M cached value, gradient (= -unit normal), curvature 1/|p-q| (0 on an edge interior) */
    double *oad, *ohd;                   /* dt parts (n*M, 4*n*M) of a dynamic obstacle's row when the footprint turns with the pose */
    double *oat, *oh3;                   /* third-variable parts of the rows: gradient entry n*M and Hessian entries 3*n*M -- heading for the
                                          * footprints that turn with the pose, dt for dynamic obstacles */
    const double* vel;                   /* obstacle velocities of this instance [O][2] or NULL */
    double ts, ty, tst, tds, tdy, tg, ta[3];   /* terminal-ball row: slack, multiplier, trial slack, steps, cached value and gradient */
    /* via-points of this instance (set per batch with oracle_set_via_points) and the grid point each one is attached to */
    int nvia; const double* via; int vidx[64];
    int rows_dropped;          /* clearance rows that did not fit into max_rows (obst_associate) */
    double* dual;              /* this instance's block of the kept multipliers or NULL */
    double *bf, *bf_g, *bf_e;  /* hessian_mode 2 (stage-structured BFGS): per stage the 4 x 4 block over (theta, v, w, dt) as 10 words [00 01 02 11 12 22 | 03 13 23 | 33],
                                * the gradient of lam' D_k at the point of the last update (4) and that point (4) */
    int bf_init;
    int convexify;             /* this factorisation: stage blocks of the Lagrangian curvature replaced by their positive semidefinite parts */
    double *inS, *inb;         /* scratch of kkt_negative_eigenvalues */
    int rhs_only;              /* assemble(): leave the factorised band alone, rebuild only the right-hand side / gradient pieces (they are linear in mu) */
} work_t;

static int iu(int k, int j) { return 8 * k + j; }
static int il(int k, int a) { return 8 * k + 2 + a; }
static int ixn(int k, int a) { return 8 * (k - 1) + 5 + a; }   /* x_k, k >= 1 */

static void band_zero(work_t* w) { if (!w->rhs_only) memset(w->AB, 0, sizeof(double) * LDAB * w->N); }
static void band_add(work_t* w, int i, int j, double v) {
    if (w->rhs_only) return;
    /* LAPACK band layout: A(i,j) at AB[kl+ku+i-j][j] */
    w->AB[(size_t)(KL + KU + i - j) + (size_t)LDAB * j] += v;
}
static void sym_add(work_t* w, int i, int j, double v) { band_add(w, i, j, v); if (i != j) band_add(w, j, i, v); }

/* unblocked banded LU with partial pivoting (same elimination order as LAPACK dgbtf2) */
static int band_factor(work_t* w) {
    const int N = w->N, kv = KU + KL;
    double* AB = w->AB;
    int ju = 0;
    for (int j = 0; j < N; ++j) {
        int km = KL < N - 1 - j ? KL : N - 1 - j;
        int jp = 0; double best = fabs(AB[kv + (size_t)LDAB * j]);
        for (int i = 1; i <= km; ++i) { double a = fabs(AB[kv + i + (size_t)LDAB * j]); if (a > best) { best = a; jp = i; } }
        w->ipiv[j] = j + jp;
        if (!(best > 0.0) || !isfinite(best)) return -1;
        int jm = j + KU + jp; if (jm > N - 1) jm = N - 1; if (jm > ju) ju = jm;
        if (jp != 0) for (int c2 = j; c2 <= ju; ++c2) {
            double* a = &AB[kv + jp + j - c2 + (size_t)LDAB * c2];
            double* b = &AB[kv + j - c2 + (size_t)LDAB * c2];
            double t = *a; *a = *b; *b = t;
        }
        double piv = 1.0 / AB[kv + (size_t)LDAB * j];
        for (int i = 1; i <= km; ++i) AB[kv + i + (size_t)LDAB * j] *= piv;
        for (int c2 = j + 1; c2 <= ju; ++c2) {
            double t = AB[kv + j - c2 + (size_t)LDAB * c2];
            if (t != 0.0) for (int i = 1; i <= km; ++i) AB[kv + i + j - c2 + (size_t)LDAB * c2] -= AB[kv + i + (size_t)LDAB * j] * t;
        }
    }
    return 0;
}
static void band_solve(const work_t* w, double* b) {
    const int N = w->N, kv = KU + KL;
    const double* AB = w->AB;
    for (int j = 0; j < N; ++j) {
        int km = KL < N - 1 - j ? KL : N - 1 - j;
        int p = w->ipiv[j];
        if (p != j) { double t = b[p]; b[p] = b[j]; b[j] = t; }
        for (int i = 1; i <= km; ++i) b[j + i] -= AB[kv + i + (size_t)LDAB * j] * b[j];
    }
    for (int j = N - 1; j >= 0; --j) {
        b[j] /= AB[kv + (size_t)LDAB * j];
        int lo = j - kv; if (lo < 0) lo = 0;
        for (int i = lo; i < j; ++i) b[i] -= AB[kv + i - j + (size_t)LDAB * j] * b[j];
    }
}

static int row_on(const work_t* w, int r, int q) { return w->ron[4 * r + q]; }
static double sgn(int q) { return q < 2 ? -1.0 : 1.0; }
static double lim(const work_t* w, int q) { return q < 2 ? w->c->du_lb[q] : w->c->du_ub[q - 2]; }

static double row_val_at(const work_t* w, const double* U, double D, int r, int q) {
    int n = w->n, j = q & 1;
    double ur = r < n - 1 ? U[2 * r + j] : 0.0;
    double um = r > 0 ? U[2 * (r - 1) + j] : w->uprev[j];
    double dtp = r > 0 ? D : w->dtprev;
    return sgn(q) * ((ur - um) - lim(w, q) * dtp);
}

/* c_k, objective at a point */

/* ---------------------------------------------------------------- clearance rows
 * distance of (px,py) to obstacle j with the teb semantics the reference links against (point / segment / closed polygon edge loop, no
 * inside test; an optional radius turns a 1-vertex obstacle into a circle): dist >= 0 before the radius is subtracted, unit normal
 * from the closest point to (px,py), hk = 1/|p-q| where the closest feature is a vertex (0 on an edge interior). */
static int obst_M(const work_t* w) { return w->ob ? w->ob->max_rows : 0; }
static void obst_eval(const work_t* w, double px, double py, int j, double* dist, double* nx, double* ny, double* hk) {
    const int V = w->ob->max_vertices;
    int nv = w->n_vert[j]; if (nv > V) nv = V;
    const double* v = w->verts + (size_t)2 * V * j;
    double best = 1e30, bx = 0, by = 0;
    int vert = 1;
    if (nv <= 1) { bx = v[0]; by = v[1]; best = (px - bx) * (px - bx) + (py - by) * (py - by); }
    else {
        const int ne = nv == 2 ? 1 : nv;        /* closed edge loop, no inside test (teb distance_point_to_polygon_2d) */
        for (int e = 0; e < ne; ++e) {
            const int e2 = (e + 1) % nv;
            const double ax = v[2 * e], ay = v[2 * e + 1], cx = v[2 * e2], cy = v[2 * e2 + 1];
            const double abx = cx - ax, aby = cy - ay, sq = abx * abx + aby * aby;
            double t = sq > 0 ? ((px - ax) * abx + (py - ay) * aby) / sq : 0.0;
            t = fmin(1.0, fmax(0.0, t));
            const double qx = ax + t * abx, qy = ay + t * aby;
            const double d2 = (px - qx) * (px - qx) + (py - qy) * (py - qy);
            if (d2 < best) { best = d2; bx = qx; by = qy; vert = !(t > 0 && t < 1); }
        }
    }
    const double dd = sqrt(best);
    if (dd > 0) { *nx = (px - bx) / dd; *ny = (py - by) / dd; *hk = vert ? 1.0 / dd : 0.0; }
    else { *nx = 0; *ny = 0; *hk = 0; }
    *dist = dd - (w->radius ? w->radius[j] : 0.0);
}
static void obst_centroids(work_t* w) {
    const int V = w->ob->max_vertices;
    for (int j = 0; j < w->n_obst; ++j) {
        int nv = w->n_vert[j]; if (nv > V) nv = V;
        const double* v = w->verts + (size_t)2 * V * j;
        double cx = 0, cy = 0;
        if (nv >= 3) {
            double a = 0, sx = 0, sy = 0, mx = 0, my = 0;
            for (int e = 0; e < nv; ++e) {
                const int e2 = (e + 1) % nv;
                const double cr = v[2 * e] * v[2 * e2 + 1] - v[2 * e2] * v[2 * e + 1];
                a += cr; sx += (v[2 * e] + v[2 * e2]) * cr; sy += (v[2 * e + 1] + v[2 * e2 + 1]) * cr;
                mx += v[2 * e]; my += v[2 * e + 1];
            }
            a *= 0.5;
            if (fabs(a) < 1e-12) { cx = mx / nv; cy = my / nv; } else { cx = sx / (6 * a); cy = sy / (6 * a); }
        } else if (nv == 2) { cx = 0.5 * (v[0] + v[2]); cy = 0.5 * (v[1] + v[3]); }
        else if (nv == 1) { cx = v[0]; cy = v[1]; }
        w->cent[2 * j] = cx; w->cent[2 * j + 1] = cy;
    }
}
static int fp_turns(const work_t* w);
static int is_dynamic(const work_t* w, int j);
static double turn_dist(const work_t* w, double px, double py, double th, int j, double a[3], double* hk, double h3[3]);
/* StageInequalitySE2::update (stage_inequality_se2.cpp:50-162) on the current vertex values.  Capacity rule of the batched solvers (M rows per
 * grid point): dynamic obstacles, forced ones in container order -- the M closest of them when they do not all fit (ties: lower index) --,
 * then nearest left / right.  Returns the number of rows that did not fit, summed over k = 1..n-2. */
static int obst_associate(work_t* w) {
    const int n = w->n, M = obst_M(w);
    const oracle_obst* o = w->ob;
    int dropped = 0;
    double fd[64];
    for (int k = 0; k < n; ++k) {
        for (int m = 0; m < M; ++m) w->oi[k * M + m] = -1;
        if (k < 1) continue;
        const double px = w->X[3 * k], py = w->X[3 * k + 1], th = w->X[3 * k + 2], co = cos(th), si = sin(th);
        double lmin = 1e30, rmin = 1e30; int lidx = -1, ridx = -1, cnt = 0, wanted = 0;
        for (int j = 0; j < w->n_obst; ++j) if (w->n_vert[j] > 0 && is_dynamic(w, j)) { ++wanted; if (cnt < M) w->oi[k * M + cnt++] = j; }   /* always kept (:99-106) */
        const int first_forced = cnt;
        for (int j = 0; j < w->n_obst; ++j) {
            if (w->n_vert[j] <= 0 || is_dynamic(w, j)) continue;
            double dist, nx, ny, hk;
            if (fp_turns(w)) { double a3[3], h3[3]; dist = turn_dist(w, px, py, th, j, a3, &hk, h3); }
            else { obst_eval(w, px, py, j, &dist, &nx, &ny, &hk); dist -= o->footprint_radius; }
            if (dist < o->force_inclusion_dist) {
                ++wanted;
                if (cnt < M) { fd[cnt] = dist; w->oi[k * M + cnt++] = j; }
                else if (first_forced < M) {
                    int far = first_forced;
                    for (int m = first_forced + 1; m < M; ++m) if (fd[m] >= fd[far]) far = m;
                    if (dist < fd[far]) {
                        for (int m = far; m + 1 < M; ++m) { w->oi[k * M + m] = w->oi[k * M + m + 1]; fd[m] = fd[m + 1]; }
                        w->oi[k * M + M - 1] = j; fd[M - 1] = dist;
                    }
                }
                continue;
            }
            if (dist > o->cutoff_dist) continue;
            if (co * w->cent[2 * j + 1] - w->cent[2 * j] * si > 0) { if (dist < lmin) { lmin = dist; lidx = j; } }   /* centroid as an ABSOLUTE vector (:121) */
            else { if (dist < rmin) { rmin = dist; ridx = j; } }
        }
        if (lidx >= 0) { ++wanted; if (cnt < M) w->oi[k * M + cnt++] = lidx; }
        if (ridx >= 0) { ++wanted; if (cnt < M) w->oi[k * M + cnt++] = ridx; }
        if (k < n - 1) dropped += wanted - cnt;
    }
    return dropped;
}
/* value / gradient / curvature of row (k,m) at position (px,py); 0 if the slot is empty */
static int obst_row(const work_t* w, int k, int m, double px, double py, double* g, double* ax, double* ay, double* hk) {
    const int j = w->oi[k * obst_M(w) + m];
    if (j < 0) return 0;
    double dist, nx, ny;
    obst_eval(w, px, py, j, &dist, &nx, &ny, hk);
    *g = w->ob->min_obstacle_dist - (dist - w->ob->footprint_radius);
    *ax = -nx; *ay = -ny;
    return 1;
}
static int fp_turns(const work_t* w) { return w->ob && w->ob->footprint_kind >= 2; }
static int is_dynamic(const work_t* w, int j) { return j >= 0 && w->ob && w->ob->dynamic && w->vel && (w->vel[2 * j] != 0.0 || w->vel[2 * j + 1] != 0.0); }
/* teb Line / PolygonRobotFootprint::calculateDistance for ONE world point (an obstacle centre or vertex): the point in the robot frame,
 * q = R(-theta)(v - p), against the fixed segment / closed edge loop (first closest edge, no inside test).  Returns the distance; a = gradient
 * of the row g = d_min - dist wrt (x, y, theta), hk as in obst_eval, h3 = hess g [x theta, y theta, theta theta]. */
static double fp_point_eval(const work_t* w, double px, double py, double th, double vwx, double vwy, double a[3], double* hk, double h3[3]) {
    const oracle_obst* o = w->ob;
    const double s = sin(th), c = cos(th), vx = vwx - px, vy = vwy - py;
    const double qx = c * vx + s * vy, qy = c * vy - s * vx;
    const int poly = o->footprint_kind == 4, nv = poly ? o->footprint_nv : 2, ne = nv <= 2 ? 1 : nv;
    double dx = 0, dy = 0, t = 0, best = 1.7976931348623157e308;
    for (int e = 0; e < ne; ++e) {
        const int e2 = nv == 1 ? 0 : (e + 1) % nv;
        const double a0 = poly ? o->footprint_poly[2 * e] : o->footprint_params[0], a1 = poly ? o->footprint_poly[2 * e + 1] : o->footprint_params[1];
        const double b0 = poly ? o->footprint_poly[2 * e2] : o->footprint_params[2], b1 = poly ? o->footprint_poly[2 * e2 + 1] : o->footprint_params[3];
        const double abx = b0 - a0, aby = b1 - a1, sq = abx * abx + aby * aby;
        double te = sq > 0 ? ((qx - a0) * abx + (qy - a1) * aby) / sq : 0.0;
        te = fmin(1.0, fmax(0.0, te));
        const double ex = qx - (a0 + te * abx), ey = qy - (a1 + te * aby), de = sqrt(ex * ex + ey * ey);
        if (de < best) { best = de; dx = ex; dy = ey; t = te; }
    }
    const double D = best;
    double nx = 0, ny = 0;
    *hk = 0;
    if (D > 0) { nx = dx / D; ny = dy / D; *hk = (t > 0 && t < 1) ? 0.0 : 1.0 / D; }
    const double nw = nx * qy - ny * qx;
    a[0] = c * nx - s * ny; a[1] = s * nx + c * ny; a[2] = -nw;
    const double hwx = *hk * (qy - nx * nw), hwy = *hk * (-qx - ny * nw);
    h3[0] = -((s * hwy - c * hwx) + (nx * s + ny * c));
    h3[1] = -((-s * hwx - c * hwy) + (ny * s - nx * c));
    h3[2] = -((qy * hwx - qx * hwy) - (nx * qx + ny * qy));
    return D;
}
static void fp_vertex(const oracle_obst* o, int i, double* ax, double* ay) {
    if (o->footprint_kind == 4) { *ax = o->footprint_poly[2 * i]; *ay = o->footprint_poly[2 * i + 1]; }
    else { *ax = o->footprint_params[2 * i]; *ay = o->footprint_params[2 * i + 1]; }
}
static int seg_intersect(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy) {
    const double o1 = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax), o2 = (bx - ax) * (dy - ay) - (by - ay) * (dx - ax);
    const double o3 = (dx - cx) * (ay - cy) - (dy - cy) * (ax - cx), o4 = (dx - cx) * (by - cy) - (dy - cy) * (bx - cx);
    return o1 * o2 < 0 && o3 * o4 < 0;
}
/* line / polygon footprint against obstacle j of any kind (teb Obstacle::getMinimumDistance(segment | polygon): distance_segment_to_segment_2d,
 * distance_segment_to_polygon_2d, distance_polygon_to_polygon_2d): minimum over the point-to-segment distances between the two edge loops --
 * every obstacle vertex against the footprint edges (fp_point_eval), every footprint vertex c_i(theta) = p + R(theta) a_i against the obstacle
 * edges (obst_eval, chain rule through theta) -- and 0 where two edges cross (no inside test). */
static double line_eval(const work_t* w, double px, double py, double th, int j, double a[3], double* hk, double h3[3]) {
    const oracle_obst* o = w->ob;
    const double* v = w->verts + (size_t)2 * o->max_vertices * j;
    int nvo = w->n_vert[j]; if (nvo > o->max_vertices) nvo = o->max_vertices;
    if (nvo <= 1) return fp_point_eval(w, px, py, th, v[0], v[1], a, hk, h3) - (w->radius ? w->radius[j] : 0.0);
    const int F = o->footprint_kind == 4 ? o->footprint_nv : 2;
    const double s = sin(th), c = cos(th);
    double best = 1.7976931348623157e308;
    for (int m = 0; m < nvo; ++m) {
        double am[3], hm, h3m[3];
        const double D = fp_point_eval(w, px, py, th, v[2 * m], v[2 * m + 1], am, &hm, h3m);
        if (D < best) { best = D; for (int i = 0; i < 3; ++i) { a[i] = am[i]; h3[i] = h3m[i]; } *hk = hm; }
    }
    for (int i = 0; i < F; ++i) {
        double ax_, ay_, D, nx, ny, ho;
        fp_vertex(o, i, &ax_, &ay_);
        const double rx = c * ax_ - s * ay_, ry = s * ax_ + c * ay_;
        obst_eval(w, px + rx, py + ry, j, &D, &nx, &ny, &ho);
        if (D < best) {
            best = D;
            const double wx = -ry, wy = rx, nw = nx * wx + ny * wy;
            a[0] = -nx; a[1] = -ny; a[2] = -nw; *hk = ho;
            const double hvx = ho * (wx - nx * nw), hvy = ho * (wy - ny * nw);
            h3[0] = -hvx; h3[1] = -hvy; h3[2] = -((wx * hvx + wy * hvy) - (nx * rx + ny * ry));
        }
    }
    const int nef = F <= 2 ? (F == 2 ? 1 : 0) : F, neo = nvo == 2 ? 1 : nvo;
    for (int e = 0; e < nef; ++e) {
        double a0x, a0y, a1x, a1y;
        fp_vertex(o, e, &a0x, &a0y); fp_vertex(o, (e + 1) % F, &a1x, &a1y);
        const double Ax = px + c * a0x - s * a0y, Ay = py + s * a0x + c * a0y, Bx = px + c * a1x - s * a1y, By = py + s * a1x + c * a1y;
        for (int q = 0; q < neo; ++q) {
            const int q2 = (q + 1) % nvo;
            if (seg_intersect(Ax, Ay, Bx, By, v[2 * q], v[2 * q + 1], v[2 * q2], v[2 * q2 + 1])) {
                a[0] = a[1] = a[2] = 0; *hk = 0; h3[0] = h3[1] = h3[2] = 0;
                return 0.0;
            }
        }
    }
    return best;
}
/* teb TwoCirclesRobotFootprint::calculateDistance: the closer of the two circles (front wins a tie), chain rule through the heading */
static double two_eval(const work_t* w, double px, double py, double th, int j, double a[3], double* hk, double h3[3]) {
    const double* fp = w->ob->footprint_params;
    const double s = sin(th), c = cos(th);
    double df, nfx, nfy, hf, dr, nrx, nry, hr;
    obst_eval(w, px + fp[0] * c, py + fp[0] * s, j, &df, &nfx, &nfy, &hf);
    obst_eval(w, px - fp[2] * c, py - fp[2] * s, j, &dr, &nrx, &nry, &hr);
    df -= fp[1]; dr -= fp[3];
    const int rear = dr < df;
    const double o = rear ? -fp[2] : fp[0], nx = rear ? nrx : nfx, ny = rear ? nry : nfy;
    *hk = rear ? hr : hf;
    const double wx = -o * s, wy = o * c, nw = nx * wx + ny * wy;
    a[0] = -nx; a[1] = -ny; a[2] = -nw;
    const double hvx = *hk * (wx - nx * nw), hvy = *hk * (wy - ny * nw);
    h3[0] = -hvx; h3[1] = -hvy; h3[2] = -((wx * hvx + wy * hvy) - o * (nx * c + ny * s));
    return rear ? dr : df;
}
static double turn_dist(const work_t* w, double px, double py, double th, int j, double a[3], double* hk, double h3[3]) {
    return w->ob->footprint_kind == 3 ? two_eval(w, px, py, th, j, a, hk, h3) : line_eval(w, px, py, th, j, a, hk, h3);
}
/* row (k,m) with its third-variable parts: heading (footprints that turn with the pose) or dt (dynamic obstacle: the obstacle moved by
 * k D v = the static row at p - k D v, stage_inequality_se2.cpp:177-189) */
static int obst_row3x(const work_t* w, int k, int m, double px, double py, double th, double D, double* g, double a[3], double* hk, double h3[3], double* ad, double hd[4]) {
    const int j = w->oi[k * obst_M(w) + m];
    if (j < 0) return 0;
    a[2] = 0; h3[0] = h3[1] = h3[2] = 0;
    *ad = 0; hd[0] = hd[1] = hd[2] = hd[3] = 0;
    if (is_dynamic(w, j)) {
        const double kvx = k * w->vel[2 * j], kvy = k * w->vel[2 * j + 1];
        if (fp_turns(w)) {
            /* a dynamic obstacle seen by a footprint that turns with the pose: a[2] / h3 are the heading parts, ad / hd = (g_dt, hess g [x dt,
             * y dt, dt dt, theta dt]) the dt parts of G(p - k dt v, theta) */
            *g = w->ob->min_obstacle_dist - turn_dist(w, px - D * kvx, py - D * kvy, th, j, a, hk, h3);
            if (w->c->dt_free) {
                const double ak = a[0] * kvx + a[1] * kvy;
                *ad = -ak;
                hd[0] = *hk * (kvx - a[0] * ak); hd[1] = *hk * (kvy - a[1] * ak);
                hd[2] = -*hk * (kvx * kvx + kvy * kvy - ak * ak);
                hd[3] = -(h3[0] * kvx + h3[1] * kvy);
            }
            return 1;
        }
        double dist, nx, ny;
        obst_eval(w, px - D * kvx, py - D * kvy, j, &dist, &nx, &ny, hk);
        *g = w->ob->min_obstacle_dist - (dist - w->ob->footprint_radius);
        a[0] = -nx; a[1] = -ny;
        if (w->c->dt_free) {
            const double nkv = nx * kvx + ny * kvy, hx = *hk * (kvx - nx * nkv), hy = *hk * (kvy - ny * nkv);
            a[2] = nkv; h3[0] = hx; h3[1] = hy; h3[2] = -(kvx * hx + kvy * hy);
        }
        return 1;
    }
    if (fp_turns(w)) { *g = w->ob->min_obstacle_dist - turn_dist(w, px, py, th, j, a, hk, h3); return 1; }
    return obst_row(w, k, m, px, py, g, &a[0], &a[1], hk);
}
static int obst_row3(const work_t* w, int k, int m, double px, double py, double th, double D, double* g, double a[3], double* hk, double h3[3]) {
    double ad, hd[4];
    return obst_row3x(w, k, m, px, py, th, D, g, a, hk, h3, &ad, hd);
}
/* sum |g + s| over the clearance rows at the point (X, D) with slacks sl */
static double obst_theta(const work_t* w, const double* X, double D, const double* sl) {
    const int n = w->n, M = obst_M(w);
    double th = 0;
    for (int k = 1; k < n - 1; ++k) for (int m = 0; m < M; ++m) {
        double g, a[3], hk, h3[3];
        if (obst_row3(w, k, m, X[3 * k], X[3 * k + 1], X[3 * k + 2], D, &g, a, &hk, h3)) th += fabs(g + sl[k * M + m] - (sl == w->ost ? w->oet[k * M + m] : w->oe[k * M + m]));
    }
    return th;
}

/* ---- via-points: MinTimeViaPointsCost::update (min_time_via_points_cost.cpp:39-117) + findClosestPose (...grid_base_se2.cpp:364-388) */
/* multipliers kept between control cycles (the product's dual_warm_start): per instance [0] grid size (0 = nothing kept), [1] pi_dt lower,
 * [2] pi_dt upper, [3] terminal-ball multiplier, lam 3(n-1), y 4n, pl 2(n-1), pu 2(n-1); a solve that is given an initial guess starts from
 * them with every inequality multiplier max(previous, mu0 / slack) and the barrier at mu0 */
static double* g_dual_state = NULL; static int g_dual_words = 0; static double g_dual_mu0 = 1e-3;
void oracle_set_dual_state(double* state, int words, double mu0) { g_dual_state = state; g_dual_words = words; g_dual_mu0 = mu0 > 0 ? mu0 : 1e-3; }
int oracle_dual_words(int n) { return 4 + 3 * (n - 1) + 4 * n + 4 * (n - 1); }
static int32_t* g_dropped_out = NULL;     /* [B] rows that did not fit (next batch), or NULL */
void oracle_set_rows_dropped_out(int32_t* out) { g_dropped_out = out; }
static const double* g_ovel = NULL;      /* obstacle velocities of the next batch [B][O][2] (dynamic obstacles) */
void oracle_set_obstacle_velocities(const double* vel) { g_ovel = vel; }
static const int32_t* g_nvia = NULL; static const double* g_via = NULL; static int g_vp_cap = 0;
void oracle_set_via_points(const int32_t* n_via, const double* via, int cap) { g_nvia = n_via; g_via = via; g_vp_cap = cap > 64 ? 64 : cap; }
static void via_associate(work_t* w) {
    const int n = w->n;
    int start = 0;
    for (int v = 0; v < w->nvia; ++v) {
        const double vx = w->via[3 * v], vy = w->via[3 * v + 1];
        double best = 1.7976931348623157e308; int idx = -1;
        for (int i = start; i < n - 1; ++i) {
            const double dx = vx - w->X[3 * i], dy = vy - w->X[3 * i + 1], d = sqrt(dx * dx + dy * dy);
            if (d < best) { best = d; idx = i; }
        }
        { const double dx = vx - w->X[3 * (n - 1)], dy = vy - w->X[3 * (n - 1) + 1]; if (sqrt(dx * dx + dy * dy) < best) idx = n - 1; }
        if (w->c->vp_ordered) start = idx + 2;
        if (idx > n - 2) idx = n - 2;
        if (idx < 1) idx = w->c->vp_ordered ? 1 : -1;
        w->vidx[v] = idx;
    }
}
/* value, gradient wrt (x, y, theta) and count of the via-points attached to grid point k; orientation term linear as coded (:139-142) */
static int via_terms(const work_t* w, int k, double px, double py, double th, double* val, double g[3]) {
    int m = 0;
    *val = 0; g[0] = g[1] = g[2] = 0;
    for (int v = 0; v < w->nvia; ++v) {
        if (w->vidx[v] != k) continue;
        const double dx = px - w->via[3 * v], dy = py - w->via[3 * v + 1];
        *val += w->c->vp_wp * (dx * dx + dy * dy);
        g[0] += 2 * w->c->vp_wp * dx; g[1] += 2 * w->c->vp_wp * dy;
        if (w->c->vp_wo > 0) { *val += w->c->vp_wo * wrap(w->via[3 * v + 2] - th); g[2] -= w->c->vp_wo; }
        ++m;
    }
    return m;
}

static void eval_point(const work_t* w, const double* X, const double* U, double D, double* cc, double* fobj) {
    const oracle_config* c = w->c;
    int n = w->n;
    double f = MINTIME(c) ? (n - 1) * D : 0.0;
    for (int k = 0; k < n - 1; ++k) {
        stage_map_t sm;
        stage_map(c, X[3 * k + 2], U[2 * k], U[2 * k + 1], D, NULL, &sm);
        cc[3 * k + 0] = sm.val[0] - (X[3 * (k + 1)] - X[3 * k]);
        cc[3 * k + 1] = sm.val[1] - (X[3 * (k + 1) + 1] - X[3 * k + 1]);
        cc[3 * k + 2] = sm.val[2] - wrap(X[3 * (k + 1) + 2] - X[3 * k + 2]);
        if (c->via && k >= 1) { double vv, vg[3]; via_terms(w, k, X[3 * k], X[3 * k + 1], X[3 * k + 2], &vv, vg); f += vv; }
        if (c->objective == 1) {
            const double w8 = c->integral ? D : 1.0, v = U[2 * k], om = U[2 * k + 1];
            f += w8 * (c->R[0] * v * v + c->R[1] * om * om + 2 * c->Ro * v * om);
        }
    }
    if (c->objective == 1) for (int k = 0; k < n; ++k) {
        const double ws = state_weight(c, n, k);
        if (ws == 0.0) continue;
        double xd[3] = {X[3 * k] - w->xf[0], X[3 * k + 1] - w->xf[1], wrap(X[3 * k + 2] - w->xf[2])}, qx[3];
        sym3_mul(c->Q, c->Qo, xd, qx);
        f += (c->integral ? D : 1.0) * ws * (xd[0] * qx[0] + xd[1] * qx[1] + xd[2] * qx[2]);
    }
    if (c->has_Qf && !(c->xf_fixed[0] && c->xf_fixed[1] && c->xf_fixed[2])) {
        const double* xl = &X[3 * (n - 1)];
        double xd[3] = {xl[0] - w->xf[0], xl[1] - w->xf[1], wrap(xl[2] - w->xf[2])}, qx[3];       /* a fixed component sits on the goal: xd = 0 */
        sym3_mul(c->Qf, c->Qfo, xd, qx);
        f += xd[0] * qx[0] + xd[1] * qx[1] + xd[2] * qx[2];
    }
    *fobj = f;
}

static int ball_on(const work_t* w) { return w->c->ball && !(w->c->xf_fixed[0] && w->c->xf_fixed[1] && w->c->xf_fixed[2]); }
static double ball_eval(const work_t* w, const double* X, double a[3]) {
    const oracle_config* c = w->c;
    double g = -c->ball_gamma, xd[3], sx[3];
    for (int i = 0; i < 3; ++i) {
        xd[i] = c->xf_fixed[i] ? 0.0 : X[3 * (w->n - 1) + i] - w->xf[i];
        if (i == 2) xd[i] = wrap(xd[i]);
    }
    sym3_mul(c->ball_S, c->So, xd, sx);
    for (int i = 0; i < 3; ++i) { a[i] = c->xf_fixed[i] ? 0.0 : 2 * sx[i]; g += xd[i] * sx[i]; }
    return g;
}
static double barrier_logs(const work_t* w, const double* U, double D, const double* s, const double* os) {
    const oracle_config* c = w->c;
    int n = w->n;
    double a = 0.0;
    for (int k = 0; k < n - 1; ++k) for (int j = 0; j < 2; ++j) a += log(U[2 * k + j] - c->u_lb[j]) + log(c->u_ub[j] - U[2 * k + j]);
    if (c->dt_free) a += log(D - c->dt_lb) + log(c->dt_ub - D);
    for (int r = 0; r < n; ++r) for (int q = 0; q < 4; ++q) if (row_on(w, r, q)) a += log(s[4 * r + q]);
    for (int k = 1, M = obst_M(w); k < n - 1; ++k) for (int m = 0; m < M; ++m) if (w->oi[k * M + m] >= 0) {
        a += log(os[k * M + m]);
        if (w->erho > 0) a += log(os == w->ost ? w->oet[k * M + m] : w->oe[k * M + m]);
    }
    return a;
}

typedef struct { double rd, rp, cmin, cmax, csum, sm, sb, theta; int nm, nb; } err_t;     /* csum: sum of the nb complementarity products */

static void kkt_terms(const work_t* w, const double* cc, err_t* e) {
    const oracle_config* c = w->c;
    int n = w->n;
    memset(e, 0, sizeof(*e));
    e->cmin = 1e30;
    double rdd = MINTIME(c) ? (double)(n - 1) : 0.0;
    for (int k = 0; k < n - 1; ++k) {
        const double* lam = &w->lam[3 * k];
        double v = w->U[2 * k], om = w->U[2 * k + 1];
        stage_map_t sm;
        stage_map(c, w->X[3 * k + 2], v, om, w->D, NULL, &sm);
        double gq[3];              /* lam^T dDk/dq */
        for (int j = 0; j < 3; ++j) gq[j] = lam[0] * sm.Jq[0][j] + lam[1] * sm.Jq[1][j] + lam[2] * sm.Jq[2][j];
        for (int i = 0; i < 3; ++i) { double a = fabs(cc[3 * k + i]); if (a > e->rp) e->rp = a; e->theta += a; e->sm += fabs(lam[i]); }
        e->nm += 3;
        rdd += lam[0] * sm.Jdt[0] + lam[1] * sm.Jdt[1] + lam[2] * sm.Jdt[2];
        double gx[3] = {0, 0, 0}, gu[2] = {0, 0};
        if (c->objective == 1) {
            double xd[3] = {w->X[3 * k] - w->xf[0], w->X[3 * k + 1] - w->xf[1], wrap(w->X[3 * k + 2] - w->xf[2])}, qx[3];
            const double w8 = c->integral ? w->D : 1.0, ws = state_weight(c, n, k);
            sym3_mul(c->Q, c->Qo, xd, qx);
            for (int i = 0; i < 3; ++i) gx[i] = 2 * ws * qx[i] * w8;
            gu[0] = 2 * (c->R[0] * v + c->Ro * om) * w8; gu[1] = 2 * (c->R[1] * om + c->Ro * v) * w8;
            if (c->integral) rdd += ws * (xd[0] * qx[0] + xd[1] * qx[1] + xd[2] * qx[2]) + c->R[0] * v * v + c->R[1] * om * om + 2 * c->Ro * v * om;
        }
        if (c->via && k >= 1) { double vv, vg[3]; via_terms(w, k, w->X[3 * k], w->X[3 * k + 1], w->X[3 * k + 2], &vv, vg); for (int i = 0; i < 3; ++i) gx[i] += vg[i]; }
        double osx = 0, osy = 0, ost = 0;
        if (k >= 1) for (int m = 0, M = obst_M(w); m < M; ++m) {
            double g, a3[3], hk, h3[3], ad, hd[4];
            work_t* wm = (work_t*)w;           /* the caches are scratch */
            if (!obst_row3x(w, k, m, w->X[3 * k], w->X[3 * k + 1], w->X[3 * k + 2], w->D, &g, a3, &hk, h3, &ad, hd)) continue;
            const double ax = a3[0], ay = a3[1];
            wm->og[k * M + m] = g; wm->oax[k * M + m] = ax; wm->oay[k * M + m] = ay; wm->ohk[k * M + m] = hk;
            wm->oat[k * M + m] = a3[2]; for (int i = 0; i < 3; ++i) wm->oh3[3 * (k * M + m) + i] = h3[i];
            wm->oad[k * M + m] = ad; for (int i = 0; i < 4; ++i) wm->ohd[4 * (k * M + m) + i] = hd[i];
            const double sl = w->os[k * M + m], y = w->oy[k * M + m], res = g + sl - w->oe[k * M + m];
            if (fabs(res) > e->rp) e->rp = fabs(res);
            e->theta += fabs(res);
            if (sl * y < e->cmin) e->cmin = sl * y; if (sl * y > e->cmax) e->cmax = sl * y; e->csum += sl * y;
            e->sb += y; e->nb += 1;
            if (w->erho > 0) {      /* the elastic variable's own complementarity, e (rho - y); a solve may only end with e negligible: it counts as infeasibility of the ORIGINAL row */
                const double ce = w->oe[k * M + m] * (w->erho - y);
                if (ce < e->cmin) e->cmin = ce; if (ce > e->cmax) e->cmax = ce; e->csum += ce;
                e->sb += w->erho - y; e->nb += 1;
                if (w->oe[k * M + m] > e->rp) e->rp = w->oe[k * M + m];
            }
            osx += y * ax; osy += y * ay;
            if (is_dynamic(w, w->oi[k * M + m]) && fp_turns(w)) { rdd += y * ad; ost += y * a3[2]; }
            else if (is_dynamic(w, w->oi[k * M + m])) rdd += y * a3[2]; else ost += y * a3[2];
        }
        if (k >= 1) {
            const double* lp = &w->lam[3 * (k - 1)];
            double r[3] = {gx[0] + osx + lam[0] - lp[0], gx[1] + osy + lam[1] - lp[1], gx[2] + ost + lam[2] + gq[0] - lp[2]};
            for (int i = 0; i < 3; ++i) if (fabs(r[i]) > e->rd) e->rd = fabs(r[i]);
        }
        for (int j = 0; j < 2; ++j) {
            double u = w->U[2 * k + j], pl = w->pl[2 * k + j], pu = w->pu[2 * k + j];
            double r = gu[j] + gq[1 + j] - pl + pu;
            for (int q = j; q < 4; q += 2) {
                if (row_on(w, k, q)) r += sgn(q) * w->y[4 * k + q];
                if (row_on(w, k + 1, q)) r -= sgn(q) * w->y[4 * (k + 1) + q];
            }
            if (fabs(r) > e->rd) e->rd = fabs(r);
            double cl = (u - c->u_lb[j]) * pl, cu = (c->u_ub[j] - u) * pu;
            if (cl < e->cmin) e->cmin = cl; if (cu < e->cmin) e->cmin = cu;
            if (cl > e->cmax) e->cmax = cl; if (cu > e->cmax) e->cmax = cu; e->csum += cl + cu;
            e->sb += pl + pu; e->nb += 2;
        }
    }
    {
        const double* lp = &w->lam[3 * (n - 2)];
        double xdf[3] = {w->X[3 * (n - 1)] - w->xf[0], w->X[3 * (n - 1) + 1] - w->xf[1], wrap(w->X[3 * (n - 1) + 2] - w->xf[2])}, qfx[3], qtx[3];
        const double wt = state_weight(c, n, n - 1);              /* trapezoidal rule: the final state carries half an interval's state cost */
        sym3_mul(c->Qf, c->Qfo, xdf, qfx);
        sym3_mul(c->Q, c->Qo, xdf, qtx);
        if (wt != 0.0) rdd += wt * (xdf[0] * qtx[0] + xdf[1] * qtx[1] + xdf[2] * qtx[2]);
        for (int i = 0; i < 3; ++i) if (!c->xf_fixed[i]) {
            double g = 0.0;
            if (c->has_Qf) g = 2 * qfx[i];
            if (wt != 0.0) g += 2 * wt * w->D * qtx[i];
            if (ball_on(w)) { double ta[3]; ball_eval(w, w->X, ta); g += w->ty * ta[i]; }
            if (fabs(g - lp[i]) > e->rd) e->rd = fabs(g - lp[i]);
        }
        if (ball_on(w)) {
            work_t* wm = (work_t*)w;
            wm->tg = ball_eval(w, w->X, wm->ta);
            const double res = w->tg + w->ts;
            if (fabs(res) > e->rp) e->rp = fabs(res);
            e->theta += fabs(res);
            if (w->ts * w->ty < e->cmin) e->cmin = w->ts * w->ty; if (w->ts * w->ty > e->cmax) e->cmax = w->ts * w->ty; e->csum += w->ts * w->ty;
            e->sb += w->ty; e->nb += 1;
        }
    }
    for (int r = 0; r < n; ++r) for (int q = 0; q < 4; ++q) if (row_on(w, r, q)) {
        double s = w->s[4 * r + q], y = w->y[4 * r + q];
        double res = row_val_at(w, w->U, w->D, r, q) + s;
        if (fabs(res) > e->rp) e->rp = fabs(res);
        e->theta += fabs(res);
        if (s * y < e->cmin) e->cmin = s * y; if (s * y > e->cmax) e->cmax = s * y; e->csum += s * y;
        e->sb += y; e->nb += 1;
        if (r > 0) rdd -= sgn(q) * lim(w, q) * y;
    }
    if (c->dt_free) {
        rdd += -w->pdl + w->pdu;
        if (fabs(rdd) > e->rd) e->rd = fabs(rdd);
        double cl = (w->D - c->dt_lb) * w->pdl, cu = (c->dt_ub - w->D) * w->pdu;
        if (cl < e->cmin) e->cmin = cl; if (cu < e->cmin) e->cmin = cu;
        if (cl > e->cmax) e->cmax = cl; if (cu > e->cmax) e->cmax = cu; e->csum += cl + cu;
        e->sb += w->pdl + w->pdu; e->nb += 2;
    }
    e->sm += e->sb; e->nm += e->nb;
}
static double err_value(const err_t* e, double mu) {
    const double smax = 100.0;
    double sd = fmax(smax, e->sm / (e->nm > 0 ? e->nm : 1)) / smax;
    double sc = fmax(smax, e->sb / (e->nb > 0 ? e->nb : 1)) / smax;
    double comp = e->nb > 0 ? fmax(e->cmax - mu, mu - e->cmin) : 0.0;
    return fmax(e->rd / sd, fmax(e->rp, comp / sc));
}

/* Assemble the condensed KKT system (banded part + dt border) and its right-hand side.
 * Unknown order per interval k: u_k(2), lambda_k(3), x_{k+1}(3).  dt is the border unknown.
 * hd / Hdd: gradient and Hessian entries of dt; bcol: coupling column K[:,dt]. */
static int g_variant = 1;          /* 1 = the algorithm (skip the delta = 0 attempt after a failed one); 0/2/3: experiments */
/* EXPERIMENT (oracle_set_variant(3)): replace the stage block [Hqq Hqd; Hqd' Hdd] of lam' D by its positive semidefinite part
 * (cyclic Jacobi on the symmetric 4x4, negative eigenvalues clipped to 0).  Not part of the algorithm the product implements. */
static void psd_project4(stage_map_t* sm, int drop_theta) {
    double A[4][4], V[4][4];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) A[i][j] = sm->Hqq[i][j]; A[i][3] = A[3][i] = sm->Hqd[i]; }
    A[3][3] = sm->Hdd;
    if (drop_theta) for (int j = 0; j < 4; ++j) A[0][j] = A[j][0] = 0.0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = i == j;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
        if (off < 1e-30) break;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) {
            if (fabs(A[p][q]) < 1e-300) continue;
            const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0)), c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < 4; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
            for (int k = 0; k < 4; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
            for (int k = 0; k < 4; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
        }
    }
    double R[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double v = 0; for (int k = 0; k < 4; ++k) v += V[i][k] * (A[k][k] > 0 ? A[k][k] : 0.0) * V[j][k]; R[i][j] = v; }
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) sm->Hqq[i][j] = R[i][j]; sm->Hqd[i] = R[i][3]; }
    sm->Hdd = R[3][3];
}

/* cheap convexification of the stage block [Hqq Hqd; Hqd' Hdd] of lam' D: every diagonal entry is raised to the sum of the absolute off-diagonal entries of
 * its row (Gershgorin: the block becomes diagonally dominant with a non-negative diagonal, hence positive semidefinite) */
static void gershgorin4(stage_map_t* sm, int drop_theta) {
    double A[4][4];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) A[i][j] = sm->Hqq[i][j]; A[i][3] = A[3][i] = sm->Hqd[i]; }
    A[3][3] = sm->Hdd;
    if (drop_theta) for (int j = 0; j < 4; ++j) A[0][j] = A[j][0] = 0.0;
    for (int i = 0; i < 4; ++i) { double r = 0; for (int j = 0; j < 4; ++j) if (j != i) r += fabs(A[i][j]); if (A[i][i] < r) A[i][i] = r; }
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) sm->Hqq[i][j] = A[i][j]; sm->Hqd[i] = A[i][3]; }
    sm->Hdd = A[3][3];
}
static void assemble(work_t* w, const double* cc, double delta, double dc, double* Hdd, double* hd) {
    const oracle_config* c = w->c;
    int n = w->n, N = w->N;
    band_zero(w);
    memset(w->rhs, 0, sizeof(double) * N);
    memset(w->bcol, 0, sizeof(double) * N);
    double D = w->D, mu = w->mu;
    double hdd = 0.0, gd = 0.0;
    if (MINTIME(c)) gd += (double)(n - 1);
    if (c->dt_free) {
        double dl = D - c->dt_lb, du = c->dt_ub - D;
        hdd += w->pdl / dl + w->pdu / du + delta;
        gd += -mu / dl + mu / du;
    }
    for (int k = 0; k < n - 1; ++k) {
        const double* lam = &w->lam[3 * k];
        double v = w->U[2 * k], om = w->U[2 * k + 1];
        stage_map_t sm;
        stage_map(c, w->X[3 * k + 2], v, om, D, lam, &sm);
        int qi[3] = {k >= 1 ? ixn(k, 2) : -1, iu(k, 0), iu(k, 1)};
        /* collocation rows: c = x_k + Dk(theta_k, u_k, dt) - x_{k+1} */
        for (int a = 0; a < 3; ++a) {
            int row = il(k, a);
            if (k >= 1) sym_add(w, row, ixn(k, a), 1.0);
            if (k + 1 < n - 1 || !c->xf_fixed[a]) sym_add(w, row, ixn(k + 1, a), -1.0);
            for (int j = 0; j < 3; ++j) if (qi[j] >= 0) sym_add(w, row, qi[j], sm.Jq[a][j]);
            w->bcol[row] += sm.Jdt[a];
            w->rhs[row] = -cc[3 * k + a];
        }
        if (k == n - 2) for (int a = 0; a < 3; ++a) if (c->xf_fixed[a]) band_add(w, il(k, a), il(k, a), -dc);
        if (c->hessian_mode == 2) {           /* stage-structured quasi-Newton block instead of the exact curvature of lam' D_k */
            const double* b = &w->bf[10 * k];
            sm.Hqq[0][0] = b[0]; sm.Hqq[0][1] = sm.Hqq[1][0] = b[1]; sm.Hqq[0][2] = sm.Hqq[2][0] = b[2]; sm.Hqq[1][1] = b[3]; sm.Hqq[1][2] = sm.Hqq[2][1] = b[4]; sm.Hqq[2][2] = b[5];
            sm.Hqd[0] = b[6]; sm.Hqd[1] = b[7]; sm.Hqd[2] = b[8]; sm.Hdd = b[9];
            if (qi[0] < 0) { sm.Hqq[0][0] = sm.Hqq[0][1] = sm.Hqq[1][0] = sm.Hqq[0][2] = sm.Hqq[2][0] = 0; sm.Hqd[0] = 0; }      /* theta_0 is not a variable */
        }
        if (g_variant == 3 || c->hessian_mode == 1 || w->convexify == 1) psd_project4(&sm, qi[0] < 0);
        else if (w->convexify == 2) gershgorin4(&sm, qi[0] < 0);      /* stage-wise convexification of the Lagrangian curvature */
        /* Lagrangian curvature */
        for (int j = 0; j < 3; ++j) {
            if (qi[j] < 0) continue;
            for (int l = j; l < 3; ++l) {
                if (qi[l] < 0) continue;
                sym_add(w, qi[j], qi[l], sm.Hqq[j][l]);
            }
            w->bcol[qi[j]] += sm.Hqd[j];
        }
        hdd += sm.Hdd;
        /* objective */
        if (c->objective == 1) {
            double xd[3] = {w->X[3 * k] - w->xf[0], w->X[3 * k + 1] - w->xf[1], wrap(w->X[3 * k + 2] - w->xf[2])}, qx[3];
            const double w8 = c->integral ? D : 1.0, ws = state_weight(c, n, k), v = w->U[2 * k], om = w->U[2 * k + 1];
            const double ru[2] = {c->R[0] * v + c->Ro * om, c->R[1] * om + c->Ro * v};
            sym3_mul(c->Q, c->Qo, xd, qx);
            if (k >= 1) for (int i = 0; i < 3; ++i) {
                band_add(w, ixn(k, i), ixn(k, i), 2 * ws * c->Q[i] * w8);
                for (int j = i + 1; j < 3; ++j) sym_add(w, ixn(k, i), ixn(k, j), 2 * ws * sym3_at(c->Q, c->Qo, i, j) * w8);
                w->rhs[ixn(k, i)] -= 2 * ws * qx[i] * w8;
            }
            for (int j = 0; j < 2; ++j) { band_add(w, iu(k, j), iu(k, j), 2 * c->R[j] * w8); w->rhs[iu(k, j)] -= 2 * ru[j] * w8; }
            sym_add(w, iu(k, 0), iu(k, 1), 2 * c->Ro * w8);
            if (c->integral) {       /* d/d dt and the mixed second derivatives of  dt * (ws xd'Q xd + u'R u) */
                if (k >= 1) for (int i = 0; i < 3; ++i) w->bcol[ixn(k, i)] += 2 * ws * qx[i];
                for (int j = 0; j < 2; ++j) w->bcol[iu(k, j)] += 2 * ru[j];
                gd += ws * (xd[0] * qx[0] + xd[1] * qx[1] + xd[2] * qx[2]) + v * ru[0] + om * ru[1];
            }
        }
        if (c->via && k >= 1) {
            double vv, vg[3];
            const int m = via_terms(w, k, w->X[3 * k], w->X[3 * k + 1], w->X[3 * k + 2], &vv, vg);
            for (int i = 0; i < 2; ++i) band_add(w, ixn(k, i), ixn(k, i), 2 * c->vp_wp * m);
            for (int i = 0; i < 3; ++i) w->rhs[ixn(k, i)] -= vg[i];
        }
        /* control box + regularisation */
        for (int j = 0; j < 2; ++j) {
            double u = w->U[2 * k + j], dl = u - c->u_lb[j], du = c->u_ub[j] - u;
            band_add(w, iu(k, j), iu(k, j), w->pl[2 * k + j] / dl + w->pu[2 * k + j] / du + delta);
            w->rhs[iu(k, j)] -= -mu / dl + mu / du;
        }
        if (k >= 1) for (int i = 0; i < 3; ++i) band_add(w, ixn(k, i), ixn(k, i), delta);
        /* clearance rows, condensed: + sigma a a^T + y hess(g),  hess(g) = -hk (I - a a^T);  gradient + a * ybar */
        if (k >= 1) for (int m = 0, M = obst_M(w); m < M; ++m) {
            if (w->oi[k * M + m] < 0) continue;
            const double sl = w->os[k * M + m], y = w->oy[k * M + m], g = w->og[k * M + m];
            const double ax = w->oax[k * M + m], ay = w->oay[k * M + m], hk = w->ohk[k * M + m];
            double sig = y / sl, ybar = mu / sl + sig * (g + sl);
            if (w->erho > 0) {      /* elastic row, (s, e) condensed: sigma = 1 / (s / y + e / (rho - y)), ybar = y + sigma (res + mu / y - s - mu / (rho - y) + e) */
                const double ee = w->oe[k * M + m], wv = w->erho - y;
                sig = 1.0 / (sl / y + ee / wv); ybar = y + sig * ((g + sl - ee) + mu / y - sl - mu / wv + ee);
            }
            band_add(w, ixn(k, 0), ixn(k, 0), sig * ax * ax - y * hk * (1.0 - ax * ax));
            sym_add(w, ixn(k, 0), ixn(k, 1), sig * ax * ay + y * hk * ax * ay);
            band_add(w, ixn(k, 1), ixn(k, 1), sig * ay * ay - y * hk * (1.0 - ay * ay));
            w->rhs[ixn(k, 0)] -= ax * ybar; w->rhs[ixn(k, 1)] -= ay * ybar;
            const double at = w->oat[k * M + m], *h3 = &w->oh3[3 * (k * M + m)];
            if (is_dynamic(w, w->oi[k * M + m]) && fp_turns(w)) {     /* heading parts in the band + dt parts in the border */
                const double ad = w->oad[k * M + m], *hd = &w->ohd[4 * (k * M + m)];
                sym_add(w, ixn(k, 0), ixn(k, 2), sig * ax * at + y * h3[0]);
                sym_add(w, ixn(k, 1), ixn(k, 2), sig * ay * at + y * h3[1]);
                band_add(w, ixn(k, 2), ixn(k, 2), sig * at * at + y * h3[2]);
                w->rhs[ixn(k, 2)] -= at * ybar;
                w->bcol[ixn(k, 0)] += sig * ax * ad + y * hd[0];
                w->bcol[ixn(k, 1)] += sig * ay * ad + y * hd[1];
                w->bcol[ixn(k, 2)] += sig * at * ad + y * hd[3];
                hdd += sig * ad * ad + y * hd[2];
                gd += ad * ybar;
            } else if (is_dynamic(w, w->oi[k * M + m])) {          /* third variable = dt: border column, dt-dt entry, dt gradient */
                w->bcol[ixn(k, 0)] += sig * ax * at + y * h3[0];
                w->bcol[ixn(k, 1)] += sig * ay * at + y * h3[1];
                hdd += sig * at * at + y * h3[2];
                gd += at * ybar;
            } else if (fp_turns(w)) {                       /* third variable = heading */
                sym_add(w, ixn(k, 0), ixn(k, 2), sig * ax * at + y * h3[0]);
                sym_add(w, ixn(k, 1), ixn(k, 2), sig * ay * at + y * h3[1]);
                band_add(w, ixn(k, 2), ixn(k, 2), sig * at * at + y * h3[2]);
                w->rhs[ixn(k, 2)] -= at * ybar;
            }
        }
    }
    /* terminal state block: free components are variables, fixed ones are pinned (dx = 0) */
    {
        double xdf[3] = {w->X[3 * (n - 1)] - w->xf[0], w->X[3 * (n - 1) + 1] - w->xf[1], wrap(w->X[3 * (n - 1) + 2] - w->xf[2])}, qfx[3], qtx[3];
        const double wt = state_weight(c, n, n - 1);              /* trapezoidal rule: wt * dt * xd' Q xd on the final state */
        sym3_mul(c->Qf, c->Qfo, xdf, qfx);
        sym3_mul(c->Q, c->Qo, xdf, qtx);
        if (wt != 0.0) gd += wt * (xdf[0] * qtx[0] + xdf[1] * qtx[1] + xdf[2] * qtx[2]);
        for (int i = 0; i < 3; ++i) {
            int id = ixn(n - 1, i);
            if (c->xf_fixed[i]) { band_add(w, id, id, 1.0); continue; }
            band_add(w, id, id, delta);
            for (int j = i; j < 3; ++j) {
                if (c->xf_fixed[j]) continue;
                double h = 0.0;
                if (c->has_Qf) h += 2 * sym3_at(c->Qf, c->Qfo, i, j);
                if (wt != 0.0) h += 2 * wt * D * sym3_at(c->Q, c->Qo, i, j);
                if (h != 0.0) sym_add(w, id, ixn(n - 1, j), h);
            }
            if (c->has_Qf) w->rhs[id] -= 2 * qfx[i];
            if (wt != 0.0) { w->rhs[id] -= 2 * wt * D * qtx[i]; w->bcol[id] += 2 * wt * qtx[i]; }
        }
    }
    if (ball_on(w)) {        /* condensed terminal-ball row: + sigma a a' + 2 y S, gradient + a ybar */
        const double sig = w->ty / w->ts, ybar = mu / w->ts + sig * (w->tg + w->ts);
        for (int i = 0; i < 3; ++i) {
            if (c->xf_fixed[i]) continue;
            band_add(w, ixn(n - 1, i), ixn(n - 1, i), 2 * w->ty * c->ball_S[i] + sig * w->ta[i] * w->ta[i]);
            for (int j = i + 1; j < 3; ++j) if (!c->xf_fixed[j]) sym_add(w, ixn(n - 1, i), ixn(n - 1, j), 2 * w->ty * sym3_at(c->ball_S, c->So, i, j) + sig * w->ta[i] * w->ta[j]);
            w->rhs[ixn(n - 1, i)] -= w->ta[i] * ybar;
        }
    }
    /* rate rows, condensed: + sigma a a^T, gradient + a * ybar */
    for (int r = 0; r < n; ++r) for (int q = 0; q < 4; ++q) if (row_on(w, r, q)) {
        int j = q & 1;
        double sg = sgn(q), L = lim(w, q);
        double s = w->s[4 * r + q], y = w->y[4 * r + q], sig = y / s;
        double res = row_val_at(w, w->U, w->D, r, q) + s;
        double ybar = mu / s + sig * res;
        int i1 = r < n - 1 ? iu(r, j) : -1, i2 = r > 0 ? iu(r - 1, j) : -1;
        double ad = r > 0 ? -sg * L : 0.0;
        if (i1 >= 0) { band_add(w, i1, i1, sig); w->rhs[i1] -= sg * ybar; w->bcol[i1] += sig * sg * ad; }
        if (i2 >= 0) { band_add(w, i2, i2, sig); w->rhs[i2] -= -sg * ybar; w->bcol[i2] += sig * (-sg) * ad; }
        if (i1 >= 0 && i2 >= 0) sym_add(w, i1, i2, -sig);
        hdd += sig * ad * ad;
        gd += ad * ybar;
    }
    *Hdd = hdd;
    *hd = gd;
}

static void ftb(double val, double dval, double tau, double* alpha) { if (dval < 0) { double a = -tau * val / dval; if (a < *alpha) *alpha = a; } }

typedef struct { double hdz, dz2, dphi, a_p, a_d, dzmax, clam, nunu; } step_t;
/* Everything that follows from a solution of the Newton system (w->rhs, w->ddt) at barrier parameter mu: primal steps, slack / multiplier steps of every
 * inequality row, fraction-to-boundary step lengths at tau, directional derivative of the barrier function, curvature terms */
static void derive_step(work_t* w, const double* cc, double mu, double tau, double* ds, double* dy, step_t* o) {
    const oracle_config* c = w->c;
    const int n = w->n;
    /* curvature dz^T (Hc + delta I) dz = -h^T dz + c^T lam+ - dc |lam+_term|^2, with h = -(rhs of the primal rows) */
    double clam = 0, nunu = 0, hdz = 0, dz2 = 0, dphi = 0, a_p = 1, a_d = 1, dzmax = 0;
    /* re-assemble gradient pieces (cheap): h^T dz = gphi.dz + ybar.(Jg dz) */
    double ddt = w->ddt;
    if (c->dt_free) {
        double dl = w->D - c->dt_lb, du = c->dt_ub - w->D, gb = -mu / dl + mu / du;
        hdz += gb * ddt; dphi += gb * ddt; dz2 += ddt * ddt; if (fabs(ddt) > dzmax) dzmax = fabs(ddt);
        ftb(dl, ddt, tau, &a_p); ftb(du, -ddt, tau, &a_p);
        ftb(w->pdl, mu / dl - w->pdl - (w->pdl / dl) * ddt, tau, &a_d);
        ftb(w->pdu, mu / du - w->pdu + (w->pdu / du) * ddt, tau, &a_d);
    }
    if (MINTIME(c)) { hdz += (n - 1) * ddt; dphi += (n - 1) * ddt; }
    if (c->objective == 1 && c->integral) for (int k = 0; k < n; ++k) {        /* d/d dt of the integral-form stage costs */
        double xd[3] = {w->X[3 * k] - w->xf[0], w->X[3 * k + 1] - w->xf[1], wrap(w->X[3 * k + 2] - w->xf[2])}, qx[3], sc;
        sym3_mul(c->Q, c->Qo, xd, qx);
        sc = state_weight(c, n, k) * (xd[0] * qx[0] + xd[1] * qx[1] + xd[2] * qx[2]);
        if (k < n - 1) { const double v = w->U[2 * k], om = w->U[2 * k + 1]; sc += c->R[0] * v * v + c->R[1] * om * om + 2 * c->Ro * v * om; }
        hdz += sc * ddt; dphi += sc * ddt;
    }
    for (int k = 0; k < n - 1; ++k) {
        for (int j = 0; j < 2; ++j) {
            double du_ = w->rhs[iu(k, j)], u = w->U[2 * k + j];
            double dl = u - c->u_lb[j], du = c->u_ub[j] - u, pl = w->pl[2 * k + j], pu = w->pu[2 * k + j];
            double gb = -mu / dl + mu / du;
            if (c->objective == 1) gb += 2 * (c->R[j] * u + c->Ro * w->U[2 * k + 1 - j]) * (c->integral ? w->D : 1.0);
            hdz += gb * du_; dphi += gb * du_; dz2 += du_ * du_; if (fabs(du_) > dzmax) dzmax = fabs(du_);
            ftb(dl, du_, tau, &a_p); ftb(du, -du_, tau, &a_p);
            ftb(pl, mu / dl - pl - (pl / dl) * du_, tau, &a_d);
            ftb(pu, mu / du - pu + (pu / du) * du_, tau, &a_d);
            w->dz_u[2 * k + j] = du_;
        }
        for (int a = 0; a < 3; ++a) {
            double l = w->rhs[il(k, a)];
            w->lamn[3 * k + a] = l;
            clam += cc[3 * k + a] * l;
            if (k == n - 2 && c->xf_fixed[a]) nunu += l * l;
            double dx = (k + 1 < n - 1 || !c->xf_fixed[a]) ? w->rhs[ixn(k + 1, a)] : 0.0;
            w->dz_x[3 * (k + 1) + a] = dx;
            dz2 += dx * dx; if (fabs(dx) > dzmax) dzmax = fabs(dx);
            {
                double g = 0;
                const double xd[3] = {w->X[3 * (k + 1)] - w->xf[0], w->X[3 * (k + 1) + 1] - w->xf[1], wrap(w->X[3 * (k + 1) + 2] - w->xf[2])};
                const double ws = state_weight(c, n, k + 1);
                double qx[3];
                if (ws != 0.0) { sym3_mul(c->Q, c->Qo, xd, qx); g = 2 * ws * qx[a] * (c->integral ? w->D : 1.0); }
                if (k + 1 == n - 1 && c->has_Qf && !c->xf_fixed[a]) { sym3_mul(c->Qf, c->Qfo, xd, qx); g += 2 * qx[a]; }
                hdz += g * dx; dphi += g * dx;
            }
        }
        if (c->via && k + 1 < n - 1) {        /* via-point gradient at grid point k+1 */
            double vv, vg[3];
            via_terms(w, k + 1, w->X[3 * (k + 1)], w->X[3 * (k + 1) + 1], w->X[3 * (k + 1) + 2], &vv, vg);
            for (int a = 0; a < 3; ++a) { hdz += vg[a] * w->dz_x[3 * (k + 1) + a]; dphi += vg[a] * w->dz_x[3 * (k + 1) + a]; }
        }
    }
    for (int r = 0; r < n; ++r) for (int q = 0; q < 4; ++q) if (row_on(w, r, q)) {
        int j = q & 1;
        double sg = sgn(q), L = lim(w, q);
        double dur = r < n - 1 ? w->dz_u[2 * r + j] : 0.0, dum = r > 0 ? w->dz_u[2 * (r - 1) + j] : 0.0;
        double jdz = sg * ((dur - dum) - (r > 0 ? L * ddt : 0.0));
        double s = w->s[4 * r + q], y = w->y[4 * r + q], sig = y / s;
        double res = row_val_at(w, w->U, w->D, r, q) + s, ybar = mu / s + sig * res;
        ds[4 * r + q] = -res - jdz;
        dy[4 * r + q] = ybar + sig * jdz - y;
        hdz += ybar * jdz;
        dphi -= (mu / s) * ds[4 * r + q];
        ftb(s, ds[4 * r + q], tau, &a_p);
        ftb(y, dy[4 * r + q], tau, &a_d);
    }
    for (int k = 1, M = obst_M(w); k < n - 1; ++k) for (int m = 0; m < M; ++m) {
        if (w->oi[k * M + m] < 0) continue;
        const int dyn_ = is_dynamic(w, w->oi[k * M + m]);
        const double jdz = w->oax[k * M + m] * w->dz_x[3 * k] + w->oay[k * M + m] * w->dz_x[3 * k + 1] +
                           ((dyn_ && fp_turns(w)) ? w->oat[k * M + m] * w->dz_x[3 * k + 2] + w->oad[k * M + m] * ddt
                                                  : w->oat[k * M + m] * (dyn_ ? ddt : w->dz_x[3 * k + 2]));
        const double sl = w->os[k * M + m], y = w->oy[k * M + m];
        if (w->erho > 0) {
            const double ee = w->oe[k * M + m], wv = w->erho - y, res = w->og[k * M + m] + sl - ee;
            const double sig = 1.0 / (sl / y + ee / wv), ybar = y + sig * (res + mu / y - sl - mu / wv + ee);
            const double dyv = ybar + sig * jdz - y;
            w->ody[k * M + m] = dyv;
            w->ods[k * M + m] = mu / y - sl - (sl / y) * dyv;
            w->ode[k * M + m] = mu / wv - ee + (ee / wv) * dyv;
            hdz += ybar * jdz;
            dphi += -(mu / sl) * w->ods[k * M + m] - (mu / ee) * w->ode[k * M + m] + w->erho * w->ode[k * M + m];
            ftb(sl, w->ods[k * M + m], tau, &a_p); ftb(ee, w->ode[k * M + m], tau, &a_p);
            ftb(y, dyv, tau, &a_d); ftb(wv, -dyv, tau, &a_d);
            continue;
        }
        const double res = w->og[k * M + m] + sl;
        const double sig = y / sl, ybar = mu / sl + sig * res;
        w->ods[k * M + m] = -res - jdz;
        w->ody[k * M + m] = ybar + sig * jdz - y;
        hdz += ybar * jdz;
        dphi -= (mu / sl) * w->ods[k * M + m];
        ftb(sl, w->ods[k * M + m], tau, &a_p);
        ftb(y, w->ody[k * M + m], tau, &a_d);
    }
    if (ball_on(w)) {
        double jdz = 0;
        for (int i = 0; i < 3; ++i) if (!c->xf_fixed[i]) jdz += w->ta[i] * w->dz_x[3 * (n - 1) + i];
        const double res = w->tg + w->ts, sig = w->ty / w->ts, ybar = mu / w->ts + sig * res;
        w->tds = -res - jdz; w->tdy = ybar + sig * jdz - w->ty;
        hdz += ybar * jdz;
        dphi -= (mu / w->ts) * w->tds;
        ftb(w->ts, w->tds, tau, &a_p);
        ftb(w->ty, w->tdy, tau, &a_d);
    }
    o->hdz = hdz; o->dz2 = dz2; o->dphi = dphi; o->a_p = a_p; o->a_d = a_d; o->dzmax = dzmax; o->clam = clam; o->nunu = nunu;
}


/* EXPERIMENT switches (oracle_set_algo; tests/tools/algo_stats.py, DESIGN.md section 3 holds the measurements).  The defaults are THE algorithm -- the one
 * oracle/ipm_dense.py and the kernel run too; everything else is Ipopt machinery that was measured on the BASELINE workloads and not adopted:
 *   mu_oracle      0 step-length rule (the product), 1 Mehrotra's probing oracle (Ipopt mu_oracle=probing: affine-scaling solve with the same factorisation),
 *                  2 LOQO rule (Ipopt mu_oracle=loqo)
 *   globalization  -1 what oracle_config.line_search says (the default; the product: the filter without second-order corrections); forced: 0 l1 merit, 1 Ipopt's filter
 *                  (Waechter & Biegler 2006, Algorithm A) with max_soc second-order corrections
 *   rho_mode       (keys 14 / 15, l1 merit only) 1 the smallest admissible penalty every iteration, 2 the penalty may fall by rho_decay per iteration: measured in r06, not adopted
 *   safeguard      adaptive mu: 1 = Ipopt's adaptive_mu_globalization=kkt-error (fixed-mu mode at fix_fact x the average complementarity when the error stalls)
 *   convex_fallback  1: a factorisation that fails the curvature test at delta = 0 is repeated with the stage blocks of lam' D replaced by their positive
 *                  semidefinite parts before any multiple of the identity is added
 *   qn_sr1         oracle_config.hessian_mode = 2 (stage-structured quasi-Newton Hessian, an experiment of this file only): 1 = symmetric rank-one element updates
 *                  (the default of that mode), 0 = damped BFGS */
typedef struct {
    int mu_oracle, globalization, max_soc, safeguard;
    double sigma_max, fix_fact;
    int convex_fallback;
    int qn_sr1;
    double elastic_rho;      /* > 0: clearance rows elastic from the start of a solve (g + s - e = 0, e >= 0, + rho e in the objective): experiment for VERDICT r04 item 3 */
    double elastic_ap;       /* the step length below which an iteration counts as jammed */
    double elastic_prog;     /* ... and the streak only triggers when the infeasibility is still above this share of its value at the streak's start */
    int elastic_trigger;     /* with elastic_rho > 0: 0 = from the start, k > 0 = entered after k iterations in a row whose fraction-to-boundary step is below 1e-2 while a row is violated */
} algo_t;
static algo_t g_algo = {0, -1, 0, 0, 100.0, 0.8, 0, 1, 1000.0, 5e-2, 0.8, 5};      /* globalization -1: what oracle_config.line_search says; 0 / 1 force the merit / the filter (experiments) */
static int g_inertia = 1;      /* 1: a factorisation is accepted when the KKT matrix has Ipopt's inertia (the algorithm); 0: the inertia-free curvature test of r01-r03 (kept for the measurements of DESIGN 3.1) */
static int g_rho_mode = 0;          /* EXPERIMENT (r06): 0 = the algorithm (the l1 penalty only grows between barrier updates), 1 = recomputed every iteration, 2 = may decay by g_rho_decay per iteration */
static double g_rho_decay = 0.5;
void oracle_set_algo(int key, double v) {
    if (key == 9) { g_inertia = (int)v; return; }
    if (key == 14) { g_rho_mode = (int)v; return; }
    if (key == 15) { g_rho_decay = v; return; }
    switch (key) {
        case 0: g_algo.mu_oracle = (int)v; break;
        case 1: g_algo.globalization = (int)v; break;
        case 2: g_algo.max_soc = (int)v; break;
        case 3: g_algo.safeguard = (int)v; break;
        case 4: g_algo.sigma_max = v; break;
        case 5: g_algo.fix_fact = v; break;
        case 6: g_algo.convex_fallback = (int)v; break;
        case 7: g_algo.qn_sr1 = (int)v; break;
        case 10: g_algo.elastic_rho = v; break;
        case 11: g_algo.elastic_trigger = (int)v; break;
        case 12: g_algo.elastic_ap = v; break;
        case 13: g_algo.elastic_prog = v; break;
    }
}

/* sum, count, smallest of the complementarity products after a step (a_p on the slacks / box distances, a_d on the multipliers; a_p = a_d = 0: the current ones).
 * mu enters the steps of the box multipliers only (d pl = mu/dl - pl - pl/dl du). */
static void compl_stats(const work_t* w, const double* ds, const double* dy, double a_p, double a_d, double mu, double* sum, int* cnt, double* cmin) {
    const oracle_config* c = w->c;
    const int n = w->n;
    double sm = 0, mn = 1e300; int k0 = 0;
#define CPAIR(sv, dsv, yv, dyv) do { double p_ = ((sv) + a_p * (dsv)) * ((yv) + a_d * (dyv)); sm += p_; if (p_ < mn) mn = p_; ++k0; } while (0)
    for (int r = 0; r < n; ++r) for (int q = 0; q < 4; ++q) if (row_on(w, r, q)) CPAIR(w->s[4 * r + q], ds[4 * r + q], w->y[4 * r + q], dy[4 * r + q]);
    for (int k = 1, M = obst_M(w); k < n - 1; ++k) for (int m = 0; m < M; ++m) if (w->oi[k * M + m] >= 0) CPAIR(w->os[k * M + m], w->ods[k * M + m], w->oy[k * M + m], w->ody[k * M + m]);
    if (ball_on(w)) CPAIR(w->ts, w->tds, w->ty, w->tdy);
    for (int k = 0; k < n - 1; ++k) for (int j = 0; j < 2; ++j) {
        const double u = w->U[2 * k + j], du_ = w->dz_u[2 * k + j], dl = u - c->u_lb[j], du = c->u_ub[j] - u, pl = w->pl[2 * k + j], pu = w->pu[2 * k + j];
        CPAIR(dl, du_, pl, mu / dl - pl - (pl / dl) * du_);
        CPAIR(du, -du_, pu, mu / du - pu + (pu / du) * du_);
    }
    if (c->dt_free) {
        const double dl = w->D - c->dt_lb, du = c->dt_ub - w->D;
        CPAIR(dl, w->ddt, w->pdl, mu / dl - w->pdl - (w->pdl / dl) * w->ddt);
        CPAIR(du, -w->ddt, w->pdu, mu / du - w->pdu + (w->pdu / du) * w->ddt);
    }
#undef CPAIR
    *sum = sm; *cnt = k0; *cmin = mn;
}


/* ---------------------------------------------------------------- inertia of the KKT matrix (Ipopt's test of a factorisation: n primal positive, m dual negative eigenvalues)
 * The banded LU above pivots by rows and carries no inertia.  This pass counts the negative eigenvalues of the SAME assembled matrix [band b; b' Hdd] by a symmetric block
 * elimination from the last stage backward -- blocks (lam_k, x_{k+1}) then u_k, k = n-2 .. 0, then dt -- with Sylvester's law: the inertia is the sum of the inertias
 * of the pivot blocks (Haynsworth), whatever the order, as long as no pivot block is singular.  Small pivot blocks: Bunch-Parlett diagonal pivoting (1x1 / 2x2, complete pivoting). */
static int small_inertia(double* A, int m, int* nneg) {            /* A: m x m symmetric, leading dimension 6, destroyed; returns -1 when singular */
    int act[6], r = m, neg = 0;
    double scale = 0;
    for (int i = 0; i < m; ++i) { act[i] = i; for (int j = 0; j < m; ++j) if (fabs(A[6 * i + j]) > scale) scale = fabs(A[6 * i + j]); }
    if (!(scale > 0) || !isfinite(scale)) return -1;
    while (r > 0) {
        double mu1 = -1, mu0 = -1; int p = 0, q0 = 0, q1 = 0;
        for (int a = 0; a < r; ++a) {
            const int i = act[a];
            if (fabs(A[6 * i + i]) > mu1) { mu1 = fabs(A[6 * i + i]); p = a; }
            for (int b = a + 1; b < r; ++b) { const int j = act[b]; if (fabs(A[6 * i + j]) > mu0) { mu0 = fabs(A[6 * i + j]); q0 = a; q1 = b; } }
        }
        if (!(fmax(mu0, mu1) > 0.0)) return -1;
        if (mu1 >= 0.6404 * mu0) {
            const int i = act[p]; const double d = A[6 * i + i];
            if (d < 0) ++neg;
            for (int a = 0; a < r; ++a) if (a != p) for (int b = 0; b < r; ++b) if (b != p) A[6 * act[a] + act[b]] -= A[6 * act[a] + i] * A[6 * i + act[b]] / d;
            act[p] = act[--r];
        } else {
            const int i = act[q0], j = act[q1];
            const double a11 = A[6 * i + i], a12 = A[6 * i + j], a22 = A[6 * j + j], det = a11 * a22 - a12 * a12;     /* < 0 by the choice of the pivot */
            if (!(det < 0)) return -1;
            ++neg;
            for (int a = 0; a < r; ++a) if (a != q0 && a != q1) for (int b = 0; b < r; ++b) if (b != q0 && b != q1) {
                const double ci = A[6 * act[a] + i], cj = A[6 * act[a] + j], ri = A[6 * i + act[b]], rj = A[6 * j + act[b]];
                A[6 * act[a] + act[b]] -= (ci * (a22 * ri - a12 * rj) + cj * (-a12 * ri + a11 * rj)) / det;
            }
            act[q1] = act[--r];            /* q1 > q0: remove the later one first */
            act[q0] = act[--r];
        }
    }
    *nneg = neg;
    return 0;
}
#define SW (2 * KL + 1)
static int kkt_negative_eigenvalues(work_t* w, double Hdd, int* nneg_out) {       /* before band_factor (the band is factorised in place) */
    const int N = w->N, n = w->n;
    double* S = w->inS; double* bb = w->inb;
    for (int i = 0; i < N; ++i) for (int d = -KL; d <= KL; ++d) {
        const int j = i + d;
        S[(size_t)SW * i + d + KL] = (j >= 0 && j < N) ? w->AB[(size_t)(KL + KU + i - j) + (size_t)LDAB * j] : 0.0;
    }
    memcpy(bb, w->bcol, sizeof(double) * N);
    double beta = Hdd;
    int neg = 0;
    for (int k = n - 2; k >= 0; --k) for (int pass = 0; pass < 2; ++pass) {
        const int b0 = pass == 0 ? 8 * k + 2 : 8 * k, m = pass == 0 ? 6 : 2, lo = b0 - KL > 0 ? b0 - KL : 0, nb = b0 - lo;
        double M[36], Mi[36], C[6][KL + 1], X[6][KL + 1];
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { M[6 * i + j] = S[(size_t)SW * (b0 + i) + (j - i) + KL]; Mi[6 * i + j] = M[6 * i + j]; }
        for (int i = 0; i < m; ++i) { for (int r = 0; r < nb; ++r) { const int d = lo + r - (b0 + i); C[i][r] = d >= -KL ? S[(size_t)SW * (b0 + i) + d + KL] : 0.0; X[i][r] = C[i][r]; } C[i][nb] = bb[b0 + i]; X[i][nb] = bb[b0 + i]; }
        int ng = 0;
        if (small_inertia(M, m, &ng) != 0) return -1;
        neg += ng;
        /* X = Mi^-1 [C | b]: Gaussian elimination with partial pivoting */
        for (int c2 = 0; c2 < m; ++c2) {
            int pv = c2; for (int r = c2 + 1; r < m; ++r) if (fabs(Mi[6 * r + c2]) > fabs(Mi[6 * pv + c2])) pv = r;
            if (Mi[6 * pv + c2] == 0.0) return -1;
            if (pv != c2) { for (int j = 0; j < m; ++j) { double t = Mi[6 * pv + j]; Mi[6 * pv + j] = Mi[6 * c2 + j]; Mi[6 * c2 + j] = t; } for (int j = 0; j <= nb; ++j) { double t = X[pv][j]; X[pv][j] = X[c2][j]; X[c2][j] = t; } }
            const double ip = 1.0 / Mi[6 * c2 + c2];
            for (int r = 0; r < m; ++r) if (r != c2) {
                const double f = Mi[6 * r + c2] * ip;
                if (f == 0.0) continue;
                for (int j = c2; j < m; ++j) Mi[6 * r + j] -= f * Mi[6 * c2 + j];
                for (int j = 0; j <= nb; ++j) X[r][j] -= f * X[c2][j];
            }
        }
        for (int i = 0; i < m; ++i) { const double ip = 1.0 / Mi[6 * i + i]; for (int j = 0; j <= nb; ++j) X[i][j] *= ip; }
        for (int r1 = 0; r1 < nb; ++r1) {
            for (int r2 = 0; r2 < nb; ++r2) { double a = 0; for (int i = 0; i < m; ++i) a += C[i][r1] * X[i][r2]; S[(size_t)SW * (lo + r1) + (r2 - r1) + KL] -= a; }
            double a = 0; for (int i = 0; i < m; ++i) a += C[i][r1] * X[i][nb];
            bb[lo + r1] -= a;
        }
        { double a = 0; for (int i = 0; i < m; ++i) a += C[i][nb] * X[i][nb]; beta -= a; }
    }
    if (w->c->dt_free) { if (!isfinite(beta) || beta == 0.0) return -1; if (beta < 0) ++neg; }
    *nneg_out = neg;
    return 0;
}

/* bordered solve with the factorised band: [K b; b' Hdd] [z; ddt] = [rhs; -hd]; y2 = K^-1 b is computed when *have_y2 == 0 */
static int border_solve(work_t* w, double* y2, int* have_y2, double Hdd, double hd) {
    const oracle_config* c = w->c;
    const int N = w->N;
    int good = 1;
    band_solve(w, w->rhs);
    double ddt = 0.0;
    if (c->dt_free) {
        if (!*have_y2) { memcpy(y2, w->bcol, sizeof(double) * N); band_solve(w, y2); *have_y2 = 1; }
        double num = -hd, den = Hdd;
        for (int i = 0; i < N; ++i) { num -= w->bcol[i] * w->rhs[i]; den -= w->bcol[i] * y2[i]; }
        ddt = num / den;
        if (!isfinite(ddt) || den == 0.0) good = 0;
        else for (int i = 0; i < N; ++i) w->rhs[i] -= y2[i] * ddt;
    }
    w->ddt = ddt;
    for (int i = 0; i < N && good; ++i) if (!isfinite(w->rhs[i])) good = 0;
    return good;
}


/* trial point X + alpha dX (heading wrapped), U + alpha dU, D + alpha dD, slacks + alpha ds: collocation residuals cct, objective, theta and the sum of the barrier logs there */
static void trial_point(work_t* w, double alpha, const double* ds, double* st, double* cct, double* ft, double* tht_out, double* logs) {
    const oracle_config* c = w->c;
    const int n = w->n;
    for (int k = 0; k < n; ++k) for (int i = 0; i < 3; ++i) {
        double x = w->X[3 * k + i];
        if (k > 0 && (k < n - 1 || !c->xf_fixed[i])) { x += alpha * w->dz_x[3 * k + i]; if (i == 2) x = wrap(x); }
        w->Xt[3 * k + i] = x;
    }
    for (int i = 0; i < 2 * (n - 1); ++i) w->Ut[i] = w->U[i] + alpha * w->dz_u[i];
    w->Dt = w->D + (c->dt_free ? alpha * w->ddt : 0.0);
    for (int i = 0; i < 4 * n; ++i) st[i] = w->s[i] + alpha * ds[i];
    for (int i = 0, nm = n * obst_M(w); i < nm; ++i) w->ost[i] = w->oi[i] >= 0 ? w->os[i] + alpha * w->ods[i] : 1.0;
    eval_point(w, w->Xt, w->Ut, w->Dt, cct, ft);
    if (w->erho > 0) for (int i = 0, nm = n * obst_M(w); i < nm; ++i) { w->oet[i] = w->oi[i] >= 0 ? w->oe[i] + alpha * w->ode[i] : 0.0; if (w->oi[i] >= 0 && i >= obst_M(w) && i < (n - 1) * obst_M(w)) *ft += w->erho * w->oet[i]; }
    double tht = 0;
    for (int i = 0; i < 3 * (n - 1); ++i) tht += fabs(cct[i]);
    for (int r = 0; r < n; ++r) for (int q = 0; q < 4; ++q) if (row_on(w, r, q)) tht += fabs(row_val_at(w, w->Ut, w->Dt, r, q) + st[4 * r + q]);
    if (obst_M(w) > 0) tht += obst_theta(w, w->Xt, w->Dt, w->ost);
    double tlog = 0;
    if (ball_on(w)) { double ta[3]; w->tst = w->ts + alpha * w->tds; tht += fabs(ball_eval(w, w->Xt, ta) + w->tst); tlog = log(w->tst); }
    *tht_out = tht;
    *logs = barrier_logs(w, w->Ut, w->Dt, st, w->ost) + tlog;
}

#define FCAP 16
static int g_trace = 0;      /* dev: print one line per iteration (single-threaded use) */
void oracle_set_trace(int on) { g_trace = on; }
/* developer counters over all solves since the last reset (tests/tools/dev): [0] iterations, [1] trial points, [2] line searches that refused every trial step, [3] trial points of those */
static long long g_cnt[4];
void oracle_counters(long long* out, int reset) { for (int i = 0; i < 4; ++i) { out[i] = g_cnt[i]; if (reset) g_cnt[i] = 0; } }
#define CNT(i, v) do { _Pragma("omp atomic") g_cnt[i] += (v); } while (0)
static long g_nfac_total = 0;
static int g_nfac_max = 0;
void oracle_set_variant(int v) { g_variant = v; g_nfac_total = 0; g_nfac_max = 0; }
long oracle_nfac_total(void) { return g_nfac_total; }
int oracle_nfac_max(void) { return g_nfac_max; }


/* hessian_mode 2 (EXPERIMENT, this file only -- the product maps `hessian_approximation: limited-memory` to MPC_HESSIAN_CONVEXIFIED, which beats both variants below at
 * the shipped tol 1e-4: DESIGN.md section 3.1): stage-structured (partitioned) quasi-Newton Hessian.  The Lagrangian is a sum of element functions, and the only non-linear ones are the collocation increments:
 * L = f + sum_k lam_k' D_k(e_k) + (linear rows),  e_k = (theta_k, v_k, w_k, dt).  Each 4 x 4 element Hessian is approximated by its own damped BFGS matrix B_k
 * (Griewank & Toint's partitioned updating; Powell's damping keeps every block positive semidefinite), which keeps the stage structure the sweeps need -- Ipopt's limited-memory
 * matrix is sigma I + a rank-2m term that couples all stages.  Secant pair of element k after an accepted step: s = e_k+ - e_k,  y = grad_e (lam+' D_k)(e_k+) - grad_e (lam+' D_k)(e_k)
 * (both gradients with the NEW multipliers).  B_k starts at 0: the first factorisations are regularised by delta_w like any singular Hessian.  The true element Hessian
 * [H_qq H_qd; H_qd' 0] is indefinite whenever H_qd != 0, so a positive semidefinite BFGS block cannot approach it (measured: 32 % of config 2 converge); symmetric rank-one
 * updates can (89 % under the curvature test of r01-r03; under the inertia test their factorisations drown in delta_w: 15 %), and delta_w treats the indefinite blocks as it treats the exact Hessian. */
static void bfgs_update(work_t* w) {
    const oracle_config* c = w->c;
    const int n = w->n;
    for (int k = 0; k < n - 1; ++k) {
        const double* lam = &w->lam[3 * k];
        stage_map_t sm;
        stage_map(c, w->X[3 * k + 2], w->U[2 * k], w->U[2 * k + 1], w->D, NULL, &sm);
        double* J = &w->bf_g[12 * k];           /* Jq (9, row major) and Jdt (3) at the point of the last update */
        double* e = &w->bf_e[4 * k];
        double* b = &w->bf[10 * k];
        if (w->bf_init) {
            double y[4], sv[4], B[4][4], Bs[4];
            for (int j = 0; j < 3; ++j) y[j] = lam[0] * (sm.Jq[0][j] - J[j]) + lam[1] * (sm.Jq[1][j] - J[3 + j]) + lam[2] * (sm.Jq[2][j] - J[6 + j]);
            y[3] = lam[0] * (sm.Jdt[0] - J[9]) + lam[1] * (sm.Jdt[1] - J[10]) + lam[2] * (sm.Jdt[2] - J[11]);
            sv[0] = k > 0 ? wrap(w->X[3 * k + 2] - e[0]) : 0.0; sv[1] = w->U[2 * k] - e[1]; sv[2] = w->U[2 * k + 1] - e[2]; sv[3] = c->dt_free ? w->D - e[3] : 0.0;
            if (k == 0) y[0] = 0.0;
            if (!c->dt_free) y[3] = 0.0;
            B[0][0] = b[0]; B[0][1] = B[1][0] = b[1]; B[0][2] = B[2][0] = b[2]; B[1][1] = b[3]; B[1][2] = B[2][1] = b[4]; B[2][2] = b[5];
            B[0][3] = B[3][0] = b[6]; B[1][3] = B[3][1] = b[7]; B[2][3] = B[3][2] = b[8]; B[3][3] = b[9];
            double sBs = 0, sy = 0, ss = 0, yy = 0;
            for (int i = 0; i < 4; ++i) { Bs[i] = 0; for (int j = 0; j < 4; ++j) Bs[i] += B[i][j] * sv[j]; }
            for (int i = 0; i < 4; ++i) { sBs += sv[i] * Bs[i]; sy += sv[i] * y[i]; ss += sv[i] * sv[i]; yy += y[i] * y[i]; }
            if (g_algo.qn_sr1) {
                double v[4], vs = 0, vv = 0;
                for (int i = 0; i < 4; ++i) { v[i] = y[i] - Bs[i]; vs += v[i] * sv[i]; vv += v[i] * v[i]; }
                if (fabs(vs) > 1e-8 * sqrt(ss * vv) && ss > 1e-24) {
                    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) B[i][j] += v[i] * v[j] / vs;
                    b[0] = B[0][0]; b[1] = B[0][1]; b[2] = B[0][2]; b[3] = B[1][1]; b[4] = B[1][2]; b[5] = B[2][2]; b[6] = B[0][3]; b[7] = B[1][3]; b[8] = B[2][3]; b[9] = B[3][3];
                }
            } else
            if (ss > 1e-24) {
                /* Powell's damping: r = th y + (1 - th) B s with s'r >= 0.2 s'B s */
                double th = 1.0;
                if (sy < 0.2 * sBs) th = 0.8 * sBs / (sBs - sy);
                double r[4], sr = 0, rr = 0;
                for (int i = 0; i < 4; ++i) { r[i] = th * y[i] + (1.0 - th) * Bs[i]; sr += sv[i] * r[i]; rr += r[i] * r[i]; }
                if (sr > 1e-12 * sqrt(ss * rr) && sr > 0) {
                    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) B[i][j] += r[i] * r[j] / sr - (sBs > 0 ? Bs[i] * Bs[j] / sBs : 0.0);
                    b[0] = B[0][0]; b[1] = B[0][1]; b[2] = B[0][2]; b[3] = B[1][1]; b[4] = B[1][2]; b[5] = B[2][2]; b[6] = B[0][3]; b[7] = B[1][3]; b[8] = B[2][3]; b[9] = B[3][3];
                }
            }
        } else for (int i = 0; i < 10; ++i) b[i] = 0.0;
        for (int a = 0; a < 3; ++a) { for (int j = 0; j < 3; ++j) J[3 * a + j] = sm.Jq[a][j]; J[9 + a] = sm.Jdt[a]; }
        e[0] = w->X[3 * k + 2]; e[1] = w->U[2 * k]; e[2] = w->U[2 * k + 1]; e[3] = w->D;
    }
    w->bf_init = 1;
}

static int solve_one(work_t* w, int warm) {
    const oracle_config* c = w->c;
    const int n = w->n, N = w->N;
    const double tol = c->tol > 0 ? c->tol : 1e-8;
    const int max_iter = c->max_iter > 0 ? c->max_iter : 100;
    const double kappa_eps = 10, kappa_mu = 0.2, theta_mu = 1.5, tau_min = 0.99, bound_push = 1e-2, slack_push = 1e-2;
    const double eta = 1e-4, rho_frac = 0.1, delta_first = 1e-4, delta_min = 1e-20, delta_max = 1e20;
    const double kplus = 8, kplus1 = 100, kminus = 1.0 / 3.0, curv_kappa = 1e-10, delta_c = 1e-8, kappa_c = 0.25;
    const int max_ls = 30;
    const double clearance_slack_push = 0.5;      /* initial slack of a clearance row, max(-g, 0.5): see Algo<T>::clearance_slack_push in csrc/mpc_core.hpp */
    int nfix = c->xf_fixed[0] + c->xf_fixed[1] + c->xf_fixed[2];
    double* cc = (double*)malloc(sizeof(double) * 3 * (n - 1));
    double* cct = (double*)malloc(sizeof(double) * 3 * (n - 1));
    double* st = (double*)malloc(sizeof(double) * 4 * n);
    double* ds = (double*)malloc(sizeof(double) * 4 * n);
    double* dy = (double*)malloc(sizeof(double) * 4 * n);
    double* y2 = (double*)malloc(sizeof(double) * N);
    double* rhs_keep = (double*)malloc(sizeof(double) * N);
    double* csoc = (double*)malloc(sizeof(double) * 3 * (n - 1));
    double* ds_keep = (double*)malloc(sizeof(double) * 4 * n);
    double* dy2 = (double*)malloc(sizeof(double) * 4 * n);
    double fth[FCAP], fph[FCAP]; int nfilt = 0, nresto = 0; double theta0 = -1, mu_filter = -1;
    int free_mode = 1, nrefs = 0; double refs[4];
    const int adaptive = c->mu_strategy != 1;
    int endgame = 0;
    const double sigma_min = 0.05, mu_err_floor = 3e-2, mu_max_fact = 1e3;
    double last_alpha = 0, last_ad = 0;
    double mu_min = 0, mu_max = 1e300;
    int status = 1, it = 0;
    int nfac = 0, fail0_streak = 0, seeded = 0;
    /* initial vertex values */
    if (!warm) {
        double dth = wrap(w->xf[2] - w->x0[2]);
        for (int k = 0; k < n; ++k) {
            double fr = (double)k / (double)(n - 1);
            w->X[3 * k] = w->x0[0] + fr * (w->xf[0] - w->x0[0]);
            w->X[3 * k + 1] = w->x0[1] + fr * (w->xf[1] - w->x0[1]);
            w->X[3 * k + 2] = wrap(w->x0[2] + fr * dth);
        }
        for (int i = 0; i < 3; ++i) w->X[3 * (n - 1) + i] = w->xf[i];
        memset(w->U, 0, sizeof(double) * 2 * (n - 1));
        w->D = c->dt_ref;
    }
    for (int i = 0; i < 3; ++i) { w->X[i] = w->x0[i]; if (c->xf_fixed[i]) w->X[3 * (n - 1) + i] = w->xf[i]; }
    if (!c->dt_free) w->D = c->dt_ref;
    /* seed controls from the state guess when all are zero (see oracle/ipm_dense.py controls_from_states) */
    {
        int any = 0;
        for (int i = 0; i < 2 * (n - 1); ++i) if (w->U[i] != 0.0) any = 1;
        seeded = !any;
        if (!any) for (int k = 0; k < n - 1; ++k) {
            double dx = w->X[3 * (k + 1)] - w->X[3 * k], dyy = w->X[3 * (k + 1) + 1] - w->X[3 * k + 1];
            double dth = wrap(w->X[3 * (k + 1) + 2] - w->X[3 * k + 2]), th = w->X[3 * k + 2];
            double v = (dx * cos(th) + dyy * sin(th)) / w->D;
            v = fmin(fmax(v, c->u_lb[0]), c->u_ub[0]);
            double rate = dth / w->D, om;
            if (c->model == 0) om = rate;
            else {
                double vv = fabs(v) > 1e-3 ? v : (v >= 0 ? 1e-3 : -1e-3);
                if (c->model == 1) om = atan(c->model_params[0] * rate / vv);
                else if (c->model == 2) om = asin(fmin(1.0, fmax(-1.0, c->model_params[0] * rate / vv)));
                else { double sb = fmin(1.0, fmax(-1.0, c->model_params[0] * rate / vv)); om = atan(tan(asin(sb)) * (c->model_params[1] + c->model_params[0]) / c->model_params[0]); }
            }
            om = fmin(fmax(om, c->u_lb[1]), c->u_ub[1]);
            w->U[2 * k] = v; w->U[2 * k + 1] = om;
        }
    }
    if (seeded) {
        /* ... and keeps the seeded controls inside the control-rate rows, as the reference's u = 0 start is (every row but the first): increments clamped
         * to rate_seed_frac x the rate limits forward from u_prev, then backward from the final row (against u_ref = 0).  A seed that jumps violates the rows
         * it crosses: their slacks start at the 1e-2 floor with a residual, and the fraction-to-boundary rule pins the first iterations. */
        const double fr = 0.9;
        for (int j = 0; j < 2; ++j) {
            if (!(c->du_lb[j] > -1e29) || !(c->du_ub[j] < 1e29)) continue;
            const double lo = c->du_lb[j] * w->D * fr, hi = c->du_ub[j] * w->D * fr;
            if (w->dtprev != 0.0) w->U[j] = fmin(fmax(w->U[j], w->uprev[j] + c->du_lb[j] * w->dtprev * fr), w->uprev[j] + c->du_ub[j] * w->dtprev * fr);
            for (int k = 1; k < n - 1; ++k) w->U[2 * k + j] = fmin(fmax(w->U[2 * k + j], w->U[2 * (k - 1) + j] + lo), w->U[2 * (k - 1) + j] + hi);
            double nxt = 0.0;
            for (int k = n - 2; k >= 0; --k) { w->U[2 * k + j] = fmin(fmax(w->U[2 * k + j], nxt - hi), nxt - lo); nxt = w->U[2 * k + j]; }
        }
    }
    for (int k = 0; k < n - 1; ++k) for (int j = 0; j < 2; ++j) {
        double lb = c->u_lb[j], ub = c->u_ub[j];
        double pl = fmin(bound_push * fmax(1.0, fabs(lb)), bound_push * (ub - lb)), pu = fmin(bound_push * fmax(1.0, fabs(ub)), bound_push * (ub - lb));
        w->U[2 * k + j] = fmin(fmax(w->U[2 * k + j], lb + pl), ub - pu);
    }
    if (c->dt_free) {
        double lb = c->dt_lb, ub = c->dt_ub;
        double pl = fmin(bound_push * fmax(1.0, fabs(lb)), bound_push * (ub - lb)), pu = fmin(bound_push * fmax(1.0, fabs(ub)), bound_push * (ub - lb));
        w->D = fmin(fmax(w->D, lb + pl), ub - pu);
    }
    const int dual_ok = warm && w->dual && (int)w->dual[0] == n;
    w->mu = dual_ok ? g_dual_mu0 : (c->mu_init > 0 ? c->mu_init : 0.1);
    w->rho = 0; w->delta_last = 0;
    for (int r = 0; r < n; ++r) for (int q = 0; q < 4; ++q) {
        int finite = q < 2 ? (c->du_lb[q] > -1e29) : (c->du_ub[q - 2] < 1e29);
        w->ron[4 * r + q] = finite && (r > 0 || w->dtprev != 0.0);
        w->s[4 * r + q] = 1.0; w->y[4 * r + q] = 0.0;
        if (w->ron[4 * r + q]) { w->s[4 * r + q] = fmax(-row_val_at(w, w->U, w->D, r, q), slack_push); w->y[4 * r + q] = w->mu / w->s[4 * r + q]; }
    }
    for (int k = 0; k < n - 1; ++k) {
        for (int j = 0; j < 2; ++j) { w->pl[2 * k + j] = w->mu / (w->U[2 * k + j] - c->u_lb[j]); w->pu[2 * k + j] = w->mu / (c->u_ub[j] - w->U[2 * k + j]); }
        for (int i = 0; i < 3; ++i) w->lam[3 * k + i] = 0.0;
    }
    if (c->via) via_associate(w);
    if (obst_M(w) > 0) {
        const int M = obst_M(w);
        obst_centroids(w);
        w->rows_dropped = obst_associate(w);
        for (int k = 0; k < n; ++k) for (int m = 0; m < M; ++m) {
            double g, a3[3], hk, h3[3];
            w->os[k * M + m] = 1.0; w->oy[k * M + m] = 0.0; w->ods[k * M + m] = 0.0; w->ody[k * M + m] = 0.0;
            if (k >= 1 && k < n - 1 && obst_row3(w, k, m, w->X[3 * k], w->X[3 * k + 1], w->X[3 * k + 2], w->D, &g, a3, &hk, h3)) {
                w->os[k * M + m] = fmax(-g, clearance_slack_push); w->oy[k * M + m] = w->mu / w->os[k * M + m];
                w->oe[k * M + m] = 0.0; w->ode[k * M + m] = 0.0; w->oet[k * M + m] = 0.0;
                if (g_algo.elastic_rho > 0 && g_algo.elastic_trigger == 0) w->oe[k * M + m] = fmax(g + w->os[k * M + m], w->mu / g_algo.elastic_rho);      /* the row starts satisfied: g + s - e = 0 */
            } else w->oi[k * M + m] = -1;
        }
        w->erho = (g_algo.elastic_rho > 0 && g_algo.elastic_trigger == 0) ? g_algo.elastic_rho : 0.0;
    }
    if (ball_on(w)) { double ta[3]; w->ts = fmax(-ball_eval(w, w->X, ta), slack_push); w->ty = w->mu / w->ts; w->tds = w->tdy = 0; }
    w->pdl = c->dt_free ? w->mu / (w->D - c->dt_lb) : 0.0;
    w->pdu = c->dt_free ? w->mu / (c->dt_ub - w->D) : 0.0;
    if (dual_ok) {
        const double* b = w->dual; const double *bl = b + 4, *by = bl + 3 * (n - 1), *bpl = by + 4 * n, *bpu = bpl + 2 * (n - 1);
        for (int i = 0; i < 3 * (n - 1); ++i) w->lam[i] = bl[i];
        for (int i = 0; i < 4 * n; ++i) if (w->ron[i]) w->y[i] = fmax(w->y[i], by[i]);
        for (int i = 0; i < 2 * (n - 1); ++i) { w->pl[i] = fmax(w->pl[i], bpl[i]); w->pu[i] = fmax(w->pu[i], bpu[i]); }
        if (c->dt_free) { w->pdl = fmax(w->pdl, b[1]); w->pdu = fmax(w->pdu, b[2]); }
        if (ball_on(w)) w->ty = fmax(w->ty, b[3]);
    }
    double fobj;
    int n_acceptable = 0;
    const double acc_tol = acc_tol_of(c);
    const int acc_it = acc_iter_of(c);
    w->bf_init = 0;
    eval_point(w, w->X, w->U, w->D, cc, &fobj);
    if (w->erho > 0) for (int k = 1, M = obst_M(w); k < n - 1; ++k) for (int m = 0; m < M; ++m) if (w->oi[k * M + m] >= 0) fobj += w->erho * w->oe[k * M + m];
    mu_min = tol / 10; mu_max = mu_max_fact * w->mu;
    int jam_streak = 0; double jam_theta0 = 0;
    while (1) {
        err_t e;
        kkt_terms(w, cc, &e);
        /* RESTORATION for clearance rows that jam (r05; Ipopt leaves such points to its restoration phase, src/controller.cpp:388-421 hands the NLP to Ipopt).  A row that
         * starts violated pulls its slack to the boundary within a few iterations; from then on the fraction-to-boundary rule admits steps of 1e-3 and the infeasibility
         * stays where it is.  Detected as elastic_trigger iterations in a row with a primal step limit below elastic_ap while the infeasibility has not fallen below
         * elastic_prog x its value at the start of the streak.  From there on the clearance rows are ELASTIC -- g + s - e = 0, e >= 0, + rho e in the objective, the exact
         * l1 penalty of the row's violation --: every row is satisfied again at once (e takes up the violation, the slack goes back to its start rule), nothing has to
         * cross a bound, and rho pushes e to zero as the trajectory moves out of the band.  (s, e) are condensed together: sigma = 1 / (s / y + e / (rho - y)).  The mode
         * stays on until the solve ends; e counts as primal infeasibility, so a solve can only end with e <= tol: the answer is a KKT point of the reference's NLP. */
        if (w->erho == 0 && obst_M(w) > 0 && g_algo.elastic_rho > 0 && g_algo.elastic_trigger > 0 && jam_streak >= g_algo.elastic_trigger && e.theta >= g_algo.elastic_prog * jam_theta0) {
            w->erho = g_algo.elastic_rho;
            for (int k = 1, M = obst_M(w); k < n - 1; ++k) for (int m = 0; m < M; ++m) if (w->oi[k * M + m] >= 0) {
                const double g = w->og[k * M + m];
                w->os[k * M + m] = fmax(fmax(-g, clearance_slack_push), w->os[k * M + m]);
                w->oe[k * M + m] = fmax(g + w->os[k * M + m], w->mu / w->erho);
                w->oy[k * M + m] = fmax(fmin(w->oy[k * M + m], 0.5 * w->erho), w->mu / w->os[k * M + m]);
                fobj += w->erho * w->oe[k * M + m];
            }
            w->rho = 0;
            nfilt = 0;           /* the objective changed: the filter's pairs are of another barrier function */
            jam_streak = 0;
            kkt_terms(w, cc, &e);
        }
        double e0 = err_value(&e, 0.0);
        if (!isfinite(e0) || !isfinite(e.theta) || !isfinite(e.sm) || !isfinite(e.csum)) { status = 4; break; }
        if (e0 <= tol) { status = 0; break; }
        /* Ipopt's acceptable-level stop, first half: acceptable_iter iterations in a row with an error of at most acceptable_tol */
        if (acc_it > 0 && acc_tol > 0) {
            n_acceptable = e0 <= acc_tol ? n_acceptable + 1 : 0;
            if (n_acceptable >= acc_it) { status = 0; break; }
        }
        if (it >= max_iter) { status = 1; break; }
        if (c->hessian_mode == 2) bfgs_update(w);
        /* barrier parameter.  Monotone (oracle_config.mu_strategy = 1, Ipopt's mu_strategy monotone): Fiacco-McCormick, mu falls when the barrier subproblem
         * is solved to kappa_eps mu.  Adaptive (the default; what corbo's SolverIpopt is believed to set): every iteration
         *     mu = sigma x (average complementarity),   sigma = clamp((1 - min(alpha, alpha_dual))^3, 0.05, 1)
         * with the step lengths the LAST iteration achieved (a full step -> the centring weight collapses, a blocked step -> mu stays: Mehrotra's
         * sigma = (mu_aff / mu)^3 read off the step that was actually taken instead of an extra affine-scaling solve; oracle_set_algo(0, 1) runs the
         * probing oracle itself, same statistics), never below min(mu, mu_err_floor x E_0): a barrier far below the optimality error is what stalls
         * the non-convex instances (measured: floors 1e-2 / 3e-2 / 1e-1 -> 93.8 / 95.3 / 95.7 % converged at 33.8 / 35.0 / 38.6 iterations
         * on configs[1]); kept inside [tol / 10, mu_max_fact x mu_init] as Ipopt's mu_min / mu_max. */
        int mu_chosen = 0, mu_changed = 0;
        if (adaptive && g_algo.safeguard == 1 && free_mode) {
            /* Ipopt adaptive_mu_globalization=kkt-error: the free mode goes on while the error is below 0.9999 x one of the last four reference values */
            int sufficient = nrefs < 4;
            for (int i = 0; i < nrefs && !sufficient; ++i) if (e0 <= 0.9999 * refs[i]) sufficient = 1;
            if (sufficient) { if (nrefs < 4) refs[nrefs++] = e0; else { refs[0] = refs[1]; refs[1] = refs[2]; refs[2] = refs[3]; refs[3] = e0; } }
            else { free_mode = 0; w->mu = fmin(fmax(g_algo.fix_fact * e.csum / e.nb, mu_min), mu_max); w->rho = 0; mu_changed = 1; }
        }
        if (!adaptive || !free_mode || endgame) {
            for (int g = 0; g < 50; ++g) {
                double emu = err_value(&e, w->mu);
                if (emu <= kappa_eps * w->mu && w->mu > tol / 10) {
                    if (adaptive && !endgame) { free_mode = 1; nrefs = 0; break; }          /* the fixed-mu subproblem is solved: back to the free mode */
                    w->mu = fmax(tol / 10, fmin(kappa_mu * w->mu, pow(w->mu, theta_mu))); w->rho = 0; mu_changed = 1;
                }
                else break;
            }
        }
        if (adaptive && free_mode && !endgame && g_algo.mu_oracle != 1 && it > 0) {      /* the first iteration keeps the start value (mu_init, or the warm start's) */
            const double avg = e.csum / e.nb;
            double sig;
            if (g_algo.mu_oracle == 2) { const double xi = e.cmin / avg, t_ = fmin(0.05 * (1 - xi) / xi, 2.0); sig = 0.1 * t_ * t_ * t_; }       /* LOQO rule (experiment) */
            else { const double a_ = 1.0 - fmin(last_alpha, last_ad); sig = fmin(fmax(a_ * a_ * a_, sigma_min), 1.0); }
            double mu_new = fmin(fmax(sig * avg, mu_min), mu_max);
            mu_new = fmax(mu_new, fmin(w->mu, mu_err_floor * e0));
            if (mu_new <= tol) { mu_new = tol; endgame = 1; }      /* end game: from mu = tol on the monotone rule takes over (tol -> tol / 10 once the barrier problem is solved to
                                                                     * kappa_eps mu), so that a solve stops at a point of the central path as the monotone strategy does */
            if (mu_new != w->mu) { w->mu = mu_new; w->rho = 0; mu_changed = 1; }
            mu_chosen = 1;
        }
        double mu = w->mu, tau = fmax(tau_min, 1.0 - mu);
        double dc = nfix > 0 ? delta_c * pow(mu, kappa_c) : 0.0;
        double delta = 0.0, Hdd, hd, curv = 0, dz2 = 0, hdz = 0, dphi = 0, a_p = 1, a_d = 1, dzmax = 0;
        int ok = 0;
        if (g_variant == 1 && fail0_streak >= 1 && w->delta_last > 0) delta = fmax(delta_min, kminus * w->delta_last);
        if (g_variant == 2 && fail0_streak >= 2 && w->delta_last > 0) delta = fmax(delta_min, kminus * w->delta_last);
        int started_zero = delta == 0.0;
        w->convexify = 0;
        for (int ntry = 0; ntry <= 40; ++ntry) {
            ++nfac;
            assemble(w, cc, delta, dc, &Hdd, &hd);
            int inertia_ok = 1;
            if (g_inertia) { int nneg = -1; inertia_ok = kkt_negative_eigenvalues(w, Hdd, &nneg) == 0 && nneg == 3 * (n - 1); }
            int good = inertia_ok && band_factor(w) == 0;
            if (good) {
                int have_y2 = 0;
                good = border_solve(w, y2, &have_y2, Hdd, hd);
                if (good && adaptive && g_algo.mu_oracle == 1 && free_mode && !endgame && !mu_chosen && it > 0) {
                    /* Mehrotra's probing oracle (Ipopt mu_oracle=probing): affine-scaling step (mu = 0) with the same factorisation, step to the boundary (tau = 1),
                     * mu_aff = average complementarity there, sigma = (mu_aff / mu_cur)^3, mu = sigma mu_cur */
                    memcpy(rhs_keep, w->rhs, sizeof(double) * N);
                    const double ddt_keep = w->ddt, mu_keep = w->mu;
                    double hd0, Hdd0;
                    w->mu = 0.0; w->rhs_only = 1; assemble(w, cc, delta, dc, &Hdd0, &hd0); w->rhs_only = 0;
                    int g2 = border_solve(w, y2, &have_y2, Hdd, hd0);
                    if (g2) {
                        step_t sa; double sum0, sum1, mn; int cnt;
                        derive_step(w, cc, 0.0, 1.0, ds, dy, &sa);
                        compl_stats(w, ds, dy, 0.0, 0.0, 0.0, &sum0, &cnt, &mn);
                        compl_stats(w, ds, dy, sa.a_p, sa.a_d, 0.0, &sum1, &cnt, &mn);
                        const double mu_cur = sum0 / cnt, mu_aff = sum1 / cnt;
                        double sigma = pow(mu_aff / mu_cur, 3.0);
                        if (!(sigma < g_algo.sigma_max)) sigma = g_algo.sigma_max;
                        double mu_new = sigma * mu_cur;
                        mu_new = fmin(fmax(mu_new, mu_min), mu_max);
                        mu_new = fmax(mu_new, fmin(mu_keep, mu_err_floor * e0));
                        if (g_trace) printf("    probing: mu_cur %.3e a_aff %.3f %.3f mu_aff %.3e sigma %.3e -> mu %.3e\n", mu_cur, sa.a_p, sa.a_d, mu_aff, sigma, mu_new);
                        w->mu = mu = mu_new; tau = fmax(tau_min, 1.0 - mu); w->rho = 0; mu_changed = 1;
                        w->rhs_only = 1; assemble(w, cc, delta, dc, &Hdd0, &hd); w->rhs_only = 0;
                        good = border_solve(w, y2, &have_y2, Hdd, hd);
                    } else { w->mu = mu_keep; memcpy(w->rhs, rhs_keep, sizeof(double) * N); w->ddt = ddt_keep; }
                    mu_chosen = 1;
                }
            }
            if (good) {
                step_t sp;
                derive_step(w, cc, mu, tau, ds, dy, &sp);
                hdz = sp.hdz; dz2 = sp.dz2; dphi = sp.dphi; a_p = sp.a_p; a_d = sp.a_d; dzmax = sp.dzmax;
                const double clam = sp.clam, nunu = sp.nunu;
                curv = -hdz + clam - dc * nunu;
                if (g_inertia ? isfinite(curv) : (isfinite(curv) && curv >= curv_kappa * dz2)) { ok = 1; break; }
            }
            if (g_algo.convex_fallback && !w->convexify) { w->convexify = g_algo.convex_fallback; continue; }
            if (delta == 0.0) delta = w->delta_last == 0.0 ? delta_first : fmax(delta_min, kminus * w->delta_last);
            else delta *= w->delta_last == 0.0 ? kplus1 : kplus;
            if (delta > delta_max) break;
        }
        if (!ok) { status = 3; break; }
        if (delta > 0) w->delta_last = delta;
        if (started_zero) fail0_streak = delta > 0 ? fail0_streak + 1 : 0;
        else if (g_variant == 2 && (it % 4) == 3) fail0_streak = 0;   /* re-probe delta = 0 now and then */
        double theta = e.theta;
        if (theta0 < 0) { theta0 = theta; }
        const double theta_max = 1e4 * fmax(1.0, theta0), theta_min = 1e-4 * fmax(1.0, theta0);
        if (mu_changed || mu != mu_filter) { nfilt = 0; mu_filter = mu; }
        const double logs0 = barrier_logs(w, w->U, w->D, w->s, w->os) + (ball_on(w) ? log(w->ts) : 0.0);
        double alpha = a_p, ft = 0, tht = 0;
        int accepted = 0, ls_used = 0, soc_used = 0;
        if ((g_algo.globalization < 0 ? (c->line_search == 1 ? 1 : 0) : g_algo.globalization) == 0) {
            if (theta > 0) {
                double sigma = curv > 0 ? 1.0 : 0.0;
                double rt = (dphi + 0.5 * sigma * curv) / ((1.0 - rho_frac) * theta);
                if (g_rho_mode == 1) w->rho = fmax(rt + 1.0, 1.0);                          /* experiment: the smallest penalty that makes this step a descent direction, every iteration */
                else if (g_rho_mode == 2) w->rho = fmax(rt + 1.0, fmax(g_rho_decay * w->rho, 1.0));      /* experiment: the penalty may fall by a factor per iteration */
                else if (w->rho < rt) w->rho = rt + 1.0;
            }
            double phi0 = fobj - mu * logs0 + w->rho * theta;
            double Dm = dphi - w->rho * theta;
            for (int ls = 0; ls < max_ls; ++ls) {
                if (ls > 0) alpha *= 0.5;
                ls_used = ls;
                double lg;
                trial_point(w, alpha, ds, st, cct, &ft, &tht, &lg); CNT(1, 1);
                double phit = ft - mu * lg + w->rho * tht;
                if (isfinite(phit) && phit - phi0 - 10 * 2.220446049250313e-16 * fabs(phi0) <= eta * alpha * Dm) { accepted = 1; break; }
            }
        } else {
            /* Ipopt's filter line search (Waechter & Biegler 2006, Algorithm A, steps A-5.1 .. A-5.10) */
            const double g_th = 1e-5, g_ph = 1e-8, s_ph = 2.3, s_th = 1.1, eta_ph = 1e-8, dlt = 1.0, g_al = 0.05, kappa_soc = 0.99;
            const double phi_cur = fobj - mu * logs0;
            double a_min = g_th;
            if (dphi < 0) {
                a_min = fmin(a_min, g_ph * theta / (-dphi));
                if (theta <= theta_min) a_min = fmin(a_min, dlt * pow(theta, s_th) / pow(-dphi, s_ph));
            }
            a_min *= g_al;
            int sw_arm = 0;
            for (int ls = 0; ls < max_ls; ++ls) {
                if (ls > 0) alpha *= 0.5;
                ls_used = ls;
                if (ls > 0 && alpha < a_min * a_p) break;
                double lg;
                trial_point(w, alpha, ds, st, cct, &ft, &tht, &lg); CNT(1, 1);
                double phit = ft - mu * lg;
                int okf = isfinite(phit) && tht <= theta_max;
                for (int f = 0; f < nfilt && okf; ++f) if (!(tht <= (1 - g_th) * fth[f] || phit <= fph[f] - g_ph * fth[f])) okf = 0;
                const int switching = dphi < 0 && alpha * pow(-dphi, s_ph) > dlt * pow(theta, s_th);
                const int armijo = phit - phi_cur - 10 * 2.220446049250313e-16 * fabs(phi_cur) <= eta_ph * alpha * dphi;
                if (okf) {
                    if (theta <= theta_min && switching) { if (armijo) { accepted = 1; sw_arm = 1; } }
                    else if (tht <= (1 - g_th) * theta || phit <= phi_cur - g_ph * theta) accepted = 1;
                }
                if (accepted) break;
                if (ls == 0 && g_algo.max_soc > 0 && tht >= theta) {
                    /* second-order correction (A-5.5 .. A-5.9): c_soc = alpha c(x) + c(x + alpha d), same matrix */
                    double th_old = theta, a_soc = alpha;
                    memcpy(csoc, cc, sizeof(double) * 3 * (n - 1));
                    memcpy(rhs_keep, w->dz_u, sizeof(double) * 2 * (n - 1)); memcpy(rhs_keep + 2 * n, w->dz_x, sizeof(double) * 3 * n);
                    memcpy(ds_keep, ds, sizeof(double) * 4 * n);
                    const double ddt_keep = w->ddt;
                    for (int p_ = 0; p_ < g_algo.max_soc && !accepted; ++p_) {
                        for (int i = 0; i < 3 * (n - 1); ++i) csoc[i] = a_soc * csoc[i] + cct[i];
                        double Hdd0, hd1; int have = 1;
                        w->rhs_only = 1; assemble(w, csoc, delta, dc, &Hdd0, &hd1); w->rhs_only = 0;
                        if (!border_solve(w, y2, &have, Hdd, hd1)) break;
                        step_t sc;
                        derive_step(w, csoc, mu, tau, ds, dy2, &sc);
                        a_soc = sc.a_p;
                        double lg2, ft2, tht2;
                        trial_point(w, a_soc, ds, st, cct, &ft2, &tht2, &lg2);
                        double phit2 = ft2 - mu * lg2;
                        int ok2 = isfinite(phit2) && tht2 <= theta_max;
                        for (int f = 0; f < nfilt && ok2; ++f) if (!(tht2 <= (1 - g_th) * fth[f] || phit2 <= fph[f] - g_ph * fth[f])) ok2 = 0;
                        const int sw2 = dphi < 0 && alpha * pow(-dphi, s_ph) > dlt * pow(theta, s_th);
                        const int ar2 = phit2 - phi_cur - 10 * 2.220446049250313e-16 * fabs(phi_cur) <= eta_ph * alpha * dphi;
                        if (ok2) {
                            if (theta <= theta_min && sw2) { if (ar2) { accepted = 1; sw_arm = 1; } }
                            else if (tht2 <= (1 - g_th) * theta || phit2 <= phi_cur - g_ph * theta) accepted = 1;
                        }
                        if (accepted) { ft = ft2; tht = tht2; soc_used = p_ + 1; alpha = a_soc; break; }
                        if (tht2 > kappa_soc * th_old) break;
                        th_old = tht2;
                    }
                    if (!accepted) {     /* back to the Newton step */
                        memcpy(w->dz_u, rhs_keep, sizeof(double) * 2 * (n - 1)); memcpy(w->dz_x, rhs_keep + 2 * n, sizeof(double) * 3 * n);
                        memcpy(ds, ds_keep, sizeof(double) * 4 * n); w->ddt = ddt_keep;
                    }
                }
                if (accepted) break;
            }
            if (accepted && !sw_arm) {
                if (nfilt == FCAP) { memmove(fth, fth + 1, sizeof(double) * (FCAP - 1)); memmove(fph, fph + 1, sizeof(double) * (FCAP - 1)); --nfilt; }
                fth[nfilt] = theta; fph[nfilt] = phi_cur; ++nfilt;
            }
            if (!accepted) {
                ++nresto; CNT(2, 1); CNT(3, ls_used + 1);
                /* no restoration phase: empty the filter and take the last trial step */
                nfilt = 0;
                double lg; trial_point(w, alpha, ds, st, cct, &ft, &tht, &lg);
                if (isfinite(ft) && isfinite(lg)) accepted = 2;
            }
        }
        /* second half (tested BEFORE the line-search failure below: Ipopt answers a failed line search at an acceptable point with success): the line search refuses every trial step, or accepts only one below 1e-6 of the fraction-to-boundary step, at a point whose
         * error is at most acceptable_tol -> the solve ends THERE (nothing is moved) with status 0 */
        if (acc_tol > 0 && (!accepted || alpha < 1e-6 * a_p) && e0 <= acc_tol) { status = 0; break; }
        if (!accepted && alpha * dzmax < 1e-14) { status = 2; break; }
        if (g_trace) printf("%3d mu %.2e e0 %.3e th %.3e a_p %.3e alpha %.3e a_d %.3e delta %.1e rho %.2e D %.5f obj %.6f acc %d ls %d soc %d nf %d | rd %.2e rp %.2e cmin/mu %.2e cmax/mu %.2e dzmax %.2e curv %.2e\n", it, mu, e0, theta, a_p, alpha, a_d, delta, w->rho, w->D, fobj, accepted, ls_used, soc_used, nfilt, e.rd, e.rp, e.cmin / mu, e.cmax / mu, dzmax, curv);
        last_alpha = alpha; last_ad = a_d; CNT(0, 1);
        /* (experiment) the restoration trigger: the fraction-to-boundary rule has held the primal step below 1e-2 for elastic_trigger iterations in a row while rows are infeasible */
        if (g_algo.elastic_rho > 0 && g_algo.elastic_trigger > 0 && w->erho == 0 && obst_M(w) > 0) {
            if (a_p < g_algo.elastic_ap && e.rp > 1e-3) { if (jam_streak == 0) jam_theta0 = e.theta; ++jam_streak; } else jam_streak = 0;
        }
        /* accept */
        const double kS = 1e10;
        for (int r = 0; r < n; ++r) for (int q = 0; q < 4; ++q) if (row_on(w, r, q)) {
            double sn = st[4 * r + q], yn = w->y[4 * r + q] + a_d * dy[4 * r + q];
            yn = fmin(fmax(yn, mu / (kS * sn)), kS * mu / sn);
            w->s[4 * r + q] = sn; w->y[4 * r + q] = yn;
        }
        for (int k = 1, M = obst_M(w); k < n - 1; ++k) for (int m = 0; m < M; ++m) if (w->oi[k * M + m] >= 0) {
            const double sn = w->ost[k * M + m];
            double yn = w->oy[k * M + m] + a_d * w->ody[k * M + m];
            yn = fmin(fmax(yn, mu / (kS * sn)), kS * mu / sn);
            if (w->erho > 0) {      /* e and its multiplier rho - y: the same safeguards */
                const double en = w->oet[k * M + m];
                yn = fmin(fmax(yn, w->erho - kS * mu / en), w->erho - mu / (kS * en));
                w->oe[k * M + m] = en;
            }
            w->os[k * M + m] = sn; w->oy[k * M + m] = yn;
        }
        for (int k = 0; k < n - 1; ++k) {
            for (int j = 0; j < 2; ++j) {
                double u = w->U[2 * k + j], du_ = w->dz_u[2 * k + j], dl = u - c->u_lb[j], du = c->u_ub[j] - u;
                double pl = w->pl[2 * k + j], pu = w->pu[2 * k + j];
                double pln = pl + a_d * (mu / dl - pl - (pl / dl) * du_), pun = pu + a_d * (mu / du - pu + (pu / du) * du_);
                double un = w->Ut[2 * k + j], dln = un - c->u_lb[j], dun = c->u_ub[j] - un;
                w->pl[2 * k + j] = fmin(fmax(pln, mu / (kS * dln)), kS * mu / dln);
                w->pu[2 * k + j] = fmin(fmax(pun, mu / (kS * dun)), kS * mu / dun);
            }
            for (int i = 0; i < 3; ++i) w->lam[3 * k + i] += alpha * (w->lamn[3 * k + i] - w->lam[3 * k + i]);
        }
        if (ball_on(w)) {
            double yn = w->ty + a_d * w->tdy;
            w->ts = w->tst; w->ty = fmin(fmax(yn, mu / (kS * w->ts)), kS * mu / w->ts);
        }
        if (c->dt_free) {
            double dl = w->D - c->dt_lb, du = c->dt_ub - w->D;
            double pln = w->pdl + a_d * (mu / dl - w->pdl - (w->pdl / dl) * w->ddt), pun = w->pdu + a_d * (mu / du - w->pdu + (w->pdu / du) * w->ddt);
            double dln = w->Dt - c->dt_lb, dun = c->dt_ub - w->Dt;
            w->pdl = fmin(fmax(pln, mu / (kS * dln)), kS * mu / dln);
            w->pdu = fmin(fmax(pun, mu / (kS * dun)), kS * mu / dun);
        }
        memcpy(w->X, w->Xt, sizeof(double) * 3 * n);
        memcpy(w->U, w->Ut, sizeof(double) * 2 * (n - 1));
        w->D = w->Dt;
        memcpy(cc, cct, sizeof(double) * 3 * (n - 1));
        fobj = ft;
        ++it;
    }
    if (w->dual) {
        double* b = w->dual;
        if (status == 0) {
            double *bl = b + 4, *by = bl + 3 * (n - 1), *bpl = by + 4 * n, *bpu = bpl + 2 * (n - 1);
            b[0] = n; b[1] = w->pdl; b[2] = w->pdu; b[3] = ball_on(w) ? w->ty : 0.0;
            memcpy(bl, w->lam, sizeof(double) * 3 * (n - 1)); memcpy(by, w->y, sizeof(double) * 4 * n);
            memcpy(bpl, w->pl, sizeof(double) * 2 * (n - 1)); memcpy(bpu, w->pu, sizeof(double) * 2 * (n - 1));
        } else b[0] = 0;
    }
    free(cc); free(cct); free(st); free(ds); free(dy); free(y2); free(rhs_keep); free(csoc); free(ds_keep); free(dy2);
#pragma omp critical
    { g_nfac_total += nfac; if (nfac > g_nfac_max) g_nfac_max = nfac; }
    return status * 100000 + it;
}

/* TEST HOOK: g = d_min - dist(footprint(pose), obstacle) with its analytic derivatives, as the solver uses them.
 * out = [dist, a0, a1, a2, hk, h3[0], h3[1], h3[2]] */
void oracle_footprint_row(const oracle_obst* ob, const double pose[3], int nv, const double* verts, double radius, double out[8]) {
    work_t w;
    memset(&w, 0, sizeof(w));
    int32_t nvv = nv;
    w.ob = ob; w.n_obst = 1; w.n_vert = &nvv; w.verts = verts; w.radius = &radius;
    double a[3] = {0, 0, 0}, hk = 0, h3[3] = {0, 0, 0}, d;
    if (ob->footprint_kind >= 2) d = turn_dist(&w, pose[0], pose[1], pose[2], 0, a, &hk, h3);
    else { double nx, ny; obst_eval(&w, pose[0], pose[1], 0, &d, &nx, &ny, &hk); d -= ob->footprint_radius; a[0] = -nx; a[1] = -ny; }
    out[0] = d; out[1] = a[0]; out[2] = a[1]; out[3] = a[2]; out[4] = hk; out[5] = h3[0]; out[6] = h3[1]; out[7] = h3[2];
}

static work_t* work_new(const oracle_config* c) {
    work_t* w = (work_t*)calloc(1, sizeof(work_t));
    int n = c->n;
    w->c = c; w->n = n; w->N = 8 * (n - 1);
    w->X = (double*)calloc(3 * n, 8); w->U = (double*)calloc(2 * (n - 1), 8);
    w->Xt = (double*)calloc(3 * n, 8); w->Ut = (double*)calloc(2 * (n - 1), 8);
    w->lam = (double*)calloc(3 * (n - 1), 8); w->lamn = (double*)calloc(3 * (n - 1), 8);
    w->s = (double*)calloc(4 * n, 8); w->y = (double*)calloc(4 * n, 8); w->ron = (int*)calloc(4 * n, sizeof(int));
    w->pl = (double*)calloc(2 * (n - 1), 8); w->pu = (double*)calloc(2 * (n - 1), 8);
    w->AB = (double*)calloc((size_t)LDAB * w->N, 8); w->inS = (double*)calloc((size_t)(2 * KL + 1) * w->N, 8); w->inb = (double*)calloc(w->N, 8); w->ipiv = (int*)calloc(w->N, sizeof(int));
    w->rhs = (double*)calloc(w->N, 8); w->bcol = (double*)calloc(w->N, 8);
    w->dz_u = (double*)calloc(2 * (n - 1), 8); w->dz_x = (double*)calloc(3 * n, 8);
    w->bf = (double*)calloc(10 * (n - 1), 8); w->bf_g = (double*)calloc(12 * (n - 1), 8); w->bf_e = (double*)calloc(4 * (n - 1), 8);
    return w;
}
static void work_obst(work_t* w, const oracle_obst* ob) {       /* clearance-row storage (only when a batch has obstacles) */
    const int n = w->n, M = ob->max_rows, O = ob->max_obstacles;
    w->ob = ob;
    w->cent = (double*)calloc((size_t)2 * O, 8);
    w->oi = (int*)calloc((size_t)n * M, sizeof(int));
    w->os = (double*)calloc((size_t)n * M, 8); w->oy = (double*)calloc((size_t)n * M, 8); w->ost = (double*)calloc((size_t)n * M, 8);
    w->ods = (double*)calloc((size_t)n * M, 8); w->ody = (double*)calloc((size_t)n * M, 8);
    w->oe = (double*)calloc((size_t)n * M + 1, 8); w->oet = (double*)calloc((size_t)n * M + 1, 8); w->ode = (double*)calloc((size_t)n * M + 1, 8); w->erho = 0;
    w->og = (double*)calloc((size_t)n * M, 8); w->oax = (double*)calloc((size_t)n * M, 8); w->oay = (double*)calloc((size_t)n * M, 8);
    w->ohk = (double*)calloc((size_t)n * M, 8);
    w->oat = (double*)calloc((size_t)n * M, 8); w->oh3 = (double*)calloc((size_t)3 * n * M, 8);
    w->oad = (double*)calloc((size_t)n * M, 8); w->ohd = (double*)calloc((size_t)4 * n * M, 8);
}
static void work_free(work_t* w) {
    free(w->X); free(w->U); free(w->Xt); free(w->Ut); free(w->lam); free(w->lamn); free(w->s); free(w->y); free(w->ron);
    free(w->pl); free(w->pu); free(w->AB); free(w->inS); free(w->inb); free(w->ipiv); free(w->rhs); free(w->bcol); free(w->dz_u); free(w->dz_x); free(w->bf); free(w->bf_g); free(w->bf_e);
    free(w->oe); free(w->oet); free(w->ode); free(w->cent); free(w->oi); free(w->os); free(w->oy); free(w->ost); free(w->ods); free(w->ody); free(w->og); free(w->oax); free(w->oay); free(w->ohk); free(w->oat); free(w->oh3); free(w->oad); free(w->ohd);
    free(w);
}

/* Batched entry point; same array layouts as include/mpc_hip.h.  nthreads <= 0: all cores.
 * Obstacles (optional, ob != NULL): n_obst[B], n_vert[B][O], verts[B][O][V][2], radius[B][O] or NULL, as struct mpc_obstacles. */
int oracle_solve_batch_obst(const oracle_config* c, int B, const double* x0, const double* xf, const double* u_prev,
                            const double* dt_prev, const double* x_init, const double* u_init, const double* dt_init,
                            const oracle_obst* ob, const int32_t* n_obst, const int32_t* n_vert, const double* verts, const double* radius,
                            double* x_out, double* u_out, double* dt_out, int32_t* status, int32_t* iters, int nthreads) {
    const int n = c->n;
    if (n < 3) return -1;
    if (ob && (ob->max_rows <= 0 || ob->max_obstacles <= 0 || !n_obst || !n_vert || !verts)) ob = NULL;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        work_t* w = work_new(c);
        if (ob) work_obst(w, ob);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            for (int i = 0; i < 3; ++i) { w->x0[i] = x0[3 * b + i]; w->xf[i] = xf[3 * b + i]; }
            w->x0[2] = wrap(w->x0[2]); w->xf[2] = wrap(w->xf[2]);
            w->uprev[0] = u_prev ? u_prev[2 * b] : 0.0; w->uprev[1] = u_prev ? u_prev[2 * b + 1] : 0.0;
            w->dtprev = dt_prev ? dt_prev[b] : 0.0;
            if (ob) {
                const size_t O = ob->max_obstacles, V = ob->max_vertices;
                w->n_obst = n_obst[b] < (int)O ? n_obst[b] : (int)O;
                w->n_vert = n_vert + (size_t)b * O; w->verts = verts + (size_t)b * O * V * 2; w->radius = radius ? radius + (size_t)b * O : NULL;
                w->vel = (ob->dynamic && g_ovel) ? g_ovel + (size_t)b * O * 2 : NULL;
            }
            w->nvia = 0;
            if (c->via && g_nvia && g_via) { w->nvia = g_nvia[b] < g_vp_cap ? g_nvia[b] : g_vp_cap; w->via = g_via + (size_t)b * g_vp_cap * 3; }
            int warm = x_init && u_init && dt_init;
            if (warm) {
                memcpy(w->X, x_init + (size_t)b * n * 3, sizeof(double) * 3 * n);
                for (int k = 0; k < n - 1; ++k) for (int j = 0; j < 2; ++j) w->U[2 * k + j] = u_init[(size_t)b * n * 2 + 2 * k + j];
                w->D = dt_init[b];
            }
            w->rows_dropped = 0;
            w->dual = (g_dual_state && g_dual_words >= oracle_dual_words(n)) ? g_dual_state + (size_t)b * g_dual_words : NULL;
            int r = solve_one(w, warm);
            if (g_dropped_out) g_dropped_out[b] = w->rows_dropped;
            memcpy(x_out + (size_t)b * n * 3, w->X, sizeof(double) * 3 * n);
            for (int k = 0; k < n; ++k) { int ks = k < n - 1 ? k : n - 2; for (int j = 0; j < 2; ++j) u_out[(size_t)b * n * 2 + 2 * k + j] = w->U[2 * ks + j]; }
            dt_out[b] = w->D;
            if (status) status[b] = r / 100000;
            if (iters) iters[b] = r % 100000;
        }
        work_free(w);
    }
    return 0;
}
/* TEST HOOK: the association of obst_associate() on GIVEN grid states (x [n][3]) for one instance's obstacles; oi_out [n][max_rows] obstacle indices
 * (-1 = empty slot; moving obstacles first, then the static ones in the reference's order).  Returns the number of rows that did not fit. */
int oracle_associate_at(const oracle_config* c, const oracle_obst* ob, const double* x, int n_obst, const int32_t* n_vert, const double* verts, const double* radius,
                        const double* vel, int32_t* oi_out) {
    work_t* w = work_new(c);
    work_obst(w, ob);
    const int n = c->n, M = obst_M(w);
    memcpy(w->X, x, sizeof(double) * 3 * n);
    w->n_obst = n_obst < ob->max_obstacles ? n_obst : ob->max_obstacles;
    w->n_vert = n_vert; w->verts = verts; w->radius = radius; w->vel = ob->dynamic ? vel : NULL;
    obst_centroids(w);
    const int dropped = obst_associate(w);
    for (int i = 0; i < n * M; ++i) oi_out[i] = w->oi[i];
    work_free(w);
    return dropped;
}

int oracle_solve_batch(const oracle_config* c, int B, const double* x0, const double* xf, const double* u_prev,
                       const double* dt_prev, const double* x_init, const double* u_init, const double* dt_init,
                       double* x_out, double* u_out, double* dt_out, int32_t* status, int32_t* iters, int nthreads) {
    return oracle_solve_batch_obst(c, B, x0, xf, u_prev, dt_prev, x_init, u_init, dt_init, NULL, NULL, NULL, NULL, NULL, x_out, u_out, dt_out,
                                   status, iters, nthreads);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

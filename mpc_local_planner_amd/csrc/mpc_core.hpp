// mpc_core.hpp -- per-instance interior-point solve of the mpc_local_planner NLP with a
// stage-structured (Riccati) KKT factor/solve.  One GPU lane owns one planner instance;
// all per-instance arrays live in an instance-minor (SoA) workspace so that the 64 lanes of a
// wavefront touch 64 consecutive words for every (field, stage, component) they load/store.
//
// What is solved (reference files under /root/reference/mpc_local_planner/):
//   variables   x_1..x_{n-1} (SE2), u_0..u_{n-2}, dt       src/optimal_control/full_discretization_grid_base_se2.cpp:564-577
//   equality    forward-difference collocation              include/.../optimal_control/fd_collocation_se2.h:54-69
//   dynamics    unicycle / simple car / bicycle             include/.../systems/*.h
//   objective   (n-1)*dt  |  quadratic form + terminal      src/controller.cpp:551-668, src/optimal_control/quadratic_cost_se2.cpp:31-52
//   rows <= 0   control-rate rows, control/dt boxes         src/optimal_control/stage_inequality_se2.cpp:191-222, src/controller.cpp:511-543
//   retraction  theta <- wrap(theta + dtheta)               include/.../optimal_control/vector_vertex_se2.h:79-96
// Rows are used in "solver form" (positive rescalings of the reference rows, same KKT points):
//   c_k = x_k + dt f(x_k,u_k) - x_{k+1}   (= dt * reference defect),  rate rows multiplied by dt_prev.
//
// Linear algebra: the Newton system of the barrier problem is an LQ problem over the augmented
// stage state xi_k = (x_k, u_{k-1}, dt) in R^6 with control u_k in R^2.  The terminal equality
// (fixed goal components) is carried through the backward sweep as 3 extra right-hand sides
// (Bryson-Ho sweep method); dt is a state whose initial value is free.  No pivoting across
// stages, no sparse solver, O(n) work and O(n) storage per instance.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>

#if defined(__HIPCC__)
#define MPC_HD __host__ __device__ __forceinline__
#else
#define MPC_HD inline
#endif

namespace mpc {

constexpr int MODEL_UNICYCLE = 0;
constexpr int MODEL_SIMPLE_CAR = 1;
constexpr int MODEL_SIMPLE_CAR_FRONT = 2;
constexpr int MODEL_KINEMATIC_BICYCLE = 3;
constexpr int OBJ_MIN_TIME = 0;
constexpr int OBJ_QUADRATIC = 1;

constexpr int ST_CONVERGED = 0;
constexpr int ST_MAX_ITER = 1;
constexpr int ST_LINESEARCH = 2;
constexpr int ST_LINSOLVE = 3;
constexpr int ST_NUMERICAL = 4;
constexpr int ST_SUPERSEDED = 5;     // internal: a candidate stopped because a higher-priority candidate of its instance converged (never returned)

// Problem description in device-friendly form (passed by value as a kernel argument).
template <typename T>
struct Problem {
    int model;
    int n;               // grid points
    int dt_free;
    int xf_fixed[3];
    int objective;
    int integral_form;
    int collocation;     // 0 forward differences, 1 midpoint differences, 2 Crank-Nicolson (wave kernel only)
    int has_Qf;
    int rate_on[4];      // slots: lo0, lo1, hi0, hi1 (finite du bound?)
    int max_iter;
    T p0, p1;            // model params: L | (lr, lf)
    T dt_ref, dt_lb, dt_ub;
    T Q[3], R[2], Qf[3];
    T u_lb[2], u_ub[2];
    T rate_lim[4];       // du_lb0, du_lb1, du_ub0, du_ub1
    T tol, mu_init, mu_init_warm;
    // collision avoidance (wave kernel only)
    int n_obst, n_vert, obst_rows, footprint_kind;
    T d_min, force_incl, cutoff, fp_radius;
    T fp_line[4];        // line footprint: start, end in the robot frame
    int dyn_obst;        // enable_dynamic_obstacles
    int fp_nv;           // polygon footprint: vertices (robot frame)
    T fp_poly[32];
    // terminal l2-ball row  xd' S xd - gamma <= 0  on the free final state (wave kernel only)
    int ball;
    T ball_S[3], ball_gamma;
    // minimum_time_via_points objective (wave kernel only): objective stays OBJ_MIN_TIME, the via-point terms are switched by `via`
    int via, n_via, vp_ordered;
    T vp_wp, vp_wo;
    // candidate initial trajectories (wave kernel only): kinds (mpc_candidate_kind), iteration caps, heading-blend length
    int n_cand, cand_kind[4], cand_max_iter[4], cand_blend;
};

// Algorithm constants (Waechter & Biegler 2006 names).  Compile-time so that they live in
// instruction immediates instead of scalar registers.
template <typename T> struct Algo;
template <> struct Algo<double> {
    static constexpr double kappa_eps = 10, kappa_mu = 0.2, theta_mu = 1.5, tau_min = 0.99, bound_push = 1e-2, slack_push = 1e-2;
    static constexpr double eta_armijo = 1e-4, rho_frac = 0.1, delta_first = 1e-4, delta_min = 1e-20, delta_max = 1e20;
    static constexpr double kappa_plus = 8, kappa_plus_first = 100, kappa_minus = 1.0 / 3.0;
    static constexpr double curv_kappa = 1e-10, s_max = 100, delta_c = 1e-8, kappa_c = 0.25, ls_eps = 10 * 2.220446049250313e-16;
    static constexpr int max_ls = 30;
};
template <> struct Algo<float> {
    static constexpr float kappa_eps = 10, kappa_mu = 0.2f, theta_mu = 1.5f, tau_min = 0.99f, bound_push = 1e-2f, slack_push = 1e-2f;
    static constexpr float eta_armijo = 1e-4f, rho_frac = 0.1f, delta_first = 1e-4f, delta_min = 1e-12f, delta_max = 1e12f;
    static constexpr float kappa_plus = 8, kappa_plus_first = 100, kappa_minus = 1.0f / 3.0f;
    static constexpr float curv_kappa = 1e-7f, s_max = 100, delta_c = 1e-5f, kappa_c = 0.25f, ls_eps = 10 * 1.1920929e-7f;
    static constexpr int max_ls = 30;
};

// Workspace layout: slot index -> word offset = slot * stride + instance.
struct Layout {
    int n;
    int X, U, D, XT, UT, DT, LAM, LAMN, SR, YR, PL, PU, PD, DX, DU, DD, GAIN, CC, TRIG, total;
    MPC_HD static Layout make(int n) {
        Layout L;
        L.n = n;
        int o = 0;
        L.X = o;    o += 3 * n;
        L.U = o;    o += 2 * (n - 1);
        L.D = o;    o += 1;
        L.XT = o;   o += 3 * n;
        L.UT = o;   o += 2 * (n - 1);
        L.DT = o;   o += 1;
        L.LAM = o;  o += 3 * (n - 1);
        L.LAMN = o; o += 3 * (n - 1);
        L.SR = o;   o += 4 * n;
        L.YR = o;   o += 4 * n;
        L.PL = o;   o += 2 * (n - 1);
        L.PU = o;   o += 2 * (n - 1);
        L.PD = o;   o += 2;
        L.DX = o;   o += 3 * n;
        L.DU = o;   o += 2 * (n - 1);
        L.DD = o;   o += 1;
        L.GAIN = o; o += 50 * (n - 1);   // K(12) kappa(2) Knu(6) | Px(18) px(3) Sx(9)
        L.CC = o;   o += 3 * (n - 1);
        L.TRIG = o; o += 4 * (n - 1);
        L.total = o;
        return L;
    }
};

template <typename T>
struct Mem {
    T* base;
    long stride;
    MPC_HD T ld(int slot) const { return base[(long)slot * stride]; }
    MPC_HD void st(int slot, T v) const { base[(long)slot * stride] = v; }
};

MPC_HD double t_abs(double a) { return __builtin_fabs(a); }      // a source modifier on the GPU (the compare-and-select form costs 3 instructions)
MPC_HD float t_abs(float a) { return __builtin_fabsf(a); }
MPC_HD double t_fmin(double a, double b) { return __builtin_fmin(a, b); }   // IEEE minNum: one instruction, drops a NaN operand
MPC_HD float t_fmin(float a, float b) { return __builtin_fminf(a, b); }
template <typename T> MPC_HD T t_max(T a, T b) { return a > b ? a : b; }
template <typename T> MPC_HD T t_min(T a, T b) { return a < b ? a : b; }
MPC_HD double t_floor(double a) { return ::floor(a); }
MPC_HD float t_floor(float a) { return ::floorf(a); }
MPC_HD double t_log(double a) { return ::log(a); }
MPC_HD float t_log(float a) { return ::logf(a); }
MPC_HD double t_pow(double a, double b) { return ::pow(a, b); }
MPC_HD float t_pow(float a, float b) { return ::powf(a, b); }
MPC_HD double t_atan(double a) { return ::atan(a); }
MPC_HD float t_atan(float a) { return ::atanf(a); }
MPC_HD double t_atan2(double y, double x) { return ::atan2(y, x); }
MPC_HD float t_atan2(float y, float x) { return ::atan2f(y, x); }
MPC_HD double t_asin(double a) { return ::asin(a); }
MPC_HD float t_asin(float a) { return ::asinf(a); }
// reciprocal: hardware seed + two Newton steps on the device (the IEEE division sequence is ~40 instructions), 1/x on the host
MPC_HD double t_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
#else
    return 1.0 / x;
#endif
}
MPC_HD float t_rcp(float x) { return 1.0f / x; }
// ---- fp64 sine / cosine / tangent for the arguments this solver produces (angles wrapped to [-pi, pi), steering angles
//      inside their box): Cody-Waite reduction by pi/2 (two FMAs, exact for the quadrant counts that occur) and the classic
//      degree-13/14 minimax kernels on [-pi/4, pi/4] (Sun fdlibm coefficients).  ~40 instructions instead of the several hundred
//      of the general libm routine, whose extended-precision reduction dominated the line-search trials; <= 1 ulp on the
//      reduced range (tests/host_harness checks it against libm).  Arguments beyond 1e5 or non-finite take the libm path.
MPC_HD void sincos_reduced(double x, double* sp, double* cp) {
    const double k = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-k, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-k, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double s = __builtin_fma(z * r, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, wv = 1.0 - hz;
    const double c = wv + (((1.0 - wv) - hz) + z * z * pc);
    const int q = (int)k & 3;
    const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    *sp = (q & 2) ? -s1 : s1;
    *cp = ((q + 1) & 2) ? -c1 : c1;
}
MPC_HD void t_sincos(double a, double* s, double* c) {
#if defined(__HIP_DEVICE_COMPILE__)
    sincos_reduced(a, s, c);        // device callers pass wrapped angles / boxed steering angles only (no large-argument path, no branch)
#else
    if (a > -1e5 && a < 1e5) sincos_reduced(a, s, c);
    else ::sincos(a, s, c);
#endif
}
MPC_HD void t_sincos(float a, float* s, float* c) { ::sincosf(a, s, c); }
MPC_HD double t_tan(double a) {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (!(a > -1e5 && a < 1e5)) return ::tan(a);
#endif
    double s, c;
    sincos_reduced(a, &s, &c);
    return s * t_rcp(c);
}
MPC_HD float t_tan(float a) { return ::tanf(a); }
// false for NaN and +-inf.  (Not `(a - a) == 0`: with FMA contraction `a` = x*y turns that into fma(x, y, -(x*y)), the rounding
// error of the product, which is not zero.)
template <typename T> MPC_HD bool t_finite(T a) { return __builtin_isfinite(a); }

// include/mpc_local_planner/utils/math_utils.h:81-91
template <typename T>
MPC_HD T normalize_theta(T th) {
    const T pi = T(3.14159265358979323846);
    if (th >= -pi && th < pi) return th;
    T m = t_floor(th / (T(2) * pi));
    th = th - m * T(2) * pi;
    if (th >= pi) th -= T(2) * pi;
    if (th < -pi) th += T(2) * pi;
    return th;
}

// sum of logs as log of a running product; the product is renormalised after every factor pair by peeling its binary
// exponent (frexp: two cheap instructions, no branch), so it never under/overflows.  sum log = log(m) + e * ln 2.
MPC_HD void t_frexp(double a, double* m, int* e) {
#if defined(__HIP_DEVICE_COMPILE__)
    *m = __builtin_amdgcn_frexp_mant(a); *e = __builtin_amdgcn_frexp_exp(a);
#else
    *m = ::frexp(a, e);
#endif
}
MPC_HD void t_frexp(float a, float* m, int* e) {
#if defined(__HIP_DEVICE_COMPILE__)
    *m = __builtin_amdgcn_frexp_mantf(a); *e = __builtin_amdgcn_frexp_expf(a);
#else
    *m = ::frexpf(a, e);
#endif
}
// log(m) for a frexp mantissa m in [0.5, 1) (or any m in ~[0.35, 1.42]): the kernel of the classic fdlibm log --
// f = m' - 1 with m' in [sqrt(1/2), sqrt(2)), s = f / (2 + f), log(1 + f) = f - hfsq + s (hfsq + R(s^2)) -- one reciprocal and
// 14 FMAs instead of the general routine (exponent handling, sub-normals, special values).  Non-positive m yields NaN.
MPC_HD double log_mantissa(double m) {
    const bool lo = m < 0.70710678118654752440;
    const double f = (lo ? m + m : m) - 1.0;
    const double s = f * t_rcp(2.0 + f), z = s * s, w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double hfsq = 0.5 * f * f;
    const double r = f - (hfsq - s * (hfsq + (t1 + t2)));
    const double out = lo ? r - 0.69314718055994530942 : r;
    return m > 0.0 ? out : (m - m) / (m - m) + __builtin_nan("");
}
MPC_HD float log_mantissa(float m) { return t_log(m); }

template <typename T>
struct LogAcc {
    T m;
    int e;
    MPC_HD LogAcc() : m(T(1)), e(0) {}
    MPC_HD void mul(T a) {
        T mm; int ee;
        t_frexp(m * a, &mm, &ee);
        m = mm; e += ee;
    }
    MPC_HD T value() const { return log_mantissa(m) + T(e) * T(0.69314718055994530942); }
};

// Model functions: f, G = df/d(theta,v,w), and the lambda-contracted second derivative.
template <typename T>
struct ModelEval {
    T f[3];
    T G[3][3];
};

// trig cache per stage: t0 = sin, t1 = cos (of theta, or theta+beta), t2 = tan(w)|sin(w)|sin(beta), t3 = model extra
template <typename T, int MODEL>
MPC_HD void model_trig(const Problem<T>& P, T th, T w, T tr[4]) {
    if (MODEL == MODEL_KINEMATIC_BICYCLE) {
        const T kap = P.p0 / (P.p1 + P.p0);
        T t = t_tan(w);
        T beta = t_atan(kap * t);
        t_sincos(th + beta, &tr[0], &tr[1]);
        tr[2] = t;
        tr[3] = beta;
    } else {
        t_sincos(th, &tr[0], &tr[1]);
        if (MODEL == MODEL_SIMPLE_CAR) { tr[2] = t_tan(w); tr[3] = T(0); }
        else if (MODEL == MODEL_SIMPLE_CAR_FRONT) { t_sincos(w, &tr[2], &tr[3]); }
        else { tr[2] = T(0); tr[3] = T(0); }
    }
}

template <typename T, int MODEL>
MPC_HD void model_f(const Problem<T>& P, const T tr[4], T v, T w, T f[3]) {
    f[0] = v * tr[1];
    f[1] = v * tr[0];
    if (MODEL == MODEL_UNICYCLE) f[2] = w;
    else if (MODEL == MODEL_SIMPLE_CAR) f[2] = v * tr[2] / P.p0;
    else if (MODEL == MODEL_SIMPLE_CAR_FRONT) f[2] = v * tr[2] / P.p0;
    else { T sb, cb; t_sincos(tr[3], &sb, &cb); f[2] = v * sb / P.p0; }
}

// G[a][j] = d f_a / d q_j, q = (theta, v, w);  Hq = sum_a lam_a d2 f_a / dq dq (symmetric 3x3)
template <typename T, int MODEL>
MPC_HD void model_derivs(const Problem<T>& P, const T tr[4], T v, T w, const T lam[3], T f[3], T G[3][3], T Hq[3][3]) {
    const T s = tr[0], c = tr[1];
    for (int a = 0; a < 3; ++a) for (int j = 0; j < 3; ++j) { G[a][j] = T(0); Hq[a][j] = T(0); }
    if (MODEL == MODEL_KINEMATIC_BICYCLE) {
        const T lr = P.p0, lf = P.p1;
        const T kap = lr / (lf + lr);
        const T t = tr[2];
        const T tp = T(1) + t * t;
        const T tpp = T(2) * t * tp;
        const T den = T(1) + kap * kap * t * t;
        const T bp = kap * tp / den;
        const T bpp = kap * (tpp * den - tp * T(2) * kap * kap * t * tp) / (den * den);
        T sb, cb;
        t_sincos(tr[3], &sb, &cb);
        f[0] = v * c; f[1] = v * s; f[2] = v * sb / lr;
        G[0][0] = -v * s; G[0][1] = c; G[0][2] = -v * s * bp;
        G[1][0] = v * c;  G[1][1] = s; G[1][2] = v * c * bp;
        G[2][1] = sb / lr; G[2][2] = v * cb * bp / lr;
        const T l0 = lam[0], l1 = lam[1], l2 = lam[2];
        Hq[0][0] = l0 * (-v * c) + l1 * (-v * s);
        Hq[0][1] = l0 * (-s) + l1 * c;
        Hq[0][2] = l0 * (-v * c * bp) + l1 * (-v * s * bp);
        Hq[1][2] = l0 * (-s * bp) + l1 * (c * bp) + l2 * (cb * bp / lr);
        Hq[2][2] = l0 * (-v * c * bp * bp - v * s * bpp) + l1 * (-v * s * bp * bp + v * c * bpp)
                 + l2 * (v * (-sb * bp * bp + cb * bpp) / lr);
    } else {
        f[0] = v * c; f[1] = v * s;
        G[0][0] = -v * s; G[0][1] = c;
        G[1][0] = v * c;  G[1][1] = s;
        Hq[0][0] = lam[0] * (-v * c) + lam[1] * (-v * s);
        Hq[0][1] = lam[0] * (-s) + lam[1] * c;
        if (MODEL == MODEL_UNICYCLE) {
            f[2] = w;
            G[2][2] = T(1);
        } else if (MODEL == MODEL_SIMPLE_CAR) {
            const T t = tr[2], tp = T(1) + t * t, iL = T(1) / P.p0;
            f[2] = v * t * iL;
            G[2][1] = t * iL;
            G[2][2] = v * tp * iL;
            Hq[1][2] = lam[2] * tp * iL;
            Hq[2][2] = lam[2] * v * T(2) * t * tp * iL;
        } else {  // front-wheel car: tr[2] = sin w, tr[3] = cos w
            const T iL = T(1) / P.p0;
            f[2] = v * tr[2] * iL;
            G[2][1] = tr[2] * iL;
            G[2][2] = v * tr[3] * iL;
            Hq[1][2] = lam[2] * tr[3] * iL;
            Hq[2][2] = -lam[2] * v * tr[2] * iL;
        }
    }
    Hq[1][0] = Hq[0][1]; Hq[2][0] = Hq[0][2]; Hq[2][1] = Hq[1][2];
}

// ---- collocation variants (include/mpc_local_planner/optimal_control/fd_collocation_se2.h) in solver form:
//        c_k = x_k + D(theta_k, u_k, dt) - x_{k+1},  theta row wrapped,   D = dt sum_e wt_e f(theta_k + ce_e dt f_2(u_k), u_k)
//      forward differences  (:54-69)    one point  (wt, ce) = (1, 0)
//      midpoint differences (:91-108)   one point  (1, 1/2):  theta_m = theta_k + dt f_2(u_k) / 2
//      Crank-Nicolson       (:130-147)  two points (1/2, 0), (3/2, 2): the reference's code evaluates to 1.5 f(x_{k+1}) + 0.5 f(x_k) - quot
//        (`error` is aliased on the right-hand side, :139-141), restated literally; its heading row reads theta_{k+1} = theta_k + 2 dt f_2.
//      The reference evaluates the dynamics at interpolate_angle(theta_k, theta_{k+1}, 0.5) resp. at theta_{k+1}; on the constraint
//      manifold theta_{k+1} is an explicit function of (theta_k, u_k, dt) (the heading rate f_2 of every model is independent of the
//      pose), so these are the same points: same feasible set and KKT points, and the rows stay explicit in x_{k+1} (stage structure
//      of the Riccati sweep).
// model_trig_colloc fills the trig cache at the angle(s) the row is evaluated at (tr: first point + steering terms, tr2: sin/cos of the
// second point); colloc_f returns sum_e wt_e f_e (so that D = dt * that); stage_map turns the model derivatives at those angles into
// the derivatives of D with respect to (theta, v, w) and dt, and of lam' D (chain rule through the evaluation angles).
enum { COLLOC_FWD = 0, COLLOC_MID = 1, COLLOC_CN = 2 };
template <typename T>
MPC_HD int colloc_points(int method, T wt[2], T ce[2]) {
    wt[0] = T(1); ce[0] = T(0); wt[1] = T(0); ce[1] = T(0);
    if (method == COLLOC_MID) { ce[0] = T(0.5); return 1; }
    if (method == COLLOC_CN) { wt[0] = T(0.5); wt[1] = T(1.5); ce[1] = T(2); return 2; }
    return 1;
}
template <typename T, int MODEL>
MPC_HD T model_heading_rate(const Problem<T>& P, T v, T w) {
    if (MODEL == MODEL_UNICYCLE) return w;
    if (MODEL == MODEL_SIMPLE_CAR) return v * t_tan(w) / P.p0;
    if (MODEL == MODEL_SIMPLE_CAR_FRONT) { T sw, cw; t_sincos(w, &sw, &cw); return v * sw / P.p0; }
    T sb, cb; t_sincos(t_atan(P.p0 / (P.p1 + P.p0) * t_tan(w)), &sb, &cb); return v * sb / P.p0;
}
template <typename T, int MODEL>
MPC_HD void model_trig_colloc(const Problem<T>& P, T th, T v, T w, T d, T tr[4], T tr2[2]) {
    tr2[0] = T(0); tr2[1] = T(0);
    if (P.collocation == COLLOC_FWD) { model_trig<T, MODEL>(P, th, w, tr); return; }
    T wt[2], ce[2];
    const int np_ = colloc_points<T>(P.collocation, wt, ce);
    const T f2 = model_heading_rate<T, MODEL>(P, v, w);
    model_trig<T, MODEL>(P, th + ce[0] * d * f2, w, tr);
    if (np_ > 1) {
        T thb = th + ce[1] * d * f2;
        if (MODEL == MODEL_KINEMATIC_BICYCLE) thb += tr[3];      // heading + slip angle beta
        t_sincos(thb, &tr2[0], &tr2[1]);
    }
}
template <typename T, int MODEL>
MPC_HD void colloc_f(const Problem<T>& P, const T tr[4], const T tr2[2], T v, T w, T f[3]) {
    model_f<T, MODEL>(P, tr, v, w, f);
    if (P.collocation == COLLOC_CN) {
        const T trb[4] = {tr2[0], tr2[1], tr[2], tr[3]};
        T fb[3];
        model_f<T, MODEL>(P, trb, v, w, fb);
        for (int a = 0; a < 3; ++a) f[a] = T(0.5) * f[a] + T(1.5) * fb[a];
    }
}
template <typename T>
struct StageMap {
    T f[3];          // sum_e wt_e f_e  (D = dt * f)
    T Jq[3][3];      // dD/d(theta, v, w)
    T Jdt[3];        // dD/d dt
    T Hqq[3][3];     // d2 (lam' D) / d(theta, v, w)^2
    T Hqd[3];        // d2 (lam' D) / d(theta, v, w) d dt
    T Hdd;           // d2 (lam' D) / d dt^2
};
template <typename T, int MODEL>
MPC_HD void stage_map(const Problem<T>& P, const T tr[4], const T tr2[2], T v, T w, T d, const T lam[3], StageMap<T>& o) {
    T G[3][3], Hq[3][3];
    model_derivs<T, MODEL>(P, tr, v, w, lam, o.f, G, Hq);
    if (P.collocation == COLLOC_FWD) {
        T gq[3];
        for (int j = 0; j < 3; ++j) gq[j] = lam[0] * G[0][j] + lam[1] * G[1][j] + lam[2] * G[2][j];
        for (int a = 0; a < 3; ++a) { o.Jdt[a] = o.f[a]; for (int j = 0; j < 3; ++j) o.Jq[a][j] = d * G[a][j]; }
        for (int j = 0; j < 3; ++j) { o.Hqd[j] = gq[j]; for (int l = 0; l < 3; ++l) o.Hqq[j][l] = d * Hq[j][l]; }
        o.Hdd = T(0);
        return;
    }
    T wt[2], ce[2];
    const int np_ = colloc_points<T>(P.collocation, wt, ce);
    // second derivatives of the heading rate f_2 wrt (v, w): the lam = e_2 slice of the model Hessian (independent of the angle)
    const T e2[3] = {T(0), T(0), T(1)};
    T f_[3], G_[3][3], H2[3][3];
    model_derivs<T, MODEL>(P, tr, v, w, e2, f_, G_, H2);
    const T f2 = o.f[2], f2v = G[2][1], f2w = G[2][2];
    T fsum[3] = {T(0), T(0), T(0)}, L[4][4];
    for (int a = 0; a < 3; ++a) { o.Jdt[a] = T(0); for (int j = 0; j < 3; ++j) o.Jq[a][j] = T(0); }
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) L[a][b] = T(0);
    for (int e = 0; e < np_; ++e) {
        T fe[3];
        if (e == 1) {                                   // second evaluation angle: same steering terms, its own sin/cos
            const T trb[4] = {tr2[0], tr2[1], tr[2], tr[3]};
            model_derivs<T, MODEL>(P, trb, v, w, lam, fe, G, Hq);
        } else { fe[0] = o.f[0]; fe[1] = o.f[1]; fe[2] = o.f[2]; }
        T gq[3];
        for (int j = 0; j < 3; ++j) gq[j] = lam[0] * G[0][j] + lam[1] * G[1][j] + lam[2] * G[2][j];
        // m = d theta_e / d(theta, v, w, dt);  mab = its second derivatives (only (u,u) and (u,dt) are non-zero)
        const T c_ = ce[e], we = wt[e];
        const T m[4] = {T(1), c_ * d * f2v, c_ * d * f2w, c_ * f2};
        T mab[4][4];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) mab[a][b] = T(0);
        for (int j = 1; j < 3; ++j) {
            for (int l = 1; l < 3; ++l) mab[j][l] = c_ * d * H2[j][l];
            mab[j][3] = mab[3][j] = c_ * (j == 1 ? f2v : f2w);
        }
        // g(theta,u,dt) = f(theta_e,u): dg_a/dz = G[a][0] m_z + [z = u_j] G[a][j]
        for (int a = 0; a < 3; ++a) {
            T dg[4];
            for (int z = 0; z < 4; ++z) dg[z] = G[a][0] * m[z];
            dg[1] += G[a][1]; dg[2] += G[a][2];
            for (int j = 0; j < 3; ++j) o.Jq[a][j] += we * d * dg[j];
            o.Jdt[a] += we * (fe[a] + d * dg[3]);
            fsum[a] += we * fe[a];
        }
        // L_e = dt phi(theta_e, u), phi = lam' f:  L_ab = [a=dt] Dphi_b + [b=dt] Dphi_a + dt D2phi_ab
        T Dphi[4];
        for (int z = 0; z < 4; ++z) Dphi[z] = gq[0] * m[z];
        Dphi[1] += gq[1]; Dphi[2] += gq[2];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) {
            T v2 = Hq[0][0] * m[a] * m[b] + gq[0] * mab[a][b];
            if (a >= 1 && a <= 2) v2 += Hq[0][a] * m[b];
            if (b >= 1 && b <= 2) v2 += Hq[0][b] * m[a];
            if (a >= 1 && a <= 2 && b >= 1 && b <= 2) v2 += Hq[a][b];
            L[a][b] += we * (d * v2 + (a == 3 ? Dphi[b] : T(0)) + (b == 3 ? Dphi[a] : T(0)));
        }
    }
    for (int a = 0; a < 3; ++a) o.f[a] = fsum[a];
    for (int j = 0; j < 3; ++j) { o.Hqd[j] = L[j][3]; for (int l = 0; l < 3; ++l) o.Hqq[j][l] = L[j][l]; }
    o.Hdd = L[3][3];
}

// ---------------------------------------------------------------------------------------------
// Structured backward (Riccati) stage step, shared by both kernels.
//
// Value function of stage k+1 over xi+ = (x+, u_k, dt):  1/2 xi'P xi + xi'(p + S nu) + 1/2 nu'W nu + nu'om.
// Stage map: x+ = x + a*theta + Bx u + f dt + c with a = (a0, a1, 0)  (forward differences: only the theta
// column of df/dx is non-zero for every model), next u-state = u_k, dt+ = dt.
// Stage cost = Lagrangian curvature of lam'(dt f) + objective + condensed barrier terms of the control box and
// of the (linear) control-rate rows; the previous control enters only through the rate rows, so all of its
// blocks are diagonal and never formed as dense matrices.  ~280 FMAs (the dense 6+2 version needs >600).
// Index of the combined stage-cost entries ("A-form"): the non-zeros of the symmetric 8x8 stage Hessian over
// (x0,x1,x2, up0,up1, d | u0,u1) and of its gradient column, already summed over Lagrangian curvature, objective,
// condensed control-box, rate-row and clearance-row terms -- everything except the regularisation delta and the
// dt-box/objective terms that live at stage 0.
enum StageAdd {
    A00 = 0, A01, A11, A22, A25, A26, A27, A33, A35, A36, A44, A45, A47, A55, A56, A57, A66, A67, A77,   // Hessian (19)
    A08, A18, A28, A38, A48, A58, A68, A78,                                                              // gradient (8)
    NADD_BASE,                                                                                           // the headline configurations stop here (27 entries)
    A02 = NADD_BASE, A12,                                                                                // position-heading coupling (clearance rows of a footprint that turns with the pose)
    A05, A15,                                                                                            // position-dt coupling (integral-form cost on the variable grid, dynamic obstacles)
    NADD
};

template <typename T>
struct StageRec {
    T a0, a1;          // dt * d f_{0,1} / d theta
    T f[3];            // f(x_k, u_k)
    T B[3][2];         // dt * d f / d u
    T c[3];            // collocation residual c_k
    T A[NADD];         // combined stage cost, see StageAdd
};

// assembles the A-form from its ingredients (used by both kernels; the raw pieces are documented at the call sites)
template <typename T>
struct StageParts {
    T h00, h01, h02, h11, h12, h22;   // dt * sum_a lam_a d2 f_a / d(theta,v,w)^2
    T g[3];                           // sum_a lam_a d f_a / d(theta,v,w)   (cross terms with dt)
    T hdd;                            // d2 (lam' D) / d dt^2 (midpoint collocation; 0 for forward differences)
    T sz[2], gb[2];                   // control box: Sigma, barrier (+objective) gradient
    T ss[2], sl[2], sll, gy[2], gyl;  // rate rows: sum sigma, sigma*lim, sigma*lim^2, sg*ybar, sg*lim*ybar
    T hx[3];                          // objective gradient wrt x_k
    T oxx, oxy, oyy, ogx, ogy;        // clearance rows
    T oxt = T(0), oyt = T(0), ott = T(0), ogt = T(0);   // ... heading parts (footprints that turn with the pose)
    T cxd[3] = {T(0), T(0), T(0)}, cud[2] = {T(0), T(0)}, gdt = T(0);   // integral-form cost with free dt: d2/dx ddt, d2/du ddt, d/ddt
};
template <typename T>
MPC_HD void assemble_adds(const StageParts<T>& s, const T q2[3], const T r2[2], T A[NADD]) {
    A[A00] = q2[0] + s.oxx; A[A01] = s.oxy; A[A11] = q2[1] + s.oyy; A[A22] = q2[2] + s.h00 + s.ott;
    A[A02] = s.oxt; A[A12] = s.oyt; A[A05] = s.cxd[0]; A[A15] = s.cxd[1];
    A[A25] = s.g[0] + s.cxd[2]; A[A26] = s.h01; A[A27] = s.h02;
    A[A33] = s.ss[0]; A[A35] = s.sl[0]; A[A36] = -s.ss[0];
    A[A44] = s.ss[1]; A[A45] = s.sl[1]; A[A47] = -s.ss[1];
    A[A55] = s.sll + s.hdd; A[A56] = s.g[1] - s.sl[0] + s.cud[0]; A[A57] = s.g[2] - s.sl[1] + s.cud[1];
    A[A66] = s.h11 + s.sz[0] + s.ss[0] + r2[0]; A[A67] = s.h12; A[A77] = s.h22 + s.sz[1] + s.ss[1] + r2[1];
    A[A08] = s.hx[0] + s.ogx; A[A18] = s.hx[1] + s.ogy; A[A28] = s.hx[2] + s.ogt;
    A[A38] = -s.gy[0]; A[A48] = -s.gy[1]; A[A58] = -s.gyl + s.gdt;
    A[A68] = s.gb[0] + s.gy[0]; A[A78] = s.gb[1] + s.gy[1];
}

template <typename T>
struct RicState {
    T P[6][6], p[6], S[6][3], W[3][3], om[3];
};

template <typename T>
struct StageGain {
    T K[2][6], kap[2], Kn[2][3];
};

// dx = regularisation of x_k (0 at k = 0), du = regularisation of u_k, add_dd / add_qd = dt-box + objective terms that
// live at stage 0.  Returns false on a singular pivot.
template <typename T>
MPC_HD bool riccati_step(RicState<T>& V, const StageRec<T>& r, T dx, T du, T add_dd, T add_qd, StageGain<T>& out) {
    T (&P)[6][6] = V.P;
    const T* A = r.A;
    const T c0 = r.c[0], c1 = r.c[1], c2 = r.c[2];
    T w[6];
    for (int i = 0; i < 6; ++i) w[i] = V.p[i] + P[i][0] * c0 + P[i][1] * c1 + P[i][2] * c2;
    for (int b = 0; b < 3; ++b) V.om[b] += V.S[0][b] * c0 + V.S[1][b] * c1 + V.S[2][b] * c2;
    T E[3][2], e[3], Pa[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 2; ++j) E[i][j] = P[i][0] * r.B[0][j] + P[i][1] * r.B[1][j] + P[i][2] * r.B[2][j] + P[i][3 + j];
        e[i] = P[i][0] * r.f[0] + P[i][1] * r.f[1] + P[i][2] * r.f[2] + P[i][5];
        Pa[i] = P[i][0] * r.a0 + P[i][1] * r.a1;
    }
    // Q~ over x (symmetric), cross x-d, d-d
    T Q00 = P[0][0] + dx + A[A00], Q01 = P[0][1] + A[A01], Q11 = P[1][1] + dx + A[A11];
    T Q02 = P[0][2] + Pa[0], Q12 = P[1][2] + Pa[1];
    T Q22 = P[2][2] + T(2) * Pa[2] + r.a0 * Pa[0] + r.a1 * Pa[1] + dx + A[A22];
    T qd0 = e[0], qd1 = e[1], qd2 = e[2] + r.a0 * e[0] + r.a1 * e[1] + A[A25];
    T Qdd = r.f[0] * (e[0] + P[0][5]) + r.f[1] * (e[1] + P[1][5]) + r.f[2] * (e[2] + P[2][5]) + P[5][5] + A[A55] + add_dd;
    // M~ (u rows): x columns, d column; the u_{k-1} columns are diag(A36, A47)
    T Mx[2][3], Md[2];
    for (int j = 0; j < 2; ++j) {
        Mx[j][0] = E[0][j];
        Mx[j][1] = E[1][j];
        Mx[j][2] = E[2][j] + r.a0 * E[0][j] + r.a1 * E[1][j];
        Md[j] = r.B[0][j] * e[0] + r.B[1][j] * e[1] + r.B[2][j] * e[2] + P[0][3 + j] * r.f[0] + P[1][3 + j] * r.f[1] + P[2][3 + j] * r.f[2]
              + P[3 + j][5] + A[A56 + j];
    }
    Mx[0][2] += A[A26];
    Mx[1][2] += A[A27];
    const T mu0 = A[A36], mu1 = A[A47];      // M~[0][up0], M~[1][up1]
    // R~
    T R00 = r.B[0][0] * E[0][0] + r.B[1][0] * E[1][0] + r.B[2][0] * E[2][0] + P[0][3] * r.B[0][0] + P[1][3] * r.B[1][0] + P[2][3] * r.B[2][0]
          + P[3][3] + A[A66] + du;
    T R01 = r.B[0][0] * E[0][1] + r.B[1][0] * E[1][1] + r.B[2][0] * E[2][1] + P[0][3] * r.B[0][1] + P[1][3] * r.B[1][1] + P[2][3] * r.B[2][1]
          + P[3][4] + A[A67];
    T R11 = r.B[0][1] * E[0][1] + r.B[1][1] * E[1][1] + r.B[2][1] * E[2][1] + P[0][4] * r.B[0][1] + P[1][4] * r.B[1][1] + P[2][4] * r.B[2][1]
          + P[4][4] + A[A77] + du;
    // gradients
    T qx0 = w[0] + A[A08], qx1 = w[1] + A[A18], qx2 = w[2] + r.a0 * w[0] + r.a1 * w[1] + A[A28];
    T ru[2];
    for (int j = 0; j < 2; ++j) ru[j] = r.B[0][j] * w[0] + r.B[1][j] * w[1] + r.B[2][j] * w[2] + w[3 + j] + A[A68 + j];
    T qdd = r.f[0] * w[0] + r.f[1] * w[1] + r.f[2] * w[2] + w[5] + A[A58] + add_qd;
    // nu columns
    T Sx2[3], Su[2][3], Sd[3];
    for (int b = 0; b < 3; ++b) {
        Sx2[b] = V.S[2][b] + r.a0 * V.S[0][b] + r.a1 * V.S[1][b];
        for (int j = 0; j < 2; ++j) Su[j][b] = r.B[0][j] * V.S[0][b] + r.B[1][j] * V.S[1][b] + r.B[2][j] * V.S[2][b] + V.S[3 + j][b];
        Sd[b] = r.f[0] * V.S[0][b] + r.f[1] * V.S[1][b] + r.f[2] * V.S[2][b] + V.S[5][b];
    }
    // eliminate u_k
    T det = R00 * R11 - R01 * R01;
    T scale = t_abs(R00 * R11) + R01 * R01;
    if (!(t_abs(det) > T(1e-14) * scale) || !t_finite(det)) return false;
    T id = T(1) / det;
    T Ri00 = R11 * id, Ri01 = -R01 * id, Ri11 = R00 * id;
    T (&K)[2][6] = out.K;
    for (int i = 0; i < 3; ++i) {
        K[0][i] = Ri00 * Mx[0][i] + Ri01 * Mx[1][i];
        K[1][i] = Ri01 * Mx[0][i] + Ri11 * Mx[1][i];
    }
    K[0][3] = Ri00 * mu0; K[0][4] = Ri01 * mu1;
    K[1][3] = Ri01 * mu0; K[1][4] = Ri11 * mu1;
    K[0][5] = Ri00 * Md[0] + Ri01 * Md[1];
    K[1][5] = Ri01 * Md[0] + Ri11 * Md[1];
    out.kap[0] = Ri00 * ru[0] + Ri01 * ru[1];
    out.kap[1] = Ri01 * ru[0] + Ri11 * ru[1];
    for (int b = 0; b < 3; ++b) {
        out.Kn[0][b] = Ri00 * Su[0][b] + Ri01 * Su[1][b];
        out.Kn[1][b] = Ri01 * Su[0][b] + Ri11 * Su[1][b];
    }
    const T Qx[3][3] = {{Q00, Q01, Q02}, {Q01, Q11, Q12}, {Q02, Q12, Q22}};
    const T qd[3] = {qd0, qd1, qd2};
    const T qx[3] = {qx0, qx1, qx2};
    T Sx0[3], Sx1[3];
    for (int b = 0; b < 3; ++b) { Sx0[b] = V.S[0][b]; Sx1[b] = V.S[1][b]; }
    for (int i = 0; i < 3; ++i) {
        for (int l = i; l < 3; ++l) { T v = Qx[i][l] - (Mx[0][i] * K[0][l] + Mx[1][i] * K[1][l]); P[i][l] = v; P[l][i] = v; }
        for (int l = 0; l < 2; ++l) { T v = -(Mx[0][i] * K[0][3 + l] + Mx[1][i] * K[1][3 + l]); P[i][3 + l] = v; P[3 + l][i] = v; }
        { T v = qd[i] - (Mx[0][i] * K[0][5] + Mx[1][i] * K[1][5]); P[i][5] = v; P[5][i] = v; }
        V.p[i] = qx[i] - (Mx[0][i] * out.kap[0] + Mx[1][i] * out.kap[1]);
        for (int b = 0; b < 3; ++b) {
            T sx = i == 0 ? Sx0[b] : (i == 1 ? Sx1[b] : Sx2[b]);
            V.S[i][b] = sx - (Mx[0][i] * out.Kn[0][b] + Mx[1][i] * out.Kn[1][b]);
        }
    }
    // u_{k-1} block: M~[a][up_j] = mu_j delta_aj
    P[3][3] = A[A33] - mu0 * K[0][3];
    P[3][4] = -mu0 * K[0][4]; P[4][3] = P[3][4];
    P[4][4] = A[A44] - mu1 * K[1][4];
    P[3][5] = A[A35] - mu0 * K[0][5]; P[5][3] = P[3][5];
    P[4][5] = A[A45] - mu1 * K[1][5]; P[5][4] = P[4][5];
    P[5][5] = Qdd - (Md[0] * K[0][5] + Md[1] * K[1][5]);
    V.p[3] = A[A38] - mu0 * out.kap[0];
    V.p[4] = A[A48] - mu1 * out.kap[1];
    V.p[5] = qdd - (Md[0] * out.kap[0] + Md[1] * out.kap[1]);
    for (int b = 0; b < 3; ++b) {
        V.S[3][b] = -mu0 * out.Kn[0][b];
        V.S[4][b] = -mu1 * out.Kn[1][b];
        V.S[5][b] = Sd[b] - (Md[0] * out.Kn[0][b] + Md[1] * out.Kn[1][b]);
    }
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) V.W[a][b] -= Su[0][a] * out.Kn[0][b] + Su[1][a] * out.Kn[1][b];
        V.om[a] -= Su[0][a] * out.kap[0] + Su[1][a] * out.kap[1];
    }
    return true;
}

// terminal value function and the final 4x4 solve, shared by both kernels
template <typename T>
MPC_HD void riccati_terminal(RicState<T>& V, const Problem<T>& P_, const T xd_f[3], T delta, T dc, const T ss[2], const T sl[2], T sll,
                             const T gy[2], T gyl) {
    for (int a = 0; a < 6; ++a) { V.p[a] = T(0); for (int b = 0; b < 6; ++b) V.P[a][b] = T(0); for (int b = 0; b < 3; ++b) V.S[a][b] = T(0); }
    for (int a = 0; a < 3; ++a) { V.om[a] = T(0); for (int b = 0; b < 3; ++b) V.W[a][b] = T(0); }
    for (int i = 0; i < 3; ++i) {
        if (P_.xf_fixed[i]) { V.S[i][i] = T(1); V.W[i][i] = -dc; }
        else {
            V.P[i][i] = delta;
            if (P_.objective == OBJ_QUADRATIC && P_.has_Qf) { V.P[i][i] += T(2) * P_.Qf[i]; V.p[i] = T(2) * P_.Qf[i] * xd_f[i]; }
        }
    }
    for (int j = 0; j < 2; ++j) {      // final rate rows: a over (up_j, d) = (-sg, -sg*lim)
        V.P[3 + j][3 + j] += ss[j];
        V.P[3 + j][5] += sl[j]; V.P[5][3 + j] += sl[j];
        V.p[3 + j] -= gy[j];
    }
    V.P[5][5] += sll;
    V.p[5] -= gyl;
}

template <typename T>
MPC_HD bool riccati_root(const RicState<T>& V, const Problem<T>& P_, T& dd_out, T nu_out[3]) {
    T A4[4][5];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 5; ++b) A4[a][b] = T(0);
    if (P_.dt_free) {
        A4[0][0] = V.P[5][5];
        for (int b = 0; b < 3; ++b) A4[0][1 + b] = P_.xf_fixed[b] ? V.S[5][b] : T(0);
        A4[0][4] = -V.p[5];
    } else { A4[0][0] = T(1); }
    for (int a = 0; a < 3; ++a) {
        if (P_.xf_fixed[a]) {
            A4[1 + a][0] = P_.dt_free ? V.S[5][a] : T(0);
            for (int b = 0; b < 3; ++b) A4[1 + a][1 + b] = P_.xf_fixed[b] ? V.W[a][b] : T(0);
            A4[1 + a][4] = -V.om[a];
        } else { A4[1 + a][1 + a] = T(1); }
    }
    // Gaussian elimination with partial pivoting, written with compile-time indices only: the pivot row is brought up by
    // compare-and-swap selects (a run-time row index would put the 4x5 tableau into scratch memory on the GPU).
    T ipiv[4];
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            const bool sw = t_abs(A4[r][c]) > t_abs(A4[c][c]);
#pragma unroll
            for (int b = 0; b < 5; ++b) { const T x = A4[c][b], y = A4[r][b]; A4[c][b] = sw ? y : x; A4[r][b] = sw ? x : y; }
        }
        const T best = t_abs(A4[c][c]);
        ok = ok && (best > T(0)) && t_finite(best);
        const T ip = t_rcp(A4[c][c]);
        ipiv[c] = ip;
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            const T m = A4[r][c] * ip;
#pragma unroll
            for (int b = c; b < 5; ++b) A4[r][b] -= m * A4[c][b];
        }
    }
    if (!ok) return false;
    T sol[4];
#pragma unroll
    for (int c = 3; c >= 0; --c) {
        T a = A4[c][4];
#pragma unroll
        for (int b = c + 1; b < 4; ++b) a -= A4[c][b] * sol[b];
        sol[c] = a * ipiv[c];
    }
    dd_out = sol[0];
    nu_out[0] = sol[1]; nu_out[1] = sol[2]; nu_out[2] = sol[3];
    return t_finite(sol[0]) && t_finite(sol[1]) && t_finite(sol[2]) && t_finite(sol[3]);
}

// rate-row slot helpers: q in 0..3 -> component j, sign sg
MPC_HD int slot_comp(int q) { return q & 1; }
template <typename T> MPC_HD T slot_sign(int q) { return q < 2 ? T(-1) : T(1); }

template <typename T>
struct SolveStats {
    int status;
    int iters;
    T kkt_error;
    T objective;
};

template <typename T, int MODEL>
struct Ipm {
    const Problem<T>& P;     // on the GPU this refers to a copy in LDS (see the kernels): uniform values read by
    const Layout& L;         // broadcast ds_read instead of occupying (and spilling) scalar registers
    Mem<T> M;
    // per-instance inputs
    T x0[3], xf[3], uprev[2], dtprev;
    // scalar state
    T mu, rho, delta_last;
    int nfix;
    bool row0_on;
    bool fail0;

    MPC_HD Ipm(const Problem<T>& p, const Layout& l, Mem<T> m) : P(p), L(l), M(m) {}

    // ---------------------------------------------------------------- accessors
    MPC_HD T X(int base, int k, int i) const { return M.ld(base + 3 * k + i); }
    MPC_HD T U(int base, int k, int j) const { return M.ld(base + 2 * k + j); }

    MPC_HD bool row_on(int r, int q) const { return P.rate_on[q] && (r > 0 || row0_on); }

    // value of rate row r, slot q at the given controls / dt      (solver form, <= 0 feasible)
    MPC_HD T rate_g(int r, int q, T ur, T um, T d) const {
        const T sg = slot_sign<T>(q);
        const T dtp = r > 0 ? d : dtprev;
        return sg * ((ur - um) - P.rate_lim[q] * dtp);
    }

    // ---------------------------------------------------------------- initial point
    MPC_HD void cold_start() {
        // Controller::generateInitialStateTrajectory + initializeSequences(xinit) for a 2-pose plan:
        // src/controller.cpp:807-857, full_discretization_grid_base_se2.cpp:192-239
        const int n = L.n;
        const T dth = normalize_theta(xf[2] - x0[2]);
        for (int k = 0; k < n; ++k) {
            T fr = T(k) / T(n - 1);
            T xk[3];
            if (k == 0) { xk[0] = x0[0]; xk[1] = x0[1]; xk[2] = x0[2]; }
            else if (k == n - 1) { xk[0] = xf[0]; xk[1] = xf[1]; xk[2] = xf[2]; }
            else {
                xk[0] = x0[0] + fr * (xf[0] - x0[0]);
                xk[1] = x0[1] + fr * (xf[1] - x0[1]);
                xk[2] = normalize_theta(x0[2] + fr * dth);
            }
            for (int i = 0; i < 3; ++i) M.st(L.X + 3 * k + i, xk[i]);
        }
        for (int k = 0; k < n - 1; ++k) { M.st(L.U + 2 * k, T(0)); M.st(L.U + 2 * k + 1, T(0)); }
        M.st(L.D, P.dt_ref);
    }

    MPC_HD void seed_controls_if_zero() {
        const int n = L.n;
        bool any = false;
        for (int k = 0; k < n - 1; ++k) any = any || (U(L.U, k, 0) != T(0)) || (U(L.U, k, 1) != T(0));
        if (any) return;
        const T d = M.ld(L.D);
        for (int k = 0; k < n - 1; ++k) {
            T dx = X(L.X, k + 1, 0) - X(L.X, k, 0);
            T dy = X(L.X, k + 1, 1) - X(L.X, k, 1);
            T dth = normalize_theta(X(L.X, k + 1, 2) - X(L.X, k, 2));
            T th = X(L.X, k, 2);
            T s, c;
            t_sincos(th, &s, &c);
            T v = (dx * c + dy * s) / d;
            v = t_min(t_max(v, P.u_lb[0]), P.u_ub[0]);
            T rate = dth / d;
            T w;
            if (MODEL == MODEL_UNICYCLE) w = rate;
            else {
                T vv = t_abs(v) > T(1e-3) ? v : (v >= T(0) ? T(1e-3) : T(-1e-3));
                if (MODEL == MODEL_SIMPLE_CAR) w = t_atan(P.p0 * rate / vv);
                else if (MODEL == MODEL_SIMPLE_CAR_FRONT) w = t_asin(t_min(T(1), t_max(T(-1), P.p0 * rate / vv)));
                else {
                    T sb = t_min(T(1), t_max(T(-1), P.p0 * rate / vv));
                    w = t_atan(t_tan(t_asin(sb)) * (P.p1 + P.p0) / P.p0);
                }
            }
            w = t_min(t_max(w, P.u_lb[1]), P.u_ub[1]);
            M.st(L.U + 2 * k, v);
            M.st(L.U + 2 * k + 1, w);
        }
    }

    MPC_HD T push_interior(T v, T lb, T ub) const {
        T pl = t_min(Algo<T>::bound_push * t_max(T(1), t_abs(lb)), Algo<T>::bound_push * (ub - lb));
        T pu = t_min(Algo<T>::bound_push * t_max(T(1), t_abs(ub)), Algo<T>::bound_push * (ub - lb));
        return t_min(t_max(v, lb + pl), ub - pu);
    }

    // ---------------------------------------------------------------- trig / residual cache at (XB,UB,DB)
    // writes trig and c_k for every interval; returns sum |c| and objective f
    MPC_HD void eval_point(int XB, int UB, int DB, int TRB, int CB, T& theta_c, T& fobj, T& cinf) const {
        const int n = L.n;
        const T d = M.ld(DB);
        theta_c = T(0);
        cinf = T(0);
        fobj = (P.objective == OBJ_MIN_TIME) ? T(n - 1) * d : T(0);
        T xk[3] = {X(XB, 0, 0), X(XB, 0, 1), X(XB, 0, 2)};
        for (int k = 0; k < n - 1; ++k) {
            T v = U(UB, k, 0), w = U(UB, k, 1);
            T tr[4], f[3];
            model_trig<T, MODEL>(P, xk[2], w, tr);
            model_f<T, MODEL>(P, tr, v, w, f);
            T xn[3] = {X(XB, k + 1, 0), X(XB, k + 1, 1), X(XB, k + 1, 2)};
            T c0 = d * f[0] - (xn[0] - xk[0]);
            T c1 = d * f[1] - (xn[1] - xk[1]);
            T c2 = d * f[2] - normalize_theta(xn[2] - xk[2]);
            for (int i = 0; i < 4; ++i) M.st(TRB + 4 * k + i, tr[i]);
            M.st(CB + 3 * k, c0); M.st(CB + 3 * k + 1, c1); M.st(CB + 3 * k + 2, c2);
            theta_c += t_abs(c0) + t_abs(c1) + t_abs(c2);
            cinf = t_max(cinf, t_max(t_abs(c0), t_max(t_abs(c1), t_abs(c2))));
            if (P.objective == OBJ_QUADRATIC) {
                T xd0 = xk[0] - xf[0], xd1 = xk[1] - xf[1], xd2 = normalize_theta(xk[2] - xf[2]);
                T sc = P.Q[0] * xd0 * xd0 + P.Q[1] * xd1 * xd1 + P.Q[2] * xd2 * xd2 + P.R[0] * v * v + P.R[1] * w * w;
                fobj += P.integral_form ? sc * d : sc;
            }
            xk[0] = xn[0]; xk[1] = xn[1]; xk[2] = xn[2];
        }
        if (P.objective == OBJ_QUADRATIC && P.has_Qf) {
            T xd[3] = {xk[0] - xf[0], xk[1] - xf[1], normalize_theta(xk[2] - xf[2])};
            for (int i = 0; i < 3; ++i) if (!P.xf_fixed[i]) fobj += P.Qf[i] * xd[i] * xd[i];
        }
    }

    // barrier terms and inequality residuals at (UB,DB) with slacks scaled implicitly:
    // for the (linear) rate rows  g(z+a dz) + (s + a ds) = (1-a)(g+s), so only log terms need the trial slacks.
    MPC_HD T barrier_logs(int UB, int DB, T alpha, bool trial) const {
        // returns  sum log(s) + sum log(u-lb) + sum log(ub-u) + logs of dt bounds  at the current (alpha=0)
        // or trial point; trial slacks s + alpha*ds are recomputed from the stored step.
        const int n = L.n;
        LogAcc<T> acc;
        const T d = M.ld(DB);
        for (int k = 0; k < n - 1; ++k) {
            for (int j = 0; j < 2; ++j) {
                T u = U(UB, k, j);
                acc.mul(u - P.u_lb[j]);
                acc.mul(P.u_ub[j] - u);
            }
        }
        if (P.dt_free) { acc.mul(d - P.dt_lb); acc.mul(P.dt_ub - d); }
        for (int r = 0; r < n; ++r) {
            for (int q = 0; q < 4; ++q) {
                if (!row_on(r, q)) continue;
                T s = M.ld(L.SR + 4 * r + q);
                if (trial) s += alpha * row_ds(r, q);
                acc.mul(s);
            }
        }
        return acc.value();
    }

    // J_g dz for rate row r, slot q, from the stored step
    MPC_HD T row_jdz(int r, int q) const {
        const int n = L.n;
        const int j = slot_comp(q);
        const T sg = slot_sign<T>(q);
        T dur = r < n - 1 ? M.ld(L.DU + 2 * r + j) : T(0);
        T dum = r > 0 ? M.ld(L.DU + 2 * (r - 1) + j) : T(0);
        T dd = r > 0 ? M.ld(L.DD) : T(0);
        return sg * ((dur - dum) - P.rate_lim[q] * dd);
    }
    MPC_HD T row_val(int r, int q) const {
        const int n = L.n;
        const int j = slot_comp(q);
        T ur = r < n - 1 ? U(L.U, r, j) : T(0);
        T um = r > 0 ? U(L.U, r - 1, j) : uprev[j];
        return rate_g(r, q, ur, um, M.ld(L.D));
    }
    MPC_HD T row_ds(int r, int q) const {
        T s = M.ld(L.SR + 4 * r + q);
        return -(row_val(r, q) + s) - row_jdz(r, q);
    }

    // ---------------------------------------------------------------- KKT error pass
    struct Err {
        T rd, rp, cmin, cmax, sum_mult, sum_bmult;
        int n_mult, n_bmult;
        T theta;   // l1 constraint violation
    };

    MPC_HD Err kkt_pass() const {
        const int n = L.n;
        Err e;
        e.rd = T(0); e.rp = T(0); e.cmin = T(1e30); e.cmax = T(0); e.sum_mult = T(0); e.sum_bmult = T(0);
        e.n_mult = 0; e.n_bmult = 0; e.theta = T(0);
        const T d = M.ld(L.D);
        T rd_d = (P.objective == OBJ_MIN_TIME) ? T(n - 1) : T(0);
        T lam_prev[3] = {T(0), T(0), T(0)};
        for (int k = 0; k < n - 1; ++k) {
            T lam[3] = {M.ld(L.LAM + 3 * k), M.ld(L.LAM + 3 * k + 1), M.ld(L.LAM + 3 * k + 2)};
            T tr[4] = {M.ld(L.TRIG + 4 * k), M.ld(L.TRIG + 4 * k + 1), M.ld(L.TRIG + 4 * k + 2), M.ld(L.TRIG + 4 * k + 3)};
            T v = U(L.U, k, 0), w = U(L.U, k, 1);
            T f[3], G[3][3], Hq[3][3];
            model_derivs<T, MODEL>(P, tr, v, w, lam, f, G, Hq);
            T gq[3];
            for (int j = 0; j < 3; ++j) gq[j] = lam[0] * G[0][j] + lam[1] * G[1][j] + lam[2] * G[2][j];
            for (int i = 0; i < 3; ++i) {
                T ci = M.ld(L.CC + 3 * k + i);
                e.rp = t_max(e.rp, t_abs(ci));
                e.theta += t_abs(ci);
                e.sum_mult += t_abs(lam[i]);
            }
            e.n_mult += 3;
            rd_d += lam[0] * f[0] + lam[1] * f[1] + lam[2] * f[2];
            // quadratic objective gradient pieces
            T gx[3] = {T(0), T(0), T(0)}, gu[2] = {T(0), T(0)};
            if (P.objective == OBJ_QUADRATIC) {
                T w8 = P.integral_form ? d : T(1);
                T xd[3] = {X(L.X, k, 0) - xf[0], X(L.X, k, 1) - xf[1], normalize_theta(X(L.X, k, 2) - xf[2])};
                for (int i = 0; i < 3; ++i) gx[i] = T(2) * P.Q[i] * xd[i] * w8;
                gu[0] = T(2) * P.R[0] * v * w8; gu[1] = T(2) * P.R[1] * w * w8;
                if (P.integral_form)
                    rd_d += P.Q[0] * xd[0] * xd[0] + P.Q[1] * xd[1] * xd[1] + P.Q[2] * xd[2] * xd[2] + P.R[0] * v * v + P.R[1] * w * w;
            }
            // x_k stationarity (k >= 1)
            if (k >= 1) {
                T r0 = gx[0] + lam[0] - lam_prev[0];
                T r1 = gx[1] + lam[1] - lam_prev[1];
                T r2 = gx[2] + lam[2] + d * gq[0] - lam_prev[2];
                e.rd = t_max(e.rd, t_max(t_abs(r0), t_max(t_abs(r1), t_abs(r2))));
#ifdef MPC_TRACE
                if (MPC_TRACE_COND && mu < T(1e-8) && (t_abs(r0) > T(1e-7) || t_abs(r1) > T(1e-7) || t_abs(r2) > T(1e-7)))
                    printf("  x-stat k=%d r=(%.3e %.3e %.3e) gx=(%.3e %.3e %.3e) lam=(%.6e %.6e %.6e) lamp=(%.6e %.6e %.6e) dgq0 %.3e\n", k, (double)r0, (double)r1, (double)r2,
                           (double)gx[0], (double)gx[1], (double)gx[2], (double)lam[0], (double)lam[1], (double)lam[2], (double)lam_prev[0], (double)lam_prev[1], (double)lam_prev[2], (double)(d * gq[0]));
#endif
            }
            // u_k stationarity
            for (int j = 0; j < 2; ++j) {
                T u = j == 0 ? v : w;
                T pl = M.ld(L.PL + 2 * k + j), pu = M.ld(L.PU + 2 * k + j);
                T r = gu[j] + d * gq[1 + j] - pl + pu;
                for (int q = j; q < 4; q += 2) {
                    const T sg = slot_sign<T>(q);
                    if (row_on(k, q)) r += sg * M.ld(L.YR + 4 * k + q);
                    if (row_on(k + 1, q)) r -= sg * M.ld(L.YR + 4 * (k + 1) + q);
                }
                e.rd = t_max(e.rd, t_abs(r));
#ifdef MPC_TRACE
                if (MPC_TRACE_COND && mu < T(1e-8) && t_abs(r) > T(1e-7)) printf("  u-stat k=%d j=%d r=%.3e\n", k, j, (double)r);
#endif
                T cl = (u - P.u_lb[j]) * pl, cu = (P.u_ub[j] - u) * pu;
                e.cmin = t_min(e.cmin, t_min(cl, cu));
                e.cmax = t_max(e.cmax, t_max(cl, cu));
                e.sum_bmult += pl + pu;
                e.n_bmult += 2;
            }
            lam_prev[0] = lam[0]; lam_prev[1] = lam[1]; lam_prev[2] = lam[2];
        }
        // free terminal components
        for (int i = 0; i < 3; ++i) {
            if (!P.xf_fixed[i]) {
                T g = T(0);
                if (P.objective == OBJ_QUADRATIC && P.has_Qf) {
                    T xd = X(L.X, n - 1, i) - xf[i];
                    if (i == 2) xd = normalize_theta(xd);
                    g = T(2) * P.Qf[i] * xd;
                }
                e.rd = t_max(e.rd, t_abs(g - lam_prev[i]));
#ifdef MPC_TRACE
                if (MPC_TRACE_COND && mu < T(1e-8) && t_abs(g - lam_prev[i]) > T(1e-7)) printf("  xf-stat i=%d g=%.9e lam=%.9e\n", i, (double)g, (double)lam_prev[i]);
#endif
            }
        }
        // rate rows
        for (int r = 0; r < n; ++r) {
            for (int q = 0; q < 4; ++q) {
                if (!row_on(r, q)) continue;
                T s = M.ld(L.SR + 4 * r + q), y = M.ld(L.YR + 4 * r + q);
                T res = row_val(r, q) + s;
                e.rp = t_max(e.rp, t_abs(res));
                e.theta += t_abs(res);
                e.cmin = t_min(e.cmin, s * y);
                e.cmax = t_max(e.cmax, s * y);
                e.sum_bmult += y;
                e.n_bmult += 1;
                if (r > 0) rd_d -= slot_sign<T>(q) * P.rate_lim[q] * y;
            }
        }
        if (P.dt_free) {
            T pl = M.ld(L.PD), pu = M.ld(L.PD + 1);
            rd_d += -pl + pu;
            e.rd = t_max(e.rd, t_abs(rd_d));
            T cl = (d - P.dt_lb) * pl, cu = (P.dt_ub - d) * pu;
            e.cmin = t_min(e.cmin, t_min(cl, cu));
            e.cmax = t_max(e.cmax, t_max(cl, cu));
            e.sum_bmult += pl + pu;
            e.n_bmult += 2;
        }
        e.sum_mult += e.sum_bmult;
        e.n_mult += e.n_bmult;
        return e;
    }

    MPC_HD T err_value(const Err& e, T mu_t) const {
        T sd = t_max(Algo<T>::s_max, e.sum_mult / T(e.n_mult > 0 ? e.n_mult : 1)) / Algo<T>::s_max;
        T sc = t_max(Algo<T>::s_max, e.sum_bmult / T(e.n_bmult > 0 ? e.n_bmult : 1)) / Algo<T>::s_max;
        T comp = e.n_bmult > 0 ? t_max(e.cmax - mu_t, mu_t - e.cmin) : T(0);
        return t_max(e.rd / sd, t_max(e.rp, comp / sc));
    }

    // ---------------------------------------------------------------- backward Riccati sweep
    // returns false if a stage pivot is (numerically) singular
    MPC_HD bool backward(T delta, T dc, T& dd_out, T nu_out[3]) const {
        const int n = L.n;
        const T d = M.ld(L.D);
        RicState<T> V;
        T q2[3] = {T(0), T(0), T(0)}, r2[2] = {T(0), T(0)};
        if (P.objective == OBJ_QUADRATIC) { for (int i = 0; i < 3; ++i) q2[i] = T(2) * P.Q[i]; for (int j = 0; j < 2; ++j) r2[j] = T(2) * P.R[j]; }
        {   // terminal stage: xi = (x_{n-1}, u_{n-2}, dt); final rate rows against u_ref = 0
            T xd[3] = {X(L.X, n - 1, 0) - xf[0], X(L.X, n - 1, 1) - xf[1], normalize_theta(X(L.X, n - 1, 2) - xf[2])};
            T ss[2] = {T(0), T(0)}, sl[2] = {T(0), T(0)}, sll = T(0), gy[2] = {T(0), T(0)}, gyl = T(0);
            rate_terms(n - 1, d, ss, sl, sll, gy, gyl);
            riccati_terminal(V, P, xd, delta, dc, ss, sl, sll, gy, gyl);
        }
        for (int k = n - 2; k >= 0; --k) {
            T lam[3] = {M.ld(L.LAM + 3 * k), M.ld(L.LAM + 3 * k + 1), M.ld(L.LAM + 3 * k + 2)};
            T tr[4] = {M.ld(L.TRIG + 4 * k), M.ld(L.TRIG + 4 * k + 1), M.ld(L.TRIG + 4 * k + 2), M.ld(L.TRIG + 4 * k + 3)};
            T v = U(L.U, k, 0), w = U(L.U, k, 1);
            T f[3], G[3][3], Hq[3][3];
            model_derivs<T, MODEL>(P, tr, v, w, lam, f, G, Hq);
            // value rows of stage k+1 needed for lambda_k in the forward sweep
            {
                const int gb = L.GAIN + 50 * k + 20;
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 6; ++b) M.st(gb + 6 * a + b, V.P[a][b]);
                for (int a = 0; a < 3; ++a) M.st(gb + 18 + a, V.p[a]);
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M.st(gb + 21 + 3 * a + b, V.S[a][b]);
            }
            StageRec<T> r;
            r.a0 = d * G[0][0]; r.a1 = d * G[1][0];
            for (int a = 0; a < 3; ++a) { r.f[a] = f[a]; r.B[a][0] = d * G[a][1]; r.B[a][1] = d * G[a][2]; r.c[a] = M.ld(L.CC + 3 * k + a); }
            StageParts<T> sp;
            sp.hdd = T(0);
            sp.h00 = d * Hq[0][0]; sp.h01 = d * Hq[0][1]; sp.h02 = d * Hq[0][2]; sp.h11 = d * Hq[1][1]; sp.h12 = d * Hq[1][2]; sp.h22 = d * Hq[2][2];
            for (int j = 0; j < 3; ++j) sp.g[j] = lam[0] * G[0][j] + lam[1] * G[1][j] + lam[2] * G[2][j];
            for (int j = 0; j < 2; ++j) {
                T u = j == 0 ? v : w;
                T dl = u - P.u_lb[j], du = P.u_ub[j] - u;
                sp.sz[j] = M.ld(L.PL + 2 * k + j) / dl + M.ld(L.PU + 2 * k + j) / du;
                sp.gb[j] = -mu / dl + mu / du + r2[j] * u;
            }
            sp.ss[0] = sp.ss[1] = sp.sl[0] = sp.sl[1] = sp.sll = sp.gy[0] = sp.gy[1] = sp.gyl = T(0);
            rate_terms(k, d, sp.ss, sp.sl, sp.sll, sp.gy, sp.gyl);
            for (int i = 0; i < 3; ++i) sp.hx[i] = T(0);
            sp.oxx = sp.oxy = sp.oyy = sp.ogx = sp.ogy = T(0);
            if (P.objective == OBJ_QUADRATIC) {
                T xd[3] = {X(L.X, k, 0) - xf[0], X(L.X, k, 1) - xf[1], normalize_theta(X(L.X, k, 2) - xf[2])};
                for (int i = 0; i < 3; ++i) sp.hx[i] = q2[i] * xd[i];
            }
            assemble_adds(sp, q2, r2, r.A);
            T add_dd = T(0), add_qd = T(0);
            if (k == 0) {
                if (P.objective == OBJ_MIN_TIME) add_qd += T(n - 1);
                if (P.dt_free) {
                    T dl = d - P.dt_lb, du = P.dt_ub - d;
                    add_dd = M.ld(L.PD) / dl + M.ld(L.PD + 1) / du + delta;
                    add_qd += -mu / dl + mu / du;
                }
            }
            StageGain<T> g;
            if (!riccati_step(V, r, k >= 1 ? delta : T(0), delta, add_dd, add_qd, g)) return false;
            const int gb = L.GAIN + 50 * k;
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 6; ++b) M.st(gb + 6 * a + b, g.K[a][b]);
            M.st(gb + 12, g.kap[0]); M.st(gb + 13, g.kap[1]);
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) M.st(gb + 14 + 3 * a + b, g.Kn[a][b]);
        }
        return riccati_root(V, P, dd_out, nu_out);
    }

    // condensed barrier terms of the rate rows of index r (lim = 0 for r = 0: dt_prev is a constant there)
    MPC_HD void rate_terms(int r, T d, T ss[2], T sl[2], T& sll, T gy[2], T& gyl) const {
        const int n = L.n;
        for (int q = 0; q < 4; ++q) {
            if (!row_on(r, q)) continue;
            const int j = slot_comp(q);
            const T sg = slot_sign<T>(q), lim = r > 0 ? P.rate_lim[q] : T(0);
            T s = M.ld(L.SR + 4 * r + q), y = M.ld(L.YR + 4 * r + q);
            T sig = y / s;
            T ur = r < n - 1 ? U(L.U, r, j) : T(0);
            T um = r > 0 ? U(L.U, r - 1, j) : uprev[j];
            T ybar = mu / s + sig * (rate_g(r, q, ur, um, d) + s);
            ss[j] += sig; sl[j] += sig * lim; sll += sig * lim * lim;
            gy[j] += sg * ybar; gyl += sg * lim * ybar;
        }
    }

    // ---------------------------------------------------------------- forward sweep
    struct Fwd {
        T hdz, clam, dz2, dphi, a_p, a_d, dzmax, nunu;
        bool finite;
    };

    MPC_HD void ftb(T val, T dval, T tau, T& alpha) const {
        if (dval < T(0)) { T a = -tau * val / dval; if (a < alpha) alpha = a; }
    }

    MPC_HD Fwd forward(T dd, const T nu[3], T tau, T dc) const {
        const int n = L.n;
        Fwd o;
        o.hdz = T(0); o.clam = T(0); o.dz2 = T(0); o.dphi = T(0); o.a_p = T(1); o.a_d = T(1); o.dzmax = T(0); o.finite = true;
        o.nunu = T(0);
        for (int i = 0; i < 3; ++i) if (P.xf_fixed[i]) o.nunu += nu[i] * nu[i];
        const T d = M.ld(L.D);
        T xi[6] = {T(0), T(0), T(0), T(0), T(0), dd};
        M.st(L.DD, dd);
        for (int i = 0; i < 3; ++i) M.st(L.DX + i, T(0));
        if (P.dt_free) {
            T dl = d - P.dt_lb, du = P.dt_ub - d;
            T pl = M.ld(L.PD), pu = M.ld(L.PD + 1);
            T gb = -mu / dl + mu / du;
            o.hdz += gb * dd; o.dphi += gb * dd;
            ftb(dl, dd, tau, o.a_p); ftb(du, -dd, tau, o.a_p);
            ftb(pl, mu / dl - pl - (pl / dl) * dd, tau, o.a_d);
            ftb(pu, mu / du - pu + (pu / du) * dd, tau, o.a_d);
            o.dz2 += dd * dd;
            o.dzmax = t_max(o.dzmax, t_abs(dd));
        }
        if (P.objective == OBJ_MIN_TIME) { o.hdz += T(n - 1) * dd; o.dphi += T(n - 1) * dd; }
        for (int k = 0; k < n - 1; ++k) {
            const int gb = L.GAIN + 50 * k;
            T du_[2];
            for (int a = 0; a < 2; ++a) {
                T acc = M.ld(gb + 12 + a);
                for (int b = 0; b < 6; ++b) acc += M.ld(gb + 6 * a + b) * xi[b];
                for (int b = 0; b < 3; ++b) acc += M.ld(gb + 14 + 3 * a + b) * nu[b];
                du_[a] = -acc;
            }
            T tr[4] = {M.ld(L.TRIG + 4 * k), M.ld(L.TRIG + 4 * k + 1), M.ld(L.TRIG + 4 * k + 2), M.ld(L.TRIG + 4 * k + 3)};
            T v = U(L.U, k, 0), w = U(L.U, k, 1);
            T lz[3] = {T(0), T(0), T(0)};
            T f[3], G[3][3], Hq[3][3];
            model_derivs<T, MODEL>(P, tr, v, w, lz, f, G, Hq);
            T ck[3] = {M.ld(L.CC + 3 * k), M.ld(L.CC + 3 * k + 1), M.ld(L.CC + 3 * k + 2)};
            // objective / barrier gradient contributions of (x_k, u_k)
            T uu[2] = {v, w};
            if (P.objective == OBJ_QUADRATIC) {
                T w8 = P.integral_form ? d : T(1);
                T xd[3] = {X(L.X, k, 0) - xf[0], X(L.X, k, 1) - xf[1], normalize_theta(X(L.X, k, 2) - xf[2])};
                T sc = T(0), g = T(0);
                for (int i = 0; i < 3; ++i) { g += T(2) * P.Q[i] * xd[i] * w8 * xi[i]; sc += P.Q[i] * xd[i] * xd[i]; }
                for (int j = 0; j < 2; ++j) { g += T(2) * P.R[j] * uu[j] * w8 * du_[j]; sc += P.R[j] * uu[j] * uu[j]; }
                if (P.integral_form) g += sc * dd;
                o.hdz += g; o.dphi += g;
            }
            for (int j = 0; j < 2; ++j) {
                T dl = uu[j] - P.u_lb[j], du = P.u_ub[j] - uu[j];
                T pl = M.ld(L.PL + 2 * k + j), pu = M.ld(L.PU + 2 * k + j);
                T gbar = -mu / dl + mu / du;
                o.hdz += gbar * du_[j]; o.dphi += gbar * du_[j];
                ftb(dl, du_[j], tau, o.a_p); ftb(du, -du_[j], tau, o.a_p);
                ftb(pl, mu / dl - pl - (pl / dl) * du_[j], tau, o.a_d);
                ftb(pu, mu / du - pu + (pu / du) * du_[j], tau, o.a_d);
                o.dz2 += du_[j] * du_[j];
                o.dzmax = t_max(o.dzmax, t_abs(du_[j]));
                M.st(L.DU + 2 * k + j, du_[j]);
            }
            // rate rows of stage k: J dz = sg*((du_j - dup_j) - lim*dd[k>0])
            for (int q = 0; q < 4; ++q) {
                if (!row_on(k, q)) continue;
                const int j = slot_comp(q);
                const T sg = slot_sign<T>(q);
                T jdz = sg * ((du_[j] - xi[3 + j]) - (k > 0 ? P.rate_lim[q] * dd : T(0)));
                T s = M.ld(L.SR + 4 * k + q), y = M.ld(L.YR + 4 * k + q);
                T um = k > 0 ? U(L.U, k - 1, j) : uprev[j];
                T res = rate_g(k, q, uu[j], um, d) + s;
                T sig = y / s;
                T ybar = mu / s + sig * res;
                T ds = -res - jdz;
                T dy = ybar + sig * jdz - y;
                o.hdz += ybar * jdz;
                o.dphi -= (mu / s) * ds;
                ftb(s, ds, tau, o.a_p);
                ftb(y, dy, tau, o.a_d);
            }
            // next state
            T xn[6];
            for (int a = 0; a < 3; ++a)
                xn[a] = xi[a] + d * G[a][0] * xi[2] + d * (G[a][1] * du_[0] + G[a][2] * du_[1]) + f[a] * dd + ck[a];
            xn[3] = du_[0]; xn[4] = du_[1]; xn[5] = dd;
            // lambda_k = Px xi+ + px + Sx nu
            for (int a = 0; a < 3; ++a) {
                T acc = M.ld(gb + 20 + 18 + a);
                for (int b = 0; b < 6; ++b) acc += M.ld(gb + 20 + 6 * a + b) * xn[b];
                for (int b = 0; b < 3; ++b) acc += M.ld(gb + 20 + 21 + 3 * a + b) * nu[b];
                M.st(L.LAMN + 3 * k + a, acc);
                o.clam += ck[a] * acc;
                if (!t_finite(acc)) o.finite = false;
            }
            for (int a = 0; a < 6; ++a) xi[a] = xn[a];
            for (int a = 0; a < 3; ++a) {
                M.st(L.DX + 3 * (k + 1) + a, xi[a]);
                if (k + 1 < n - 1 || !P.xf_fixed[a]) { o.dz2 += xi[a] * xi[a]; o.dzmax = t_max(o.dzmax, t_abs(xi[a])); }
            }
        }
        // terminal: Qf gradient, final rate rows
        if (P.objective == OBJ_QUADRATIC && P.has_Qf) {
            for (int i = 0; i < 3; ++i) if (!P.xf_fixed[i]) {
                T xd = X(L.X, n - 1, i) - xf[i];
                if (i == 2) xd = normalize_theta(xd);
                T g = T(2) * P.Qf[i] * xd * xi[i];
                o.hdz += g; o.dphi += g;
            }
        }
        for (int q = 0; q < 4; ++q) {
            const int r = n - 1;
            if (!row_on(r, q)) continue;
            const int j = slot_comp(q);
            const T sg = slot_sign<T>(q);
            T jdz = sg * ((T(0) - xi[3 + j]) - P.rate_lim[q] * dd);
            T s = M.ld(L.SR + 4 * r + q), y = M.ld(L.YR + 4 * r + q);
            T res = row_val(r, q) + s;
            T sig = y / s;
            T ybar = mu / s + sig * res;
            T ds = -res - jdz;
            T dy = ybar + sig * jdz - y;
            o.hdz += ybar * jdz;
            o.dphi -= (mu / s) * ds;
            ftb(s, ds, tau, o.a_p);
            ftb(y, dy, tau, o.a_d);
        }
        if (!t_finite(o.hdz) || !t_finite(o.dz2)) o.finite = false;
        return o;
    }

    // ---------------------------------------------------------------- trial point
    MPC_HD void make_trial(T alpha) const {
        const int n = L.n;
        for (int k = 0; k < n; ++k) {
            for (int i = 0; i < 3; ++i) {
                T x = X(L.X, k, i);
                if (k > 0 && (k < n - 1 || !P.xf_fixed[i])) {
                    x += alpha * M.ld(L.DX + 3 * k + i);
                    if (i == 2) x = normalize_theta(x);
                }
                M.st(L.XT + 3 * k + i, x);
            }
        }
        for (int k = 0; k < n - 1; ++k)
            for (int j = 0; j < 2; ++j) M.st(L.UT + 2 * k + j, U(L.U, k, j) + alpha * M.ld(L.DU + 2 * k + j));
        M.st(L.DT, M.ld(L.D) + (P.dt_free ? alpha * M.ld(L.DD) : T(0)));
    }

    // ---------------------------------------------------------------- accept: duals, slacks, copy trial -> current
    MPC_HD void accept(T alpha, T a_d) const {
        const int n = L.n;
        const T kS = T(1e10);
        const T d_old = M.ld(L.D);
        // slacks and inequality multipliers first (they need the OLD point through row_val)
        for (int r = 0; r < n; ++r) {
            for (int q = 0; q < 4; ++q) {
                if (!row_on(r, q)) continue;
                T s = M.ld(L.SR + 4 * r + q), y = M.ld(L.YR + 4 * r + q);
                T res = row_val(r, q) + s;
                T jdz = row_jdz(r, q);
                T sig = y / s;
                T ds = -res - jdz;
                T dy = mu / s + sig * res + sig * jdz - y;
                T sn = s + alpha * ds;
                T yn = y + a_d * dy;
                yn = t_min(t_max(yn, mu / (kS * sn)), kS * mu / sn);
                M.st(L.SR + 4 * r + q, sn);
                M.st(L.YR + 4 * r + q, yn);
            }
        }
        for (int k = 0; k < n - 1; ++k) {
            for (int j = 0; j < 2; ++j) {
                T u = U(L.U, k, j), du_ = M.ld(L.DU + 2 * k + j);
                T dl = u - P.u_lb[j], du = P.u_ub[j] - u;
                T pl = M.ld(L.PL + 2 * k + j), pu = M.ld(L.PU + 2 * k + j);
                T pln = pl + a_d * (mu / dl - pl - (pl / dl) * du_);
                T pun = pu + a_d * (mu / du - pu + (pu / du) * du_);
                T un = U(L.UT, k, j);
                T dln = un - P.u_lb[j], dun = P.u_ub[j] - un;
                pln = t_min(t_max(pln, mu / (kS * dln)), kS * mu / dln);
                pun = t_min(t_max(pun, mu / (kS * dun)), kS * mu / dun);
                M.st(L.PL + 2 * k + j, pln);
                M.st(L.PU + 2 * k + j, pun);
                M.st(L.U + 2 * k + j, un);
            }
            for (int i = 0; i < 3; ++i) {
                T lo = M.ld(L.LAM + 3 * k + i);
                M.st(L.LAM + 3 * k + i, lo + alpha * (M.ld(L.LAMN + 3 * k + i) - lo));
            }
        }
        if (P.dt_free) {
            T dd = M.ld(L.DD);
            T dl = d_old - P.dt_lb, du = P.dt_ub - d_old;
            T pl = M.ld(L.PD), pu = M.ld(L.PD + 1);
            T pln = pl + a_d * (mu / dl - pl - (pl / dl) * dd);
            T pun = pu + a_d * (mu / du - pu + (pu / du) * dd);
            T dn = M.ld(L.DT);
            T dln = dn - P.dt_lb, dun = P.dt_ub - dn;
            pln = t_min(t_max(pln, mu / (kS * dln)), kS * mu / dln);
            pun = t_min(t_max(pun, mu / (kS * dun)), kS * mu / dun);
            M.st(L.PD, pln); M.st(L.PD + 1, pun);
        }
        M.st(L.D, M.ld(L.DT));
        for (int k = 0; k < n; ++k) for (int i = 0; i < 3; ++i) M.st(L.X + 3 * k + i, M.ld(L.XT + 3 * k + i));
    }

    // ---------------------------------------------------------------- driver
    // Expects X/U/D filled with the initial vertex values (or call cold_start() first).
    MPC_HD SolveStats<T> solve() {
        const int n = L.n;
        SolveStats<T> out;
        nfix = P.xf_fixed[0] + P.xf_fixed[1] + P.xf_fixed[2];
        row0_on = dtprev != T(0);
        // x_0 := measured state, fixed goal components := xf   (full_discretization_grid_base_se2.cpp:101-110)
        for (int i = 0; i < 3; ++i) {
            M.st(L.X + i, x0[i]);
            if (P.xf_fixed[i]) M.st(L.X + 3 * (n - 1) + i, xf[i]);
        }
        seed_controls_if_zero();
        for (int k = 0; k < n - 1; ++k)
            for (int j = 0; j < 2; ++j) M.st(L.U + 2 * k + j, push_interior(U(L.U, k, j), P.u_lb[j], P.u_ub[j]));
        if (P.dt_free) M.st(L.D, push_interior(M.ld(L.D), P.dt_lb, P.dt_ub));
        else M.st(L.D, P.dt_ref);
        mu = P.mu_init;
        rho = T(0);
        delta_last = T(0);
        fail0 = false;
        // duals / slacks
        for (int r = 0; r < n; ++r) {
            for (int q = 0; q < 4; ++q) {
                T s = T(1), y = T(0);
                if (row_on(r, q)) { s = t_max(-row_val(r, q), Algo<T>::slack_push); y = mu / s; }
                M.st(L.SR + 4 * r + q, s);
                M.st(L.YR + 4 * r + q, y);
            }
        }
        for (int k = 0; k < n - 1; ++k) {
            for (int j = 0; j < 2; ++j) {
                T u = U(L.U, k, j);
                M.st(L.PL + 2 * k + j, mu / (u - P.u_lb[j]));
                M.st(L.PU + 2 * k + j, mu / (P.u_ub[j] - u));
            }
            for (int i = 0; i < 3; ++i) M.st(L.LAM + 3 * k + i, T(0));
        }
        if (P.dt_free) { T d = M.ld(L.D); M.st(L.PD, mu / (d - P.dt_lb)); M.st(L.PD + 1, mu / (P.dt_ub - d)); }
        else { M.st(L.PD, T(0)); M.st(L.PD + 1, T(0)); }

        T theta_c, fobj, cinf;
        eval_point(L.X, L.U, L.D, L.TRIG, L.CC, theta_c, fobj, cinf);

        int it = 0;
        int status = ST_MAX_ITER;
        T e0 = T(0);
        while (true) {
            Err er = kkt_pass();
            e0 = err_value(er, T(0));
            if (!t_finite(e0)) { status = ST_NUMERICAL; break; }
            if (e0 <= P.tol) { status = ST_CONVERGED; break; }
            if (it >= P.max_iter) { status = ST_MAX_ITER; break; }
            // monotone barrier update (Waechter & Biegler eq. 7)
            for (int guard = 0; guard < 50; ++guard) {
                T emu = err_value(er, mu);
                if (emu <= Algo<T>::kappa_eps * mu && mu > P.tol / T(10)) {
                    mu = t_max(P.tol / T(10), t_min(Algo<T>::kappa_mu * mu, t_pow(mu, Algo<T>::theta_mu)));
                    rho = T(0);
                } else break;
            }
            const T tau = t_max(Algo<T>::tau_min, T(1) - mu);
            const T dc = nfix > 0 ? Algo<T>::delta_c * t_pow(mu, Algo<T>::kappa_c) : T(0);
            // ---- factor/solve with inertia-free regularisation
            // first trial delta = 0 unless the previous iteration's delta = 0 attempt failed (then continue from the decayed value)
            T delta = (fail0 && delta_last > T(0)) ? t_max(Algo<T>::delta_min, Algo<T>::kappa_minus * delta_last) : T(0);
            const bool started_zero = delta == T(0);
            bool ok = false;
            Fwd fw;
            T dd = T(0), nu[3] = {T(0), T(0), T(0)};
            T curv = T(0);
            for (int ntry = 0; ntry <= 40; ++ntry) {
                bool good = backward(delta, dc, dd, nu);
                if (good) {
                    fw = forward(dd, nu, tau, dc);
                    good = fw.finite;
                    if (good) {
                        curv = -fw.hdz + fw.clam - dc * fw.nunu;     // = dz^T (H + delta I) dz
                        if (curv >= Algo<T>::curv_kappa * fw.dz2) { ok = true; break; }
                    }
                }
                if (delta == T(0)) delta = (delta_last == T(0)) ? Algo<T>::delta_first : t_max(Algo<T>::delta_min, Algo<T>::kappa_minus * delta_last);
                else delta *= (delta_last == T(0)) ? Algo<T>::kappa_plus_first : Algo<T>::kappa_plus;
                if (delta > Algo<T>::delta_max) break;
            }
            if (!ok) { status = ST_LINSOLVE; break; }
            if (delta > T(0)) delta_last = delta;
            if (started_zero) fail0 = delta > T(0);
            // ---- l1 merit, backtracking
            const T theta = er.theta;
            if (theta > T(0)) {
                T sigma = curv > T(0) ? T(1) : T(0);
                T rho_trial = (fw.dphi + T(0.5) * sigma * curv) / ((T(1) - Algo<T>::rho_frac) * theta);
                if (rho < rho_trial) rho = rho_trial + T(1);
            }
            const T phi0 = fobj - mu * barrier_logs(L.U, L.D, T(0), false) + rho * theta;
            const T Dm = fw.dphi - rho * theta;
            const T theta_rows = theta - theta_c;      // linear rows: scales with (1 - alpha)
            T alpha = fw.a_p;
            bool accepted = false;
            T th_t = T(0), f_t = T(0), cinf_t = T(0);
            for (int ls = 0; ls < Algo<T>::max_ls; ++ls) {
                if (ls > 0) alpha *= T(0.5);
                make_trial(alpha);
                eval_point(L.XT, L.UT, L.DT, L.TRIG, L.CC, th_t, f_t, cinf_t);   // overwrites the caches of the current point
                T tht = th_t + (T(1) - alpha) * theta_rows;
                T phit = f_t - mu * barrier_logs(L.UT, L.DT, alpha, true) + rho * tht;
                // round-off relaxed Armijo test (Waechter & Biegler 2006, sec. 3.3: 10*eps_mach*|phi|)
                if (t_finite(phit) && phit - phi0 - Algo<T>::ls_eps * t_abs(phi0) <= Algo<T>::eta_armijo * alpha * Dm) { accepted = true; break; }
            }
            if (!accepted && alpha * fw.dzmax < T(1e-14)) {
                // restore caches of the current point before leaving
                eval_point(L.X, L.U, L.D, L.TRIG, L.CC, theta_c, fobj, cinf);
                status = ST_LINESEARCH;
                break;
            }
#ifdef MPC_TRACE
            if (MPC_TRACE_COND) printf("it %d e0 %.3e mu %.1e theta %.3e alpha %.4f a_d %.4f delta %.1e curv %.3e dz2 %.3e dphi %.3e rho %.2e acc %d phi0 %.17g | rd %.3e rp %.3e cmin %.3e cmax %.3e sm %.3e sb %.3e nm %d nb %d\n",
                                       it, (double)e0, (double)mu, (double)theta, (double)alpha, (double)fw.a_d, (double)delta, (double)curv, (double)fw.dz2, (double)fw.dphi, (double)rho, (int)accepted, (double)phi0, (double)er.rd, (double)er.rp, (double)er.cmin, (double)er.cmax, (double)er.sum_mult, (double)er.sum_bmult, er.n_mult, er.n_bmult);
#endif
            accept(alpha, fw.a_d);
            theta_c = th_t; fobj = f_t; cinf = cinf_t;
            ++it;
        }
        out.status = status;
        out.iters = it;
        out.kkt_error = e0;
        out.objective = fobj;
        return out;
    }
};

}  // namespace mpc

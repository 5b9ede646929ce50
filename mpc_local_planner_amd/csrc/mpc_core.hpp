// mpc_core.hpp -- arithmetic shared by the interior-point solve of the mpc_local_planner NLP: problem record, algorithm
// constants, math helpers (wrap, fast sincos / tan / log kernels), dynamics + collocation maps with their derivatives, the
// stage-structured (Riccati) step and its root solve.  The solver itself is mpc_wave.hpp (one wavefront per planner instance).
// A serial restatement that drives the same arithmetic one instance at a time lives in tests/host_harness/ipm_serial.hpp
// (tests only: it lets the formulas here be checked against the oracle without a GPU).
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
constexpr int ST_TIME_LIMIT = 5;      // MPC_TIME_LIMIT
constexpr int ST_SUPERSEDED = 1000;  // internal: a candidate stopped because a higher-priority candidate of its instance converged (never returned; outside the range of enum mpc_status)
static_assert(ST_SUPERSEDED != ST_CONVERGED && ST_SUPERSEDED != ST_MAX_ITER && ST_SUPERSEDED != ST_LINESEARCH && ST_SUPERSEDED != ST_LINSOLVE && ST_SUPERSEDED != ST_NUMERICAL &&
              ST_SUPERSEDED != ST_TIME_LIMIT, "the internal status must differ from every status a caller can see");

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
    T cand_param[4];     // tangent scale of the Hermite kinds
    int hess_mode;       // 0 exact Lagrangian Hessian, 1 convexified (stage-wise positive semidefinite part; EXT kernel instantiation)
    T mu_init_dual;      // barrier start of a solve that starts from the multipliers kept in the handle (dual_warm_start)
    // cost variants (EXT kernel instantiation, except `hybrid`): off-diagonal terms (01, 02, 12) of full weight matrices, trapezoidal rule for the
    // integral-form cost on the variable grid, minimum time added to the quadratic form
    T Qo[3], Ro, Qfo[3], So[3];
    int trapz, hybrid, costx;      // costx: any of Qo / Ro / Qfo / So non-zero, or trapz
    // Ipopt's acceptable-level stop (mpc_config.acceptable_tol / acceptable_iter): level (0 = rule off) and iterations in a row (0 = counting half off)
    T acc_tol;
    int acc_iter;
    int line_search;     // mpc_config.line_search resolved: 0 l1 merit, 1 Ipopt's filter line search (sits in the padding after acc_iter in the fp64 form)
    T pit_mu_min;        // ... while the barrier parameter is above this (the last iterations of a solve take the serial sweeps: see DESIGN.md)
    int pit;             // partitioned (parallel-in-time) sweeps for grids of 40 points and more (1; 0 = the serial sweeps everywhere: developer switch MPC_NO_PIT)
    int mu_strategy;     // mpc_config.mu_strategy: 0 adaptive barrier parameter (the default), 1 monotone Fiacco-McCormick
    long long max_ticks; // mpc_config.max_time_us in ticks of the device's constant 100 MHz clock (wall_clock64); 0 = no limit
};

// Algorithm constants (Waechter & Biegler 2006 names).  Compile-time so that they live in
// instruction immediates instead of scalar registers.
#ifndef MPC_ELASTIC_TRIGGER
#define MPC_ELASTIC_TRIGGER 5      // developer builds: a huge value switches the restoration mode off (A/B against the C oracle's experiment switch)
#endif
template <typename T> struct Algo;
template <> struct Algo<double> {
    static constexpr double kappa_eps = 10, kappa_mu = 0.2, theta_mu = 1.5, tau_min = 0.99, bound_push = 1e-2, slack_push = 1e-2;
    static constexpr double eta_armijo = 1e-4, rho_frac = 0.1, delta_first = 1e-4, delta_min = 1e-20, delta_max = 1e20;
    static constexpr double kappa_plus = 8, kappa_plus_first = 100, kappa_minus = 1.0 / 3.0;
    static constexpr double curv_kappa = 1e-10, s_max = 100, delta_c = 1e-8, kappa_c = 0.25, ls_eps = 10 * 2.220446049250313e-16;
    static constexpr int max_ls = 30;
    // Ipopt's filter line search (Waechter & Biegler 2006, Algorithm A; the constants of section 3.4 / Ipopt's defaults)
    static constexpr double flt_gth = 1e-5, flt_gph = 1e-8, flt_sph = 2.3, flt_sth = 1.1, flt_eta = 1e-8, flt_delta = 1.0, flt_gal = 0.05, flt_thmax = 1e4, flt_thmin = 1e-4;
    static constexpr int flt_cap = 16;
    // initial slack of a CLEARANCE row: max(-g, 0.5) (metres of clearance) instead of the 1e-2 of the linear rows.  With 1e-2 a row that starts active or
    // violated pins the fraction-to-boundary rule from the first iteration on (steps of 1e-3) and, without Ipopt's restoration phase, the solve never
    // leaves that corner: obstacles inside the clearance band converge in 81 % of the instances with 1e-2 and in 98 % with 0.5
    static constexpr double clearance_slack_push = 0.5;
    // restoration for clearance rows that jam (mpc_wave.hpp::solve; same constants in the two CPU restatements the tests check against): penalty of the elastic variables, the
    // primal step limit below which an iteration counts as jammed, the share of the streak's initial infeasibility that has to be left, the length of the streak
    static constexpr double elastic_rho = 1000.0, elastic_ap = 5e-2, elastic_prog = 0.8;
    static constexpr int elastic_trigger = MPC_ELASTIC_TRIGGER;
    // adaptive barrier parameter (mpc_wave.hpp::solve): sigma = clamp((1 - min(alpha, alpha_dual))^3, sigma_min, 1) from the last iteration's step lengths,
    // mu = sigma x average complementarity, never below min(mu, mu_err_floor x E_0), inside [tol / 10, mu_max_fact x the solve's first mu]
    static constexpr double sigma_min = 0.05, mu_err_floor = 3e-2, mu_max_fact = 1e3;
    // control seed of a cold start: increments kept inside this fraction of the control-rate limits
    static constexpr double rate_seed_frac = 0.9;
};
template <> struct Algo<float> {
    static constexpr float kappa_eps = 10, kappa_mu = 0.2f, theta_mu = 1.5f, tau_min = 0.99f, bound_push = 1e-2f, slack_push = 1e-2f;
    static constexpr float eta_armijo = 1e-4f, rho_frac = 0.1f, delta_first = 1e-4f, delta_min = 1e-12f, delta_max = 1e12f;
    static constexpr float kappa_plus = 8, kappa_plus_first = 100, kappa_minus = 1.0f / 3.0f;
    static constexpr float curv_kappa = 1e-7f, s_max = 100, delta_c = 1e-5f, kappa_c = 0.25f, ls_eps = 10 * 1.1920929e-7f;
    static constexpr int max_ls = 30;
    static constexpr float flt_gth = 1e-5f, flt_gph = 1e-8f, flt_sph = 2.3f, flt_sth = 1.1f, flt_eta = 1e-8f, flt_delta = 1.0f, flt_gal = 0.05f, flt_thmax = 1e4f, flt_thmin = 1e-4f;
    static constexpr int flt_cap = 16;
    static constexpr float clearance_slack_push = 0.5f;
    static constexpr float elastic_rho = 1000.0f, elastic_ap = 5e-2f, elastic_prog = 0.8f;
    static constexpr int elastic_trigger = MPC_ELASTIC_TRIGGER;
    static constexpr float sigma_min = 0.05f, mu_err_floor = 3e-2f, mu_max_fact = 1e3f;
    static constexpr float rate_seed_frac = 0.9f;
};

MPC_HD double t_abs(double a) { return __builtin_fabs(a); }      // a source modifier on the GPU (the compare-and-select form costs 3 instructions)
MPC_HD float t_abs(float a) { return __builtin_fabsf(a); }
MPC_HD double t_fmin(double a, double b) { return __builtin_fmin(a, b); }   // IEEE minNum: one instruction, drops a NaN operand
MPC_HD float t_fmin(float a, float b) { return __builtin_fminf(a, b); }
template <typename T> MPC_HD T t_max(T a, T b) { return a > b ? a : b; }
template <typename T> MPC_HD T t_min(T a, T b) { return a < b ? a : b; }
// floating point: IEEE maxNum / minNum -- ONE instruction (v_max_f64) where the compare-and-select form is a compare, a wait state and two v_cndmask per fp64 value.
// Same value for ordered operands; a NaN operand is dropped (the select form kept a NaN in b and dropped one in a): non-finite iterates are caught through the sums
// (theta, the step norms, the multiplier sums), which propagate them.
MPC_HD double t_max(double a, double b) { return __builtin_fmax(a, b); }
MPC_HD double t_min(double a, double b) { return __builtin_fmin(a, b); }
MPC_HD float t_max(float a, float b) { return __builtin_fmaxf(a, b); }
MPC_HD float t_min(float a, float b) { return __builtin_fminf(a, b); }
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
// hardware reciprocal seed alone (~1e-8 relative in fp64; callers add the Newton steps they need)
MPC_HD double t_rcp_approx(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(x);
#else
    return 1.0 / x;
#endif
}
MPC_HD float t_rcp_approx(float x) { return 1.0f / x; }
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
#if defined(__HIP_DEVICE_COMPILE__)
    T m = t_floor(th * T(0.15915494309189533577));      // (a multiple count that is off by one at an exact multiple of 2 pi is put right by the two corrections below; no IEEE division)
#else
    T m = t_floor(th / (T(2) * pi));
#endif
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
// MPC_HESSIAN_CONVEXIFIED: the stage block [Hqq Hqd; Hqd' Hdd] of the Lagrangian curvature lam' D over (theta, v, w, dt) is replaced by its
// positive semidefinite part (cyclic Jacobi eigen-decomposition of the 4 x 4 block, negative eigenvalues set to 0); drop_theta: the heading
// of this stage is not a variable (stage 0), its row / column is removed first.  The C restatement used by the tests runs the same arithmetic.
template <typename T>
MPC_HD void psd_project4(StageMap<T>& sm, bool drop_theta) {
    T A[4][4], V[4][4];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) A[i][j] = sm.Hqq[i][j]; A[i][3] = A[3][i] = sm.Hqd[i]; }
    A[3][3] = sm.Hdd;
    if (drop_theta) for (int j = 0; j < 4; ++j) A[0][j] = A[j][0] = T(0);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = i == j ? T(1) : T(0);
    for (int sweep = 0; sweep < 30; ++sweep) {
        T off = T(0);
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
        if (off < T(1e-30)) break;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) {
            if (!(t_abs(A[p][q]) >= T(1e-300))) continue;      // (fp32: the constant rounds to 0, exact zeros are skipped)
            if (A[p][q] == T(0)) continue;
            const T th = (A[q][q] - A[p][p]) / (T(2) * A[p][q]);
            const T t = (th >= T(0) ? T(1) : T(-1)) / (t_abs(th) + sqrt(th * th + T(1))), c = T(1) / sqrt(t * t + T(1)), sn = t * c;
            for (int k = 0; k < 4; ++k) { const T akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
            for (int k = 0; k < 4; ++k) { const T apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
            for (int k = 0; k < 4; ++k) { const T vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
        }
    }
    T R[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { T v = T(0); for (int k = 0; k < 4; ++k) v += V[i][k] * (A[k][k] > T(0) ? A[k][k] : T(0)) * V[j][k]; R[i][j] = v; }
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) sm.Hqq[i][j] = R[i][j]; sm.Hqd[i] = R[i][3]; }
    sm.Hdd = R[3][3];
}

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
    int neg;          // negative eigenvalues of the control pivots R_k eliminated so far (inertia of the factorisation, riccati_root)
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
    V.neg += det < T(0) ? 1 : (R00 < T(0) ? 2 : 0);      // sign changes of (1, R00, det): the negative eigenvalues of this pivot
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
    V.neg = 0;
    for (int i = 0; i < 3; ++i) {
        if (P_.xf_fixed[i]) { V.S[i][i] = T(1); V.W[i][i] = -dc; }
        else {
            V.P[i][i] = delta;
            if (P_.has_Qf) { V.P[i][i] += T(2) * P_.Qf[i]; V.p[i] = T(2) * P_.Qf[i] * xd_f[i]; }
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
// returns 1: solved, the factorisation has the right inertia; -1: wrong inertia (the caller raises delta_w); 0: a pivot broke down / non-finite values
MPC_HD int riccati_root(const RicState<T>& V, const Problem<T>& P_, T& dd_out, T nu_out[3]) {
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
    // ---- inertia (r04): Ipopt accepts a factorisation when the KKT matrix has as many positive eigenvalues as primal variables and as many negative ones as
    // equality rows, and raises delta_w otherwise.  By Sylvester's law the inertia is the sum over the pivot blocks of ANY symmetric block elimination (Haynsworth): the
    // sweep eliminates, per stage, the pair (x_{k+1}, lambda_k) -- pivot [[H, -I], [-I, 0]], three positive and three negative eigenvalues whatever H is -- and the
    // control u_k with the 2 x 2 pivot R_k (V.neg counts its negative eigenvalues); what is left is this symmetric system over (dt, nu): one negative eigenvalue per FIXED
    // final component is what the inertia asks for.  Counted by Jacobi's signature rule -- the negative pivots of the elimination WITHOUT exchanges, order nu then dt; the
    // rows of free components / a fixed dt are identity rows (pivot +1) -- and a vanishing pivot counts as a failed factorisation.  The test replaces the inertia-free
    // curvature test of r01-r03, which lets Newton's iteration converge to saddle points (18 % of the config-2 answers).
    {
        T B4[4][4];
        T scl = T(0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) { B4[a][b] = A4[(a + 1) & 3][(b + 1) & 3]; scl = t_max(scl, t_abs(B4[a][b])); }      // order nu_0 nu_1 nu_2 dt
        int neg = V.neg, want = 0;
        bool okp = t_finite(scl);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const T pv = B4[c][c];
            // a pivot counts as vanished against the scale of ITS OWN row (r05), not against the largest entry of the system: with active clearance rows the dt-dt entry reaches
            // 1e11 while the nu block is -1e-3 .. -delta_c -- a legitimate, badly scaled system that the test against the global scale rejected for every delta_w (such a solve
            // ended with MPC_LINSOLVE in its last iterations, where the banded-LU oracle converges: 6 of 64 instances with point obstacles 0.05 .. 0.5 m beside the path)
            T rs = t_abs(pv);
#pragma unroll
            for (int b = c + 1; b < 4; ++b) rs = t_max(rs, t_abs(B4[c][b]));
            okp = okp && (t_abs(pv) > T(1e-14) * rs) && (rs > T(0));
            neg += pv < T(0) ? 1 : 0;
            const T ip = t_rcp(pv);
            okp = okp && t_finite(ip);      // (a denormal pivot whose reciprocal overflows passes the row-relative test when its row holds nothing else: ADVICE r05)
#pragma unroll
            for (int r = c + 1; r < 4; ++r) {
                const T m = B4[r][c] * ip;
#pragma unroll
                for (int b = c + 1; b < 4; ++b) B4[r][b] -= m * B4[c][b];
            }
        }
        for (int a = 0; a < 3; ++a) want += P_.xf_fixed[a] ? 1 : 0;
        if (!okp) return 0;
        if (neg != want) return -1;
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
    if (!ok) return 0;
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
    return (t_finite(sol[0]) && t_finite(sol[1]) && t_finite(sol[2]) && t_finite(sol[3])) ? 1 : 0;
}

// Excess of negative eigenvalues of the pivot block of a combine step of the partitioned sweep (mpc_wave.hpp::backward_pit): n-(W) + n-(P+ - W^-1) - 5 for symmetric 5 x 5
// matrices given by their upper triangles, index u(a, b) = a (9 - a) / 2 + b for a <= b.  The block [[W, -I], [-I, P+]] over the five components with a costate column has the
// inertia In(W) + In(P+ - W^-1) (Haynsworth), five negative eigenvalues when all is well.  Jacobi's signature rule: the negative eigenvalues of a symmetric matrix are the negative
// pivots of its elimination without exchanges.  W is swept in place -- the symmetric sweep operator shows the same pivots as the elimination and leaves -W^-1 --, then
// g = P+ + (-W^-1) is eliminated.  One Newton step per reciprocal is plenty for signs.  ok = false when a pivot vanishes (both arrays are destroyed).
template <typename T>
MPC_HD int pit_block_inertia_tri(T (&w)[15], T (&g)[15], bool& ok) {
    T scl = T(0);
#pragma unroll
    for (int u = 0; u < 15; ++u) scl = t_max(scl, t_abs(w[u]));
    auto rcp1 = [](T x) { T r = t_rcp_approx(x); return r + r * (T(1) - x * r); };
    int neg = -5;
    bool okp = t_finite(scl) && scl > T(0);
#pragma unroll
    for (int k = 0; k < 5; ++k) {                 // symmetric sweep on pivot k: the pivot is the diagonal of the Schur complement; after five sweeps w = -W^-1
        const T d = w[k * (9 - k) / 2 + k];
        okp = okp && (t_abs(d) > T(1e-13) * scl);
        neg += d < T(0) ? 1 : 0;
        const T id = rcp1(d);
        T col[5], tc[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) { col[i] = i <= k ? w[i * (9 - i) / 2 + k] : w[k * (9 - k) / 2 + i]; tc[i] = col[i] * id; }
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = i; j < 5; ++j) if (i != k && j != k) w[i * (9 - i) / 2 + j] -= tc[i] * col[j];
#pragma unroll
        for (int i = 0; i < 5; ++i) if (i != k) { if (i < k) w[i * (9 - i) / 2 + k] = tc[i]; else w[k * (9 - k) / 2 + i] = tc[i]; }
        w[k * (9 - k) / 2 + k] = -id;
    }
    T gs = T(0);
#pragma unroll
    for (int u = 0; u < 15; ++u) { g[u] += w[u]; gs = t_max(gs, t_abs(g[u])); }
    okp = okp && t_finite(gs) && gs > T(0);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const T d = g[k * (9 - k) / 2 + k];
        okp = okp && (t_abs(d) > T(1e-13) * gs);
        neg += d < T(0) ? 1 : 0;
        const T id = rcp1(d);
#pragma unroll
        for (int i = k + 1; i < 5; ++i) {
            const T m = g[k * (9 - k) / 2 + i] * id;
#pragma unroll
            for (int j = i; j < 5; ++j) g[i * (9 - i) / 2 + j] -= m * g[k * (9 - k) / 2 + j];
        }
    }
    ok = ok && okp;
    return neg;
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

}  // namespace mpc

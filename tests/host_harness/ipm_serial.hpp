// ipm_serial.hpp -- DEVELOPER HARNESS (tests only).  Serial, one-instance-at-a-time driver of the arithmetic in
// mpc_local_planner_amd/csrc/mpc_core.hpp (same interior-point iteration, same Riccati step) over a flat workspace.  It was the
// first GPU kernel of this repository (one lane per instance); as a GPU kernel it was retired in favour of mpc_wave.hpp, and it
// is kept here only so that tests/test_host_core.py can exercise mpc_core.hpp's formulas against the oracle on the host.
// Nothing in mpc_local_planner_amd/ includes this file.
#pragma once
#include "../../mpc_local_planner_amd/csrc/mpc_core.hpp"

namespace mpc {

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
        // rate-feasible seed (as mpc_wave.hpp::init_point): increments clamped to rate_seed_frac x the rate limits forward from u_prev, then backward from the final row
        const T fr = Algo<T>::rate_seed_frac;
        for (int j = 0; j < 2; ++j) {
            if (!P.rate_on[j] || !P.rate_on[2 + j]) continue;
            const T lo = P.rate_lim[j] * d * fr, hi = P.rate_lim[2 + j] * d * fr;
            T prev = U(L.U, 0, j);
            if (dtprev != T(0)) { prev = t_min(t_max(prev, uprev[j] + P.rate_lim[j] * dtprev * fr), uprev[j] + P.rate_lim[2 + j] * dtprev * fr); M.st(L.U + j, prev); }
            for (int k = 1; k < n - 1; ++k) { prev = t_min(t_max(U(L.U, k, j), prev + lo), prev + hi); M.st(L.U + 2 * k + j, prev); }
            T nxt = T(0);
            for (int k = n - 2; k >= 0; --k) { nxt = t_min(t_max(U(L.U, k, j), nxt - hi), nxt - lo); M.st(L.U + 2 * k + j, nxt); }
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
        if (P.has_Qf) {
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
        T rd, rp, cmin, cmax, csum, sum_mult, sum_bmult;     // csum: sum of the n_bmult complementarity products
        int n_mult, n_bmult;
        T theta;   // l1 constraint violation
    };

    MPC_HD Err kkt_pass() const {
        const int n = L.n;
        Err e;
        e.rd = T(0); e.rp = T(0); e.cmin = T(1e30); e.cmax = T(0); e.csum = T(0); e.sum_mult = T(0); e.sum_bmult = T(0);
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
                e.csum += cl + cu;
                e.sum_bmult += pl + pu;
                e.n_bmult += 2;
            }
            lam_prev[0] = lam[0]; lam_prev[1] = lam[1]; lam_prev[2] = lam[2];
        }
        // free terminal components
        for (int i = 0; i < 3; ++i) {
            if (!P.xf_fixed[i]) {
                T g = T(0);
                if (P.has_Qf) {
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
                e.csum += s * y;
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
            e.csum += cl + cu;
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
    // returns false if a stage pivot is (numerically) singular or the factorisation has the wrong inertia
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
        return riccati_root(V, P, dd_out, nu_out) > 0;
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
        if (P.has_Qf) {
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

        int it = 0, n_acc = 0;
        T f_th[Algo<T>::flt_cap], f_ph[Algo<T>::flt_cap], theta0 = T(-1), mu_filter = T(-1);      // Ipopt's filter (mpc_config.line_search): a ring of (theta, phi) pairs
        int nfilt = 0, fpos = 0;
        int status = ST_MAX_ITER;
        T e0 = T(0), last_alpha = T(0), last_ad = T(0);
        bool endgame = false;
        const T mu_max = Algo<T>::mu_max_fact * mu;
        while (true) {
            Err er = kkt_pass();
            e0 = err_value(er, T(0));
            if (!t_finite(e0) || !t_finite(er.theta) || !t_finite(er.sum_mult) || !t_finite(er.csum)) { status = ST_NUMERICAL; break; }
            if (e0 <= P.tol) { status = ST_CONVERGED; break; }
            n_acc = (P.acc_iter > 0 && e0 <= P.acc_tol) ? n_acc + 1 : 0;       // Ipopt's acceptable-level stop (counting half), as mpc_wave.hpp
            if (P.acc_iter > 0 && n_acc >= P.acc_iter) { status = ST_CONVERGED; break; }
            if (it >= P.max_iter) { status = ST_MAX_ITER; break; }
            if (P.mu_strategy == 1 || endgame) {
                // monotone barrier update (Waechter & Biegler eq. 7)
                for (int guard = 0; guard < 50; ++guard) {
                    T emu = err_value(er, mu);
                    if (emu <= Algo<T>::kappa_eps * mu && mu > P.tol / T(10)) {
                        mu = t_max(P.tol / T(10), t_min(Algo<T>::kappa_mu * mu, t_pow(mu, Algo<T>::theta_mu)));
                        rho = T(0);
                    } else break;
                }
            } else if (it > 0) {
                // adaptive barrier update, as mpc_wave.hpp::solve (the first iteration keeps the start value)
                const T avg = er.csum / T(er.n_bmult > 0 ? er.n_bmult : 1);
                const T a_ = T(1) - t_min(last_alpha, last_ad);
                const T sig = t_min(t_max(a_ * a_ * a_, Algo<T>::sigma_min), T(1));
                T mu_new = t_min(t_max(sig * avg, P.tol / T(10)), mu_max);
                mu_new = t_max(mu_new, t_min(mu, Algo<T>::mu_err_floor * e0));
                if (mu_new <= P.tol) { mu_new = P.tol; endgame = true; }      // end game: from mu = tol on the monotone rule takes over (tol -> tol / 10 once the barrier problem is solved to kappa_eps mu): a solve stops at a point of the central path, as with the monotone strategy
                if (mu_new != mu) { mu = mu_new; rho = T(0); }
            }
            const T tau = t_max(Algo<T>::tau_min, T(1) - mu);
            const T dc = nfix > 0 ? Algo<T>::delta_c * t_pow(mu, Algo<T>::kappa_c) : T(0);
            // ---- factor/solve, delta_w raised until the factorisation has Ipopt's inertia
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
                    if (good) {      // backward() has checked the inertia of this factorisation (mpc_core.hpp::riccati_root); the curvature only feeds the penalty update
                        curv = -fw.hdz + fw.clam - dc * fw.nunu;     // = dz^T (H + delta I) dz
                        ok = true; break;
                    }
                }
                if (delta == T(0)) delta = (delta_last == T(0)) ? Algo<T>::delta_first : t_max(Algo<T>::delta_min, Algo<T>::kappa_minus * delta_last);
                else delta *= (delta_last == T(0)) ? Algo<T>::kappa_plus_first : Algo<T>::kappa_plus;
                if (delta > Algo<T>::delta_max) break;
            }
            if (!ok) { status = ST_LINSOLVE; break; }
            if (delta > T(0)) delta_last = delta;
            if (started_zero) fail0 = delta > T(0);
            // ---- l1 merit backtracking, or Ipopt's filter line search (mpc_config.line_search; the statements of mpc_wave_solve.inc)
            const bool filter = P.line_search == 1;
            const T theta = er.theta;
            if (!filter && theta > T(0)) {
                T sigma = curv > T(0) ? T(1) : T(0);
                T rho_trial = (fw.dphi + T(0.5) * sigma * curv) / ((T(1) - Algo<T>::rho_frac) * theta);
                if (rho < rho_trial) rho = rho_trial + T(1);
            }
            const T rho_ls = filter ? T(0) : rho;
            const T phi0 = fobj - mu * barrier_logs(L.U, L.D, T(0), false) + rho_ls * theta;
            const T Dm = fw.dphi - rho_ls * theta;
            const T theta_rows = theta - theta_c;      // linear rows: scales with (1 - alpha)
            T alpha = fw.a_p;
            T a_min = T(0), p_ph = T(0), p_th = T(0), theta_max = T(0), theta_min = T(0);
            if (filter) {
                if (theta0 < T(0)) theta0 = theta;
                if (mu != mu_filter) { nfilt = 0; fpos = 0; mu_filter = mu; }
                theta_max = Algo<T>::flt_thmax * t_max(T(1), theta0); theta_min = Algo<T>::flt_thmin * t_max(T(1), theta0);
                a_min = Algo<T>::flt_gth;
                if (fw.dphi < T(0)) {
                    a_min = t_min(a_min, Algo<T>::flt_gph * theta / (-fw.dphi));
                    if (theta <= theta_min) {      // (the powers are only consulted at such a point)
                        p_ph = t_pow(-fw.dphi, Algo<T>::flt_sph); p_th = t_pow(theta, Algo<T>::flt_sth);
                        a_min = t_min(a_min, Algo<T>::flt_delta * p_th / p_ph);
                    }
                }
                a_min = a_min * Algo<T>::flt_gal * fw.a_p;
            }
            bool accepted = false, sw_arm = false;
            T th_t = T(0), f_t = T(0), cinf_t = T(0), lg_t = T(0);
            for (int ls = 0; ls < Algo<T>::max_ls; ++ls) {
                if (ls > 0) alpha *= T(0.5);
                const bool below = filter && ls > 0 && alpha < a_min;
                make_trial(alpha);
                eval_point(L.XT, L.UT, L.DT, L.TRIG, L.CC, th_t, f_t, cinf_t);   // overwrites the caches of the current point
                T tht = th_t + (T(1) - alpha) * theta_rows;
                lg_t = barrier_logs(L.UT, L.DT, alpha, true);
                T phit = f_t - mu * lg_t + rho_ls * tht;
                if (!filter) {
                    // round-off relaxed Armijo test (Waechter & Biegler 2006, sec. 3.3: 10*eps_mach*|phi|)
                    if (t_finite(phit) && phit - phi0 - Algo<T>::ls_eps * t_abs(phi0) <= Algo<T>::eta_armijo * alpha * Dm) { accepted = true; break; }
                } else {
                    if (below) break;
                    bool okf = t_finite(phit) && tht <= theta_max;
                    for (int f = 0; f < nfilt && okf; ++f) if (!(tht <= (T(1) - Algo<T>::flt_gth) * f_th[f] || phit <= f_ph[f] - Algo<T>::flt_gph * f_th[f])) okf = false;
                    if (okf) {
                        const bool switching = fw.dphi < T(0) && alpha * p_ph > Algo<T>::flt_delta * p_th;
                        if (theta <= theta_min && switching) {
                            if (phit - phi0 - Algo<T>::ls_eps * t_abs(phi0) <= Algo<T>::flt_eta * alpha * fw.dphi) { accepted = true; sw_arm = true; }
                        } else if (tht <= (T(1) - Algo<T>::flt_gth) * theta || phit <= phi0 - Algo<T>::flt_gph * theta) accepted = true;
                    }
                    if (accepted) break;
                }
            }
            if (filter) {
                if (accepted && !sw_arm) {
                    f_th[fpos] = theta; f_ph[fpos] = phi0;
                    fpos = (fpos + 1) & (Algo<T>::flt_cap - 1); nfilt = nfilt < Algo<T>::flt_cap ? nfilt + 1 : Algo<T>::flt_cap;
                }
                if (!accepted) { nfilt = 0; fpos = 0; accepted = t_finite(f_t) && t_finite(lg_t); }
            }
            if (P.acc_tol > T(0) && (!accepted || alpha < T(1e-6) * fw.a_p) && e0 <= P.acc_tol) {      // refused-step half of the acceptable-level stop
                eval_point(L.X, L.U, L.D, L.TRIG, L.CC, theta_c, fobj, cinf);
                status = ST_CONVERGED;
                break;
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
            last_alpha = alpha; last_ad = fw.a_d;
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

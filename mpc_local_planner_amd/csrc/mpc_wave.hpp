// mpc_wave.hpp -- wavefront-per-instance variant of the interior-point solve (device only).
//
// One 64-lane wavefront (= one workgroup) owns ONE planner instance.  Every per-instance array
// (iterate, duals, slacks, step, Riccati gains, per-stage LQ data) lives in LDS for the whole solve
// (~38 KB at n = 50 in fp64, so 4 instances per CU = 1024 per MI355X, one wave per SIMD); HBM is
// touched only to read the inputs / initial guess and to write the result.
//
// Work split inside an interior-point iteration:
//   lane-parallel over the horizon (lane k <-> interval k / grid point k / rate row k):
//       residuals + KKT error, per-stage LQ data (dynamics Jacobians, Lagrangian curvature, condensed
//       barrier terms), step post-processing (slack/dual steps, fraction-to-boundary), line-search
//       trial evaluation, acceptance.  Scalars are combined with wavefront reductions (DPP/bpermute).
//   wave-uniform (every lane executes the same scalar recurrence on broadcast LDS reads, lane 0 stores):
//       backward Riccati sweep over the augmented stage state (x_k, u_{k-1}, dt), forward state
//       recurrence, costate (multiplier) recurrence.
// The arithmetic is the same as mpc_core.hpp (lane-per-instance variant); see that file for the
// reference citations of every formula.
#pragma once
#include <hip/hip_runtime.h>

#include "mpc_core.hpp"

namespace mpc {

constexpr int kWave = 64;
constexpr int NSTG = 35;   // per-stage LQ record
constexpr int NGAIN = 20;  // K(2x6) kappa(2) Knu(2x3)

struct WaveLayout {
    int n, NS;
    int X, U, XT, UT, LAM, LAMN, SR, YR, PL, PU, DX, DU, CC, TRIG, GAIN, STG, SC, total;
    __host__ __device__ static WaveLayout make(int n) {
        WaveLayout L;
        L.n = n;
        L.NS = n;
        int o = 0;
        auto take = [&](int comps) { int b = o; o += comps * n; return b; };
        L.X = take(3); L.U = take(2); L.XT = take(3); L.UT = take(2);
        L.LAM = take(3); L.LAMN = take(3);
        L.SR = take(4); L.YR = take(4);
        L.PL = take(2); L.PU = take(2);
        L.DX = take(3); L.DU = take(2);
        L.CC = take(3); L.TRIG = take(4);
        L.GAIN = take(NGAIN); L.STG = take(NSTG);
        L.SC = o; o += 8;     // scalars: D, DT, DD, PDL, PDU
        L.total = o;
        return L;
    }
};

enum { SC_D = 0, SC_DT = 1, SC_DD = 2, SC_PDL = 3, SC_PDU = 4 };

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <typename T> __device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o); v = w < v ? w : v; }
    return v;
}
template <typename T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o); v = w > v ? w : v; }
    return v;
}

template <typename T, int MODEL>
struct IpmWave {
    const Problem<T>& P;
    const WaveLayout& L;
    T* sm;
    const int lane;
    T x0[3], xf[3], uprev[2], dtprev;
    T mu, rho, delta_last;
    bool row0_on;
    int nfix;

    __device__ IpmWave(const Problem<T>& p, const WaveLayout& l, T* s, int ln) : P(p), L(l), sm(s), lane(ln) {}

    // ---- LDS accessors: component-major, stage-minor (conflict-free for lane == stage)
    __device__ __forceinline__ T& F(int base, int comp, int k) const { return sm[base + comp * L.NS + k]; }
    __device__ __forceinline__ T& SCL(int i) const { return sm[L.SC + i]; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }

    __device__ __forceinline__ bool row_on(int r, int q) const { return P.rate_on[q] && (r > 0 || row0_on); }

    // rate row r, slot q at controls from base UB and dt d (solver form, <= 0 feasible)
    __device__ __forceinline__ T row_val(int UB, T d, int r, int q) const {
        const int n = L.n, j = q & 1;
        T ur = r < n - 1 ? F(UB, j, r) : T(0);
        T um = r > 0 ? F(UB, j, r - 1) : uprev[j];
        T dtp = r > 0 ? d : dtprev;
        return slot_sign<T>(q) * ((ur - um) - P.rate_lim[q] * dtp);
    }
    __device__ __forceinline__ T row_jdz(int r, int q, T dd) const {
        const int n = L.n, j = q & 1;
        T dur = r < n - 1 ? F(L.DU, j, r) : T(0);
        T dum = r > 0 ? F(L.DU, j, r - 1) : T(0);
        return slot_sign<T>(q) * ((dur - dum) - (r > 0 ? P.rate_lim[q] * dd : T(0)));
    }

    __device__ __forceinline__ T push_interior(T v, T lb, T ub) const {
        T pl = t_min(Algo<T>::bound_push * t_max(T(1), t_abs(lb)), Algo<T>::bound_push * (ub - lb));
        T pu = t_min(Algo<T>::bound_push * t_max(T(1), t_abs(ub)), Algo<T>::bound_push * (ub - lb));
        return t_min(t_max(v, lb + pl), ub - pu);
    }

    // ---------------------------------------------------------------- point evaluation (parallel)
    // trig cache + c_k for the point (XB, UB, d); returns wave-reduced sum|c|, objective
    __device__ void eval_point(int XB, int UB, T d, T& theta_c, T& fobj) const {
        const int n = L.n;
        T th = T(0), fo = T(0);
        for (int k = lane; k < n - 1; k += kWave) {
            T xk[3] = {F(XB, 0, k), F(XB, 1, k), F(XB, 2, k)};
            T xn[3] = {F(XB, 0, k + 1), F(XB, 1, k + 1), F(XB, 2, k + 1)};
            T v = F(UB, 0, k), w = F(UB, 1, k);
            T tr[4], f[3];
            model_trig<T, MODEL>(P, xk[2], w, tr);
            model_f<T, MODEL>(P, tr, v, w, f);
            T c0 = d * f[0] - (xn[0] - xk[0]);
            T c1 = d * f[1] - (xn[1] - xk[1]);
            T c2 = d * f[2] - normalize_theta(xn[2] - xk[2]);
            for (int i = 0; i < 4; ++i) F(L.TRIG, i, k) = tr[i];
            F(L.CC, 0, k) = c0; F(L.CC, 1, k) = c1; F(L.CC, 2, k) = c2;
            th += t_abs(c0) + t_abs(c1) + t_abs(c2);
            if (P.objective == OBJ_QUADRATIC) {
                T xd0 = xk[0] - xf[0], xd1 = xk[1] - xf[1], xd2 = normalize_theta(xk[2] - xf[2]);
                fo += P.Q[0] * xd0 * xd0 + P.Q[1] * xd1 * xd1 + P.Q[2] * xd2 * xd2 + P.R[0] * v * v + P.R[1] * w * w;
            }
        }
        if (lane == 0) {
            if (P.objective == OBJ_MIN_TIME) fo += T(n - 1) * d;
            else if (P.has_Qf) {
                for (int i = 0; i < 3; ++i) if (!P.xf_fixed[i]) {
                    T xd = F(XB, i, n - 1) - xf[i];
                    if (i == 2) xd = normalize_theta(xd);
                    fo += P.Qf[i] * xd * xd;
                }
            }
        }
        theta_c = wave_sum(th);
        fobj = wave_sum(fo);
    }

    // sum of barrier logs at the current (alpha = 0) or trial point; wave-reduced
    __device__ T barrier_logs(int UB, T d, T alpha, bool trial, T dd) const {
        const int n = L.n;
        LogAcc<T> acc;
        for (int k = lane; k < n; k += kWave) {
            if (k < n - 1) {
                for (int j = 0; j < 2; ++j) { T u = F(UB, j, k); acc.mul(u - P.u_lb[j]); acc.mul(P.u_ub[j] - u); }
            }
            for (int q = 0; q < 4; ++q) {
                if (!row_on(k, q)) continue;
                T s = F(L.SR, q, k);
                if (trial) s += alpha * (-(row_val(L.U, SCL(SC_D), k, q) + s) - row_jdz(k, q, dd));
                acc.mul(s);
            }
        }
        if (lane == 0 && P.dt_free) { acc.mul(d - P.dt_lb); acc.mul(P.dt_ub - d); }
        return wave_sum(acc.value());
    }

    // ---------------------------------------------------------------- KKT error + stage records
    struct Err { T rd, rp, cmin, cmax, sum_mult, sum_bmult, theta; int n_mult, n_bmult; };

    __device__ T err_value(const Err& e, T mu_t) const {
        T sd = t_max(Algo<T>::s_max, e.sum_mult / T(e.n_mult > 0 ? e.n_mult : 1)) / Algo<T>::s_max;
        T sc = t_max(Algo<T>::s_max, e.sum_bmult / T(e.n_bmult > 0 ? e.n_bmult : 1)) / Algo<T>::s_max;
        T comp = e.n_bmult > 0 ? t_max(e.cmax - mu_t, mu_t - e.cmin) : T(0);
        return t_max(e.rd / sd, t_max(e.rp, comp / sc));
    }

    // parallel: KKT error pieces (needs LAM of the neighbours) ; also writes the mu-independent part of STG
    __device__ Err kkt_pass() const {
        const int n = L.n;
        const T d = SCL(SC_D);
        T rd = T(0), rp = T(0), cmin = T(1e30), cmax = T(0), smult = T(0), sb = T(0), th = T(0), rdd = T(0);
        int nm = 0, nb = 0;
        for (int k = lane; k < n; k += kWave) {
            if (k < n - 1) {
                T lam[3] = {F(L.LAM, 0, k), F(L.LAM, 1, k), F(L.LAM, 2, k)};
                T tr[4] = {F(L.TRIG, 0, k), F(L.TRIG, 1, k), F(L.TRIG, 2, k), F(L.TRIG, 3, k)};
                T v = F(L.U, 0, k), w = F(L.U, 1, k);
                T f[3], G[3][3], Hq[3][3];
                model_derivs<T, MODEL>(P, tr, v, w, lam, f, G, Hq);
                T gq[3];
                for (int j = 0; j < 3; ++j) gq[j] = lam[0] * G[0][j] + lam[1] * G[1][j] + lam[2] * G[2][j];
                // stage record, mu-independent part
                F(L.STG, 0, k) = d * G[0][0]; F(L.STG, 1, k) = d * G[1][0];
                F(L.STG, 2, k) = f[0]; F(L.STG, 3, k) = f[1]; F(L.STG, 4, k) = f[2];
                for (int a = 0; a < 3; ++a) { F(L.STG, 5 + 2 * a, k) = d * G[a][1]; F(L.STG, 6 + 2 * a, k) = d * G[a][2]; }
                F(L.STG, 11, k) = d * Hq[0][0]; F(L.STG, 12, k) = d * Hq[0][1]; F(L.STG, 13, k) = d * Hq[0][2];
                F(L.STG, 14, k) = d * Hq[1][1]; F(L.STG, 15, k) = d * Hq[1][2]; F(L.STG, 16, k) = d * Hq[2][2];
                F(L.STG, 17, k) = gq[0]; F(L.STG, 18, k) = gq[1]; F(L.STG, 19, k) = gq[2];
                for (int i = 0; i < 3; ++i) {
                    T ci = F(L.CC, i, k);
                    rp = t_max(rp, t_abs(ci)); th += t_abs(ci); smult += t_abs(lam[i]);
                }
                nm += 3;
                rdd += lam[0] * f[0] + lam[1] * f[1] + lam[2] * f[2];
                T gx[3] = {T(0), T(0), T(0)}, gu[2] = {T(0), T(0)};
                if (P.objective == OBJ_QUADRATIC) {
                    T xd[3] = {F(L.X, 0, k) - xf[0], F(L.X, 1, k) - xf[1], normalize_theta(F(L.X, 2, k) - xf[2])};
                    for (int i = 0; i < 3; ++i) gx[i] = T(2) * P.Q[i] * xd[i];
                    gu[0] = T(2) * P.R[0] * v; gu[1] = T(2) * P.R[1] * w;
                }
                F(L.STG, 32, k) = gx[0]; F(L.STG, 33, k) = gx[1]; F(L.STG, 34, k) = gx[2];
                if (k >= 1) {
                    T r0 = gx[0] + lam[0] - F(L.LAM, 0, k - 1);
                    T r1 = gx[1] + lam[1] - F(L.LAM, 1, k - 1);
                    T r2 = gx[2] + lam[2] + d * gq[0] - F(L.LAM, 2, k - 1);
                    rd = t_max(rd, t_max(t_abs(r0), t_max(t_abs(r1), t_abs(r2))));
                }
                for (int j = 0; j < 2; ++j) {
                    T u = j == 0 ? v : w;
                    T pl = F(L.PL, j, k), pu = F(L.PU, j, k);
                    T r = gu[j] + d * gq[1 + j] - pl + pu;
                    for (int q = j; q < 4; q += 2) {
                        const T sg = slot_sign<T>(q);
                        if (row_on(k, q)) r += sg * F(L.YR, q, k);
                        if (row_on(k + 1, q)) r -= sg * F(L.YR, q, k + 1);
                    }
                    rd = t_max(rd, t_abs(r));
                    T cl = (u - P.u_lb[j]) * pl, cu = (P.u_ub[j] - u) * pu;
                    cmin = t_min(cmin, t_min(cl, cu)); cmax = t_max(cmax, t_max(cl, cu));
                    sb += pl + pu; nb += 2;
                }
                if (k == n - 2) {
                    for (int i = 0; i < 3; ++i) if (!P.xf_fixed[i]) {
                        T g = T(0);
                        if (P.objective == OBJ_QUADRATIC && P.has_Qf) {
                            T xd = F(L.X, i, n - 1) - xf[i];
                            if (i == 2) xd = normalize_theta(xd);
                            g = T(2) * P.Qf[i] * xd;
                        }
                        rd = t_max(rd, t_abs(g - lam[i]));
                    }
                }
            }
            for (int q = 0; q < 4; ++q) {
                if (!row_on(k, q)) continue;
                T s = F(L.SR, q, k), y = F(L.YR, q, k);
                T res = row_val(L.U, d, k, q) + s;
                rp = t_max(rp, t_abs(res)); th += t_abs(res);
                cmin = t_min(cmin, s * y); cmax = t_max(cmax, s * y);
                sb += y; nb += 1;
                if (k > 0) rdd -= slot_sign<T>(q) * P.rate_lim[q] * y;
            }
        }
        if (lane == 0) {
            if (P.objective == OBJ_MIN_TIME) rdd += T(n - 1);
            if (P.dt_free) {
                T pl = SCL(SC_PDL), pu = SCL(SC_PDU);
                rdd += -pl + pu;
                T cl = (d - P.dt_lb) * pl, cu = (P.dt_ub - d) * pu;
                cmin = t_min(cmin, t_min(cl, cu)); cmax = t_max(cmax, t_max(cl, cu));
                sb += pl + pu; nb += 2;
            }
        }
        Err e;
        rdd = wave_sum(rdd);
        e.rd = wave_max(rd);
        if (P.dt_free) e.rd = t_max(e.rd, t_abs(rdd));
        e.rp = wave_max(rp);
        e.cmin = wave_min(cmin);
        e.cmax = wave_max(cmax);
        e.sum_bmult = wave_sum(sb);
        e.sum_mult = wave_sum(smult) + e.sum_bmult;
        e.theta = wave_sum(th);
        e.n_bmult = (int)wave_sum((T)nb);
        e.n_mult = (int)wave_sum((T)nm) + e.n_bmult;
        return e;
    }

    // parallel: mu-dependent part of the stage records (box + rate-row condensation); record n-1 = final rate rows
    __device__ void stage_barrier_terms() const {
        const int n = L.n;
        const T d = SCL(SC_D);
        for (int k = lane; k < n; k += kWave) {
            if (k < n - 1) {
                for (int j = 0; j < 2; ++j) {
                    T u = F(L.U, j, k);
                    T dl = u - P.u_lb[j], du = P.u_ub[j] - u;
                    F(L.STG, 20 + j, k) = F(L.PL, j, k) / dl + F(L.PU, j, k) / du;
                    T g = -mu / dl + mu / du;
                    if (P.objective == OBJ_QUADRATIC) g += T(2) * P.R[j] * u;
                    F(L.STG, 22 + j, k) = g;
                }
            }
            T ss[2] = {T(0), T(0)}, ssl[2] = {T(0), T(0)}, sll = T(0), gy[2] = {T(0), T(0)}, gyl = T(0);
            for (int q = 0; q < 4; ++q) {
                if (!row_on(k, q)) continue;
                const int j = q & 1;
                const T sg = slot_sign<T>(q), lim = k > 0 ? P.rate_lim[q] : T(0);
                T s = F(L.SR, q, k), y = F(L.YR, q, k);
                T sig = y / s;
                T ybar = mu / s + sig * (row_val(L.U, d, k, q) + s);
                ss[j] += sig; ssl[j] += sig * lim; sll += sig * lim * lim;
                gy[j] += sg * ybar; gyl += sg * lim * ybar;
            }
            F(L.STG, 24, k) = ss[0]; F(L.STG, 25, k) = ss[1];
            F(L.STG, 26, k) = ssl[0]; F(L.STG, 27, k) = ssl[1];
            F(L.STG, 28, k) = sll;
            F(L.STG, 29, k) = gy[0]; F(L.STG, 30, k) = gy[1];
            F(L.STG, 31, k) = gyl;
        }
    }

    // ---------------------------------------------------------------- wave-uniform backward Riccati sweep
    __device__ bool backward(T delta, T dc, T& dd_out, T nu_out[3]) const {
        const int n = L.n;
        const T d = SCL(SC_D);
        T Pm[6][6], pv[6], S[6][3], W[3][3], om[3];
        for (int a = 0; a < 6; ++a) { pv[a] = T(0); for (int b = 0; b < 6; ++b) Pm[a][b] = T(0); for (int b = 0; b < 3; ++b) S[a][b] = T(0); }
        for (int a = 0; a < 3; ++a) { om[a] = T(0); for (int b = 0; b < 3; ++b) W[a][b] = T(0); }
        for (int i = 0; i < 3; ++i) {
            if (P.xf_fixed[i]) { S[i][i] = T(1); W[i][i] = -dc; }
            else {
                Pm[i][i] = delta;
                if (P.objective == OBJ_QUADRATIC && P.has_Qf) {
                    T xd = F(L.X, i, n - 1) - xf[i];
                    if (i == 2) xd = normalize_theta(xd);
                    Pm[i][i] += T(2) * P.Qf[i];
                    pv[i] = T(2) * P.Qf[i] * xd;
                }
            }
        }
        {   // final rate rows (record n-1): a over (up_j, d) = (-sg, -sg*lim)
            const int r = n - 1;
            for (int j = 0; j < 2; ++j) {
                T ssj = F(L.STG, 24 + j, r), sslj = F(L.STG, 26 + j, r);
                Pm[3 + j][3 + j] += ssj;
                Pm[3 + j][5] += sslj; Pm[5][3 + j] += sslj;
                pv[3 + j] -= F(L.STG, 29 + j, r);
            }
            Pm[5][5] += F(L.STG, 28, r);
            pv[5] -= F(L.STG, 31, r);
        }
        for (int k = n - 2; k >= 0; --k) {
            T R_[NSTG];
#pragma unroll
            for (int i = 0; i < NSTG; ++i) R_[i] = F(L.STG, i, k);
            T ck[3] = {F(L.CC, 0, k), F(L.CC, 1, k), F(L.CC, 2, k)};
            T pt[6];
            for (int a = 0; a < 6; ++a) pt[a] = pv[a] + Pm[a][0] * ck[0] + Pm[a][1] * ck[1] + Pm[a][2] * ck[2];
            for (int b = 0; b < 3; ++b) om[b] += S[0][b] * ck[0] + S[1][b] * ck[1] + S[2][b] * ck[2];
            T Gx[3][4], Bx[3][2];
            for (int a = 0; a < 3; ++a) {
                Gx[a][0] = a == 0 ? T(1) : T(0);
                Gx[a][1] = a == 1 ? T(1) : T(0);
                Gx[a][2] = (a == 2 ? T(1) : T(0)) + (a == 0 ? R_[0] : (a == 1 ? R_[1] : T(0)));
                Gx[a][3] = R_[2 + a];
                Bx[a][0] = R_[5 + 2 * a];
                Bx[a][1] = R_[6 + 2 * a];
            }
            T Z[6][4], Y[6][2];
            for (int a = 0; a < 6; ++a) {
                for (int c4 = 0; c4 < 4; ++c4) {
                    T z = Pm[a][0] * Gx[0][c4] + Pm[a][1] * Gx[1][c4] + Pm[a][2] * Gx[2][c4];
                    if (c4 == 3) z += Pm[a][5];
                    Z[a][c4] = z;
                }
                for (int c2 = 0; c2 < 2; ++c2)
                    Y[a][c2] = Pm[a][0] * Bx[0][c2] + Pm[a][1] * Bx[1][c2] + Pm[a][2] * Bx[2][c2] + Pm[a][3 + c2];
            }
            const int im[4] = {0, 1, 2, 5};
            T Qt[6][6], Mt[2][6], Rt[2][2], qt[6], rt[2], Sx[6][3], Su[2][3];
            for (int a = 0; a < 6; ++a) { qt[a] = T(0); for (int b = 0; b < 6; ++b) Qt[a][b] = T(0); for (int b = 0; b < 3; ++b) Sx[a][b] = T(0); }
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 6; ++b) Mt[a][b] = T(0);
            for (int r4 = 0; r4 < 4; ++r4) {
                for (int c4 = 0; c4 < 4; ++c4) {
                    T z = Gx[0][r4] * Z[0][c4] + Gx[1][r4] * Z[1][c4] + Gx[2][r4] * Z[2][c4];
                    if (r4 == 3) z += Z[5][c4];
                    Qt[im[r4]][im[c4]] = z;
                }
                T g = Gx[0][r4] * pt[0] + Gx[1][r4] * pt[1] + Gx[2][r4] * pt[2];
                if (r4 == 3) g += pt[5];
                qt[im[r4]] = g;
                for (int b = 0; b < 3; ++b) {
                    T sgs = Gx[0][r4] * S[0][b] + Gx[1][r4] * S[1][b] + Gx[2][r4] * S[2][b];
                    if (r4 == 3) sgs += S[5][b];
                    Sx[im[r4]][b] = sgs;
                }
            }
            for (int a = 0; a < 2; ++a) {
                for (int c4 = 0; c4 < 4; ++c4)
                    Mt[a][im[c4]] = Bx[0][a] * Z[0][c4] + Bx[1][a] * Z[1][c4] + Bx[2][a] * Z[2][c4] + Z[3 + a][c4];
                for (int b = 0; b < 2; ++b)
                    Rt[a][b] = Bx[0][a] * Y[0][b] + Bx[1][a] * Y[1][b] + Bx[2][a] * Y[2][b] + Y[3 + a][b];
                rt[a] = Bx[0][a] * pt[0] + Bx[1][a] * pt[1] + Bx[2][a] * pt[2] + pt[3 + a];
                for (int b = 0; b < 3; ++b)
                    Su[a][b] = Bx[0][a] * S[0][b] + Bx[1][a] * S[1][b] + Bx[2][a] * S[2][b] + S[3 + a][b];
            }
            // stage cost from the record
            Qt[2][2] += R_[11];
            Mt[0][2] += R_[12]; Mt[1][2] += R_[13];
            Rt[0][0] += R_[14]; Rt[0][1] += R_[15]; Rt[1][0] += R_[15]; Rt[1][1] += R_[16];
            Qt[2][5] += R_[17]; Qt[5][2] += R_[17];
            Mt[0][5] += R_[18]; Mt[1][5] += R_[19];
            if (P.objective == OBJ_QUADRATIC) {
                for (int i = 0; i < 3; ++i) { Qt[i][i] += T(2) * P.Q[i]; qt[i] += R_[32 + i]; }
                for (int j = 0; j < 2; ++j) Rt[j][j] += T(2) * P.R[j];
            } else if (k == 0) {
                qt[5] += T(n - 1);
            }
            for (int j = 0; j < 2; ++j) { Rt[j][j] += R_[20 + j] + delta; rt[j] += R_[22 + j]; }
            if (k == 0 && P.dt_free) {
                T dl = d - P.dt_lb, du = P.dt_ub - d;
                Qt[5][5] += SCL(SC_PDL) / dl + SCL(SC_PDU) / du + delta;
                qt[5] += -mu / dl + mu / du;
            }
            if (k >= 1) { Qt[0][0] += delta; Qt[1][1] += delta; Qt[2][2] += delta; }
            for (int j = 0; j < 2; ++j) {
                const T ssj = R_[24 + j], sslj = R_[26 + j];
                Rt[j][j] += ssj;
                Mt[j][3 + j] -= ssj;
                Mt[j][5] -= sslj;
                Qt[3 + j][3 + j] += ssj;
                Qt[3 + j][5] += sslj; Qt[5][3 + j] += sslj;
                rt[j] += R_[29 + j];
                qt[3 + j] -= R_[29 + j];
            }
            Qt[5][5] += R_[28];
            qt[5] -= R_[31];
            // eliminate u_k
            T det = Rt[0][0] * Rt[1][1] - Rt[0][1] * Rt[1][0];
            T scale = t_abs(Rt[0][0] * Rt[1][1]) + t_abs(Rt[0][1] * Rt[1][0]);
            if (!(t_abs(det) > T(1e-14) * scale) || !t_finite(det)) return false;
            T id = T(1) / det;
            T Ri[2][2] = {{Rt[1][1] * id, -Rt[0][1] * id}, {-Rt[1][0] * id, Rt[0][0] * id}};
            T K[2][6], kap[2], Kn[2][3];
            for (int a = 0; a < 2; ++a) {
                for (int b = 0; b < 6; ++b) K[a][b] = Ri[a][0] * Mt[0][b] + Ri[a][1] * Mt[1][b];
                kap[a] = Ri[a][0] * rt[0] + Ri[a][1] * rt[1];
                for (int b = 0; b < 3; ++b) Kn[a][b] = Ri[a][0] * Su[0][b] + Ri[a][1] * Su[1][b];
            }
            if (lane == 0) {
                for (int a = 0; a < 2; ++a) for (int b = 0; b < 6; ++b) F(L.GAIN, 6 * a + b, k) = K[a][b];
                F(L.GAIN, 12, k) = kap[0]; F(L.GAIN, 13, k) = kap[1];
                for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) F(L.GAIN, 14 + 3 * a + b, k) = Kn[a][b];
            }
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) Pm[a][b] = Qt[a][b] - (Mt[0][a] * K[0][b] + Mt[1][a] * K[1][b]);
                pv[a] = qt[a] - (Mt[0][a] * kap[0] + Mt[1][a] * kap[1]);
                for (int b = 0; b < 3; ++b) S[a][b] = Sx[a][b] - (Mt[0][a] * Kn[0][b] + Mt[1][a] * Kn[1][b]);
            }
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) W[a][b] -= Su[0][a] * Kn[0][b] + Su[1][a] * Kn[1][b];
                om[a] -= Su[0][a] * kap[0] + Su[1][a] * kap[1];
            }
            for (int a = 0; a < 6; ++a) for (int b = a + 1; b < 6; ++b) { T m = T(0.5) * (Pm[a][b] + Pm[b][a]); Pm[a][b] = m; Pm[b][a] = m; }
        }
        T A4[4][5];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 5; ++b) A4[a][b] = T(0);
        if (P.dt_free) {
            A4[0][0] = Pm[5][5];
            for (int b = 0; b < 3; ++b) A4[0][1 + b] = P.xf_fixed[b] ? S[5][b] : T(0);
            A4[0][4] = -pv[5];
        } else { A4[0][0] = T(1); }
        for (int a = 0; a < 3; ++a) {
            if (P.xf_fixed[a]) {
                A4[1 + a][0] = P.dt_free ? S[5][a] : T(0);
                for (int b = 0; b < 3; ++b) A4[1 + a][1 + b] = P.xf_fixed[b] ? W[a][b] : T(0);
                A4[1 + a][4] = -om[a];
            } else { A4[1 + a][1 + a] = T(1); }
        }
        for (int c = 0; c < 4; ++c) {
            int piv = c; T best = t_abs(A4[c][c]);
            for (int r = c + 1; r < 4; ++r) if (t_abs(A4[r][c]) > best) { best = t_abs(A4[r][c]); piv = r; }
            if (!(best > T(0)) || !t_finite(best)) return false;
            if (piv != c) for (int b = 0; b < 5; ++b) { T t = A4[c][b]; A4[c][b] = A4[piv][b]; A4[piv][b] = t; }
            T ip = T(1) / A4[c][c];
            for (int r = c + 1; r < 4; ++r) {
                T m = A4[r][c] * ip;
                for (int b = c; b < 5; ++b) A4[r][b] -= m * A4[c][b];
            }
        }
        T sol[4];
        for (int c = 3; c >= 0; --c) {
            T a = A4[c][4];
            for (int b = c + 1; b < 4; ++b) a -= A4[c][b] * sol[b];
            sol[c] = a / A4[c][c];
        }
        dd_out = sol[0];
        nu_out[0] = sol[1]; nu_out[1] = sol[2]; nu_out[2] = sol[3];
        return t_finite(sol[0]) && t_finite(sol[1]) && t_finite(sol[2]) && t_finite(sol[3]);
    }

    // wave-uniform: state recurrence (writes DU, DX) then costate recurrence (writes LAMN)
    __device__ void forward_states(T dd, const T nu[3], T delta) const {
        const int n = L.n;
        T xi[6] = {T(0), T(0), T(0), T(0), T(0), dd};
        if (lane == 0) { SCL(SC_DD) = dd; F(L.DX, 0, 0) = T(0); F(L.DX, 1, 0) = T(0); F(L.DX, 2, 0) = T(0); }
        for (int k = 0; k < n - 1; ++k) {
            T du_[2];
            for (int a = 0; a < 2; ++a) {
                T acc = F(L.GAIN, 12 + a, k);
                for (int b = 0; b < 6; ++b) acc += F(L.GAIN, 6 * a + b, k) * xi[b];
                for (int b = 0; b < 3; ++b) acc += F(L.GAIN, 14 + 3 * a + b, k) * nu[b];
                du_[a] = -acc;
            }
            T xn[3];
            for (int a = 0; a < 3; ++a) {
                T ax = a < 2 ? F(L.STG, a, k) : T(0);
                xn[a] = xi[a] + ax * xi[2] + F(L.STG, 5 + 2 * a, k) * du_[0] + F(L.STG, 6 + 2 * a, k) * du_[1] + F(L.STG, 2 + a, k) * dd + F(L.CC, a, k);
            }
            if (lane == 0) {
                F(L.DU, 0, k) = du_[0]; F(L.DU, 1, k) = du_[1];
                F(L.DX, 0, k + 1) = xn[0]; F(L.DX, 1, k + 1) = xn[1]; F(L.DX, 2, k + 1) = xn[2];
            }
            xi[0] = xn[0]; xi[1] = xn[1]; xi[2] = xn[2]; xi[3] = du_[0]; xi[4] = du_[1];
        }
        // costate: lam+_{k-1} = A_x,k^T lam+_k + (H dz)_{x_k} + h_{x_k}
        T lp[3];
        for (int i = 0; i < 3; ++i) {
            if (P.xf_fixed[i]) lp[i] = nu[i];
            else {
                T g = delta * xi[i];
                if (P.objective == OBJ_QUADRATIC && P.has_Qf) {
                    T xd = F(L.X, i, n - 1) - xf[i];
                    if (i == 2) xd = normalize_theta(xd);
                    g += T(2) * P.Qf[i] * (xi[i] + xd);
                }
                lp[i] = g;
            }
        }
        if (lane == 0) { F(L.LAMN, 0, n - 2) = lp[0]; F(L.LAMN, 1, n - 2) = lp[1]; F(L.LAMN, 2, n - 2) = lp[2]; }
        sync();
        for (int k = n - 2; k >= 1; --k) {
            T dx[3] = {F(L.DX, 0, k), F(L.DX, 1, k), F(L.DX, 2, k)};
            T duv = F(L.DU, 0, k), duw = F(L.DU, 1, k);
            T t[3];
            for (int i = 0; i < 3; ++i) {
                T qd = delta + (P.objective == OBJ_QUADRATIC ? T(2) * P.Q[i] : T(0));
                t[i] = qd * dx[i] + F(L.STG, 32 + i, k);
            }
            t[2] += F(L.STG, 11, k) * dx[2] + F(L.STG, 12, k) * duv + F(L.STG, 13, k) * duw + F(L.STG, 17, k) * dd
                  + F(L.STG, 0, k) * lp[0] + F(L.STG, 1, k) * lp[1];
            lp[0] += t[0]; lp[1] += t[1]; lp[2] += t[2];
            if (lane == 0) { F(L.LAMN, 0, k - 1) = lp[0]; F(L.LAMN, 1, k - 1) = lp[1]; F(L.LAMN, 2, k - 1) = lp[2]; }
        }
    }

    // ---------------------------------------------------------------- parallel post-processing of the step
    struct Fwd { T hdz, clam, dz2, dphi, a_p, a_d, dzmax, nunu; bool finite; };

    __device__ __forceinline__ void ftb(T val, T dval, T tau, T& alpha) const {
        if (dval < T(0)) { T a = -tau * val / dval; if (a < alpha) alpha = a; }
    }

    __device__ Fwd post_pass(T dd, const T nu[3], T tau) const {
        const int n = L.n;
        const T d = SCL(SC_D);
        T hdz = T(0), clam = T(0), dz2 = T(0), dphi = T(0), a_p = T(1), a_d = T(1), dzmax = T(0);
        bool fin = true;
        if (lane == 0) {
            if (P.dt_free) {
                T dl = d - P.dt_lb, du = P.dt_ub - d;
                T pl = SCL(SC_PDL), pu = SCL(SC_PDU);
                T gb = -mu / dl + mu / du;
                hdz += gb * dd; dphi += gb * dd;
                ftb(dl, dd, tau, a_p); ftb(du, -dd, tau, a_p);
                ftb(pl, mu / dl - pl - (pl / dl) * dd, tau, a_d);
                ftb(pu, mu / du - pu + (pu / du) * dd, tau, a_d);
                dz2 += dd * dd; dzmax = t_max(dzmax, t_abs(dd));
            }
            if (P.objective == OBJ_MIN_TIME) { hdz += T(n - 1) * dd; dphi += T(n - 1) * dd; }
        }
        for (int k = lane; k < n; k += kWave) {
            if (k < n - 1) {
                for (int j = 0; j < 2; ++j) {
                    T u = F(L.U, j, k), du_ = F(L.DU, j, k);
                    T dl = u - P.u_lb[j], du = P.u_ub[j] - u;
                    T pl = F(L.PL, j, k), pu = F(L.PU, j, k);
                    T gbar = F(L.STG, 22 + j, k);         // barrier (+ quadratic objective) gradient wrt u
                    hdz += gbar * du_; dphi += gbar * du_;
                    ftb(dl, du_, tau, a_p); ftb(du, -du_, tau, a_p);
                    ftb(pl, mu / dl - pl - (pl / dl) * du_, tau, a_d);
                    ftb(pu, mu / du - pu + (pu / du) * du_, tau, a_d);
                    dz2 += du_ * du_; dzmax = t_max(dzmax, t_abs(du_));
                }
                for (int i = 0; i < 3; ++i) {
                    T l = F(L.LAMN, i, k);
                    clam += F(L.CC, i, k) * l;
                    if (!t_finite(l)) fin = false;
                }
            }
            if (k >= 1) {
                for (int i = 0; i < 3; ++i) {
                    if (k < n - 1 || !P.xf_fixed[i]) {
                        T dx = F(L.DX, i, k);
                        dz2 += dx * dx; dzmax = t_max(dzmax, t_abs(dx));
                        T g = T(0);
                        if (P.objective == OBJ_QUADRATIC) {
                            if (k < n - 1) g = F(L.STG, 32 + i, k);
                            else if (P.has_Qf) { T xd = F(L.X, i, k) - xf[i]; if (i == 2) xd = normalize_theta(xd); g = T(2) * P.Qf[i] * xd; }
                        }
                        hdz += g * dx; dphi += g * dx;
                    }
                }
            }
            for (int q = 0; q < 4; ++q) {
                if (!row_on(k, q)) continue;
                T jdz = row_jdz(k, q, dd);
                T s = F(L.SR, q, k), y = F(L.YR, q, k);
                T res = row_val(L.U, d, k, q) + s;
                T sig = y / s;
                T ybar = mu / s + sig * res;
                T ds = -res - jdz;
                T dy = ybar + sig * jdz - y;
                hdz += ybar * jdz;
                dphi -= (mu / s) * ds;
                ftb(s, ds, tau, a_p);
                ftb(y, dy, tau, a_d);
            }
        }
        Fwd o;
        o.hdz = wave_sum(hdz); o.clam = wave_sum(clam); o.dz2 = wave_sum(dz2); o.dphi = wave_sum(dphi);
        o.a_p = wave_min(a_p); o.a_d = wave_min(a_d); o.dzmax = wave_max(dzmax);
        o.nunu = T(0);
        for (int i = 0; i < 3; ++i) if (P.xf_fixed[i]) o.nunu += nu[i] * nu[i];
        o.finite = (wave_min(fin ? T(1) : T(0)) > T(0.5)) && t_finite(o.hdz) && t_finite(o.dz2);
        return o;
    }

    // ---------------------------------------------------------------- trial point / acceptance (parallel)
    __device__ void make_trial(T alpha) const {
        const int n = L.n;
        for (int k = lane; k < n; k += kWave) {
            for (int i = 0; i < 3; ++i) {
                T x = F(L.X, i, k);
                if (k > 0 && (k < n - 1 || !P.xf_fixed[i])) {
                    x += alpha * F(L.DX, i, k);
                    if (i == 2) x = normalize_theta(x);
                }
                F(L.XT, i, k) = x;
            }
            if (k < n - 1) for (int j = 0; j < 2; ++j) F(L.UT, j, k) = F(L.U, j, k) + alpha * F(L.DU, j, k);
        }
        if (lane == 0) SCL(SC_DT) = SCL(SC_D) + (P.dt_free ? alpha * SCL(SC_DD) : T(0));
    }

    __device__ void accept(T alpha, T a_d) const {
        const int n = L.n;
        const T kS = T(1e10);
        const T d_old = SCL(SC_D), dd = SCL(SC_DD), d_new = SCL(SC_DT);
        // phase 1: everything that reads the OLD point
        T sn[4], yn[4];
        for (int k = lane; k < n; k += kWave) {      // (n <= 64 + ... handled by the loop; registers reused per chunk)
            for (int q = 0; q < 4; ++q) {
                if (!row_on(k, q)) continue;
                T s = F(L.SR, q, k), y = F(L.YR, q, k);
                T res = row_val(L.U, d_old, k, q) + s;
                T jdz = row_jdz(k, q, dd);
                T sig = y / s;
                T ds = -res - jdz;
                T dy = mu / s + sig * res + sig * jdz - y;
                sn[q] = s + alpha * ds;
                T yv = y + a_d * dy;
                yn[q] = t_min(t_max(yv, mu / (kS * sn[q])), kS * mu / sn[q]);
            }
            sync();      // all lanes of this chunk have read their neighbours' old controls
            for (int q = 0; q < 4; ++q) if (row_on(k, q)) { F(L.SR, q, k) = sn[q]; F(L.YR, q, k) = yn[q]; }
            if (k < n - 1) {
                for (int j = 0; j < 2; ++j) {
                    T u = F(L.U, j, k), du_ = F(L.DU, j, k);
                    T dl = u - P.u_lb[j], du = P.u_ub[j] - u;
                    T pl = F(L.PL, j, k), pu = F(L.PU, j, k);
                    T pln = pl + a_d * (mu / dl - pl - (pl / dl) * du_);
                    T pun = pu + a_d * (mu / du - pu + (pu / du) * du_);
                    T un = F(L.UT, j, k);
                    T dln = un - P.u_lb[j], dun = P.u_ub[j] - un;
                    F(L.PL, j, k) = t_min(t_max(pln, mu / (kS * dln)), kS * mu / dln);
                    F(L.PU, j, k) = t_min(t_max(pun, mu / (kS * dun)), kS * mu / dun);
                    F(L.U, j, k) = un;
                }
                for (int i = 0; i < 3; ++i) {
                    T lo = F(L.LAM, i, k);
                    F(L.LAM, i, k) = lo + alpha * (F(L.LAMN, i, k) - lo);
                }
            }
            for (int i = 0; i < 3; ++i) F(L.X, i, k) = F(L.XT, i, k);
        }
        if (lane == 0) {
            if (P.dt_free) {
                T dl = d_old - P.dt_lb, du = P.dt_ub - d_old;
                T pl = SCL(SC_PDL), pu = SCL(SC_PDU);
                T pln = pl + a_d * (mu / dl - pl - (pl / dl) * dd);
                T pun = pu + a_d * (mu / du - pu + (pu / du) * dd);
                T dln = d_new - P.dt_lb, dun = P.dt_ub - d_new;
                SCL(SC_PDL) = t_min(t_max(pln, mu / (kS * dln)), kS * mu / dln);
                SCL(SC_PDU) = t_min(t_max(pun, mu / (kS * dun)), kS * mu / dun);
            }
            SCL(SC_D) = d_new;
        }
    }

    // ---------------------------------------------------------------- initial point (parallel)
    __device__ void cold_start() const {
        const int n = L.n;
        const T dth = normalize_theta(xf[2] - x0[2]);
        for (int k = lane; k < n; k += kWave) {
            T fr = T(k) / T(n - 1);
            T xk[3];
            if (k == 0) { xk[0] = x0[0]; xk[1] = x0[1]; xk[2] = x0[2]; }
            else if (k == n - 1) { xk[0] = xf[0]; xk[1] = xf[1]; xk[2] = xf[2]; }
            else {
                xk[0] = x0[0] + fr * (xf[0] - x0[0]);
                xk[1] = x0[1] + fr * (xf[1] - x0[1]);
                xk[2] = normalize_theta(x0[2] + fr * dth);
            }
            for (int i = 0; i < 3; ++i) F(L.X, i, k) = xk[i];
            if (k < n - 1) { F(L.U, 0, k) = T(0); F(L.U, 1, k) = T(0); }
        }
        if (lane == 0) SCL(SC_D) = P.dt_ref;
    }

    __device__ void init_point() {
        const int n = L.n;
        if (lane == 0) {
            for (int i = 0; i < 3; ++i) {
                F(L.X, i, 0) = x0[i];
                if (P.xf_fixed[i]) F(L.X, i, n - 1) = xf[i];
            }
            if (!P.dt_free) SCL(SC_D) = P.dt_ref;
        }
        sync();
        // seed controls from the state guess when every control is zero
        T nz = T(0);
        for (int k = lane; k < n - 1; k += kWave) nz += (F(L.U, 0, k) != T(0) || F(L.U, 1, k) != T(0)) ? T(1) : T(0);
        nz = wave_sum(nz);
        const T d0 = SCL(SC_D);
        if (nz == T(0)) {
            for (int k = lane; k < n - 1; k += kWave) {
                T dx = F(L.X, 0, k + 1) - F(L.X, 0, k), dy = F(L.X, 1, k + 1) - F(L.X, 1, k);
                T dth = normalize_theta(F(L.X, 2, k + 1) - F(L.X, 2, k));
                T s, c;
                t_sincos(F(L.X, 2, k), &s, &c);
                T v = (dx * c + dy * s) / d0;
                v = t_min(t_max(v, P.u_lb[0]), P.u_ub[0]);
                T rate = dth / d0, w;
                if (MODEL == MODEL_UNICYCLE) w = rate;
                else {
                    T vv = t_abs(v) > T(1e-3) ? v : (v >= T(0) ? T(1e-3) : T(-1e-3));
                    if (MODEL == MODEL_SIMPLE_CAR) w = t_atan(P.p0 * rate / vv);
                    else if (MODEL == MODEL_SIMPLE_CAR_FRONT) w = t_asin(t_min(T(1), t_max(T(-1), P.p0 * rate / vv)));
                    else { T sb = t_min(T(1), t_max(T(-1), P.p0 * rate / vv)); w = t_atan(t_tan(t_asin(sb)) * (P.p1 + P.p0) / P.p0); }
                }
                w = t_min(t_max(w, P.u_lb[1]), P.u_ub[1]);
                F(L.U, 0, k) = v; F(L.U, 1, k) = w;
            }
        }
        sync();
        for (int k = lane; k < n - 1; k += kWave)
            for (int j = 0; j < 2; ++j) F(L.U, j, k) = push_interior(F(L.U, j, k), P.u_lb[j], P.u_ub[j]);
        if (lane == 0 && P.dt_free) SCL(SC_D) = push_interior(SCL(SC_D), P.dt_lb, P.dt_ub);
        sync();
        mu = P.mu_init; rho = T(0); delta_last = T(0);
        const T d = SCL(SC_D);
        for (int k = lane; k < n; k += kWave) {
            for (int q = 0; q < 4; ++q) {
                T s = T(1), y = T(0);
                if (row_on(k, q)) { s = t_max(-row_val(L.U, d, k, q), Algo<T>::slack_push); y = mu / s; }
                F(L.SR, q, k) = s; F(L.YR, q, k) = y;
            }
            if (k < n - 1) {
                for (int j = 0; j < 2; ++j) {
                    T u = F(L.U, j, k);
                    F(L.PL, j, k) = mu / (u - P.u_lb[j]);
                    F(L.PU, j, k) = mu / (P.u_ub[j] - u);
                }
                for (int i = 0; i < 3; ++i) F(L.LAM, i, k) = T(0);
            }
        }
        if (lane == 0) {
            SCL(SC_PDL) = P.dt_free ? mu / (d - P.dt_lb) : T(0);
            SCL(SC_PDU) = P.dt_free ? mu / (P.dt_ub - d) : T(0);
        }
        sync();
    }

    // ---------------------------------------------------------------- driver (all lanes, uniform control flow)
    __device__ SolveStats<T> solve() {
        SolveStats<T> out;
        nfix = P.xf_fixed[0] + P.xf_fixed[1] + P.xf_fixed[2];
        row0_on = dtprev != T(0);
        init_point();
        T theta_c, fobj;
        eval_point(L.X, L.U, SCL(SC_D), theta_c, fobj);
        sync();
        int it = 0, status = ST_MAX_ITER;
        T e0 = T(0);
        while (true) {
            Err er = kkt_pass();
            e0 = err_value(er, T(0));
            if (!t_finite(e0)) { status = ST_NUMERICAL; break; }
            if (e0 <= P.tol) { status = ST_CONVERGED; break; }
            if (it >= P.max_iter) { status = ST_MAX_ITER; break; }
            for (int guard = 0; guard < 50; ++guard) {
                T emu = err_value(er, mu);
                if (emu <= Algo<T>::kappa_eps * mu && mu > P.tol / T(10)) {
                    mu = t_max(P.tol / T(10), t_min(Algo<T>::kappa_mu * mu, t_pow(mu, Algo<T>::theta_mu)));
                    rho = T(0);
                } else break;
            }
            stage_barrier_terms();
            sync();
            const T tau = t_max(Algo<T>::tau_min, T(1) - mu);
            const T dc = nfix > 0 ? Algo<T>::delta_c * t_pow(mu, Algo<T>::kappa_c) : T(0);
            T delta = T(0);
            bool ok = false;
            Fwd fw;
            T dd = T(0), nu[3] = {T(0), T(0), T(0)}, curv = T(0);
            for (int ntry = 0; ntry <= 40; ++ntry) {
                bool good = backward(delta, dc, dd, nu);
                sync();
                if (good) {
                    forward_states(dd, nu, delta);
                    sync();
                    fw = post_pass(dd, nu, tau);
                    good = fw.finite;
                    if (good) {
                        curv = -fw.hdz + fw.clam - dc * fw.nunu;
                        if (curv >= Algo<T>::curv_kappa * fw.dz2) { ok = true; break; }
                    }
                }
                if (delta == T(0)) delta = (delta_last == T(0)) ? Algo<T>::delta_first : t_max(Algo<T>::delta_min, Algo<T>::kappa_minus * delta_last);
                else delta *= (delta_last == T(0)) ? Algo<T>::kappa_plus_first : Algo<T>::kappa_plus;
                if (delta > Algo<T>::delta_max) break;
            }
            if (!ok) { status = ST_LINSOLVE; break; }
            if (delta > T(0)) delta_last = delta;
            const T theta = er.theta;
            if (theta > T(0)) {
                T sigma = curv > T(0) ? T(1) : T(0);
                T rho_trial = (fw.dphi + T(0.5) * sigma * curv) / ((T(1) - Algo<T>::rho_frac) * theta);
                if (rho < rho_trial) rho = rho_trial + T(1);
            }
            const T phi0 = fobj - mu * barrier_logs(L.U, SCL(SC_D), T(0), false, dd) + rho * theta;
            const T Dm = fw.dphi - rho * theta;
            const T theta_rows = theta - theta_c;
            T alpha = fw.a_p;
            bool accepted = false;
            T th_t = T(0), f_t = T(0);
            for (int ls = 0; ls < Algo<T>::max_ls; ++ls) {
                if (ls > 0) alpha *= T(0.5);
                make_trial(alpha);
                sync();
                eval_point(L.XT, L.UT, SCL(SC_DT), th_t, f_t);
                T tht = th_t + (T(1) - alpha) * theta_rows;
                T phit = f_t - mu * barrier_logs(L.UT, SCL(SC_DT), alpha, true, dd) + rho * tht;
                sync();
                if (t_finite(phit) && phit - phi0 - Algo<T>::ls_eps * t_abs(phi0) <= Algo<T>::eta_armijo * alpha * Dm) { accepted = true; break; }
            }
            if (!accepted && alpha * fw.dzmax < T(1e-14)) { status = ST_LINESEARCH; break; }
            accept(alpha, fw.a_d);
            sync();
            theta_c = th_t; fobj = f_t;
            ++it;
        }
        out.status = status; out.iters = it; out.kkt_error = e0; out.objective = fobj;
        return out;
    }
};

}  // namespace mpc

// DEVELOPER HARNESS (tests only; never loaded by the package or the C-ABI library).
// Compiles the device solver core (mpc_core.hpp) for the HOST so that its logic can be
// debugged against the oracle in a container without a GPU.  It is NOT a CPU fallback:
// nothing in mpc_local_planner_amd/ links or loads this file.
#include <cstring>
#include <vector>

#include "../../include/mpc_hip.h"
#include "../../mpc_local_planner_amd/csrc/mpc_core.hpp"
#include "ipm_serial.hpp"
#include "../../mpc_local_planner_amd/csrc/mpc_problem.hpp"

template <typename T, int MODEL>
static void run(const mpc_config& cfg, int B, const double* x0, const double* xf, const double* up, const double* dtp,
                const double* xi, const double* ui, const double* dti, double* xo, double* uo, double* dto, int* st, int* it,
                double* kkt) {
    mpc::Problem<T> P;
    mpc::fill_problem<T>(cfg, P);
    mpc::Layout L = mpc::Layout::make(cfg.n);
    const int n = cfg.n;
    std::vector<T> ws((size_t)L.total * B, T(0));
    for (int inst = 0; inst < B; ++inst) {
        mpc::Mem<T> M{ws.data() + inst, (long)B};
        mpc::Ipm<T, MODEL> S(P, L, M);
        for (int i = 0; i < 3; ++i) { S.x0[i] = T(x0[3 * inst + i]); S.xf[i] = T(xf[3 * inst + i]); }
        S.x0[2] = mpc::normalize_theta(S.x0[2]);
        S.xf[2] = mpc::normalize_theta(S.xf[2]);
        S.uprev[0] = up ? T(up[2 * inst]) : T(0);
        S.uprev[1] = up ? T(up[2 * inst + 1]) : T(0);
        S.dtprev = dtp ? T(dtp[inst]) : T(0);
        if (xi && ui && dti) {
            for (int k = 0; k < n; ++k) for (int i = 0; i < 3; ++i) M.st(L.X + 3 * k + i, T(xi[(size_t)inst * n * 3 + 3 * k + i]));
            for (int k = 0; k < n - 1; ++k) for (int j = 0; j < 2; ++j) M.st(L.U + 2 * k + j, T(ui[(size_t)inst * n * 2 + 2 * k + j]));
            M.st(L.D, T(dti[inst]));
        } else {
            S.cold_start();
        }
        mpc::SolveStats<T> r = S.solve();
        for (int k = 0; k < n; ++k) for (int i = 0; i < 3; ++i) xo[(size_t)inst * n * 3 + 3 * k + i] = double(M.ld(L.X + 3 * k + i));
        for (int k = 0; k < n; ++k) { int ks = k < n - 1 ? k : n - 2; for (int j = 0; j < 2; ++j) uo[(size_t)inst * n * 2 + 2 * k + j] = double(M.ld(L.U + 2 * ks + j)); }
        dto[inst] = double(M.ld(L.D));
        if (st) st[inst] = r.status;
        if (it) it[inst] = r.iters;
        if (kkt) kkt[inst] = double(r.kkt_error);
    }
}

template <typename T>
static void run_prec(const mpc_config& cfg, int B, const double* x0, const double* xf, const double* up, const double* dtp,
                     const double* xi, const double* ui, const double* dti, double* xo, double* uo, double* dto, int* st, int* it,
                     double* kkt) {
    switch (cfg.model) {
        case 0: run<T, 0>(cfg, B, x0, xf, up, dtp, xi, ui, dti, xo, uo, dto, st, it, kkt); break;
        case 1: run<T, 1>(cfg, B, x0, xf, up, dtp, xi, ui, dti, xo, uo, dto, st, it, kkt); break;
        case 2: run<T, 2>(cfg, B, x0, xf, up, dtp, xi, ui, dti, xo, uo, dto, st, it, kkt); break;
        default: run<T, 3>(cfg, B, x0, xf, up, dtp, xi, ui, dti, xo, uo, dto, st, it, kkt); break;
    }
}

extern "C" int hostdbg_solve(const mpc_config* cfg, int B, const double* x0, const double* xf, const double* up, const double* dtp,
                             const double* xi, const double* ui, const double* dti, double* xo, double* uo, double* dto, int* st,
                             int* it, double* kkt) {
    if (cfg->precision == MPC_FP32) run_prec<float>(*cfg, B, x0, xf, up, dtp, xi, ui, dti, xo, uo, dto, st, it, kkt);
    else run_prec<double>(*cfg, B, x0, xf, up, dtp, xi, ui, dti, xo, uo, dto, st, it, kkt);
    return 0;
}

// trig kernels of the device code (mpc_core.hpp: sincos_reduced / t_tan), exposed for tests/test_host_core.py
extern "C" void hostdbg_trig(int n, const double* x, double* s, double* c, double* t) {
    for (int i = 0; i < n; ++i) { mpc::t_sincos(x[i], &s[i], &c[i]); t[i] = mpc::t_tan(x[i]); }
}
extern "C" void hostdbg_log_mantissa(int n, const double* m, double* out) { for (int i = 0; i < n; ++i) out[i] = mpc::log_mantissa(m[i]); }

// the collocation rows as the kernel's core forms them (mpc_core.hpp: model_trig_colloc + colloc_f; mpc_wave.hpp::eval_point does exactly this):
// c = dt * F(theta_1, u, dt) - (x_2 - x_1), heading difference wrapped.  tests/test_host_core.py compares them with the numpy oracle's
// reference-form collocation rules on the heading manifold, where the two formulations coincide.
template <int MODEL>
static void colloc_rows(const mpc::Problem<double>& P, int count, const double* x1, const double* u, const double* x2, const double* dt, double* c) {
    for (int i = 0; i < count; ++i) {
        double tr[4], tr2[2], f[3];
        const double th = x1[3 * i + 2], v = u[2 * i], w = u[2 * i + 1], d = dt[i];
        mpc::model_trig_colloc<double, MODEL>(P, th, v, w, d, tr, tr2);
        mpc::colloc_f<double, MODEL>(P, tr, tr2, v, w, f);
        c[3 * i] = d * f[0] - (x2[3 * i] - x1[3 * i]);
        c[3 * i + 1] = d * f[1] - (x2[3 * i + 1] - x1[3 * i + 1]);
        c[3 * i + 2] = d * f[2] - mpc::normalize_theta(x2[3 * i + 2] - x1[3 * i + 2]);
    }
}
extern "C" void hostdbg_colloc(const mpc_config* cfg, int count, const double* x1, const double* u, const double* x2, const double* dt, double* c) {
    mpc::Problem<double> P;
    mpc::fill_problem(*cfg, P);
    switch (cfg->model) {
        case 0: colloc_rows<0>(P, count, x1, u, x2, dt, c); break;
        case 1: colloc_rows<1>(P, count, x1, u, x2, dt, c); break;
        case 2: colloc_rows<2>(P, count, x1, u, x2, dt, c); break;
        default: colloc_rows<3>(P, count, x1, u, x2, dt, c); break;
    }
}
extern "C" double hostdbg_normalize_theta(double th) { return mpc::normalize_theta(th); }
// the accept step of the kernel on ONE state vertex, as mpc_wave.hpp::xt / accept() write it: x + alpha dx per component, the heading wrapped (SURVEY.md 8 row a15:
// VectorVertexSE2::plus, include/mpc_local_planner/optimal_control/vector_vertex_se2.h:79-96 -- tests/test_host_core.py holds this to the numpy oracle's retraction)
extern "C" void hostdbg_retract(int count, const double* x, const double* dx, double alpha, double* out) {
    for (int i = 0; i < count; ++i)
        for (int a = 0; a < 3; ++a) {
            double v = x[3 * i + a];
            v += alpha * dx[3 * i + a];
            if (a == 2) v = mpc::normalize_theta(v);
            out[3 * i + a] = v;
        }
}


// mpc_core.hpp::pit_block_inertia_tri on the host (tests/test_pit_math.py): full symmetric 5 x 5 matrices in, excess of negative eigenvalues out (ok = 0 when a pivot vanished)
extern "C" int host_pit_block_inertia(const double* W, const double* P, int* ok_out) {
    double w[15], g[15];
    for (int a = 0; a < 5; ++a) for (int b = a; b < 5; ++b) { w[a * (9 - a) / 2 + b] = W[5 * a + b]; g[a * (9 - a) / 2 + b] = P[5 * a + b]; }
    bool ok = true;
    const int r = mpc::pit_block_inertia_tri(w, g, ok);
    *ok_out = ok ? 1 : 0;
    return r;
}

// mpc_wave_debug.hpp -- the developer diagnostics of the wave kernel behind ONE set of macros: every macro is empty in the product build.
//   -DMPC_ASM_MARK            comment markers in the assembly around the phases of an iteration (scripts/dev/asm_phase_count.py counts instructions between them)
//   -DMPC_PROFILE             per-wave cycle counters per phase, read back through mpc_debug_profile (scripts/dev/phase_profile.py)
//   -DMPC_PIT_CHECK=<blocks>  both sweeps on every factorisation of the first <blocks> workgroups, differences printed
//   -DMPC_NANCHECK=<block>    non-finite words per field of that workgroup's record after every sweep, line-search traces
// The macros expand INSIDE IpmWave::solve() and use its local names.
#pragma once

#ifdef MPC_ASM_MARK
#define MPC_MARK(name) asm volatile("; " name)
#else
#define MPC_MARK(name) do { } while (0)
#endif

#ifdef MPC_PROFILE
#define MPC_PROFILE_BEGIN \
        long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int nfac = 0, ntrial = 0; \
        const long long t_begin = __builtin_readcyclecounter(); \
        const long long w_begin = wall_clock64();
#define MPC_TICK(i, stmt) { long long t0_ = __builtin_readcyclecounter(); stmt; tk[i] += __builtin_readcyclecounter() - t0_; }
#define MPC_PROFILE_COUNT(c) ++c
#define MPC_PROFILE_END \
        if (lane == 0 && blockIdx.x < 4096) { \
            long long* o = g_mpc_prof[blockIdx.x]; \
            o[0] = __builtin_readcyclecounter() - t_begin; o[1] = wall_clock64() - w_begin; o[2] = it; o[3] = nfac; o[4] = ntrial; \
            for (int i = 0; i < 8; ++i) o[5 + i] = tk[i]; \
            o[13] = prof_loop; o[14] = prof_setup; o[15] = prof_fwd_loop; if (prof_mult) o[10] = prof_mult;      /* (-DMPC_PROFILE_MULT: the multiplier recurrence instead of logs0) */ \
        }
#else
#define MPC_PROFILE_BEGIN
#define MPC_TICK(i, stmt) { stmt; }
#define MPC_PROFILE_COUNT(c) do { } while (0)
#define MPC_PROFILE_END
#endif

#ifdef MPC_PIT_CHECK
#define MPC_DBG_PIT_CHECK \
                if (pit && (int)blockIdx.x < MPC_PIT_CHECK) { \
                    T dd1 = T(0), nu1[3] = {T(0), T(0), T(0)}, dd2 = T(0), nu2[3] = {T(0), T(0), T(0)}; \
                    const int g1c = backward_pit(delta, dc, dd1, nu1); sync(); \
                    const bool g1 = g1c > 0; \
                    if (g1) { forward_pit(dd1, nu1, delta); sync(); } \
                    T keepx = T(0), keepu = T(0), keepl = T(0); \
                    const int kk = lane < L.n - 1 ? lane : 0; \
                    keepx = F(L.DX, 2, kk); keepu = F(L.DU, 1, kk); keepl = F(L.LAMN, 2, kk); \
                    sync(); \
                    const bool g2 = backward_dpp(delta, dc, dd2, nu2); sync(); \
                    if (g2) { forward_states(dd2, nu2, delta); sync(); } \
                    const T ex = wave_max(t_abs(keepx - F(L.DX, 2, kk))), eu = wave_max(t_abs(keepu - F(L.DU, 1, kk))), el = wave_max(t_abs(keepl - F(L.LAMN, 2, kk))); \
                    const T sx = wave_max(t_abs(F(L.DX, 2, kk))), su = wave_max(t_abs(F(L.DU, 1, kk))), sl = wave_max(t_abs(F(L.LAMN, 2, kk))); \
                    if (lane == 0 && (g1 != g2 || ex > T(1e-7) * (sx + T(1e-3)) || eu > T(1e-7) * (su + T(1e-3)) || el > T(1e-7) * (sl + T(1e-3)))) \
                        printf("blk %d it %d try %d delta %.2e mu %.1e: pit ok %d (code %d) serial ok %d | dd %.9e vs %.9e | max diff dx %.2e (of %.2e) du %.2e (of %.2e) lam %.2e (of %.2e)\n", (int)blockIdx.x, it, ntry, (double)delta, (double)mu, \
                               (int)g1, g1c, (int)g2, (double)dd1, (double)dd2, (double)ex, (double)sx, (double)eu, (double)su, (double)el, (double)sl); \
                    sync(); \
                }
#else
#define MPC_DBG_PIT_CHECK
#endif

#ifdef MPC_NANCHECK
#define MPC_DBG_NANCHECK_FIELDS \
                if (blockIdx.x == MPC_NANCHECK && lane == 0) { \
                    const int offs[] = {L.X, L.U, L.LAM, L.LAMN, L.SR, L.YR, L.PL, L.PU, L.DX, L.DU, L.CC, L.TRIG, L.GAIN, L.STG, L.SC, L.VP, L.ZC, L.total}; \
                    const char* nm[] = {"X", "U", "LAM", "LAMN", "SR", "YR", "PL", "PU", "DX", "DU", "CC", "TRIG", "GAIN", "STG", "SC", "VP", "ZC"}; \
                    printf("it %d try %d good %d delta %g dd %g nu %g %g %g mu %g |", it, ntry, (int)good, (double)delta, (double)dd, (double)nu[0], (double)nu[1], (double)nu[2], (double)mu); \
                    for (int f = 0; f < 17; ++f) { \
                        int cnt = 0, first = -1; \
                        for (int q = offs[f]; q < offs[f + 1]; ++q) if (!t_finite(sm[q])) { ++cnt; if (first < 0) first = q - offs[f]; } \
                        if (cnt) printf(" %s:%d@%d", nm[f], cnt, first); \
                    } \
                    printf("\n"); \
                }
#define MPC_DBG_NANCHECK_STEP \
                    if (blockIdx.x == MPC_NANCHECK) { \
                        for (int k = lane; k < L.n; k += kWave) { \
                            bool f = t_finite(F(L.DX, 0, k)) && t_finite(F(L.DX, 1, k)) && t_finite(F(L.DX, 2, k)); \
                            if (k < L.n - 1) f = f && t_finite(F(L.DU, 0, k)) && t_finite(F(L.DU, 1, k)) && t_finite(F(L.LAMN, 0, k)) && t_finite(F(L.LAMN, 1, k)) && t_finite(F(L.LAMN, 2, k)); \
                            if (!f) printf("it %d try %d fwd nonfinite at k %d: dx %g %g %g du %g %g lamn %g %g %g\n", it, ntry, k, (double)F(L.DX, 0, k), (double)F(L.DX, 1, k), (double)F(L.DX, 2, k), (double)F(L.DU, 0, k), (double)F(L.DU, 1, k), (double)F(L.LAMN, 0, k), (double)F(L.LAMN, 1, k), (double)F(L.LAMN, 2, k)); \
                        } \
                    }
#define MPC_DBG_NANCHECK_LS_FAILED \
            if (blockIdx.x == MPC_NANCHECK && lane == 0 && !accepted) \
                printf("   ls FAILED: it %d phi0 %.12e Dm %.6e rho %.6e mu %g theta %.6e theta_c %.6e fobj %.9f logs_cur %.9f | last alpha %g f_t %.9f th_t %.6e lg_t %.9f dzmax %g a_p %g\n", \
                       it, (double)phi0, (double)Dm, (double)rho, (double)mu, (double)theta, (double)theta_c, (double)fobj, (double)logs_cur, (double)alpha, (double)f_t, (double)th_t, (double)lg_t, (double)fw.dzmax, (double)fw.a_p);
#define MPC_DBG_NANCHECK_LS \
            if (blockIdx.x == MPC_NANCHECK && lane == 0) \
                printf("   ls: it %d f %.6f -> %.6f theta_c %.3e alpha %.4f a_p %.4f a_d %.4f rho %.3e Dm %.4e hdz %.6e clam %.6e dz2 %.6e dphi %.6e\n", it, (double)fobj, (double)f_t, (double)th_t, (double)alpha, (double)fw.a_p, (double)fw.a_d, (double)rho, (double)Dm, (double)fw.hdz, (double)fw.clam, (double)fw.dz2, (double)fw.dphi);
#else
#define MPC_DBG_NANCHECK_FIELDS
#define MPC_DBG_NANCHECK_STEP
#define MPC_DBG_NANCHECK_LS_FAILED
#define MPC_DBG_NANCHECK_LS
#endif

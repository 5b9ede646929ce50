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
//       trial evaluation (register-resident), acceptance.  Scalars are combined with DPP wavefront reductions.
//   serial over the horizon, parallel INSIDE a stage (registers + DP-ALU DPP broadcasts, no LDS hand-offs):
//       backward Riccati sweep over the augmented stage state (x_k, u_{k-1}, dt): lane c owns column c of the
//       value block and of the stage Hessian (backward_dpp); forward state recurrence: lane c owns component c
//       of (dx, du) (forward_states); the costate (multiplier) recurrence is two wave suffix scans.
//       The backward sweep also counts the negative eigenvalues of its pivots: a factorisation is accepted on its inertia, as Ipopt does it (mpc_core.hpp::riccati_root).
//   partitioned over the four 16-lane DPP rows (four time segments at once, backward_pit / forward_pit) while the barrier parameter is above pit_floor().
// The arithmetic is the same as mpc_core.hpp (lane-per-instance variant); see that file for the
// reference citations of every formula.
//
// File map (r05: one struct, six parts -- the .inc files are included INSIDE struct IpmWave):
//   mpc_wave_layout.hpp   WaveLayout / FixedLayout / GlobalStage, wavefront reductions
//   mpc_wave.hpp          this file: members, accessors, sweep-pointer abstraction (LDS or global memory)
//   mpc_wave_rows.inc     terminal ball, via-points, clearance rows (geometry, footprints, association)
//   mpc_wave_passes.inc   point evaluation, line-search trials, KKT error + stage records, condensed barrier terms
//   mpc_wave_sweeps.inc   serial Riccati sweeps, multiplier recurrence
//   mpc_wave_pit.inc      partitioned (parallel-in-time) sweeps
//   mpc_wave_step.inc     slack / dual steps, fraction to the boundary, acceptance, initial point, seeds, kept multipliers
//   mpc_wave_solve.inc    the interior-point loop
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "mpc_core.hpp"
#include "mpc_dpp_blocks.inc"

#ifdef MPC_PROFILE
__device__ long long g_mpc_prof[4096][16];
#endif

#include "mpc_wave_layout.hpp"
#include "mpc_wave_debug.hpp"

namespace mpc {

// EXT = false compiles the rarely used rows / objective terms (terminal l2-ball, via-points) out of the kernel: the headline
// configurations keep their instruction count and register budget
// EXT: 0 = headline instantiation, 1 = + the rarely used rows / terms / coupling slots, 2 = + the cost variants (off-diagonal weights, trapezoidal rule)
// OBST = false compiles every clearance-row path out (solvers created without obstacles: the headline configurations): less code, and the two dozen layout
// words of the obstacle arrays leave the scalar registers (the headline kernel spilled ~430 of them)
// NSC > 0: the layout is a compile-time constant for the stride NS = NSC (FixedLayout; needs EXT == 0, OBST == false, no Crank-Nicolson trig words)
// GS: the factorisation data (STG, GAIN) in global memory instead of LDS (GlobalStage): same arithmetic, same results bit for bit, a third of the LDS record
template <typename T, int MODEL, int EXT = 1, bool OBST = true, int NSC = 0, bool GS = false, bool W2 = false>
struct IpmWave {
    static constexpr int NSTG = EXT ? NSTG_EXT : NSTG_BASE;        // words per stage record
    // trig-cache words per stage: sin, cos, steering term(s); Crank-Nicolson appends sin/cos of its second evaluation angle
    static constexpr int NTRB = (MODEL == MODEL_KINEMATIC_BICYCLE || MODEL == MODEL_SIMPLE_CAR_FRONT) ? 4 : 3;
    static_assert(NSC == 0 || (EXT == 0 && !OBST), "the fixed layout exists for the headline instantiation only");
    static_assert(!GS || NSC == 0, "the fixed layout keeps its factorisation data in LDS");
    using LayoutT = typename LayoutOf<NSC, NTRB, NSTG>::type;
    const Problem<T>& P;     // lives in LDS (copied once per workgroup): wave-uniform constants are fetched with
    const LayoutT L;         // broadcast ds_reads instead of being pinned in (and spilled from) scalar registers; the layout
                             // (45 small ints, used by every accessor) is held by value = in scalar registers -- or is a compile-time constant (NSC > 0)
    T* sm;
    T* gmb = nullptr;        // GS: this workgroup's block of global memory (GlobalStage), wave-uniform
    const int lane;
    T x0[3], xf[3], uprev[2], dtprev;
    T mu, rho, delta_last;
    T erho = T(0);           // > 0: the clearance rows are elastic with this penalty (restoration mode, see solve()); wave-uniform
    bool row0_on, fail0;
    bool warm_guess = false;     // the caller supplied an initial guess (second and later control cycles)
    mutable int cnt_mult = -1, cnt_bmult = -1;      // number of equality / bound multipliers (cached by kkt_pass)
    mutable T inv_cnt_mult = T(1), inv_cnt_bmult = T(1);   // 1 / max(count, 1)
    int nvia = 0;   // via-points of this instance
    int flags;      // bits 0..2 xf_fixed, 3 dt_free, 4 quadratic objective, 5 has_Qf, 6..9 rate_on, 10 terminal ball, 11 via-points, 12 footprint that turns with the pose (line, two circles), 13 integral form with dt free, 14 dynamic obstacles: the problem record lives in LDS and every
                    // P.x costs a ds_read (+ wait) that the compiler cannot hoist over LDS stores; one scalar register holds the switches
#ifdef MPC_PROFILE
    mutable long long prof_mult = 0;
    mutable long long prof_loop = 0, prof_setup = 0, prof_fwd_loop = 0;    // ticks inside the backward stage loop / before it / inside the forward loop
#endif
    int nfix;
    // candidate initial trajectories: this wave's candidate index, its iteration cap and the instance's winner word (global memory; NULL when
    // the solver runs a single candidate).  A lower winner index than ours = a higher-priority candidate has converged: we stop.
    int my_cand = 0, iter_cap = 0;
    int rows_dropped = 0;        // clearance rows that did not fit into max_obstacle_rows (associate_obstacles)
    const double* dual_in = nullptr;   // multipliers of this instance's last converged solve (handle state; dual_warm_start) or NULL
    const int* win_ptr = nullptr;

    __device__ IpmWave(const Problem<T>& p, const WaveLayout& l, T* s, int ln) : P(p), L(LayoutOf<NSC, NTRB, NSTG>::from(l)), sm(s), lane(ln) {}

    // ---- LDS accessors: component-major, stage-minor (conflict-free for lane == stage)
    __device__ __forceinline__ T& F(int base, int comp, int k) const { return sm[base + comp * L.NS + k]; }
    // stage records: stage-major in LDS
    // (GS: the same records in the workgroup's global block, in tiles of four stages (GlobalStage) -- byte offsets are formed in 32 bits and
    //  zero-extended, so that the accesses compile to global_load / global_store with the block's base in a scalar register pair)
    typedef __attribute__((address_space(1))) T GlbT;
    typedef __attribute__((address_space(1))) char GlbC;
    using SwT = std::conditional_t<GS, GlbT, T>;                         // a word of the sweeps' storage class
    __device__ __forceinline__ GlbT& gw(unsigned word) const { return *(GlbT*)((GlbC*)gmb + (size_t)(word * (unsigned)sizeof(T))); }
    // entry e of stage k's tile slot (GlobalStage): [0, NSTG) the stage record, then c_k, the constants 0 0 0 1 0 0
    static constexpr int TE_CC = NSTG, TE_Z = NSTG + 3, TNT = GlobalStage::nt(NSTG);
    __device__ __forceinline__ int tile_k(int k) const { return GlobalStage::TILE(L.NS) + ((k + GlobalStage::kGuard) >> 2) * (4 * TNT) + ((k + GlobalStage::kGuard) & 3); }      // word of entry 0 of stage k
    __device__ __forceinline__ GlbT& TL_(int e, int k) const { return gw((unsigned)(tile_k(k) + 4 * e)); }
    __device__ __forceinline__ SwT& G_(int i, int k) const { if constexpr (GS) return gw((unsigned)(GlobalStage::GAIN + k * NGAIN + i)); else return sm[L.GAIN + k * NGAIN + i]; }
    static constexpr int NADDv = EXT ? (int)NADD : (int)NADD_BASE;
    __device__ __forceinline__ SwT& S_(int i, int k) const { if constexpr (GS) return TL_(i, k); else return sm[L.STG + k * NSTG + i]; }
    __device__ __forceinline__ T& C_(int i, int k) const { return sm[L.CC + k * 3 + i]; }
    // cached value (0), gradient (1, 2) and curvature (3) of clearance row m at grid point k: written by kkt_pass, read by the other lane-parallel passes
    __device__ __forceinline__ SwT& OB_(int which, int m, int k) const {
        if constexpr (GS) return gw((unsigned)(GlobalStage::OBC(L.NS, NSTG) + (which * L.M + m) * L.NS + k)); else return sm[(which == 0 ? L.OG : (which == 1 ? L.OAX : (which == 2 ? L.OAY : L.OHK))) + m * L.NS + k];
    }
    // third-variable caches of clearance row m at grid point k -- which: 0 OAT (gradient part), 1 OHXT 2 OHYT 3 OHTT (curvature parts) of the heading (footprints that turn with the
    // pose) or of dt (dynamic obstacles); 4 OAD 5 OHXD 6 OHYD 7 OHDD 8 OHTD the dt parts when both apply.  LDS arrays, or (GS, r06) regions of the workgroup's global block
    __device__ __forceinline__ SwT& OX_(int which, int m, int k) const {
        if constexpr (GS) return gw((unsigned)(L.OXB + ((which < 4 ? which * L.MT : 4 * L.MT + (which - 4) * L.MD) + m) * L.NS + k));
        else {
            const int base = which == 0 ? L.OAT : (which == 1 ? L.OHXT : (which == 2 ? L.OHYT : (which == 3 ? L.OHTT : (which == 4 ? L.OAD : (which == 5 ? L.OHXD : (which == 6 ? L.OHYD : (which == 7 ? L.OHDD : L.OHTD)))))));
            return sm[base + m * L.NS + k];
        }
    }
    // elastic variable e (0) and its step de (1) of clearance row m at grid point k (restoration mode): always in the workgroup's global block
    __device__ __forceinline__ GlbT& OE_(int which, int m, int k) const { return gw((unsigned)(L.OEB + (which * L.M + m) * L.NS + k)); }
    // c^_k = c_k + f_k dd, what the forward sweeps read (component-major): parked in LAMN in both forms
    __device__ __forceinline__ T& CH_(int i, int k) const { return sm[L.LAMN + i * L.NS + k]; }
    __device__ __forceinline__ T& SCL(int i) const { return sm[L.SC + i]; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ int nM() const { return OBST ? L.M : 0; }      // clearance rows per grid point
    // obstacle index of clearance row m at grid point k (-1 = no row): 16-bit words, component-major like the T arrays
    __device__ __forceinline__ unsigned short* oi_base() const { return reinterpret_cast<unsigned short*>(sm + L.OI); }
    __device__ __forceinline__ int oi(int m, int k) const { const unsigned v = oi_base()[m * L.NS + k]; return v == 0xffffu ? -1 : (int)v; }
    __device__ __forceinline__ void set_oi(int m, int k, int j) const { oi_base()[m * L.NS + k] = (unsigned short)(j < 0 ? 0xffff : j); }
    __device__ __forceinline__ bool fx(int i) const { return (flags >> i) & 1; }
    __device__ __forceinline__ bool dtf() const { return (flags >> 3) & 1; }
    __device__ __forceinline__ bool quad() const { return (flags >> 4) & 1; }
    __device__ __forceinline__ bool hasqf() const { return (flags >> 5) & 1; }
    // cost variants: minimum-time term in the objective (minimum-time objectives and the hybrid quadratic form); off-diagonal weights /
    // trapezoidal rule (EXT instantiation only: everything below `costx()` is delta code on top of the diagonal left-sum arithmetic)
    __device__ __forceinline__ bool mintime() const { return (flags >> 16) & 1; }
    __device__ __forceinline__ bool costx() const { return EXT >= 2; }      // that instantiation is only launched for such problems (mpc_capi.hip::solver_ext)
    // off-diagonal part of W x and of x' W x for the symmetric matrix with off-diagonal terms o = (01, 02, 12)
    __device__ __forceinline__ void offmul(const T o[3], const T x[3], T y[3]) const {
        y[0] = o[0] * x[1] + o[1] * x[2]; y[1] = o[0] * x[0] + o[2] * x[2]; y[2] = o[1] * x[0] + o[2] * x[1];
    }
    __device__ __forceinline__ T offquad(const T o[3], const T x[3]) const { return T(2) * (o[0] * x[0] * x[1] + o[1] * x[0] * x[2] + o[2] * x[1] * x[2]); }
    __device__ __forceinline__ T fullquad(const T dg[3], const T o[3], const T x[3]) const { return dg[0] * x[0] * x[0] + dg[1] * x[1] * x[1] + dg[2] * x[2] * x[2] + offquad(o, x); }
    // error of the final state at z + alpha dz (a fixed component sits on the goal: 0)
    __device__ __forceinline__ void xd_final(T alpha, T xd[3]) const {
        for (int i = 0; i < 3; ++i) { xd[i] = fx(i) ? T(0) : xt(i, L.n - 1, alpha) - xf[i]; }
        xd[2] = normalize_theta(xd[2]);
    }
    // delta of the objective at the final state against the diagonal left-sum arithmetic: off-diagonal terminal cost + trapezoid term
    __device__ __forceinline__ T final_cost_extra(T alpha, T d) const {
        T xd[3]; xd_final(alpha, xd);
        T f = T(0);
        if (hasqf()) f += offquad(P.Qfo, xd);
        if (P.trapz) f += T(0.5) * d * fullquad(P.Q, P.Qo, xd);
        return f;
    }
    __device__ __forceinline__ bool ron(int q) const { return (flags >> (6 + q)) & 1; }
    __device__ __forceinline__ bool ball() const { return EXT && ((flags >> 10) & 1); }
    __device__ __forceinline__ bool via() const { return EXT && ((flags >> 11) & 1); }
    __device__ __forceinline__ bool fpline() const { return EXT && ((flags >> 12) & 1); }
    __device__ __forceinline__ bool intf() const { return EXT && ((flags >> 13) & 1); }     // integral-form cost, dt free
    __device__ __forceinline__ bool dynobs() const { return EXT && ((flags >> 14) & 1); }
    __device__ __forceinline__ bool hessm() const { return EXT && ((flags >> 15) & 1); }      // convexified Hessian
    // explicit LDS pointers for the running-pointer loops (address-space inference gives up on per-lane selected pointers)
    typedef __attribute__((address_space(3))) T LdsT;
    __device__ __forceinline__ LdsT* lds(int word) const { return (LdsT*)sm + word; }
    // ---- running pointers of the sweeps, in the sweeps' storage class: an LDS pointer, or (GS) a BYTE offset into the workgroup's global block.  W_*(): where the
    //      regions the sweeps stream through start (word index in that storage class); sw_step(): a stride in the units such a pointer advances by
    using SwRef = std::conditional_t<GS, unsigned, LdsT*>;
    using SwCRef = std::conditional_t<GS, unsigned, const LdsT*>;
    __device__ __forceinline__ int W_ZC() const { if constexpr (GS) return GlobalStage::ZC; else return L.ZC; }
    // LDS form: entry i of stage 0's record and the distance between stages, for the arrays the sweeps stream through.  (GS: the gains alike -- stage-major in both forms --; the
    // stage records, c_k, c^_k and the constant triples are entries of a stage's tile slot, addressed as tile_k(k) + 4 e)
    __device__ __forceinline__ int STG_W(int i) const { return L.STG + i; }
    __device__ __forceinline__ int GAIN_W(int i) const { if constexpr (GS) return GlobalStage::GAIN + i; else return L.GAIN + i; }
    __device__ __forceinline__ int CC_W(int i) const { return L.CC + i; }
    __device__ __forceinline__ int CH_W(int i) const { return L.LAMN + i * L.NS; }      // c^_k of the forward sweeps (LDS form: parked in LAMN, component-major)
    static constexpr int STG_S = NSTG, GAIN_S = NGAIN, CC_S = 3;
    // constant coefficient triple of the backward sweeps (LDS form, read with stride 0): kind 0 (0,0,0)  1 (1,0,0)  2 (0,1,0); in the tile slot's constants 0 0 0 1 0 0 the same triples start at entries 0, 3, 2
    __device__ __forceinline__ int ZT_W(int kind) const { return L.ZC + (kind == 1 ? 4 : (kind == 2 ? 3 : 0)); }
    __device__ __forceinline__ static constexpr int zt_entry(int kind) { return TE_Z + (kind == 1 ? 3 : (kind == 2 ? 2 : 0)); }
    // where the idle lanes of a gain store go: the record behind the last stage / the sweep scratch
    __device__ __forceinline__ int GAIN_DUMMY_W() const { if constexpr (GS) return GlobalStage::GAIN + L.NS * NGAIN; else return L.VP; }
    // a word of the global block at a byte offset that is the sum of a (wave-uniform or per-row) stage part and a per-lane entry part
    __device__ __forceinline__ T gld(unsigned stage_bytes, unsigned lane_bytes) const { return *(const GlbT*)((const GlbC*)gmb + (size_t)stage_bytes + (size_t)lane_bytes); }
    __device__ __forceinline__ SwRef sw(int word) const { if constexpr (GS) return (unsigned)word * (unsigned)sizeof(T); else return lds(word); }
    __device__ __forceinline__ static constexpr int sw_step(int words) { return GS ? words * (int)sizeof(T) : words; }
    // word i (a compile-time constant at every call site) behind a running pointer / the word `step` pointer units behind it / entry i of the record the pointer is in
    __device__ __forceinline__ T sw_ld(SwCRef p, int i = 0) const { if constexpr (GS) return *(const GlbT*)((const GlbC*)gmb + (size_t)p + (size_t)(i * (int)sizeof(T))); else return p[i]; }
    __device__ __forceinline__ T sw_ld_at(SwCRef p, int step) const { if constexpr (GS) return *(const GlbT*)((const GlbC*)gmb + (size_t)(p + (unsigned)step)); else return p[step]; }
    __device__ __forceinline__ void sw_st(SwRef p, int i, T v) const { if constexpr (GS) *(GlbT*)((GlbC*)gmb + (size_t)p + (size_t)(i * (int)sizeof(T))) = v; else p[i] = v; }

    // trial point z + alpha*dz, evaluated on the fly (no trial copy in LDS)
    // (alpha == 0 must not touch the step arrays: they are unwritten before the first factorisation, and 0 * garbage can be NaN)
    __device__ __forceinline__ T xt(int i, int k, T alpha) const {
        T x = F(L.X, i, k);
        if (alpha != T(0) && k > 0 && (k < L.n - 1 || !fx(i))) { x += alpha * F(L.DX, i, k); if (i == 2) x = normalize_theta(x); }
        return x;
    }
    __device__ __forceinline__ T ut(int j, int k, T alpha) const {
        T u = F(L.U, j, k);
        if (alpha != T(0)) u += alpha * F(L.DU, j, k);
        return u;
    }

    __device__ __forceinline__ bool row_on(int r, int q) const { return ron(q) && (r > 0 || row0_on); }

    // rate row r, slot q at controls from base UB and dt d (solver form, <= 0 feasible)
    __device__ __forceinline__ T row_val(int UB, T d, int r, int q) const {
        const int n = L.n, j = q & 1;
        T ur = r < n - 1 ? F(UB, j, r) : T(0);
        const T up0 = uprev[0], up1 = uprev[1];      // (constant indices and a value select: a pointer select between LDS and this object would put the object into scratch memory)
        T um = j ? up1 : up0;
        if (r > 0) um = F(UB, j, r - 1);
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

    // lane index that the optimiser must treat as unknown HERE: keeps the per-lane address arithmetic of a phase inside the phase (hoisted out of the
    // interior-point loop as loop invariants it would occupy registers for the whole solve)
    __device__ __forceinline__ int local_lane() const { int l = lane; asm volatile("" : "+v"(l)); return l; }
    // MPC_PHASE_LANE opens every phase of an iteration: in the two-waves-per-SIMD kernels (W2: 256 registers) the phase works on such a lane index of its own
#define MPC_PHASE_LANE const int lane = W2 ? this->local_lane() : this->lane; (void)lane;
    // A wave-uniform fp64 value that lives across phases of an iteration occupies TWO vector registers in every lane; read back through v_readfirstlane it is a scalar register
    // pair (which the compiler parks in a lane of a spill register when it runs out: 1/32 of the space), and the phases in between have the vector registers for themselves.
    // The 256-register kernels (W2) do that with the solve loop's scalars -- some sixty of them: barrier parameter, penalty, step data, the KKT error's pieces --, which is where
    // their scratch traffic came from (profiles/r06_wave_kernel_n20_two_waves.md).  Pure copies: results unchanged.
    static constexpr bool kUniformScalars = W2
#ifdef MPC_UNIFORM_SCALARS_ALL      // developer A/B: the same in the one-wave kernels
        || true
#endif
        ;
    __device__ __forceinline__ static double uni_(double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); }
    __device__ __forceinline__ static float uni_(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
    // pow() of the device library is some 300 instructions and three dozen fp64 literals; inlined at its two (rarely executed) call sites of the solve loop the compiler
    // materialises those literals ONCE in front of the loop and -- in the 256-register kernels -- spills them to scratch for the whole solve.  There it is a call.
    __device__ __attribute__((noinline)) static T pow_cold(T a, T b) { return t_pow(a, b); }
    __device__ __forceinline__ static T pow_(T a, T b) { if constexpr (W2) return pow_cold(a, b); else return t_pow(a, b); }
    __device__ __forceinline__ static void U(T& v) { if constexpr (kUniformScalars) v = uni_(v); }
    template <typename... Ts> __device__ __forceinline__ static void U(T& v, Ts&... rest) { U(v); U(rest...); }

#include "mpc_wave_rows.inc"
#include "mpc_wave_passes.inc"
#include "mpc_wave_sweeps.inc"
#include "mpc_wave_pit.inc"
#include "mpc_wave_step.inc"
#include "mpc_wave_solve.inc"
#undef MPC_PHASE_LANE
};

}  // namespace mpc

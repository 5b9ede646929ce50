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
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "mpc_core.hpp"
#include "mpc_dpp_blocks.inc"

#ifdef MPC_PROFILE
__device__ long long g_mpc_prof[4096][16];
#endif

namespace mpc {

constexpr int kWave = 64;
// per-stage LQ record: 0..2 a0 a1 1 | 3..5 f | 6..8 Bx[:,0] | 9..11 Bx[:,1] | 12.. combined stage cost A[StageAdd] (27 entries, 31 with the extra coupling slots)
constexpr int RA = 12;     // first A slot (words 0..11: a0 a1 1 | f | Bx[:,0] | Bx[:,1])
constexpr int NSTG_EXT = RA + NADD;         // 43: record of the kernel instantiation with the extra coupling slots (A02 A12 A05 A15)
constexpr int NSTG_BASE = RA + NADD_BASE;   // 39: record of the headline kernel (odd strides: conflict-free for lane == stage)
constexpr int NGAIN = 24;  // negated gains: nK0(6) nkappa0 nKnu0(5) | nK1(6) nkappa1 nKnu1(5)   (5 border columns: the partitioned sweep's segments end in the
                           // costate of (x, u_prev); the serial sweep and the last segment use the first 3 = the fixed goal components)
constexpr int NGH = NGAIN / 2;

// Factorisation data in global memory (IpmWave<..., GS = true>): what the Riccati sweeps stream through -- the stage records STG, the gains GAIN, and copies of
// the little else their running pointers touch (the constant triples ZC, the residuals c_k, the folded residuals c^_k, a dummy store target) -- sits in ONE block of
// global memory per workgroup, stage-major exactly like the LDS arrays it replaces, so that the sweeps' pointer arithmetic is the same in both storage classes.
// The LDS record shrinks from 97 to 34 words per grid point (n = 120 in fp64: 95 KB -> 33 KB, four workgroups per CU instead of one); the block is written and
// re-read by the same CU within one interior-point iteration (L2 / Infinity-Cache resident: 63 n words per resident wave).  Word offsets inside the block:
struct GlobalStage {
    static constexpr int ZC = 0;          // 8 words: constants 0 0 0 0 1 0 0 0
    static constexpr int VP = 8;          // 16 words: dummy store targets of the idle lanes
    static constexpr int CC = 32;         // 3 NS words, stage-major: c_k (copy of the LDS array, written by kkt_pass)
    __host__ __device__ static constexpr int CH(int ns) { return CC + 3 * ns; }            // 3 NS words, component-major: c^_k = c_k + f_k dd (forward sweeps)
    __host__ __device__ static constexpr int GAIN(int ns) { return CC + 6 * ns; }          // NGAIN NS words, stage-major
    __host__ __device__ static constexpr int STG(int ns) { return CC + (6 + NGAIN) * ns; }    // nstg NS words, stage-major
    __host__ __device__ static constexpr int OBC(int ns, int nstg) { return STG(ns) + nstg * ns; }   // 4 M NS words, component-major [OG | OAX | OAY | OHK][m][k]: the clearance rows' cached value,
                                                                                                     // gradient and curvature (touched by the lane-parallel passes only: coalesced)
    __host__ __device__ static constexpr int OEL(int ns, int nstg, int M) { return OBC(ns, nstg) + 4 * M * ns; }   // 2 M NS words [OE | ODE][m][k]: the elastic variables of the clearance rows and
                                                                                                     // their steps (restoration mode, IpmWave::solve)
    __host__ __device__ static constexpr int words(int ns, int nstg, int M) { return ((OEL(ns, nstg, M) + 2 * M * ns + 15) / 16) * 16; }     // (128-byte multiple in fp64)
    // a layout that keeps its factorisation data in LDS still has a block when it has clearance rows: the elastic arrays alone (touched by the lane-parallel passes only, and
    // only in the restoration mode: not worth 2 M words of LDS per grid point)
    static constexpr int OEL_ONLY = 16;
    __host__ __device__ static constexpr int words_elastic_only(int ns, int M) { return ((OEL_ONLY + 2 * M * ns + 15) / 16) * 16; }
};

struct WaveLayout {
    int n, NS;
    int NTR;                                      // trig-cache words per stage (3, or 4 for the bicycle / front-wheel car; +2 for Crank-Nicolson)
    int X, U, LAM, LAMN, SR, YR, PL, PU, DX, DU, CC, TRIG, GAIN, STG, SC, VP, ZC, ZI, total;
    int M, O, V;                                  // clearance rows per grid point, obstacles, vertices per obstacle
    int OS, OY, OI, OG, OAX, OAY, OHK;            // per-row slack, multiplier, obstacle index, cached g, gradient, curvature
    int GV, GNV, GR, GC;                          // obstacle geometry: vertices, vertex counts, radii, centroids
    int OAT, OHXT, OHYT, OHTT;                    // third-variable parts of the clearance rows: heading (footprints that turn with the pose) or
                                                  // dt (dynamic obstacles); MT = M when either is configured, else 0 words
    int OAD, OHXD, OHYD, OHDD, OHTD;              // dt parts when BOTH apply (dynamic obstacles + a turning footprint): gradient, hess [x dt, y dt, dt dt, theta dt]
    int GVEL;                                     // obstacle velocities (dynamic obstacles; 2 * OD words)
    int NV, VIA, VIDX;                            // via-points: capacity, poses (x, y, theta), attached grid point (-1 = skipped)
    int GSW;                                      // > 0: the workgroup has a block of GSW words of GLOBAL memory (GlobalStage): the elastic arrays of the clearance rows, and with GSF the factorisation data
    int GSF;                                      // 1: the factorisation data (GAIN, STG) and the clearance rows' caches live in that block instead of LDS (IpmWave<..., GS = true>)
    int OEB;                                      // word offset of the elastic arrays [OE | ODE] inside the block
    // tsize = sizeof(T) of the kernel that uses the layout (the obstacle indices of the clearance rows are 16-bit words, M * n of them, packed into T-sized words)
    __host__ __device__ static constexpr WaveLayout make(int n, int M = 0, int O = 0, int V = 1, int ntrig = 4, int NV = 0, int MT = 0, int OD = 0, int nstg = NSTG_BASE, int MD = 0, int tsize = 8, bool gs = false) {
        WaveLayout L{};
        L.n = n;
        L.NS = n;
        int o = 0;
        auto take = [&o, n](int comps) constexpr { int b = o; o += comps * n; return b; };
        L.NTR = ntrig;
        L.X = take(3); L.U = take(2);
        L.LAM = take(3); L.LAMN = take(3);
        L.SR = take(4); L.YR = take(4);
        L.PL = take(2); L.PU = take(2);
        L.DX = take(3); L.DU = take(2);
        L.CC = take(3); L.TRIG = take(ntrig);
        L.GAIN = take(gs ? 0 : NGAIN); L.STG = take(gs ? 0 : nstg);
        L.GSW = gs ? GlobalStage::words(n, nstg, M) : (M > 0 ? GlobalStage::words_elastic_only(n, M) : 0);
        L.GSF = gs ? 1 : 0;
        L.OEB = gs ? GlobalStage::OEL(n, nstg, M) : GlobalStage::OEL_ONLY;
        L.SC = o; o += 16;    // scalars: D, DT, DD, PDL, PDU | terminal-ball row: slack, multiplier, cached value and gradient
        L.VP = o; o += 16;    // dummy store targets of the idle lanes in the sweeps
        L.ZC = o; o += 8;     // constants 0 0 0 0 1 0 0 0 (coefficient triples of the constant columns)
        L.ZI = o; o += 12;    // constants 0 0 0 0 0 0 1 0 0 0 0 0: the unit vector e_c (6 words) starts at ZI + 6 - c, six zeros at ZI (partitioned sweep)
        L.M = M; L.O = O; L.V = V;
        L.OS = take(M); L.OY = take(M);
        L.OI = o; o += (M * n * 2 + tsize - 1) / tsize;      // uint16 per row and grid point (0xffff = no row): a quarter of a T word each -- what lets BASELINE configs[2] (n = 80, 16 polygons) keep TWO workgroups per CU
        L.OG = take(gs ? 0 : M); L.OAX = take(gs ? 0 : M); L.OAY = take(gs ? 0 : M); L.OHK = take(gs ? 0 : M);      // (gs: the cached row values / gradients / curvatures live in the global block too)
        L.GV = o; o += 2 * O * V; L.GNV = o; o += O; L.GR = o; o += O; L.GC = o; o += 2 * O;
        L.NV = NV; L.VIA = o; o += 3 * NV; L.VIDX = o; o += NV;
        L.OAT = take(MT); L.OHXT = take(MT); L.OHYT = take(MT); L.OHTT = take(MT);
        L.OAD = take(MD); L.OHXD = take(MD); L.OHYD = take(MD); L.OHDD = take(MD); L.OHTD = take(MD);
        L.GVEL = o; o += 2 * OD;
        L.total = o;
        return L;
    }
};

// The same layout with every offset a COMPILE-TIME constant (only the instance's own grid size n stays a run-time value): the kernel instantiation for a fixed
// stride NS, without clearance rows / via-points (IpmWave<..., NSC>).  With run-time offsets the ~45 layout words compete for the scalar registers (the headline
// kernel spilled some 270 of them to VGPR lanes, a v_readlane per use) and every LDS access of the lane-parallel passes carries its address arithmetic; with
// constants the offsets fold into the 16-bit immediates of the ds instructions.  Every field IS what make() returns for the same arguments (evaluated at compile time);
// tests/test_gpu_parity.py::test_fixed_layout_kernel_equals_the_generic_kernel_bit_for_bit holds the two instantiations against each other.
template <int NSC, int NTRIG, int NSTGW>
struct FixedLayout {
    int n;
    static constexpr WaveLayout c() { return WaveLayout::make(NSC, 0, 0, 1, NTRIG, 0, 0, 0, NSTGW, 0); }
    static constexpr int NS = NSC, NTR = NTRIG;
    static constexpr int X = c().X, U = c().U, LAM = c().LAM, LAMN = c().LAMN, SR = c().SR, YR = c().YR, PL = c().PL, PU = c().PU, DX = c().DX, DU = c().DU, CC = c().CC,
                         TRIG = c().TRIG, GAIN = c().GAIN, STG = c().STG, SC = c().SC, VP = c().VP, ZC = c().ZC, ZI = c().ZI, total = c().total;
    static constexpr int M = 0, O = 0, V = 1, NV = 0, GSW = 0, GSF = 0, OEB = 0;
    static constexpr int OS = c().OS, OY = c().OY, OI = c().OI, OG = c().OG, OAX = c().OAX, OAY = c().OAY, OHK = c().OHK, GV = c().GV, GNV = c().GNV, GR = c().GR, GC = c().GC,
                         OAT = c().OAT, OHXT = c().OHXT, OHYT = c().OHYT, OHTT = c().OHTT, OAD = c().OAD, OHXD = c().OHXD, OHYD = c().OHYD, OHDD = c().OHDD, OHTD = c().OHTD,
                         GVEL = c().GVEL, VIA = c().VIA, VIDX = c().VIDX;
    // does a run-time layout describe the same record?  (n and V -- the vertex capacity, unused without obstacles -- aside)
    __host__ __device__ static bool matches(const WaveLayout& l) {
        const WaveLayout f = c();
        return l.NS == f.NS && l.NTR == f.NTR && l.M == 0 && l.O == 0 && l.NV == 0 && l.X == f.X && l.U == f.U && l.LAM == f.LAM && l.LAMN == f.LAMN && l.SR == f.SR && l.YR == f.YR &&
               l.PL == f.PL && l.PU == f.PU && l.DX == f.DX && l.DU == f.DU && l.CC == f.CC && l.TRIG == f.TRIG && l.GAIN == f.GAIN && l.STG == f.STG && l.SC == f.SC && l.VP == f.VP &&
               l.ZC == f.ZC && l.ZI == f.ZI && l.total == f.total;
    }
};
template <int NSC, int NTRIG, int NSTGW> struct LayoutOf { using type = FixedLayout<NSC, NTRIG, NSTGW>; __host__ __device__ static type from(const WaveLayout& l) { return type{l.n}; } };
template <int NTRIG, int NSTGW> struct LayoutOf<0, NTRIG, NSTGW> { using type = WaveLayout; __host__ __device__ static const WaveLayout& from(const WaveLayout& l) { return l; } };

enum { SC_D = 0, SC_DT = 1, SC_DD = 2, SC_PDL = 3, SC_PDU = 4, SC_TS = 5, SC_TY = 6, SC_TG = 7, SC_TA = 8 /* 8..10 */ };

// A slot (StageAdd) of the entry (r, c) of the symmetric 8x8 stage cost block [x(3) u_prev(2) dt u(2)] and of its gradient
// column c = 8; -1 where the block is structurally zero.  Packed per row as 12 x 5 bits (slot + 1) so that a lane looks its
// column up with one 64-bit shift instead of a cascade of divergent branches.
constexpr int stage_add_slot(int r, int c, bool ext) {
    if (c == 8) return A08 + r;
    if (c > 8) return -1;
    const int a = r < c ? r : c, b = r < c ? c : r;
    if (a == 0) return b == 0 ? A00 : (b == 1 ? A01 : (b == 2 && ext ? A02 : (b == 5 && ext ? A05 : -1)));
    if (a == 1) return b == 1 ? A11 : (b == 2 && ext ? A12 : (b == 5 && ext ? A15 : -1));
    if (a == 2) return b == 2 ? A22 : (b == 5 ? A25 : (b == 6 ? A26 : (b == 7 ? A27 : -1)));
    if (a == 3) return b == 3 ? A33 : (b == 5 ? A35 : (b == 6 ? A36 : -1));
    if (a == 4) return b == 4 ? A44 : (b == 5 ? A45 : (b == 7 ? A47 : -1));
    if (a == 5) return b == 5 ? A55 : (b == 6 ? A56 : (b == 7 ? A57 : -1));
    if (a == 6) return b == 6 ? A66 : (b == 7 ? A67 : -1);
    return b == 7 ? A77 : -1;
}
constexpr unsigned long long stage_add_row(int r, bool ext) {
    unsigned long long v = 0;
    for (int c = 0; c < 12; ++c) v |= (unsigned long long)(stage_add_slot(r, c, ext) + 1) << (5 * c);
    return v;
}


// ---- wavefront reductions on the DPP path, no LDS traffic: row rotations inside each 16-lane row (row_ror 8, 4, 2, 1: every lane of a row then holds the row's
//      reduction), row_bcast:15 into rows 1 and 3, row_bcast:31 into row 3, and ONE v_readlane pair of lane 63: 20 VALU instructions per fp64 reduction, wave-uniform
//      result.  The moves name no "old" value (an undefined register: every lane that is read later is written), which spares a copy per move; row 3 ends up with
//      (r3 + r2) + (r1 + r0) -- the association the earlier readlane version had.
__device__ __forceinline__ int undef_vgpr() { int x; asm volatile("" : "=v"(x)); return x; }
template <int CTRL, int ROWS = 0xf> __device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(undef_vgpr(), lo, CTRL, ROWS, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(undef_vgpr(), hi, CTRL, ROWS, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWS = 0xf> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(undef_vgpr(), __float_as_int(v), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ double rd_lane(double v, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float rd_lane(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

struct OpSum { template <typename T> __device__ __forceinline__ static T f(T a, T b) { return a + b; } };
struct OpMin { template <typename T> __device__ __forceinline__ static T f(T a, T b) { return b < a ? b : a; } };
struct OpMax { template <typename T> __device__ __forceinline__ static T f(T a, T b) { return b > a ? b : a; } };

template <typename Op, typename T> __device__ __forceinline__ T wave_reduce(T v) {
    // row_ror:8,4,2,1 (dpp_ctrl 0x120 + n): afterwards every lane of a row holds the row's reduction
    v = Op::f(v, dpp_mov<0x128>(v));
    v = Op::f(v, dpp_mov<0x124>(v));
    v = Op::f(v, dpp_mov<0x122>(v));
    v = Op::f(v, dpp_mov<0x121>(v));
    v = Op::f(v, dpp_mov<0x142, 0xa>(v));      // row_bcast:15 -> rows 1, 3 (the other rows' lanes are never read again)
    v = Op::f(v, dpp_mov<0x143, 0x8>(v));      // row_bcast:31 -> row 3
    return rd_lane(v, 63);
}
template <typename T> __device__ __forceinline__ T wave_sum(T v) { return wave_reduce<OpSum>(v); }
template <typename T> __device__ __forceinline__ T wave_min(T v) { return wave_reduce<OpMin>(v); }
template <typename T> __device__ __forceinline__ T wave_max(T v) { return wave_reduce<OpMax>(v); }

__device__ __forceinline__ double lane_bcast(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// EXT = false compiles the rarely used rows / objective terms (terminal l2-ball, via-points) out of the kernel: the headline
// configurations keep their instruction count and register budget
// EXT: 0 = headline instantiation, 1 = + the rarely used rows / terms / coupling slots, 2 = + the cost variants (off-diagonal weights, trapezoidal rule)
// OBST = false compiles every clearance-row path out (solvers created without obstacles: the headline configurations): less code, and the two dozen layout
// words of the obstacle arrays leave the scalar registers (the headline kernel spilled ~430 of them)
// NSC > 0: the layout is a compile-time constant for the stride NS = NSC (FixedLayout; needs EXT == 0, OBST == false, no Crank-Nicolson trig words)
// GS: the factorisation data (STG, GAIN) in global memory instead of LDS (GlobalStage): same arithmetic, same results bit for bit, a third of the LDS record
template <typename T, int MODEL, int EXT = 1, bool OBST = true, int NSC = 0, bool GS = false>
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
    // stage-major records
    // (GS: the same records in the workgroup's global block -- byte offsets are formed in 32 bits and zero-extended, so that the accesses compile to
    //  global_load / global_store with the block's base in a scalar register pair and constant parts in the immediate offset)
    typedef __attribute__((address_space(1))) T GlbT;
    typedef __attribute__((address_space(1))) char GlbC;
    using SwT = std::conditional_t<GS, GlbT, T>;                         // a word of the sweeps' storage class
    __device__ __forceinline__ GlbT& gw(unsigned word) const { return *(GlbT*)((GlbC*)gmb + (size_t)(word * (unsigned)sizeof(T))); }
    __device__ __forceinline__ SwT& G_(int i, int k) const { if constexpr (GS) return gw((unsigned)(GlobalStage::GAIN(L.NS) + k * NGAIN + i)); else return sm[L.GAIN + k * NGAIN + i]; }
    static constexpr int NADDv = EXT ? (int)NADD : (int)NADD_BASE;
    __device__ __forceinline__ SwT& S_(int i, int k) const { if constexpr (GS) return gw((unsigned)(GlobalStage::STG(L.NS) + k * NSTG + i)); else return sm[L.STG + k * NSTG + i]; }
    __device__ __forceinline__ T& C_(int i, int k) const { return sm[L.CC + k * 3 + i]; }
    // cached value (0), gradient (1, 2) and curvature (3) of clearance row m at grid point k: written by kkt_pass, read by the other lane-parallel passes
    __device__ __forceinline__ SwT& OB_(int which, int m, int k) const {
        if constexpr (GS) return gw((unsigned)(GlobalStage::OBC(L.NS, NSTG) + (which * L.M + m) * L.NS + k)); else return sm[(which == 0 ? L.OG : (which == 1 ? L.OAX : (which == 2 ? L.OAY : L.OHK))) + m * L.NS + k];
    }
    // elastic variable e (0) and its step de (1) of clearance row m at grid point k (restoration mode): always in the workgroup's global block
    __device__ __forceinline__ GlbT& OE_(int which, int m, int k) const { return gw((unsigned)(L.OEB + (which * L.M + m) * L.NS + k)); }
    // c^_k = c_k + f_k dd, what the forward sweeps read (component-major): parked in LAMN, or (GS) in the global block
    __device__ __forceinline__ SwT& CH_(int i, int k) const { if constexpr (GS) return gw((unsigned)(GlobalStage::CH(L.NS) + i * L.NS + k)); else return sm[L.LAMN + i * L.NS + k]; }
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
    __device__ __forceinline__ int W_VP() const { if constexpr (GS) return GlobalStage::VP; else return L.VP; }
    __device__ __forceinline__ int W_CC() const { if constexpr (GS) return GlobalStage::CC; else return L.CC; }
    __device__ __forceinline__ int W_CH() const { if constexpr (GS) return GlobalStage::CH(L.NS); else return L.LAMN; }      // c^_k of the forward sweeps (LDS: parked in LAMN)
    __device__ __forceinline__ int W_GAIN() const { if constexpr (GS) return GlobalStage::GAIN(L.NS); else return L.GAIN; }
    __device__ __forceinline__ int W_STG() const { if constexpr (GS) return GlobalStage::STG(L.NS); else return L.STG; }
    __device__ __forceinline__ SwRef sw(int word) const { if constexpr (GS) return (unsigned)word * (unsigned)sizeof(T); else return lds(word); }
    __device__ __forceinline__ static constexpr int sw_step(int words) { return GS ? words * (int)sizeof(T) : words; }
    // word i (a compile-time constant at every call site) behind a running pointer / the word `step` pointer units behind it
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

    // ---------------------------------------------------------------- terminal l2-ball row (TerminalBallSE2, final_state_conditions_se2.cpp:54-64)
    // g = xd' S xd - gamma on the final state at z + alpha dz, gradient a = 2 S xd over the free components (fixed ones: xd = 0)
    __device__ __forceinline__ T ball_eval(T alpha, T a[3]) const {
        T g = -P.ball_gamma;
        for (int i = 0; i < 3; ++i) {
            a[i] = T(0);
            if (fx(i)) continue;
            T xd = xt(i, L.n - 1, alpha) - xf[i];
            if (i == 2) xd = normalize_theta(xd);
            g += P.ball_S[i] * xd * xd;
            a[i] = T(2) * P.ball_S[i] * xd;
        }
        if (costx()) {
            T xd[3], sx[3]; xd_final(alpha, xd); offmul(P.So, xd, sx);
            g += offquad(P.So, xd);
            for (int i = 0; i < 3; ++i) if (!fx(i)) a[i] += T(2) * sx[i];
        }
        return g;
    }
    // a' dx_T from the gradient cached by kkt_pass
    __device__ __forceinline__ T ball_jdz() const {
        T j = T(0);
        for (int i = 0; i < 3; ++i) if (!fx(i)) j += SCL(SC_TA + i) * F(L.DX, i, L.n - 1);
        return j;
    }
    // slack at the trial point (linear in alpha, like every other row's slack)
    __device__ __forceinline__ T ball_slack(T alpha, bool trial) const {
        T s = SCL(SC_TS);
        if (trial) s += alpha * (-(SCL(SC_TG) + s) - ball_jdz());
        return s;
    }

    // ---------------------------------------------------------------- via-points (MinTimeViaPointsCost, min_time_via_points_cost.cpp)
    __device__ __forceinline__ void load_via_points(const int32_t* n_via, const double* viap, int inst) {
        const int NV = L.NV;
        int nv = n_via ? n_via[inst] : 0;
        nv = nv < 0 ? 0 : (nv > NV ? NV : nv);
        for (int v = lane; v < NV; v += kWave) {
            for (int i = 0; i < 3; ++i) sm[L.VIA + 3 * v + i] = v < nv ? T(viap[((long)inst * NV + v) * 3 + i]) : T(0);
            sm[L.VIDX + v] = T(-1);
        }
        nvia = __builtin_amdgcn_readfirstlane(nv);
    }
    // MinTimeViaPointsCost::update (:39-117) with findClosestPose (full_discretization_grid_base_se2.cpp:364-388): every via-point is
    // attached to the first closest state of the CURRENT vertex values (states start..n-2, then the final state if strictly closer);
    // ordered mode restarts the search two states behind the previous match; a match at the goal moves to n-2, one at the start is
    // moved to 1 (ordered) or skipped.
    __device__ __forceinline__ void associate_via_points() const {
        const int n = L.n;
        int start = 0;
        for (int v = 0; v < nvia; ++v) {
            const T vx = sm[L.VIA + 3 * v], vy = sm[L.VIA + 3 * v + 1];
            T best = T(3e38);
            int bidx = 1 << 30;
            for (int i = lane; i < n - 1; i += kWave) {
                if (i < start) continue;
                const T dx = vx - F(L.X, 0, i), dy = vy - F(L.X, 1, i);
                const T dist = sqrt(dx * dx + dy * dy);
                if (dist < best) { best = dist; bidx = i; }
            }
            const T dmin = wave_min(best);
            int idx = (int)wave_min(best == dmin ? T(bidx) : T(1 << 30));
            {
                const T dx = vx - F(L.X, 0, n - 1), dy = vy - F(L.X, 1, n - 1);
                if (sqrt(dx * dx + dy * dy) < dmin || idx >= n) idx = n - 1;
            }
            if (P.vp_ordered) start = idx + 2;
            if (idx > n - 2) idx = n - 2;
            if (idx < 1) idx = P.vp_ordered ? 1 : -1;
            if (lane == 0) sm[L.VIDX + v] = T(idx);
        }
    }
    // via-point terms of grid point k at (px, py, th): value, gradient, number of attached points (Hessian 2 w_p m on x and y);
    // the orientation term is linear as coded (:139-142)
    __device__ __forceinline__ int via_terms(int k, T px, T py, T th, T& val, T g[3]) const {
        val = T(0); g[0] = g[1] = g[2] = T(0);
        int m = 0;
        const T wp = P.vp_wp, wo = P.vp_wo;
        for (int v = 0; v < nvia; ++v) {
            if ((int)sm[L.VIDX + v] != k) continue;
            const T dx = px - sm[L.VIA + 3 * v], dy = py - sm[L.VIA + 3 * v + 1];
            val += wp * (dx * dx + dy * dy);
            g[0] += T(2) * wp * dx; g[1] += T(2) * wp * dy;
            if (wo > T(0)) { val += wo * normalize_theta(sm[L.VIA + 3 * v + 2] - th); g[2] -= wo; }
            ++m;
        }
        return m;
    }

    // ---------------------------------------------------------------- clearance rows
    // distance of the point (px,py) to obstacle j (teb semantics: point / segment / closed polygon edge loop -- no inside test, as
    // distance_point_to_polygon_2d);
    // returns dist (>= 0, obstacle radius already subtracted), unit normal from the closest point to (px,py) and
    // hk = 1/|p-q| if the closest feature is a vertex (or a point/circle obstacle), 0 on an edge interior.
    // obst_closest: the closest point (bx, by) of obstacle j to (px, py), its squared distance, and whether the closest feature is a vertex
    // (what the callers that compare several candidates need; roots and normals only for the winner).
    __device__ __forceinline__ T obst_closest(T px, T py, int j, T& bx, T& by, bool& vert) const {
        const int nv = (int)sm[L.GNV + j];
        const T* v = sm + L.GV + 2 * L.V * j;
        T best = T(1e30);
        bx = T(0); by = T(0); vert = true;
        if (nv <= 1) { bx = v[0]; by = v[1]; T dx = px - bx, dy = py - by; best = dx * dx + dy * dy; }
        else {
            const int ne = nv == 2 ? 1 : nv;
            for (int e = 0; e < ne; ++e) {
                const int e2 = e + 1 < nv ? e + 1 : 0;
                const T ax = v[2 * e], ay = v[2 * e + 1], cx = v[2 * e2], cy = v[2 * e2 + 1];
                const T abx = cx - ax, aby = cy - ay;
                const T sq = abx * abx + aby * aby;
                T t = sq > T(0) ? ((px - ax) * abx + (py - ay) * aby) * t_rcp(sq) : T(0);
                t = t_min(T(1), t_max(T(0), t));
                const T qx = ax + t * abx, qy = ay + t * aby;
                const T d2 = (px - qx) * (px - qx) + (py - qy) * (py - qy);
                if (d2 < best) { best = d2; bx = qx; by = qy; vert = !(t > T(0) && t < T(1)); }
            }
        }
        return best;
    }
    __device__ __forceinline__ void obst_eval(T px, T py, int j, T& dist, T& nx, T& ny, T& hk) const {
        T bx, by;
        bool vert;
        const T best = obst_closest(px, py, j, bx, by, vert);
        const T dd = sqrt(best);
        if (dd > T(0)) { const T idd = t_rcp(dd); nx = (px - bx) * idd; ny = (py - by) * idd; hk = vert ? idd : T(0); }
        else { nx = T(0); ny = T(0); hk = T(0); }
        dist = dd - sm[L.GR + j];
    }

    // copies the instance's obstacles into LDS, computes centroids (teb Obstacle::getCentroid)
    __device__ __forceinline__ void load_obstacles(const int32_t* n_obst, const int32_t* n_vert, const double* verts, const double* radius, const double* vel, int inst) {
        const int O = L.O, V = L.V;
        const int no = n_obst ? n_obst[inst] : 0;
        if (EXT && P.dyn_obst)
            for (int e = lane; e < 2 * O; e += kWave) sm[L.GVEL + e] = (vel && e / 2 < no) ? T(vel[(long)inst * O * 2 + e]) : T(0);
        for (int j = lane; j < O; j += kWave) {
            int nv = j < no ? n_vert[(long)inst * O + j] : 0;
            if (nv > V) nv = V;
            sm[L.GNV + j] = T(nv);
            sm[L.GR + j] = (radius && j < no) ? T(radius[(long)inst * O + j]) : T(0);
            const double* vv = verts + ((long)inst * O + j) * V * 2;
            T cx = T(0), cy = T(0);
            for (int e = 0; e < V; ++e) {
                T x = e < nv ? T(vv[2 * e]) : T(0), y = e < nv ? T(vv[2 * e + 1]) : T(0);
                sm[L.GV + 2 * V * j + 2 * e] = x; sm[L.GV + 2 * V * j + 2 * e + 1] = y;
            }
            const T* v = sm + L.GV + 2 * V * j;
            if (nv >= 3) {
                T a = T(0), sx = T(0), sy = T(0), mx = T(0), my = T(0);
                for (int e = 0; e < nv; ++e) {
                    const int e2 = (e + 1) % nv;
                    T cr = v[2 * e] * v[2 * e2 + 1] - v[2 * e2] * v[2 * e + 1];
                    a += cr; sx += (v[2 * e] + v[2 * e2]) * cr; sy += (v[2 * e + 1] + v[2 * e2 + 1]) * cr;
                    mx += v[2 * e]; my += v[2 * e + 1];
                }
                a *= T(0.5);
                if (t_abs(a) < T(1e-12)) { cx = mx / T(nv); cy = my / T(nv); }
                else { cx = sx / (T(6) * a); cy = sy / (T(6) * a); }
            } else if (nv == 2) { cx = T(0.5) * (v[0] + v[2]); cy = T(0.5) * (v[1] + v[3]); }
            else if (nv == 1) { cx = v[0]; cy = v[1]; }
            sm[L.GC + 2 * j] = cx; sm[L.GC + 2 * j + 1] = cy;
        }
    }

    // StageInequalitySE2::update (src/optimal_control/stage_inequality_se2.cpp:50-162): relevant obstacles of every
    // grid point from the current vertex values.  The reference keeps EVERY obstacle closer than force_inclusion_dist plus the nearest one
    // on the left and on the right; this record has room for M rows per grid point.  Order of the kept rows: dynamic obstacles, forced
    // ones (container order), left, right.  When more than M rows are wanted, the forced ones that stay are the M CLOSEST (ties: lower
    // index) -- a deviation from the reference that the caller can see: the number of rows that did not fit is returned (summed over the
    // grid points; mpc_last_rows_dropped) so that max_obstacle_rows can be raised.
    __device__ __forceinline__ int associate_obstacles() const {
        const int n = L.n, M = nM();
        int dropped = 0;
        for (int k = lane; k < n; k += kWave) {
            int cnt = 0;
            for (int m = 0; m < M; ++m) set_oi(m, k, -1);
            if (k >= 1) {
                const T px = F(L.X, 0, k), py = F(L.X, 1, k), th = F(L.X, 2, k);
                T s, c;
                t_sincos(th, &s, &c);
                T lmin = T(1e30), rmin = T(1e30);
                int lidx = -1, ridx = -1, wanted = 0;
                if (dynobs())        // every dynamic obstacle is kept at every grid point (:99-106)
                    for (int j = 0; j < L.O; ++j) if ((int)sm[L.GNV + j] > 0 && is_dynamic(j)) { ++wanted; if (cnt < M) { set_oi(cnt, k, j); ++cnt; } }
                const int first_forced = cnt;
                for (int j = 0; j < L.O; ++j) {
                    if ((int)sm[L.GNV + j] <= 0) continue;
                    if (is_dynamic(j)) continue;
                    T dist, nx, ny, hk;
                    if (fpline()) { T a3[3], h3[3]; dist = turn_eval(px, py, th, j, a3, hk, h3); }
                    else { obst_eval(px, py, j, dist, nx, ny, hk); dist -= P.fp_radius; }
                    if (dist < P.force_incl) {
                        ++wanted;
                        if (cnt < M) { set_oi(cnt, k, j); OB_(0, cnt, k) = dist; ++cnt; }       // OG doubles as the distance of a kept forced row
                        else if (first_forced < M) {
                            // full: the farthest forced row kept so far gives way if this obstacle is closer (the later ones keep their order)
                            int far = first_forced;
                            for (int m = first_forced + 1; m < M; ++m) if (OB_(0, m, k) >= OB_(0, far, k)) far = m;
                            if (dist < OB_(0, far, k)) {
                                for (int m = far; m + 1 < M; ++m) { set_oi(m, k, oi(m + 1, k)); OB_(0, m, k) = OB_(0, m + 1, k); }
                                set_oi(M - 1, k, j); OB_(0, M - 1, k) = dist;
                            }
                        }
                        continue;
                    }
                    if (dist > P.cutoff) continue;
                    // cross2d(orientation, centroid) with the centroid as an ABSOLUTE vector (:121)
                    if (c * sm[L.GC + 2 * j + 1] - sm[L.GC + 2 * j] * s > T(0)) { if (dist < lmin) { lmin = dist; lidx = j; } }
                    else { if (dist < rmin) { rmin = dist; ridx = j; } }
                }
                if (lidx >= 0) { ++wanted; if (cnt < M) { set_oi(cnt, k, lidx); ++cnt; } }
                if (ridx >= 0) { ++wanted; if (cnt < M) { set_oi(cnt, k, ridx); ++cnt; } }
                if (k < n - 1) dropped += wanted - cnt;       // the final state carries no rows (finite_differences_grid_se2.cpp:47-56)
            }
        }
        return (int)wave_sum(T(dropped));
    }

    // teb Line / PolygonRobotFootprint::calculateDistance for ONE world point (vwx, vwy) (an obstacle centre or an obstacle vertex): distance
    // of the point to the footprint segment / closed edge loop, evaluated in the ROBOT frame q = R(-theta)(v - p) where the footprint is
    // fixed (teb distance_point_to_polygon_2d: first closest edge wins, no inside test; 1 vertex = a point, 2 vertices = one edge).
    // Returns the distance, the row gradient a = d g / d(x, y, theta) of g = d_min - dist, hk (the (x,y) block of hess g is
    // -hk (I - a_xy a_xy'), |a_xy| = 1) and the heading parts h3 = hess g [x theta, y theta, theta theta].
    // closest point of the footprint (robot frame) to the world point (vwx, vwy): squared distance, offset (dx, dy) from the closest point to
    // q = R(-theta)(v - p), edge parameter t.  First closest edge wins (squared distances compared; the root is taken once by the caller).
    __device__ __forceinline__ T fp_point_closest(T px, T py, T s, T c, T vwx, T vwy, T& qx, T& qy, T& dx, T& dy, T& t) const {
        const T vx = vwx - px, vy = vwy - py;
        qx = c * vx + s * vy; qy = c * vy - s * vx;
        dx = T(0); dy = T(0); t = T(0);
        T best = T(3e38);
        const bool poly = P.footprint_kind == 4;
        const int nv = poly ? P.fp_nv : 2;
        const int ne = nv <= 2 ? 1 : nv;
        for (int e = 0; e < ne; ++e) {
            const int e2 = nv == 1 ? 0 : (e + 1 < nv ? e + 1 : 0);
            const T a0 = poly ? P.fp_poly[2 * e] : P.fp_line[0], a1 = poly ? P.fp_poly[2 * e + 1] : P.fp_line[1];
            const T b0 = poly ? P.fp_poly[2 * e2] : P.fp_line[2], b1 = poly ? P.fp_poly[2 * e2 + 1] : P.fp_line[3];
            const T abx = b0 - a0, aby = b1 - a1;
            const T sq = abx * abx + aby * aby;
            T te = sq > T(0) ? ((qx - a0) * abx + (qy - a1) * aby) * t_rcp(sq) : T(0);
            te = t_min(T(1), t_max(T(0), te));
            const T ex = qx - (a0 + te * abx), ey = qy - (a1 + te * aby);
            const T e2d = ex * ex + ey * ey;
            if (e2d < best) { best = e2d; dx = ex; dy = ey; t = te; }
        }
        return best;
    }
    // row gradient a = d g / d(x, y, theta) of g = d_min - D, hk and the heading parts h3 for the closest-point data of fp_point_closest
    __device__ __forceinline__ void fp_point_derivs(T s, T c, T qx, T qy, T dx, T dy, T t, T D, T a[3], T& hk, T h3[3]) const {
        T nx = T(0), ny = T(0);
        hk = T(0);
        if (D > T(0)) { const T iD = t_rcp(D); nx = dx * iD; ny = dy * iD; hk = (t > T(0) && t < T(1)) ? T(0) : iD; }
        const T nw = nx * qy - ny * qx;                                   // n' dq/dtheta,  dq/dtheta = (qy, -qx)
        a[0] = c * nx - s * ny; a[1] = s * nx + c * ny; a[2] = -nw;        // -(Jq' n)
        const T hwx = hk * (qy - nx * nw), hwy = hk * (-qx - ny * nw);    // H_D dq/dtheta,  H_D = hk (I - n n')
        h3[0] = -((s * hwy - c * hwx) + (nx * s + ny * c));
        h3[1] = -((-s * hwx - c * hwy) + (ny * s - nx * c));
        h3[2] = -((qy * hwx - qx * hwy) - (nx * qx + ny * qy));
    }
    __device__ __forceinline__ T fp_point_eval(T px, T py, T s, T c, T vwx, T vwy, T a[3], T& hk, T h3[3]) const {
        T qx, qy, dx, dy, t;
        const T D = sqrt(fp_point_closest(px, py, s, c, vwx, vwy, qx, qy, dx, dy, t));
        fp_point_derivs(s, c, qx, qy, dx, dy, t, D, a, hk, h3);
        return D;
    }
    // sin / cos of the heading, kept per lane: the rows of one grid point (and the trials of one row) ask for the same angle again and again
    mutable T sc_th = T(1e30), sc_s = T(0), sc_c = T(1);
    __device__ __forceinline__ void heading_sincos(T th, T& s_, T& c_) const {
        if (th != sc_th) { t_sincos(th, &sc_s, &sc_c); sc_th = th; }
        s_ = sc_s; c_ = sc_c;
    }
    // footprint vertex i in the robot frame (line: start, end; polygon: the vertex list)
    __device__ __forceinline__ void fp_vertex(int i, T& ax, T& ay) const {
        const bool poly = P.footprint_kind == 4;
        ax = poly ? P.fp_poly[2 * i] : P.fp_line[2 * i]; ay = poly ? P.fp_poly[2 * i + 1] : P.fp_line[2 * i + 1];
    }
    __device__ __forceinline__ static bool seg_intersect(T ax, T ay, T bx, T by, T cx, T cy, T dx, T dy) {
        const T o1 = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax), o2 = (bx - ax) * (dy - ay) - (by - ay) * (dx - ax);
        const T o3 = (dx - cx) * (ay - cy) - (dy - cy) * (ax - cx), o4 = (dx - cx) * (by - cy) - (dy - cy) * (bx - cx);
        return o1 * o2 < T(0) && o3 * o4 < T(0);
    }
    // line / polygon footprint against obstacle j of ANY kind (teb LineRobotFootprint / PolygonRobotFootprint::calculateDistance ->
    // Obstacle::getMinimumDistance(segment | polygon): distance_segment_to_segment_2d / distance_segment_to_polygon_2d /
    // distance_polygon_to_polygon_2d).  All of them are the minimum over point-to-segment distances between the two closed edge loops
    // (0 where two edges cross, no inside test), i.e. the minimum over
    //     every obstacle vertex against the footprint edges  (fp_point_eval: the footprint moves with the pose), and
    //     every footprint vertex c_i(theta) = p + R(theta) a_i against the obstacle edges (obst_eval, chain rule through theta).
    // Point / circular obstacles take the first family only (their single vertex).  Same outputs as fp_point_eval.
    __device__ __forceinline__ T line_eval(T px, T py, T th, int j, T a[3], T& hk, T h3[3]) const {
        const T* v = sm + L.GV + 2 * L.V * j;
        T s, c;
        heading_sincos(th, s, c);
        const int nvo = (int)sm[L.GNV + j];
        if (nvo <= 1) return fp_point_eval(px, py, s, c, v[0], v[1], a, hk, h3) - sm[L.GR + j];
        const int F_ = P.footprint_kind == 4 ? P.fp_nv : 2;
        // the candidates are compared by their squared distances (first minimum wins); roots and derivatives only for the winner of each family
        T bestA = T(3e38), aqx = T(0), aqy = T(0), adx = T(0), ady = T(0), at = T(0);
        for (int m = 0; m < nvo; ++m) {
            T qx, qy, dx, dy, t;
            const T d2 = fp_point_closest(px, py, s, c, v[2 * m], v[2 * m + 1], qx, qy, dx, dy, t);
            if (d2 < bestA) { bestA = d2; aqx = qx; aqy = qy; adx = dx; ady = dy; at = t; }
        }
        T bestB = T(3e38), brx = T(0), bry = T(0), bbx = T(0), bby = T(0);
        bool bvert = true;
        for (int i = 0; i < F_; ++i) {
            T ax_, ay_, qbx, qby;
            bool vert;
            fp_vertex(i, ax_, ay_);
            const T rx = c * ax_ - s * ay_, ry = s * ax_ + c * ay_;
            const T d2 = obst_closest(px + rx, py + ry, j, qbx, qby, vert);
            if (d2 < bestB) { bestB = d2; brx = rx; bry = ry; bbx = qbx; bby = qby; bvert = vert; }
        }
        const T DA = sqrt(bestA), DB = sqrt(bestB) - sm[L.GR + j];
        T best;
        if (DB < DA) {             // a footprint vertex c_i(theta) = p + R(theta) a_i against the obstacle's edges: chain rule through theta
            best = DB;
            const T dd = sqrt(bestB);
            T nx = T(0), ny = T(0), ho = T(0);
            if (dd > T(0)) { const T idd = t_rcp(dd); nx = (px + brx - bbx) * idd; ny = (py + bry - bby) * idd; ho = bvert ? idd : T(0); }
            const T wx = -bry, wy = brx, nw = nx * wx + ny * wy;         // w = dc/dtheta
            a[0] = -nx; a[1] = -ny; a[2] = -nw;
            hk = ho;
            const T hvx = ho * (wx - nx * nw), hvy = ho * (wy - ny * nw);
            h3[0] = -hvx; h3[1] = -hvy; h3[2] = -((wx * hvx + wy * hvy) - (nx * brx + ny * bry));
        } else {                   // an obstacle vertex against the footprint's edges
            best = DA;
            fp_point_derivs(s, c, aqx, aqy, adx, ady, at, DA, a, hk, h3);
        }
        // crossing edges: distance 0 (and no gradient), as distance_segment_to_segment_2d returns it
        const int nef = F_ <= 2 ? (F_ == 2 ? 1 : 0) : F_, neo = nvo == 2 ? 1 : nvo;
        for (int e = 0; e < nef; ++e) {
            T a0x, a0y, a1x, a1y;
            fp_vertex(e, a0x, a0y); fp_vertex(e + 1 < F_ ? e + 1 : 0, a1x, a1y);
            const T Ax = px + c * a0x - s * a0y, Ay = py + s * a0x + c * a0y, Bx = px + c * a1x - s * a1y, By = py + s * a1x + c * a1y;
            for (int o = 0; o < neo; ++o) {
                const int o2 = o + 1 < nvo ? o + 1 : 0;
                if (seg_intersect(Ax, Ay, Bx, By, v[2 * o], v[2 * o + 1], v[2 * o2], v[2 * o2 + 1])) {
                    a[0] = a[1] = a[2] = T(0); hk = T(0); h3[0] = h3[1] = h3[2] = T(0);
                    return T(0);
                }
            }
        }
        return best;
    }

    // teb TwoCirclesRobotFootprint::calculateDistance: min(dist(front centre) - r_front, dist(rear centre) - r_rear) with the centres at
    // +front_offset / -rear_offset along the heading: the point evaluation at c(theta) = p + o (cos, sin), chain rule through theta.
    // Same outputs as line_eval (|a_xy| = 1 again, so the (x,y) block keeps the -hk (I - a_xy a_xy') form).
    __device__ __forceinline__ T two_eval(T px, T py, T th, int j, T a[3], T& hk, T h3[3]) const {
        T s, c;
        heading_sincos(th, s, c);
        T df, nfx, nfy, hf, dr, nrx, nry, hr;
        obst_eval(px + P.fp_line[0] * c, py + P.fp_line[0] * s, j, df, nfx, nfy, hf);
        obst_eval(px - P.fp_line[2] * c, py - P.fp_line[2] * s, j, dr, nrx, nry, hr);
        df -= P.fp_line[1]; dr -= P.fp_line[3];
        const bool rear = dr < df;                                   // std::min(front, rear): the front one wins a tie
        const T o = rear ? -P.fp_line[2] : P.fp_line[0];
        const T nx = rear ? nrx : nfx, ny = rear ? nry : nfy;
        hk = rear ? hr : hf;
        const T wx = -o * s, wy = o * c;                             // dc/dtheta
        const T nw = nx * wx + ny * wy;
        a[0] = -nx; a[1] = -ny; a[2] = -nw;
        const T hvx = hk * (wx - nx * nw), hvy = hk * (wy - ny * nw);    // H_D dc/dtheta
        h3[0] = -hvx; h3[1] = -hvy; h3[2] = -((wx * hvx + wy * hvy) - o * (nx * c + ny * s));
        return rear ? dr : df;
    }
    __device__ __forceinline__ T turn_eval(T px, T py, T th, int j, T a[3], T& hk, T h3[3]) const {
        return P.footprint_kind == 3 ? two_eval(px, py, th, j, a, hk, h3) : line_eval(px, py, th, j, a, hk, h3);      // line_eval: line and polygon
    }

    // value / gradient / curvature cache of the clearance rows of grid point k at position (px,py); returns row count
    __device__ __forceinline__ bool obst_row(int k, int m, T px, T py, T& g, T& ax, T& ay, T& hk) const {
        const int j = oi(m, k);
        if (j < 0) return false;
        T dist, nx, ny;
        obst_eval(px, py, j, dist, nx, ny, hk);
        g = P.d_min - (dist - P.fp_radius);
        ax = -nx; ay = -ny;
        return true;
    }
    __device__ __forceinline__ bool is_dynamic(int j) const { return dynobs() && (sm[L.GVEL + 2 * j] != T(0) || sm[L.GVEL + 2 * j + 1] != T(0)); }
    // dynamic obstacle j at grid point k (computeNonIntegralStateDtTerm, stage_inequality_se2.cpp:177-189): the obstacle moved by k dt v,
    // i.e. the static evaluation at the point p - k dt v; a[2] = d g / d dt, h3 = hess g [x dt, y dt, dt dt] (chain rule, c linear in dt)
    __device__ __forceinline__ void dyn_row(int k, int j, T px, T py, T d, T& g, T a[3], T& hk, T h3[3]) const {
        const T kvx = T(k) * sm[L.GVEL + 2 * j], kvy = T(k) * sm[L.GVEL + 2 * j + 1];
        T dist, nx, ny;
        obst_eval(px - d * kvx, py - d * kvy, j, dist, nx, ny, hk);
        g = P.d_min - (dist - P.fp_radius);
        a[0] = -nx; a[1] = -ny;
        const T nkv = nx * kvx + ny * kvy;
        a[2] = nkv;                                                   // -(n . (-k v))
        // hess dist [xy, d] = H_D (-k v), H_D = hk (I - n n');  hess g = -hess dist
        const T hx = hk * (kvx - nx * nkv), hy = hk * (kvy - ny * nkv);
        h3[0] = hx; h3[1] = hy; h3[2] = -(kvx * hx + kvy * hy);
        if (!dtf()) { a[2] = T(0); h3[0] = h3[1] = h3[2] = T(0); }   // fixed grid: dt is not a variable
    }
    // row (k, m) with its third-variable parts.  a[2] / h3: heading (footprints that turn with the pose) or, with a point / circular
    // footprint, dt (dynamic obstacles; d = dt of the evaluated point).  With BOTH (a dynamic obstacle seen by a turning footprint) a[2] / h3
    // carry the heading parts and ad / hd = (d g / d dt, hess g [x dt, y dt, dt dt, theta dt]) the dt parts: the row is the static row
    // G(p - k dt v, theta), so with kappa = k v:  g_dt = -a_xy . kappa,  g_{p dt} = -H_xy kappa,  g_{theta dt} = -g_{p theta} . kappa,
    // g_{dt dt} = kappa' H_xy kappa,  H_xy = -hk (I - a_xy a_xy').
    __device__ __forceinline__ bool obst_row3(int k, int m, T px, T py, T th, T& g, T a[3], T& hk, T h3[3], T d, T& ad, T hd[4]) const {
        ad = T(0); hd[0] = hd[1] = hd[2] = hd[3] = T(0);
        if (dynobs()) {
            const int j = oi(m, k);
            if (j < 0) return false;
            if (is_dynamic(j)) {
                if (!fpline()) { dyn_row(k, j, px, py, d, g, a, hk, h3); return true; }
                const T kx = T(k) * sm[L.GVEL + 2 * j], ky = T(k) * sm[L.GVEL + 2 * j + 1];
                g = P.d_min - turn_eval(px - d * kx, py - d * ky, th, j, a, hk, h3);
                if (dtf()) {
                    const T ak = a[0] * kx + a[1] * ky;
                    ad = -ak;
                    hd[0] = hk * (kx - a[0] * ak); hd[1] = hk * (ky - a[1] * ak);
                    hd[2] = -hk * (kx * kx + ky * ky - ak * ak);
                    hd[3] = -(h3[0] * kx + h3[1] * ky);
                }
                return true;
            }
        }
        if (!fpline()) { a[2] = T(0); h3[0] = h3[1] = h3[2] = T(0); return obst_row(k, m, px, py, g, a[0], a[1], hk); }
        const int j = oi(m, k);
        if (j < 0) return false;
        g = P.d_min - turn_eval(px, py, th, j, a, hk, h3);
        return true;
    }
    __device__ __forceinline__ bool obst_row3(int k, int m, T px, T py, T th, T& g, T a[3], T& hk, T h3[3], T d = T(0)) const {
        T ad, hd[4];
        return obst_row3(k, m, px, py, th, g, a, hk, h3, d, ad, hd);
    }
    __device__ __forceinline__ bool dynturn() const { return fpline() && dynobs(); }
    // restoration mode, one clearance row g + s - e = 0 with (s, e) condensed together (derivation: DESIGN.md section 3.3):
    //     sigma = 1 / (s / y + e / (rho - y)),   ybar = y + sigma (res + mu / y - s - mu / (rho - y) + e),   res = g + s - e
    __device__ __forceinline__ void elastic_condense(T s, T y, T g, T ee, T& sig, T& ybar) const {
        const T iy = t_rcp(y), iw = t_rcp(erho - y);
        sig = t_rcp(s * iy + ee * iw);
        ybar = y + sig * ((g + s - ee) + mu * iy - s - mu * iw + ee);
    }
    // ... and the steps of its slack, elastic variable and multiplier for a' dz = jdz
    __device__ __forceinline__ void elastic_steps(T s, T y, T g, T ee, T jdz, T& ds, T& de, T& dy, T& ybar) const {
        T sig;
        elastic_condense(s, y, g, ee, sig, ybar);
        dy = ybar + sig * jdz - y;
        const T iy = t_rcp(y), iw = t_rcp(erho - y);
        ds = mu * iy - s - (s * iy) * dy;
        de = mu * iw - ee + (ee * iw) * dy;
    }
    // a' dz of row (k, m) from the cached gradient
    __device__ __forceinline__ T obst_jdz(int k, int m) const {
        T j = OB_(1, m, k) * F(L.DX, 0, k) + OB_(2, m, k) * F(L.DX, 1, k);
        if (fpline()) j += F(L.OAT, m, k) * F(L.DX, 2, k);
        if (dynturn()) j += F(L.OAD, m, k) * SCL(SC_DD);
        else if (dynobs()) j += F(L.OAT, m, k) * SCL(SC_DD);
        return j;
    }

    // ---------------------------------------------------------------- point evaluation (parallel)
    // trig cache + c_k for the point (XB, UB, d); returns wave-reduced sum|c|, objective
    __device__ __forceinline__ void eval_point(T d, T& theta_c, T& fobj, T alpha = T(0), bool trial = false) const {
        const int n = L.n;
        const T al = trial ? alpha : T(0);
        T th = T(0), fo = T(0);
        // clearance rows (non-linear): |g(x_k) + s| with the trial slack s + alpha*ds
        if (nM() > 0) {
            for (int k = lane; k < n - 1; k += kWave) {
                if (k < 1) continue;
                const T px = xt(0, k, al), py = xt(1, k, al), pth = fpline() ? xt(2, k, al) : T(0);
                for (int m = 0; m < nM(); ++m) {
                    T g, a3[3], hk, h3[3];
                    if (!obst_row3(k, m, px, py, pth, g, a3, hk, h3, d)) continue;
                    T s = F(L.OS, m, k);
                    if (erho > T(0)) {      // restoration mode: g + s - e with the trial values of both, + rho e in the objective
                        const T ee = OE_(0, m, k), de = trial ? T(OE_(1, m, k)) : T(0);
                        if (trial) s += alpha * (-(OB_(0, m, k) + s - ee) - obst_jdz(k, m) + de);
                        const T et = ee + (trial ? alpha * de : T(0));
                        th += t_abs(g + s - et);
                        fo += erho * et;
                        continue;
                    }
                    if (trial) s += alpha * (-(OB_(0, m, k) + s) - obst_jdz(k, m));
                    th += t_abs(g + s);
                }
            }
        }
        for (int k = lane; k < n - 1; k += kWave) {
            T xk[3] = {xt(0, k, al), xt(1, k, al), xt(2, k, al)};
            T xn[3] = {xt(0, k + 1, al), xt(1, k + 1, al), xt(2, k + 1, al)};
            T v = ut(0, k, al), w = ut(1, k, al);
            T tr[4], tr2[2], f[3];
            model_trig_colloc<T, MODEL>(P, xk[2], v, w, d, tr, tr2);
            colloc_f<T, MODEL>(P, tr, tr2, v, w, f);
            T c0 = d * f[0] - (xn[0] - xk[0]);
            T c1 = d * f[1] - (xn[1] - xk[1]);
            T c2 = d * f[2] - normalize_theta(xn[2] - xk[2]);
            for (int i = 0; i < NTRB; ++i) F(L.TRIG, i, k) = tr[i];
            if (P.collocation == COLLOC_CN) { F(L.TRIG, NTRB, k) = tr2[0]; F(L.TRIG, NTRB + 1, k) = tr2[1]; }
            C_(0, k) = c0; C_(1, k) = c1; C_(2, k) = c2;
            th += t_abs(c0) + t_abs(c1) + t_abs(c2);
            if (quad()) {
                T xd0 = xk[0] - xf[0], xd1 = xk[1] - xf[1], xd2 = normalize_theta(xk[2] - xf[2]);
                fo += (P.Q[0] * xd0 * xd0 + P.Q[1] * xd1 * xd1 + P.Q[2] * xd2 * xd2 + P.R[0] * v * v + P.R[1] * w * w) * (intf() ? d : T(1));
                if (costx()) {
                    const T xd[3] = {xd0, xd1, xd2};
                    fo += (offquad(P.Qo, xd) + T(2) * P.Ro * v * w) * (intf() ? d : T(1));
                    if (k == 0 && P.trapz) fo -= T(0.5) * d * fullquad(P.Q, P.Qo, xd);
                }
            }
            if (via()) { T vv, vg[3]; via_terms(k, xk[0], xk[1], xk[2], vv, vg); fo += vv; }
        }
        if (lane == 0) {
            if (mintime()) fo += T(n - 1) * d;
            if (costx()) fo += final_cost_extra(al, d);
            if (hasqf()) {       // the terminal cost does not depend on the stage cost's type (src/controller.cpp:641-672)
                for (int i = 0; i < 3; ++i) if (!fx(i)) {
                    T xd = xt(i, n - 1, al) - xf[i];
                    if (i == 2) xd = normalize_theta(xd);
                    fo += P.Qf[i] * xd * xd;
                }
            }
            if (ball()) { T a[3]; th += t_abs(ball_eval(al, a) + ball_slack(alpha, trial)); }
        }
        theta_c = wave_sum(th);
        fobj = wave_sum(fo);
    }

    // sum of barrier logs at the current (alpha = 0) or trial point; wave-reduced
    __device__ __forceinline__ T barrier_logs(T d, T alpha, bool trial, T dd) const {
        const int n = L.n;
        LogAcc<T> acc;
        for (int k = lane; k < n; k += kWave) {
            if (k < n - 1) {
                for (int j = 0; j < 2; ++j) { T u = ut(j, k, trial ? alpha : T(0)); acc.mul(u - P.u_lb[j]); acc.mul(P.u_ub[j] - u); }
            }
            for (int q = 0; q < 4; ++q) {
                if (!row_on(k, q)) continue;
                T s = F(L.SR, q, k);
                if (trial) s += alpha * (-(row_val(L.U, SCL(SC_D), k, q) + s) - row_jdz(k, q, dd));
                acc.mul(s);
            }
            if (nM() > 0 && k >= 1 && k < n - 1) {
                for (int m = 0; m < nM(); ++m) {
                    if (oi(m, k) < 0) continue;
                    T s = F(L.OS, m, k);
                    if (erho > T(0)) {
                        const T ee = OE_(0, m, k), de = trial ? T(OE_(1, m, k)) : T(0);
                        if (trial) s += alpha * (-(OB_(0, m, k) + s - ee) - obst_jdz(k, m) + de);
                        acc.mul(s); acc.mul(ee + (trial ? alpha * de : T(0)));
                        continue;
                    }
                    if (trial) s += alpha * (-(OB_(0, m, k) + s) - obst_jdz(k, m));
                    acc.mul(s);
                }
            }
        }
        if (lane == 0 && dtf()) { acc.mul(d - P.dt_lb); acc.mul(P.dt_ub - d); }
        if (lane == 0 && ball()) acc.mul(ball_slack(alpha, trial));
        return wave_sum(acc.value());
    }

    // ---------------------------------------------------------------- line-search trials, register-resident fast path
    // (n <= 64, no clearance rows: one interval / one rate row per lane).  Everything a trial needs that does not depend on the
    // step length is pulled into registers once per iteration; a trial is then ~300 instructions with no LDS reads.  Same
    // arithmetic as eval_point() + barrier_logs() at (alpha, trial = true).
    struct TrialRegs {
        T xk[3], dxk[3], xn[3], dxn[3], u[2], du[2], s[4], ds[4];
        T ulb[2], uub[2], dt_lb, dt_ub, nm1;      // (the quadratic weights are read from the problem record when needed: ten registers less across the line search)
        bool stage, on[4], quad, mint, dtf;
        int k;
    };
    __device__ __forceinline__ bool trial_fast_ok() const { return EXT == 0 && !GS && !OBST && !(sizeof(T) == 8 && NTRB > 3) && L.n <= kWave; }      // (a kernel with clearance-row code only runs for handles that have rows)      // (the extended instantiations and the fp64 bicycle / front-wheel models keep those registers for what they add: no scratch memory anywhere)
    __device__ __forceinline__ void trial_setup(TrialRegs& r, T dd) const {
        const int n = L.n, k = lane;
        const T d = SCL(SC_D);
        r.k = k; r.stage = k < n - 1;
        r.quad = quad(); r.mint = mintime(); r.dtf = dtf(); r.nm1 = T(n - 1);
        r.dt_lb = P.dt_lb; r.dt_ub = P.dt_ub;
        for (int i = 0; i < 3; ++i) {
            r.xk[i] = r.dxk[i] = r.xn[i] = r.dxn[i] = T(0);
            if (r.stage) {
                r.xk[i] = F(L.X, i, k); r.xn[i] = F(L.X, i, k + 1);
                if (k > 0) r.dxk[i] = F(L.DX, i, k);
                if (k + 1 < n - 1 || !fx(i)) r.dxn[i] = F(L.DX, i, k + 1);
            }
        }
        for (int j = 0; j < 2; ++j) {
            r.ulb[j] = P.u_lb[j]; r.uub[j] = P.u_ub[j];
            r.u[j] = r.stage ? F(L.U, j, k) : T(0); r.du[j] = r.stage ? F(L.DU, j, k) : T(0);
        }
        for (int q = 0; q < 4; ++q) {
            r.on[q] = k < n && row_on(k, q);
            r.s[q] = T(1); r.ds[q] = T(0);
            if (r.on[q]) { const T s = F(L.SR, q, k); r.s[q] = s; r.ds[q] = -(row_val(L.U, d, k, q) + s) - row_jdz(k, q, dd); }
        }
    }
    // returns wave-reduced sum|c| (th), objective (fo) and the sum of the barrier logs at z + alpha dz, dt = d
    __device__ __forceinline__ void trial_eval(const TrialRegs& r, T alpha, T d, T& th_out, T& fo_out, T& logs_out) const {
        T th = T(0), fo = T(0);
        LogAcc<T> acc;
        if (r.stage) {
            const T x0_ = r.xk[0] + alpha * r.dxk[0], x1_ = r.xk[1] + alpha * r.dxk[1];
            const T x2_ = r.k > 0 ? normalize_theta(r.xk[2] + alpha * r.dxk[2]) : r.xk[2];
            const T n0 = r.xn[0] + alpha * r.dxn[0], n1 = r.xn[1] + alpha * r.dxn[1];
            const T n2 = r.dxn[2] != T(0) ? normalize_theta(r.xn[2] + alpha * r.dxn[2]) : r.xn[2];
            const T v = r.u[0] + alpha * r.du[0], w = r.u[1] + alpha * r.du[1];
            T tr[4], tr2[2], f[3];
            model_trig_colloc<T, MODEL>(P, x2_, v, w, d, tr, tr2);
            colloc_f<T, MODEL>(P, tr, tr2, v, w, f);
            const T c0 = d * f[0] - (n0 - x0_), c1 = d * f[1] - (n1 - x1_), c2 = d * f[2] - normalize_theta(n2 - x2_);
            for (int i = 0; i < NTRB; ++i) F(L.TRIG, i, r.k) = tr[i];
            if (P.collocation == COLLOC_CN) { F(L.TRIG, NTRB, r.k) = tr2[0]; F(L.TRIG, NTRB + 1, r.k) = tr2[1]; }
            C_(0, r.k) = c0; C_(1, r.k) = c1; C_(2, r.k) = c2;
            th = t_abs(c0) + t_abs(c1) + t_abs(c2);
            if (r.quad) {
                const T xd0 = x0_ - xf[0], xd1 = x1_ - xf[1], xd2 = normalize_theta(x2_ - xf[2]);
                fo = (P.Q[0] * xd0 * xd0 + P.Q[1] * xd1 * xd1 + P.Q[2] * xd2 * xd2 + P.R[0] * v * v + P.R[1] * w * w) * (intf() ? d : T(1));
            }
            if (via()) { T vv, vg[3]; via_terms(r.k, x0_, x1_, x2_, vv, vg); fo += vv; }
            acc.mul(v - r.ulb[0]); acc.mul(r.uub[0] - v); acc.mul(w - r.ulb[1]); acc.mul(r.uub[1] - w);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc.mul(r.s[q] + alpha * r.ds[q]);      // rows that are off carry s = 1, ds = 0
        if (lane == 0) {
            if (r.mint) fo += r.nm1 * d;
            if (hasqf()) {
                for (int i = 0; i < 3; ++i) if (!fx(i)) {
                    T xd = xt(i, L.n - 1, alpha) - xf[i];
                    if (i == 2) xd = normalize_theta(xd);
                    fo += P.Qf[i] * xd * xd;
                }
            }
            if (r.dtf) { acc.mul(d - r.dt_lb); acc.mul(r.dt_ub - d); }
            if (ball()) { T a[3]; const T st = ball_slack(alpha, true); th += t_abs(ball_eval(alpha, a) + st); acc.mul(st); }
        }
        th_out = wave_sum(th); fo_out = wave_sum(fo); logs_out = wave_sum(acc.value());
    }

    // ---------------------------------------------------------------- KKT error + stage records
    struct Err { T rd, rp, cmin, cmax, csum, sum_mult, sum_bmult, theta; int n_mult, n_bmult; };     // csum: sum of the n_bmult complementarity products

    // Ipopt's scaled optimality error E_mu.  (reciprocals instead of IEEE divisions -- ~100 ticks each for a lone wave --: 1/count is cached with the counts,
    // 1/s_max is a constant, the two scalings are inverted once per call)
    __device__ __forceinline__ T err_value(const Err& e, T mu_t) const {
        const T sd = t_max(Algo<T>::s_max, e.sum_mult * inv_cnt_mult) * (T(1) / Algo<T>::s_max);
        const T sc = t_max(Algo<T>::s_max, e.sum_bmult * inv_cnt_bmult) * (T(1) / Algo<T>::s_max);
        T comp = e.n_bmult > 0 ? t_max(e.cmax - mu_t, mu_t - e.cmin) : T(0);
        return t_max(e.rd * t_rcp(sd), t_max(e.rp, comp * t_rcp(sc)));
    }

    // parallel: KKT error pieces (needs LAM of the neighbours) ; also writes the mu-independent part of STG
    __device__ __forceinline__ Err kkt_pass() const {
        const int n = L.n;
        const T d = SCL(SC_D);
        T rd = T(0), rp = T(0), cmin = T(1e30), cmax = T(0), smult = T(0), sb = T(0), th = T(0), rdd = T(0), cs = T(0);
        int nm = 0, nb = 0;
        for (int k = lane; k < n; k += kWave) {
            T rec[21];
            if (k < n - 1) {
                T lam[3] = {F(L.LAM, 0, k), F(L.LAM, 1, k), F(L.LAM, 2, k)};
                T tr[4] = {F(L.TRIG, 0, k), F(L.TRIG, 1, k), F(L.TRIG, 2, k), NTRB > 3 ? F(L.TRIG, 3, k) : T(0)};
                T tr2[2] = {T(0), T(0)};
                if (P.collocation == COLLOC_CN) { tr2[0] = F(L.TRIG, NTRB, k); tr2[1] = F(L.TRIG, NTRB + 1, k); }
                T v = F(L.U, 0, k), w = F(L.U, 1, k);
                // derivatives of the collocation increment D(theta, u, dt) (forward or midpoint differences, mpc_core.hpp::stage_map);
                // gJ[j] = lam' dD/dq_j is what the dual residuals need, Jdt = dD/d dt the dt column of the row
                StageMap<T> sm_;
                stage_map<T, MODEL>(P, tr, tr2, v, w, d, lam, sm_);
                T gJ[3];
                for (int j = 0; j < 3; ++j) gJ[j] = lam[0] * sm_.Jq[0][j] + lam[1] * sm_.Jq[1][j] + lam[2] * sm_.Jq[2][j];
                // stage record, mu-independent part: kept in registers and stored at the END of the loop body -- every LDS store
                // forces the loads that follow it in program order to be re-issued and waited for (possible aliasing)
                rec[0] = sm_.Jq[0][0]; rec[1] = sm_.Jq[1][0];
                rec[2] = sm_.Jdt[0]; rec[3] = sm_.Jdt[1]; rec[4] = sm_.Jdt[2];
                for (int a = 0; a < 3; ++a) { rec[5 + a] = sm_.Jq[a][1]; rec[8 + a] = sm_.Jq[a][2]; }
                rec[11] = sm_.Hqq[0][0]; rec[12] = sm_.Hqq[0][1]; rec[13] = sm_.Hqq[0][2];
                rec[14] = sm_.Hqq[1][1]; rec[15] = sm_.Hqq[1][2]; rec[16] = sm_.Hqq[2][2];
                rec[17] = sm_.Hqd[0]; rec[18] = sm_.Hqd[1]; rec[19] = sm_.Hqd[2];
                rec[20] = sm_.Hdd;
                for (int i = 0; i < 3; ++i) {
                    T ci = C_(i, k);
                    rp = t_max(rp, t_abs(ci)); th += t_abs(ci); smult += t_abs(lam[i]);
                }
                nm += 3;
                rdd += lam[0] * sm_.Jdt[0] + lam[1] * sm_.Jdt[1] + lam[2] * sm_.Jdt[2];
                T gx[3] = {T(0), T(0), T(0)}, gu[2] = {T(0), T(0)};
                if (quad()) {
                    T xd[3] = {F(L.X, 0, k) - xf[0], F(L.X, 1, k) - xf[1], normalize_theta(F(L.X, 2, k) - xf[2])};
                    const T w8 = intf() ? d : T(1);
                    for (int i = 0; i < 3; ++i) gx[i] = T(2) * P.Q[i] * xd[i] * w8;
                    gu[0] = T(2) * P.R[0] * v * w8; gu[1] = T(2) * P.R[1] * w * w8;
                    if (intf()) rdd += P.Q[0] * xd[0] * xd[0] + P.Q[1] * xd[1] * xd[1] + P.Q[2] * xd[2] * xd[2] + P.R[0] * v * v + P.R[1] * w * w;
                    if (costx()) {
                        T qo[3]; offmul(P.Qo, xd, qo);
                        for (int i = 0; i < 3; ++i) gx[i] += T(2) * qo[i] * w8;
                        gu[0] += T(2) * P.Ro * w * w8; gu[1] += T(2) * P.Ro * v * w8;
                        if (intf()) rdd += offquad(P.Qo, xd) + T(2) * P.Ro * v * w;
                        if (k == 0 && P.trapz) rdd -= T(0.5) * fullquad(P.Q, P.Qo, xd);
                    }
                }
                if (via()) { T vv; via_terms(k, F(L.X, 0, k), F(L.X, 1, k), F(L.X, 2, k), vv, gx); }
                T osx = T(0), osy = T(0), ost = T(0);
                if (nM() > 0 && k >= 1) {
                    const T px = F(L.X, 0, k), py = F(L.X, 1, k);
                    for (int m = 0; m < nM(); ++m) {
                        T g, a3[3], hk, h3[3], ad, hd[4];
                        if (!obst_row3(k, m, px, py, F(L.X, 2, k), g, a3, hk, h3, d, ad, hd)) continue;
                        const T ax = a3[0], ay = a3[1];
                        OB_(0, m, k) = g; OB_(1, m, k) = ax; OB_(2, m, k) = ay; OB_(3, m, k) = hk;
                        if (fpline() || dynobs()) { F(L.OAT, m, k) = a3[2]; F(L.OHXT, m, k) = h3[0]; F(L.OHYT, m, k) = h3[1]; F(L.OHTT, m, k) = h3[2]; }
                        if (dynturn()) { F(L.OAD, m, k) = ad; F(L.OHXD, m, k) = hd[0]; F(L.OHYD, m, k) = hd[1]; F(L.OHDD, m, k) = hd[2]; F(L.OHTD, m, k) = hd[3]; }
                        const T s = F(L.OS, m, k), y = F(L.OY, m, k);
                        const T ee = erho > T(0) ? T(OE_(0, m, k)) : T(0);            // restoration mode: the row reads g + s - e = 0
                        const T res = g + s - ee;
                        rp = t_max(rp, t_abs(res)); th += t_abs(res);
                        cmin = t_min(cmin, s * y); cmax = t_max(cmax, s * y); cs += s * y;
                        sb += y; nb += 1;
                        if (erho > T(0)) {      // the elastic variable's own complementarity, e (rho - y); e itself counts as infeasibility of the ORIGINAL row
                            const T ce = ee * (erho - y);
                            cmin = t_min(cmin, ce); cmax = t_max(cmax, ce); cs += ce;
                            sb += erho - y; nb += 1;
                            rp = t_max(rp, ee);
                        }
                        osx += y * ax; osy += y * ay;
                        if (dynturn()) { rdd += y * ad; ost += y * a3[2]; }
                        else if (dynobs()) rdd += y * a3[2]; else ost += y * a3[2];
                    }
                }
                if (k >= 1) {
                    T r0 = gx[0] + osx + lam[0] - F(L.LAM, 0, k - 1);
                    T r1 = gx[1] + osy + lam[1] - F(L.LAM, 1, k - 1);
                    T r2 = gx[2] + ost + lam[2] + gJ[0] - F(L.LAM, 2, k - 1);
                    rd = t_max(rd, t_max(t_abs(r0), t_max(t_abs(r1), t_abs(r2))));
                }
                for (int j = 0; j < 2; ++j) {
                    T u = j == 0 ? v : w;
                    T pl = F(L.PL, j, k), pu = F(L.PU, j, k);
                    T r = gu[j] + gJ[1 + j] - pl + pu;
                    for (int q = j; q < 4; q += 2) {
                        const T sg = slot_sign<T>(q);
                        if (row_on(k, q)) r += sg * F(L.YR, q, k);
                        if (row_on(k + 1, q)) r -= sg * F(L.YR, q, k + 1);
                    }
                    rd = t_max(rd, t_abs(r));
                    T cl = (u - P.u_lb[j]) * pl, cu = (P.u_ub[j] - u) * pu;
                    cmin = t_min(cmin, t_min(cl, cu)); cmax = t_max(cmax, t_max(cl, cu)); cs += cl + cu;
                    sb += pl + pu; nb += 2;
                }
                if (k == n - 2) {
                    T ta[3] = {T(0), T(0), T(0)}, ty = T(0);
                    if (ball()) {        // terminal l2-ball row: value / gradient cache for the step, residuals
                        const T tg = ball_eval(T(0), ta), ts = SCL(SC_TS);
                        ty = SCL(SC_TY);
                        SCL(SC_TG) = tg; SCL(SC_TA) = ta[0]; SCL(SC_TA + 1) = ta[1]; SCL(SC_TA + 2) = ta[2];
                        rp = t_max(rp, t_abs(tg + ts)); th += t_abs(tg + ts);
                        cmin = t_min(cmin, ts * ty); cmax = t_max(cmax, ts * ty); cs += ts * ty;
                        sb += ty; nb += 1;
                    }
                    T gext[3] = {T(0), T(0), T(0)};
                    if (costx()) {       // off-diagonal terminal cost, trapezoid term of the final state (gradient, and its share of d/d dt)
                        T xdT[3], y3[3]; xd_final(T(0), xdT);
                        if (hasqf()) { offmul(P.Qfo, xdT, y3); for (int i = 0; i < 3; ++i) gext[i] += T(2) * y3[i]; }
                        if (P.trapz) {
                            offmul(P.Qo, xdT, y3);
                            for (int i = 0; i < 3; ++i) gext[i] += d * (P.Q[i] * xdT[i] + y3[i]);
                            rdd += T(0.5) * fullquad(P.Q, P.Qo, xdT);
                        }
                    }
                    for (int i = 0; i < 3; ++i) if (!fx(i)) {
                        T g = gext[i];
                        if (hasqf()) {
                            T xd = F(L.X, i, n - 1) - xf[i];
                            if (i == 2) xd = normalize_theta(xd);
                            g += T(2) * P.Qf[i] * xd;
                        }
                        rd = t_max(rd, t_abs(g + ty * ta[i] - lam[i]));
                    }
                }
            }
            for (int q = 0; q < 4; ++q) {
                if (!row_on(k, q)) continue;
                T s = F(L.SR, q, k), y = F(L.YR, q, k);
                T res = row_val(L.U, d, k, q) + s;
                rp = t_max(rp, t_abs(res)); th += t_abs(res);
                cmin = t_min(cmin, s * y); cmax = t_max(cmax, s * y); cs += s * y;
                sb += y; nb += 1;
                if (k > 0) rdd -= slot_sign<T>(q) * P.rate_lim[q] * y;
            }
            if (k < n - 1) {
                if constexpr (GS) { for (int i = 0; i < 3; ++i) gw((unsigned)(GlobalStage::CC + 3 * k + i)) = C_(i, k); }      // c_k where the sweeps' pointers live
                S_(0, k) = rec[0]; S_(1, k) = rec[1]; S_(2, k) = T(1);                 // column 2 of Ghat: (a0, a1, 1)
                S_(3, k) = rec[2]; S_(4, k) = rec[3]; S_(5, k) = rec[4];
                for (int a = 0; a < 3; ++a) { S_(6 + a, k) = rec[5 + a]; S_(9 + a, k) = rec[8 + a]; }   // Bx column-major
                // raw (mu-independent) pieces parked in their A slots; stage_barrier_terms() turns them into the combined entries
                S_(RA + A22, k) = rec[11]; S_(RA + A26, k) = rec[12]; S_(RA + A27, k) = rec[13];
                S_(RA + A66, k) = rec[14]; S_(RA + A67, k) = rec[15]; S_(RA + A77, k) = rec[16];
                S_(RA + A25, k) = rec[17]; S_(RA + A56, k) = rec[18]; S_(RA + A57, k) = rec[19]; S_(RA + A55, k) = rec[20];
            }
        }
        if (lane == 0) {
            if (mintime()) rdd += T(n - 1);
            if (dtf()) {
                T pl = SCL(SC_PDL), pu = SCL(SC_PDU);
                rdd += -pl + pu;
                T cl = (d - P.dt_lb) * pl, cu = (P.dt_ub - d) * pu;
                cmin = t_min(cmin, t_min(cl, cu)); cmax = t_max(cmax, t_max(cl, cu)); cs += cl + cu;
                sb += pl + pu; nb += 2;
            }
        }
        Err e;
        rdd = wave_sum(rdd);
        e.rd = wave_max(rd);
        if (dtf()) e.rd = t_max(e.rd, t_abs(rdd));
        e.rp = wave_max(rp);
        e.cmin = wave_min(cmin);
        e.cmax = wave_max(cmax);
        e.csum = wave_sum(cs);
        e.sum_bmult = wave_sum(sb);
        e.sum_mult = wave_sum(smult) + e.sum_bmult;
        e.theta = wave_sum(th);
        if (nM() == 0 && cnt_bmult >= 0) { e.n_bmult = cnt_bmult; e.n_mult = cnt_mult; }      // without clearance rows the counts never change
        else {
            e.n_bmult = (int)wave_sum((T)nb);
            e.n_mult = (int)wave_sum((T)nm) + e.n_bmult;
            cnt_bmult = e.n_bmult; cnt_mult = e.n_mult;
            inv_cnt_bmult = T(1) / T(e.n_bmult > 0 ? e.n_bmult : 1); inv_cnt_mult = T(1) / T(e.n_mult > 0 ? e.n_mult : 1);
        }
        return e;
    }

    // parallel: combine the raw pieces with the mu-dependent condensed barrier terms into the A-form; record n-1 = final rate rows
    __device__ __forceinline__ void stage_barrier_terms() const {
        const int n = L.n;
        const T d = SCL(SC_D);
        const bool quad = this->quad();
        T q2[3] = {T(0), T(0), T(0)}, r2[2] = {T(0), T(0)};
        if (quad) { const T w8 = intf() ? d : T(1); for (int i = 0; i < 3; ++i) q2[i] = T(2) * P.Q[i] * w8; for (int j = 0; j < 2; ++j) r2[j] = T(2) * P.R[j] * w8; }
        for (int k = lane; k < n; k += kWave) {
            StageParts<T> sp;
            sp.h00 = S_(RA + A22, k); sp.h01 = S_(RA + A26, k); sp.h02 = S_(RA + A27, k);
            sp.h11 = S_(RA + A66, k); sp.h12 = S_(RA + A67, k); sp.h22 = S_(RA + A77, k);
            sp.g[0] = S_(RA + A25, k); sp.g[1] = S_(RA + A56, k); sp.g[2] = S_(RA + A57, k);
            sp.hdd = k < n - 1 ? S_(RA + A55, k) : T(0);
            if (EXT && hessm() && k < n - 1) {       // MPC_HESSIAN_CONVEXIFIED: positive semidefinite part of the stage's curvature block (theta, v, w, dt)
                StageMap<T> pm;
                pm.Hqq[0][0] = sp.h00; pm.Hqq[0][1] = pm.Hqq[1][0] = sp.h01; pm.Hqq[0][2] = pm.Hqq[2][0] = sp.h02;
                pm.Hqq[1][1] = sp.h11; pm.Hqq[1][2] = pm.Hqq[2][1] = sp.h12; pm.Hqq[2][2] = sp.h22;
                pm.Hqd[0] = sp.g[0]; pm.Hqd[1] = sp.g[1]; pm.Hqd[2] = sp.g[2]; pm.Hdd = sp.hdd;
                psd_project4(pm, k == 0);
                sp.h00 = pm.Hqq[0][0]; sp.h01 = pm.Hqq[0][1]; sp.h02 = pm.Hqq[0][2]; sp.h11 = pm.Hqq[1][1]; sp.h12 = pm.Hqq[1][2]; sp.h22 = pm.Hqq[2][2];
                sp.g[0] = pm.Hqd[0]; sp.g[1] = pm.Hqd[1]; sp.g[2] = pm.Hqd[2]; sp.hdd = pm.Hdd;
            }
            sp.hx[0] = sp.hx[1] = sp.hx[2] = T(0);
            if (quad && k < n - 1) {
                sp.hx[0] = q2[0] * (F(L.X, 0, k) - xf[0]); sp.hx[1] = q2[1] * (F(L.X, 1, k) - xf[1]);
                sp.hx[2] = q2[2] * normalize_theta(F(L.X, 2, k) - xf[2]);
            }
            sp.sz[0] = sp.sz[1] = sp.gb[0] = sp.gb[1] = T(0);
            if (k < n - 1) {
                for (int j = 0; j < 2; ++j) {
                    T u = F(L.U, j, k);
                    const T idl = t_rcp(u - P.u_lb[j]), idu = t_rcp(P.u_ub[j] - u);      // one reciprocal per bound, shared by all quotients
                    sp.sz[j] = F(L.PL, j, k) * idl + F(L.PU, j, k) * idu;
                    sp.gb[j] = mu * idu - mu * idl + r2[j] * u;
                }
            }
            sp.ss[0] = sp.ss[1] = sp.sl[0] = sp.sl[1] = sp.sll = sp.gy[0] = sp.gy[1] = sp.gyl = T(0);
            for (int q = 0; q < 4; ++q) {
                if (!row_on(k, q)) continue;
                const int j = q & 1;
                const T sg = slot_sign<T>(q), lim = k > 0 ? P.rate_lim[q] : T(0);
                T s = F(L.SR, q, k), y = F(L.YR, q, k);
                const T is = t_rcp(s);
                T sig = y * is;
                T ybar = mu * is + sig * (row_val(L.U, d, k, q) + s);
                sp.ss[j] += sig; sp.sl[j] += sig * lim; sp.sll += sig * lim * lim;
                sp.gy[j] += sg * ybar; sp.gyl += sg * lim * ybar;
            }
            sp.oxx = sp.oxy = sp.oyy = sp.ogx = sp.ogy = T(0);
            if (nM() > 0 && k >= 1 && k < n - 1) {
                for (int m = 0; m < nM(); ++m) {
                    if (oi(m, k) < 0) continue;
                    const T s = F(L.OS, m, k), y = F(L.OY, m, k), g = OB_(0, m, k);
                    const T ax = OB_(1, m, k), ay = OB_(2, m, k), hk = OB_(3, m, k);
                    const T is = t_rcp(s);
                    T sig = y * is;
                    T ybar = mu * is + sig * (g + s);
                    if (erho > T(0)) elastic_condense(s, y, g, T(OE_(0, m, k)), sig, ybar);
                    // hess(g) = -hk (I - a a')
                    sp.oxx += sig * ax * ax - y * hk * (T(1) - ax * ax);
                    sp.oxy += sig * ax * ay + y * hk * ax * ay;
                    sp.oyy += sig * ay * ay - y * hk * (T(1) - ay * ay);
                    sp.ogx += ax * ybar; sp.ogy += ay * ybar;
                    if (fpline()) {
                        const T at = F(L.OAT, m, k);
                        sp.oxt += sig * ax * at + y * F(L.OHXT, m, k);
                        sp.oyt += sig * ay * at + y * F(L.OHYT, m, k);
                        sp.ott += sig * at * at + y * F(L.OHTT, m, k);
                        sp.ogt += at * ybar;
                    }
                    if (dynturn()) {      // dt parts next to the heading parts: x dt, y dt, theta dt, dt dt
                        const T ad = F(L.OAD, m, k), at = F(L.OAT, m, k);
                        sp.cxd[0] += sig * ax * ad + y * F(L.OHXD, m, k);
                        sp.cxd[1] += sig * ay * ad + y * F(L.OHYD, m, k);
                        sp.cxd[2] += sig * at * ad + y * F(L.OHTD, m, k);
                        sp.hdd += sig * ad * ad + y * F(L.OHDD, m, k);
                        sp.gdt += ad * ybar;
                    } else if (dynobs()) {       // dt parts of the rows of dynamic obstacles (zero for the static ones)
                        const T ad = F(L.OAT, m, k);
                        sp.cxd[0] += sig * ax * ad + y * F(L.OHXT, m, k);
                        sp.cxd[1] += sig * ay * ad + y * F(L.OHYT, m, k);
                        sp.hdd += sig * ad * ad + y * F(L.OHTT, m, k);
                        sp.gdt += ad * ybar;
                    }
                }
            }
            if (intf() && k < n - 1) {       // d/ddt and the mixed second derivatives of  dt * (xd'Q xd + u'R u)
                const T xd0 = F(L.X, 0, k) - xf[0], xd1 = F(L.X, 1, k) - xf[1], xd2 = normalize_theta(F(L.X, 2, k) - xf[2]);
                const T v = F(L.U, 0, k), w = F(L.U, 1, k);
                if (k > 0) { sp.cxd[0] += T(2) * P.Q[0] * xd0; sp.cxd[1] += T(2) * P.Q[1] * xd1; sp.cxd[2] += T(2) * P.Q[2] * xd2; }   // x_0 is not a variable
                sp.cud[0] = T(2) * P.R[0] * v; sp.cud[1] = T(2) * P.R[1] * w;
                sp.gdt += P.Q[0] * xd0 * xd0 + P.Q[1] * xd1 * xd1 + P.Q[2] * xd2 * xd2 + P.R[0] * v * v + P.R[1] * w * w;
            }
            if (via() && k >= 1 && k < n - 1) {
                T vv, vg[3];
                const int m = via_terms(k, F(L.X, 0, k), F(L.X, 1, k), F(L.X, 2, k), vv, vg);
                const T h = T(2) * P.vp_wp * T(m);
                sp.oxx += h; sp.oyy += h; sp.ogx += vg[0]; sp.ogy += vg[1]; sp.hx[2] += vg[2];
            }
            if (costx() && quad && k < n - 1) {       // off-diagonal weights: Hessian slots A01 A02 A12 A67, gradients, dt coupling; trapezoid: x_0's half term
                const T w8 = intf() ? d : T(1);
                const T xd[3] = {F(L.X, 0, k) - xf[0], F(L.X, 1, k) - xf[1], normalize_theta(F(L.X, 2, k) - xf[2])};
                const T v = F(L.U, 0, k), w = F(L.U, 1, k);
                T qo[3]; offmul(P.Qo, xd, qo);
                sp.oxy += T(2) * P.Qo[0] * w8; sp.oxt += T(2) * P.Qo[1] * w8; sp.oyt += T(2) * P.Qo[2] * w8;
                sp.h12 += T(2) * P.Ro * w8;
                for (int i = 0; i < 3; ++i) sp.hx[i] += T(2) * qo[i] * w8;
                sp.gb[0] += T(2) * P.Ro * w * w8; sp.gb[1] += T(2) * P.Ro * v * w8;
                if (intf()) {
                    if (k > 0) for (int i = 0; i < 3; ++i) sp.cxd[i] += T(2) * qo[i];
                    sp.cud[0] += T(2) * P.Ro * w; sp.cud[1] += T(2) * P.Ro * v;
                    sp.gdt += offquad(P.Qo, xd) + T(2) * P.Ro * v * w;
                    if (k == 0 && P.trapz) sp.gdt -= T(0.5) * fullquad(P.Q, P.Qo, xd);
                }
            }
            T A[NADD];
            assemble_adds(sp, q2, r2, A);
#pragma unroll
            for (int i = 0; i < NADDv; ++i) S_(RA + i, k) = A[i];
        }
    }

    // ---------------------------------------------------------------- backward Riccati sweep
    // terminal value function of the backward sweeps (serial and partitioned), straight into the owning lanes' registers: column c of [P | p | S];
    // a3 / a4 / a5 = this lane's entries of the rows 3..5 of the final rate rows' record (stage n-1: the A slots (i, c) for c in {3, 4, 5, 8})
    __device__ __forceinline__ void terminal_value(T (&V)[6], const int c, const T delta, const T d, const T a3, const T a4, const T a5) const {
        const int n = L.n;
        // condensed terminal l2-ball row: + sigma a a' + 2 y S on the final-state block, + a ybar on its gradient
        T tsig = T(0), tyb = T(0), ty = T(0), ta[3] = {T(0), T(0), T(0)};
        if (ball()) {
            const T ts = SCL(SC_TS), tg = SCL(SC_TG);
            ty = SCL(SC_TY);
            const T its = t_rcp(ts);
            tsig = ty * its; tyb = mu * its + tsig * (tg + ts);
            ta[0] = SCL(SC_TA); ta[1] = SCL(SC_TA + 1); ta[2] = SCL(SC_TA + 2);
        }
        const T tac = c < 3 ? (c == 0 ? ta[0] : (c == 1 ? ta[1] : ta[2])) : T(0);
        // cost variants: off-diagonal terminal weights (Qf, S) and the trapezoid term 0.5 dt xd' Q xd of the final state (x-x, x-dt, gradients)
        T xdT[3] = {T(0), T(0), T(0)}, qT[3] = {T(0), T(0), T(0)};       // qT = Q xd_T (full) when the trapezoid term exists
        if (costx()) { xd_final(T(0), xdT); if (P.trapz) { offmul(P.Qo, xdT, qT); for (int i = 0; i < 3; ++i) qT[i] += P.Q[i] * xdT[i]; } }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (fx(i)) V[i] = c == 9 + i ? T(1) : T(0);
            else {
                T pii = delta, pi_ = T(0);
                if (hasqf()) {
                    T xd = F(L.X, i, n - 1) - xf[i];
                    if (i == 2) xd = normalize_theta(xd);
                    pii += T(2) * P.Qf[i]; pi_ = T(2) * P.Qf[i] * xd;
                }
                if (ball()) { pii += T(2) * ty * P.ball_S[i]; pi_ += ta[i] * tyb; }
                V[i] = c == i ? pii : (c == 8 ? pi_ : T(0));
                if (ball() && c < 3) V[i] += tsig * ta[i] * tac;       // a of a fixed component is 0
                if (costx()) {
                    if (c < 3 && !fx(c)) {         // Hessian entry (i, c) over the free final components
                        const int oi = i + c - 1;  // (0,1) -> 0, (0,2) -> 1, (1,2) -> 2
                        if (c != i) { if (hasqf()) V[i] += T(2) * P.Qfo[oi]; if (ball()) V[i] += T(2) * ty * P.So[oi]; if (P.trapz) V[i] += d * P.Qo[oi]; }
                        else if (P.trapz) V[i] += d * P.Q[i];
                    }
                    if (c == 8) {
                        T y3[3];
                        if (hasqf()) { offmul(P.Qfo, xdT, y3); V[i] += T(2) * y3[i]; }
                        if (P.trapz) V[i] += d * qT[i];
                    }
                    if (c == 5 && P.trapz) V[i] += qT[i];
                }
            }
        }
        {
            const bool t3 = c == 3 || c == 5 || c == 8, t4 = c == 4 || c == 5 || c == 8, t5 = c == 3 || c == 4 || c == 5 || c == 8;
            V[3] = t3 ? a3 : T(0); V[4] = t4 ? a4 : T(0); V[5] = t5 ? a5 : T(0);
            if (costx() && P.trapz) {
                if (c < 3 && !fx(c)) V[5] += c == 0 ? qT[0] : (c == 1 ? qT[1] : qT[2]);
                if (c == 8) V[5] += T(0.5) * (xdT[0] * qT[0] + xdT[1] * qT[1] + xdT[2] * qT[2]);
            }
        }
    }

    __device__ __forceinline__ T fast_rcp(double x) const {
        double r = __builtin_amdgcn_rcp(x);
        r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
        return r;
    }
    __device__ __forceinline__ T fast_rcp(float x) const { return 1.0f / x; }
    __device__ __forceinline__ static int sign_word(double x) { return __double2hiint(x); }      // the word that holds the sign bit
    __device__ __forceinline__ static int sign_word(float x) { return __float_as_int(x); }

    // ---- register-resident backward sweep: lane c (0..11 of every 16-lane DPP row) owns COLUMN c of the value block
    //      [P | . . | p | S] (6x12) and of Hhat (8x12).  Entries of other columns are fetched with the DP-ALU DPP broadcast
    //      (v_fmac_f64_dpp ... row_newbcast:m), so a stage is ~90 fp64 VALU instructions and NO LDS hand-off; the stage
    //      record (3 coefficients + 8 cost entries per lane) is prefetched from LDS one stage ahead.
    //        T1 = V+ Ghat           T1[i][c] = sum_{m<3} V+[i][m] G[m][c] + V+[i][x(c)]
    //        Hhat = Ghat' T1 + cost Hhat[r][c] = sum_{m<3} G[m][r] T1[m][c] + T1[x(r)][c] + A[r][c]
    //        V = Hhat_xx - Hhat_xu R^-1 Hhat_ux   (R = Hhat[6:8][6:8], closed-form inverse; symmetry: Hhat[i][6] = lane i's Hhat[6][.])
    //      Cost model on gfx950 with one wave per SIMD (scripts/ubench/issue_rate.hip): EVERY instruction, VALU or not, costs
    //      ~4.5 cycles of issue, an LDS write->read hand-off ~115 cycles, s_nop 1 ~8 cycles: the sweep is written to minimise
    //      the instruction count.  The DPP arithmetic lives in four inline-asm blocks per stage (mpc_dpp_blocks.inc, generated
    //      by scripts/gen_dpp_blocks.py): the compiler cannot see a DPP operand inside inline asm and therefore does not insert
    //      the 2 wait states of the VALU-write -> DPP-read hazard; each block opens with s_nop 1 and orders its own instructions.
#ifdef MPC_DPP_DEBUG     // developer aid: single-instruction DPP helpers to bisect the generated blocks (each pays its own s_nop)
#define MPC_BC_CASE(N) else if constexpr (LANE == N) { \
        if constexpr (sizeof(T) == 8) asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own)); \
        else asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own)); }
    template <int LANE> __device__ __forceinline__ static void fmac_bc(T& acc, T src, T own) {
        if constexpr (LANE < 0) {}
        MPC_BC_CASE(0) MPC_BC_CASE(1) MPC_BC_CASE(2) MPC_BC_CASE(3) MPC_BC_CASE(4) MPC_BC_CASE(5)
        MPC_BC_CASE(6) MPC_BC_CASE(7) MPC_BC_CASE(8) MPC_BC_CASE(9) MPC_BC_CASE(10) MPC_BC_CASE(11)
    }
#undef MPC_BC_CASE
#endif
    __device__ __forceinline__ bool backward_dpp(T delta, T dc, T& dd_out, T nu_out[3]) const {
#ifdef MPC_PROFILE
        const long long ts0 = __builtin_readcyclecounter();
#endif
#ifdef MPC_ASM_MARK
        asm volatile("; BWD_SETUP_BEGIN");
#endif
        const int n = L.n;
        const T d = SCL(SC_D);
        const int ZC = W_ZC();                          // constants 0 0 0 0 1 0 0 0 (written once per solve): (0,0,0) @0, (0,1,0) @3, (1,0,0) @4
        const int c = lane & 15;                    // column owned by this lane (12..15 idle: they carry zeros)
        const bool act = c < 12;
        // coefficient triple of column c: three consecutive words, running pointer (stride 0 for the constant triples).
        // kind: 0 (0,0,0)  1 (1,0,0)  2 (0,1,0)  3 (a0,a1,1)  4 f  5 Bx[:,0]  6 Bx[:,1]  7 c_k   -- columns 0..11, 3 bits each
        constexpr unsigned long long KIND = 1ull | (2ull << 3) | (3ull << 6) | (4ull << 15) | (5ull << 18) | (6ull << 21) | (7ull << 24);
        const int kind = act ? (int)((KIND >> (3 * c)) & 7) : 0;
        const int gb = kind < 3 ? ZC + (kind == 1 ? 4 : (kind == 2 ? 3 : 0)) : (kind < 7 ? W_STG() + 3 * (kind - 3) : W_CC());
        const int gsw = kind < 3 ? 0 : (kind < 7 ? NSTG : 3);
        const int gs = sw_step(gsw);
        SwCRef gp = sw(gb + (n - 2) * gsw);
        SwCRef ap[8];
        int as_[8];
        constexpr unsigned long long rows[8] = {stage_add_row(0, EXT), stage_add_row(1, EXT), stage_add_row(2, EXT), stage_add_row(3, EXT),
                                                stage_add_row(4, EXT), stage_add_row(5, EXT), stage_add_row(6, EXT), stage_add_row(7, EXT)};
        const int sh = act ? 5 * c : 60;               // idle lanes: shift the row word out (slot -1)
        const int abase = W_STG() + RA - 1 + (n - 2) * NSTG;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int slot1 = (int)((rows[r] >> sh) & 31);       // slot + 1, 0 = structurally zero
            as_[r] = sw_step(slot1 ? NSTG : 0);
            ap[r] = sw(slot1 ? abase + slot1 : ZC);
        }
        const T ec = (c == 5 || (c >= 8 && c < 12)) ? T(1) : T(0);       // own column enters T1 (dt, p, S)
        const T E3 = c == 6 ? T(1) : T(0), E4 = c == 7 ? T(1) : T(0);     // the u columns pick up the u_prev columns of V+
        const T dA0 = c == 0 ? delta : T(0), dA1 = c == 1 ? delta : T(0), dA2 = c == 2 ? delta : T(0);   // x diagonal, k >= 1
        const T dA6 = c == 6 ? delta : T(0), dA7 = c == 7 ? delta : T(0);                                 // u diagonal
        // negated gains go to GAIN as [nK0 (cols 0..5) | nkappa0 | nKnu0 (3 of 5) | nK1 ... ] : g1 = g0 + NGH; idle lanes hit a dummy pair
        const bool wrG = lane < 12 && c != 6 && c != 7;
        const int g0 = c < 6 ? c : (c == 8 ? 6 : 7 + (c - 9));
        SwRef kp = sw(wrG ? W_GAIN() + g0 + (n - 2) * NGAIN : W_VP());      // idle lanes: dummy pair in the scratch area
        const int ks = sw_step(wrG ? NGAIN : 0);
        // ---- terminal value function, straight into the owning lanes' registers (rows 3..5: the u_prev / dt entries of the
        //      final rate rows = the A slots (i, c) of stage n-1 for c in {3, 4, 5, 8})
        T V[6];
        terminal_value(V, c, delta, d, sw_ld_at(ap[3], as_[3]), sw_ld_at(ap[4], as_[4]), sw_ld_at(ap[5], as_[5]));       // rows 3..5: one stage above the running pointers
        T add_dd0 = T(0), add_qd0 = T(0);
        if (mintime()) add_qd0 += T(n - 1);
        if (dtf()) {
            const T dl = d - P.dt_lb, du = P.dt_ub - d;
            const T idl = fast_rcp(dl), idu = fast_rcp(du);
            add_dd0 = SCL(SC_PDL) * idl + SCL(SC_PDU) * idu + delta;
            add_qd0 += mu * idu - mu * idl;
        }
        const T s05 = c == 5 ? add_dd0 : (c == 8 ? add_qd0 : T(0));       // stage-0 extras of row 5
        T om = T(0), wn[3] = {T(0), T(0), T(0)};
        T worst = T(1);                                                   // min over the stages of |det R| - 1e-14 * scale
        int negc = 0;                                                     // negative eigenvalues of the control pivots = sign changes of (1, R00, det R), summed over the stages
        auto load_stage = [&](T (&g)[3], T (&a)[8]) {                     // reads the stage the running pointers are at, then steps them
            g[0] = sw_ld(gp, 0); g[1] = sw_ld(gp, 1); g[2] = sw_ld(gp, 2);
            gp -= gs;
#pragma unroll
            for (int r = 0; r < 8; ++r) { a[r] = sw_ld(ap[r]); ap[r] -= as_[r]; }
        };
        auto stage = [&](T dk0, T dk1, T dk2, T s5, T (&G)[3], T (&A)[8], T (&Gn)[3], T (&An)[8]) {
            load_stage(Gn, An);                                           // prefetch of the next stage (k - 1)
            // ---- T1 = V+ Ghat (own column, then the five broadcast columns) and
            //      omega[b] += S+[0][b] c0 + S+[1][b] c1 + S+[2][b] c2  (lanes 9..11; c_k is lane 8's coefficient triple)
            T t[6];
#if defined(MPC_DPP_DEBUG) && (MPC_DPP_DEBUG & 1)
            for (int i = 0; i < 6; ++i) t[i] = V[i] * ec;
            for (int i = 0; i < 6; ++i) fmac_bc<0>(t[i], V[i], G[0]);
            for (int i = 0; i < 6; ++i) fmac_bc<1>(t[i], V[i], G[1]);
            for (int i = 0; i < 6; ++i) fmac_bc<2>(t[i], V[i], G[2]);
            for (int i = 0; i < 6; ++i) fmac_bc<3>(t[i], V[i], E3);
            for (int i = 0; i < 6; ++i) fmac_bc<4>(t[i], V[i], E4);
            fmac_bc<8>(om, G[0], V[0]); fmac_bc<8>(om, G[1], V[1]); fmac_bc<8>(om, G[2], V[2]);
#else
            MPC_DPP_BLOCK_T1
#endif
            // ---- Hhat = Ghat' T1 + cost entries (+ regularisation on this lane's diagonal entry)
            T h[8];
            h[0] = (A[0] + dk0) + t[0]; h[1] = (A[1] + dk1) + t[1]; h[2] = (A[2] + dk2) + t[2];
            h[3] = A[3]; h[4] = A[4];
            h[5] = (A[5] + s5) + t[5];
            h[6] = (A[6] + dA6) + t[3]; h[7] = (A[7] + dA7) + t[4];
#if defined(MPC_DPP_DEBUG) && (MPC_DPP_DEBUG & 2)
            fmac_bc<6>(h[6], G[0], t[0]); fmac_bc<7>(h[7], G[0], t[0]); fmac_bc<5>(h[5], G[0], t[0]); fmac_bc<2>(h[2], G[0], t[0]);
            fmac_bc<6>(h[6], G[1], t[1]); fmac_bc<7>(h[7], G[1], t[1]); fmac_bc<5>(h[5], G[1], t[1]); fmac_bc<2>(h[2], G[1], t[1]);
            fmac_bc<6>(h[6], G[2], t[2]); fmac_bc<7>(h[7], G[2], t[2]); fmac_bc<5>(h[5], G[2], t[2]);
#else
            MPC_DPP_BLOCK_H
#endif
            // ---- Schur complement on the control block
            T R00, R01, R11;
            MPC_DPP_BLOCK_R
            const T r2 = R01 * R01;
            const T det = R00 * R11 - r2;
            worst = t_fmin(worst, t_abs(det) - T(1e-14) * (t_abs(R00 * R11) + r2));   // a NaN determinant poisons V and is caught by the root solve
            { const int s0 = sign_word(R00), s1 = sign_word(det); negc += (int)((unsigned)s0 >> 31) + (int)((unsigned)(s0 ^ s1) >> 31); }
            const T nid = -fast_rcp(det);
            const T nRi00 = R11 * nid, Ri01 = -(R01 * nid), nRi11 = R00 * nid;  // -R^-1 = [nRi00 Ri01; Ri01 nRi11]
            const T nK0 = nRi00 * h[6] + Ri01 * h[7], nK1 = Ri01 * h[6] + nRi11 * h[7];
            sw_st(kp, 0, nK0); sw_st(kp, NGH, nK1);
            kp -= ks;
            // W[a][b] -= Su[:,a]' Knu[:,b] in lane 9+b, omega[a] -= Su[:,a]' kappa in lane 8 (Su[j][a] = Hhat[6+j][9+a]);
            // V = Hhat_xx + Hhat_xu nK   (row i of Hhat[:,6:8] = lane i's Hhat[6:8][.])
            V[0] = h[0]; V[1] = h[1]; V[2] = h[2]; V[3] = h[3]; V[4] = h[4]; V[5] = h[5];
#if defined(MPC_DPP_DEBUG) && (MPC_DPP_DEBUG & 4)
            fmac_bc<9>(wn[0], h[6], nK0); fmac_bc<10>(wn[1], h[6], nK0); fmac_bc<11>(wn[2], h[6], nK0);
            fmac_bc<9>(wn[0], h[7], nK1); fmac_bc<10>(wn[1], h[7], nK1); fmac_bc<11>(wn[2], h[7], nK1);
            fmac_bc<0>(V[0], h[6], nK0); fmac_bc<1>(V[1], h[6], nK0); fmac_bc<2>(V[2], h[6], nK0);
            fmac_bc<3>(V[3], h[6], nK0); fmac_bc<4>(V[4], h[6], nK0); fmac_bc<5>(V[5], h[6], nK0);
            fmac_bc<0>(V[0], h[7], nK1); fmac_bc<1>(V[1], h[7], nK1); fmac_bc<2>(V[2], h[7], nK1);
            fmac_bc<3>(V[3], h[7], nK1); fmac_bc<4>(V[4], h[7], nK1); fmac_bc<5>(V[5], h[7], nK1);
#else
            MPC_DPP_BLOCK_V
#endif
        };
        T Ga[3], Aa[8], Gb[3], Ab[8];
#ifdef MPC_PROFILE
        const long long tl0 = __builtin_readcyclecounter();
        prof_setup += tl0 - ts0;
#endif
        load_stage(Ga, Aa);
        int k = n - 2;
        for (; k >= 2; k -= 2) {        // stages k and k-1 (both >= 1): the two register sets swap roles, no copies
#ifdef MPC_ASM_MARK
            asm volatile("; MAT_LOOP_BEGIN");
#endif
            stage(dA0, dA1, dA2, T(0), Ga, Aa, Gb, Ab);
            stage(dA0, dA1, dA2, T(0), Gb, Ab, Ga, Aa);
#ifdef MPC_ASM_MARK
            asm volatile("; MAT_LOOP_END");
#endif
        }
        if (k == 1) { stage(dA0, dA1, dA2, T(0), Ga, Aa, Gb, Ab); stage(T(0), T(0), T(0), s05, Gb, Ab, Ga, Aa); }
        else stage(T(0), T(0), T(0), s05, Ga, Aa, Gb, Ab);
#ifdef MPC_PROFILE
        prof_loop += __builtin_readcyclecounter() - tl0;
#endif
        // row 5 of the value block, omega and the W / omega corrections are gathered with v_readlane (no LDS round trip)
        if (!(rd_lane(worst, 0) > T(0))) return false;
        RicState<T> Vr;
        Vr.neg = __builtin_amdgcn_readlane(negc, 0);
        Vr.P[5][5] = rd_lane(V[5], 5);
        Vr.p[5] = rd_lane(V[5], 8);
        Vr.S[5][0] = rd_lane(V[5], 9); Vr.S[5][1] = rd_lane(V[5], 10); Vr.S[5][2] = rd_lane(V[5], 11);
        Vr.om[0] = rd_lane(om, 9) + rd_lane(wn[0], 8);
        Vr.om[1] = rd_lane(om, 10) + rd_lane(wn[1], 8);
        Vr.om[2] = rd_lane(om, 11) + rd_lane(wn[2], 8);
        {
            // W[a][b] partial lives in lane 9+b as wn[a]; symmetrise; the fixed components carry -delta_c on the diagonal
            const T w00 = rd_lane(wn[0], 9), w01 = rd_lane(wn[0], 10), w02 = rd_lane(wn[0], 11);
            const T w10 = rd_lane(wn[1], 9), w11 = rd_lane(wn[1], 10), w12 = rd_lane(wn[1], 11);
            const T w20 = rd_lane(wn[2], 9), w21 = rd_lane(wn[2], 10), w22 = rd_lane(wn[2], 11);
            Vr.W[0][0] = w00 - (fx(0) ? dc : T(0)); Vr.W[1][1] = w11 - (fx(1) ? dc : T(0)); Vr.W[2][2] = w22 - (fx(2) ? dc : T(0));
            Vr.W[0][1] = Vr.W[1][0] = T(0.5) * (w01 + w10);
            Vr.W[0][2] = Vr.W[2][0] = T(0.5) * (w02 + w20);
            Vr.W[1][2] = Vr.W[2][1] = T(0.5) * (w12 + w21);
        }
#ifdef MPC_PIT_VERBOSE
        if (blockIdx.x == MPC_PIT_CHECK && lane == 0)
            printf("  %s root: P55 %.10e p5 %.10e S5 %.10e %.10e %.10e om %.10e %.10e %.10e W %.10e %.10e %.10e %.10e %.10e %.10e\n", "serial", (double)Vr.P[5][5], (double)Vr.p[5],
                   (double)Vr.S[5][0], (double)Vr.S[5][1], (double)Vr.S[5][2], (double)Vr.om[0], (double)Vr.om[1], (double)Vr.om[2],
                   (double)Vr.W[0][0], (double)Vr.W[0][1], (double)Vr.W[0][2], (double)Vr.W[1][1], (double)Vr.W[1][2], (double)Vr.W[2][2]);
#endif
#ifdef MPC_NANCHECK
        {
            const bool okr = riccati_root(Vr, P, dd_out, nu_out) > 0;
            if (blockIdx.x == MPC_NANCHECK && lane == 0)
                printf("  root: ok %d worst %g P55 %g p5 %g S5 %g %g %g om %g %g %g W %g %g %g %g %g %g\n", (int)okr, (double)rd_lane(worst, 0), (double)Vr.P[5][5], (double)Vr.p[5],
                       (double)Vr.S[5][0], (double)Vr.S[5][1], (double)Vr.S[5][2], (double)Vr.om[0], (double)Vr.om[1], (double)Vr.om[2],
                       (double)Vr.W[0][0], (double)Vr.W[0][1], (double)Vr.W[0][2], (double)Vr.W[1][1], (double)Vr.W[1][2], (double)Vr.W[2][2]);
            return okr;
        }
#endif
        return riccati_root(Vr, P, dd_out, nu_out) > 0;
    }

    // inclusive suffix sum over the wave (lane i gets sum_{j >= i} v_j): Hillis-Steele inside each 16-lane row with DPP row
    // shifts (row_shl:n = 0x100 + n, out-of-row sources read as 0), then the totals of the higher rows via v_readlane.
    // ~27 VALU instructions and no LDS traffic (the ds_bpermute version was 12 LDS round trips).
    template <int CTRL> __device__ __forceinline__ static double dpp_shl0(double v) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
        hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    template <int CTRL> __device__ __forceinline__ static float dpp_shl0(float v) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
    }
    __device__ __forceinline__ T wave_suffix_sum(T v) const {
        v += dpp_shl0<0x101>(v);
        v += dpp_shl0<0x102>(v);
        v += dpp_shl0<0x104>(v);
        v += dpp_shl0<0x108>(v);
        const T r1 = rd_lane(v, 16), r2 = rd_lane(v, 32), r3 = rd_lane(v, 48);
        const T c2 = r3, c1 = r2 + r3, c0 = r1 + c1;
        const int row = lane >> 4;
        return v + (row == 0 ? c0 : (row == 1 ? c1 : (row == 2 ? c2 : T(0))));
    }

    // state recurrence (one component per lane, DPP broadcasts, prefetched coefficients) then multipliers by lane-parallel suffix scans
    __device__ __forceinline__ void forward_states(T dd, const T nu[3], T delta) const {
        const int n = L.n;
        // ---- lane-parallel: fold nu and dd into the affine terms so that the serial loop only carries (x, u_prev)
        //      kappa^ = kappa + Knu nu + K[:,5] dd  (stored over kappa),  c^ = c + f dd  (stored in LAMN, rewritten below)
        //      (gains are stored negated: [nK0 (6) | nkappa0 | nKnu0 (3) | nK1 (6) | nkappa1 | nKnu1 (3)])
        for (int k = lane; k < n - 1; k += kWave) {
            for (int a = 0; a < 2; ++a)
                G_(NGH * a + 6, k) += G_(NGH * a + 7, k) * nu[0] + G_(NGH * a + 8, k) * nu[1] + G_(NGH * a + 9, k) * nu[2] + G_(NGH * a + 5, k) * dd;
            for (int i = 0; i < 3; ++i) CH_(i, k) = C_(i, k) + S_(3 + i, k) * dd;
        }
        if (lane == 0) { SCL(SC_DD) = dd; F(L.DX, 0, 0) = T(0); F(L.DX, 1, 0) = T(0); F(L.DX, 2, 0) = T(0); }
        sync();
        // ---- serial recurrence, one COMPONENT per lane: lanes 0..2 carry dx_k, lanes 3,4 carry du_{k-1}; a stage is
        //        s  = cst + sum_j q_j * xi[j]        lanes 3,4: du_k = nkappa^ + nK xi ;  lanes 0..2: dx_k[c] + a_c dx_k[2] + c^_k[c]
        //        xn = s + b0 * s[3] + b1 * s[4]      lanes 0..2: + Bx[c][:] du_k
        //      i.e. 7 DPP-broadcast FMAs; the 8 per-lane coefficients are prefetched one stage ahead through running pointers.
        {
            const int c = lane & 15;
            const int ZC = W_ZC();
            int qw[8], qs[8];        // word index and stride of: cst, q0..q4, b0, b1
#pragma unroll
            for (int j = 0; j < 8; ++j) { qw[j] = ZC; qs[j] = 0; }
            if (c < 3) {
                qw[0] = W_CH() + c * L.NS; qs[0] = 1;
                qw[1 + c] = ZC + 4;                                          // xi[c] itself
                if (c < 2) { qw[3] = W_STG() + c; qs[3] = NSTG; }            // a_c * xi[2]
                qw[6] = W_STG() + 6 + c; qs[6] = NSTG;
                qw[7] = W_STG() + 9 + c; qs[7] = NSTG;
            } else if (c < 5) {
                const int a = c - 3;
                qw[0] = W_GAIN() + NGH * a + 6; qs[0] = NGAIN;
#pragma unroll
                for (int j = 0; j < 5; ++j) { qw[1 + j] = W_GAIN() + NGH * a + j; qs[1 + j] = NGAIN; }
            }
            SwCRef qp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { qp[j] = sw(qw[j]); qs[j] = sw_step(qs[j]); }
            // where the result goes: dx_{k+1}[c] / du_k[c-3]; idle lanes write a dummy word of the sweep scratch
            LdsT* op = lds(c < 3 ? L.DX + c * L.NS + 1 : (c < 5 ? L.DU + (c - 3) * L.NS : L.VP + 12));
            const int os = c < 5 ? 1 : 0;
            auto load_q = [&](T (&q)[8]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { q[j] = sw_ld(qp[j]); qp[j] += qs[j]; }
            };
            T xi = T(0);
            auto stage = [&](T (&q)[8], T (&qn)[8]) {
                load_q(qn);
                T s = q[0], s2 = T(0), xn;
#if defined(MPC_DPP_DEBUG) && (MPC_DPP_DEBUG & 8)
                fmac_bc<0>(s, xi, q[1]); fmac_bc<1>(s2, xi, q[2]); fmac_bc<2>(s, xi, q[3]); fmac_bc<3>(s2, xi, q[4]); fmac_bc<4>(s, xi, q[5]);
                s += s2;
                xn = s;
                fmac_bc<3>(xn, s, q[6]); fmac_bc<4>(xn, s, q[7]);
#else
                MPC_DPP_BLOCK_FWD
#endif
                *op = xn; op += os;
                xi = xn;
            };
#ifdef MPC_PROFILE
            const long long tf0 = __builtin_readcyclecounter();
#endif
            T qa[8], qb[8];
            load_q(qa);
            int k = 0;
            for (; k + 1 < n - 1; k += 2) {
#ifdef MPC_ASM_MARK
                asm volatile("; FWD_LOOP_BEGIN");
#endif
                stage(qa, qb); stage(qb, qa);
#ifdef MPC_ASM_MARK
                asm volatile("; FWD_LOOP_END");
#endif
            }
            if (k < n - 1) stage(qa, qb);
#ifdef MPC_PROFILE
            prof_fwd_loop += __builtin_readcyclecounter() - tf0;
#endif
        }
        multipliers(dd, nu, delta);
    }

    // multipliers of the collocation rows for ALL stages at once, from the primal step in DX / DU (lane-parallel suffix scans); the tail of both
    // forward passes (serial and partitioned)
    __device__ __forceinline__ void multipliers(T dd, const T nu[3], T delta) const {
        const int n = L.n;
        sync();
        const T xi[3] = {F(L.DX, 0, n - 1), F(L.DX, 1, n - 1), F(L.DX, 2, n - 1)};
        // ---- multipliers: lam+_{k-1} = lam+_k + t_k + e_theta (a0_k lam+_k[0] + a1_k lam+_k[1]),  k = n-2 .. 1,
        //      lam+_{n-2} from the terminal condition.  Components 0,1 are plain suffix sums, component 2 a second one.
        T lp[3];
        for (int i = 0; i < 3; ++i) {
            if (fx(i)) lp[i] = nu[i];
            else {
                T g = delta * xi[i];
                if (hasqf()) {
                    T xd = F(L.X, i, n - 1) - xf[i];
                    if (i == 2) xd = normalize_theta(xd);
                    g += T(2) * P.Qf[i] * (xi[i] + xd);
                }
                if (ball()) {
                    const T ts = SCL(SC_TS), ty = SCL(SC_TY), sig = ty / ts;
                    T adx = T(0);
                    for (int j = 0; j < 3; ++j) if (!fx(j)) adx += SCL(SC_TA + j) * xi[j];
                    g += SCL(SC_TA + i) * (sig * adx + mu / ts + sig * (SCL(SC_TG) + ts)) + T(2) * ty * P.ball_S[i] * xi[i];
                }
                if (costx()) {       // the rows of the terminal block that the variants add: (xi + xd) through the off-diagonal / trapezoid Hessian, dt coupling
                    T xdT[3], z[3], y3[3];
                    xd_final(T(0), xdT);
                    const T xif[3] = {fx(0) ? T(0) : xi[0], fx(1) ? T(0) : xi[1], fx(2) ? T(0) : xi[2]};
                    for (int j = 0; j < 3; ++j) z[j] = xif[j] + xdT[j];
                    if (hasqf()) { offmul(P.Qfo, z, y3); g += T(2) * y3[i]; }
                    if (ball()) { offmul(P.So, xif, y3); g += T(2) * SCL(SC_TY) * y3[i]; }
                    if (P.trapz) {
                        offmul(P.Qo, z, y3);
                        g += SCL(SC_D) * (P.Q[i] * z[i] + y3[i]);
                        offmul(P.Qo, xdT, y3);
                        g += (P.Q[i] * xdT[i] + y3[i]) * SCL(SC_DD);
                    }
                }
                lp[i] = g;
            }
        }
        sync();
        T carry[3] = {T(0), T(0), T(0)};                 // sum of t_m over the chunks already processed (m larger)
        const int last = n - 2;                          // stages m = 1 .. n-2 carry a term t_m
        for (int base = ((last) / kWave) * kWave; base >= 0; base -= kWave) {
            const int m = base + lane;
            const bool act = m >= 1 && m <= last;
            T t0 = T(0), t1 = T(0), t2 = T(0), a0 = T(0), a1 = T(0);
            if (act) {
                const T dx0 = F(L.DX, 0, m), dx1 = F(L.DX, 1, m), dx2 = F(L.DX, 2, m);
                const T duv = F(L.DU, 0, m), duw = F(L.DU, 1, m);
                t0 = delta * dx0 + S_(RA + A00, m) * dx0 + S_(RA + A01, m) * dx1 + S_(RA + A08, m);
                t1 = delta * dx1 + S_(RA + A01, m) * dx0 + S_(RA + A11, m) * dx1 + S_(RA + A18, m);
                t2 = delta * dx2 + S_(RA + A22, m) * dx2 + S_(RA + A26, m) * duv + S_(RA + A27, m) * duw + S_(RA + A25, m) * dd + S_(RA + A28, m);
                if (intf() || dynobs()) { t0 += S_(RA + A05, m) * dd; t1 += S_(RA + A15, m) * dd; }
                if (fpline() || costx()) {        // position-heading coupling of the clearance rows / of a full state weight matrix
                    const T c02 = S_(RA + A02, m), c12 = S_(RA + A12, m);
                    t0 += c02 * dx2; t1 += c12 * dx2; t2 += c02 * dx0 + c12 * dx1;
                }
                a0 = S_(0, m); a1 = S_(1, m);
            }
            const T s0 = wave_suffix_sum(t0) + carry[0];     // sum_{m' >= m} t_m'[0]
            const T s1 = wave_suffix_sum(t1) + carry[1];
            // lam+_m[0..1] = lp + sum_{m' >= m+1} t_m'  = lp + s - t_m
            const T l0 = lp[0] + s0 - t0, l1 = lp[1] + s1 - t1;
            const T t2f = act ? t2 + a0 * l0 + a1 * l1 : T(0);
            const T s2 = wave_suffix_sum(t2f) + carry[2];
            if (act) {       // lam+_{m-1} = lp + sum_{m' >= m} t_m'
                F(L.LAMN, 0, m - 1) = lp[0] + s0; F(L.LAMN, 1, m - 1) = lp[1] + s1; F(L.LAMN, 2, m - 1) = lp[2] + s2;
            }
            carry[0] = lane_bcast(s0, 0); carry[1] = lane_bcast(s1, 0); carry[2] = lane_bcast(s2, 0);
        }
        if (lane == 0) { F(L.LAMN, 0, last) = lp[0]; F(L.LAMN, 1, last) = lp[1]; F(L.LAMN, 2, last) = lp[2]; }
        }

    // ================================================================ partitioned ("parallel-in-time") sweeps
    // The serial sweeps above use 12 of 64 lanes and repeat the same arithmetic in all four 16-lane DPP rows.  Here the four rows work on four TIME
    // SEGMENTS of the horizon at once (tests/test_pit_math.py holds the algebra against a dense KKT solve):
    //   backward  row s < 3 sweeps its segment [s Lm, (s+1) Lm) from the identity border (P, p, S, W, om) = (0, 0, I, 0, 0): that yields the segment's
    //             scattering element, lam_a = P xi_a + S lam_b + p, xi_b = S' xi_a + W lam_b + om (the border multiplier is the costate at the segment's
    //             end: five columns, lanes 9..13; the dt column, lane 14, stays e_5); row 3 sweeps the last segment from the terminal value function, after
    //             the N mod 4 leftover stages have been swept by all rows together.  Same stage code as the serial sweep (only the V block differs).
    //   combine   right to left, three times: element + value function at its end -> value function at its start (combine()): every row does the same
    //             arithmetic; the element travels from its row to all rows through one LDS tile (in the step arrays, which are free during a factorisation)
    //   root      as in the serial sweep, from the value function at stage 0
    //   forward   boundary states / costates from the combine's maps (wave-uniform), gains folded per stage with the costate of the stage's own segment,
    //             then every row runs the state recurrence of its segment; the leftover stages follow in row 3
    // What the backward half leaves for the forward half, per mid-segment s: the eliminated tile [X | y | Z] (xi_{b_{s+1}} = X xi_{b_s} + y + Z nu) and the
    // value function [P+ | p+ | S+] at the boundary b_{s+1}, rows 0..4 x 10 columns each, in LDS words that are free during a factorisation: s = 2 in
    // LAMN, s = 1 in the trig cache (read by kkt_pass, rewritten by every line-search trial), s = 0 in DX (over the hand-off tile, dead by then).
    __device__ __forceinline__ int pit_tile(int s) const { return s == 2 ? L.LAMN : (s == 1 ? L.TRIG : L.DX); }
    // Per-lane constants of a combine step, branch-free integer arithmetic (computed once per factorisation for both lane-set variants).
    // Lane sets: A = lanes 0..5, B = lanes 6, 7, 12..15; lp_a: the value function's P+ columns sit in A and the result's in B, else the other way round.
    struct CombLane {
        int bb;        // base tile: S' in the result's lanes, om in lane 8, the identity in the value function's lanes, zeros elsewhere (row i = word i from here)
        int pb;        // the element's own P in the result's lanes, p in lane 8 (row i = 16 i words from here); the other lanes are masked by keep_p
        int sx, sp;    // slot (0..9, -1 = none) of this lane in a saved tile whose block sits in the result's / the value function's lane set
        T sig;         // -1 in the value function's lanes (M = I - W P+), +1 elsewhere
        T keep;        // 1 in the result's lanes and in lanes 8..11
        T keep_p;      // 1 in the result's lanes and in lane 8
    };
    __device__ __forceinline__ CombLane comb_lane(int c, bool lp_a, int TB) const {
        const int in_a = c < 6 ? 1 : 0, in_b = ((c == 6) | (c == 7) | (c >= 12)) ? 1 : 0, is8 = c == 8 ? 1 : 0, is_s = ((c >= 9) & (c < 12)) ? 1 : 0;
        const int qa = c, qb = c < 8 ? c - 6 : c - 10;                     // position inside A / B (meaningful where in_a / in_b)
        const int in_p = lp_a ? in_a : in_b, in_x = lp_a ? in_b : in_a, pp = lp_a ? qa : qb, px = lp_a ? qb : qa;
        CombLane r;
        r.bb = in_x * (TB + 16 * px + 9) + is8 * (TB + 176 + 9) + in_p * (L.ZI + 6 - pp) + (1 - in_x - is8 - in_p) * L.ZI;
        r.pb = TB + in_x * px + is8 * 8 + (1 - in_x - is8) * 15;
        const int tail = (is8 | is_s) * (c - 2) - (1 - (is8 | is_s));     // lanes 8..11 -> slots 6..9, else -1
        r.sx = in_x * px + (1 - in_x) * tail;
        r.sp = in_p * pp + (1 - in_p) * tail;
        r.sig = T(1 - 2 * in_p);
        r.keep = T(in_x | is8 | is_s);
        r.keep_p = T(in_x | is8);
        return r;
    }
    // saved tile: rows 0..4 x 10 slots; every row of the wave holds the same data, so all of them store (same words, same values)
    __device__ __forceinline__ void pit_save(int base, int slot, const T (&M)[6]) const {
        if (slot >= 0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) sm[base + 10 * i + slot] = M[i];
        }
    }
    __device__ __forceinline__ void pit_load(int base, int slot, T (&M)[5]) const {
        const int a = base + (slot >= 0 ? slot : 0);
#pragma unroll
        for (int i = 0; i < 5; ++i) M[i] = sm[a + 10 * i];
    }
    // (EXT = 2, the cost variants: serial sweeps only -- register budget.)  The partitioned sweep parks its tiles in live solver arrays (pit_tile): the hand-off
    // tile (192 words) in DX | DU (5 NS words), the saved tile pairs (100 words each) in LAMN (3 NS words), the trig cache (NTR NS words) and DX.  The capacities
    // are part of the condition, not a consequence of the grid-size threshold.  Second invariant: a saved tile overwrites the trig cache, which
    // kkt_pass reads -- every path from a factorisation to the next kkt_pass rewrites TRIG for all k < n - 1 (eval_point / trial_eval of the accepted trial;
    // the solve ends without another kkt_pass when no trial is evaluated).
    // r04, inertia: a factorisation is accepted when the KKT matrix has Ipopt's inertia (riccati_root).  The serial sweep reads it off the signs of its control pivots R_k.  Here
    // every row counts the negative eigenvalues of the pivots of ITS segment (those of the segment's own cost-to-go from the identity border, not the serial ones), and
    // every combine eliminates the pair (costate, state) at a boundary: the block [[W, -I], [-I, P+]] over the five components that have a costate column has the inertia
    // In(W) + In(P+ - W^-1) (Haynsworth), five negative eigenvalues when all is well -- pit_block_inertia() returns the excess.  The sum of all of it plus the root system's
    // count is the matrix's, whatever the elimination order (Sylvester); tests/test_pit_math.py::test_inertia_of_the_kkt_matrix_from_the_sweeps holds both counts to the
    // eigenvalues of the dense matrix, including the cases where a negative pivot in one place is made up for in another.
    // Where the serial sweeps take over (r04, measured on the MI355X with the inertia test in both, scripts/dev/pit_*_sweep.py): the combines' I - W P+ is eliminated without
    // exchanges and loses the last digits the end game needs.  Headline kernel, three seeds x 1024 cold starts: partitioned sweeps down to mu = tol = 1e-8 converge 990 / 980 / 989
    // instances, down to 1e-6 992 / 985 / 998 -- exactly the serial sweeps' counts and iteration numbers, at the same kernel time.  Kernels with clearance rows (active rows put
    // 1e8-sized entries into the position block of the value functions): car-like footprints x 192 instances 172 / 147 / 163 with the partitioned sweeps down to 1e-8,
    // 179 / 163 / 175 = the serial sweeps' from 1e-6 up; config 3 at 4096 instances 4004 (1e-8), 4015 (1e-6), 4016 = serial (1e-4) at 21.5 ms against 23.8 ms serial.
    // So: partitioned while mu > max(tol, 1e-6) (Problem::pit_mu_min), max(tol, 1e-4) in the kernels with clearance rows (pit_floor()).
    static constexpr bool kPartitionedSweeps = true;
    __device__ __forceinline__ T pit_floor() const { return OBST ? t_max(P.pit_mu_min, T(1e-4)) : P.pit_mu_min; }
    __device__ __forceinline__ bool pit_enabled() const { return kPartitionedSweeps && EXT < 2 && P.pit != 0 && L.n >= 40 && 3 * L.NS >= 100 && L.NTR * L.NS >= 100 && 5 * L.NS >= 192; }
    // lane index that the optimiser must treat as unknown HERE: keeps the per-lane address arithmetic of a phase inside the phase (hoisted out of the
    // interior-point loop as loop invariants it would occupy registers for the whole solve)
    __device__ __forceinline__ int local_lane() const { int l = lane; asm volatile("" : "+v"(l)); return l; }
    // excess of negative eigenvalues of a combine's pivot block: n-(W) + n-(P+ - W^-1) - 5, by Jacobi's signature rule (negative pivots of the elimination without exchanges).
    // W (the element's, 5 x 5, in the hand-off tile) is swept in place -- the symmetric sweep operator leaves -W^-1 and shows the same pivots as the elimination --, then
    // G = P+ + (-W^-1) is eliminated.  Wave-uniform arithmetic on the upper triangles, values read with uniform LDS addresses / v_readlane: a few hundred instructions per
    // combine against the ~4 k a partitioned factorisation saves.  ok = false when a pivot vanishes (the caller then repeats the factorisation with the serial sweep).
    template <bool LP_A>
    __device__ __forceinline__ int pit_block_inertia(const int TB, const T (&Vp)[6], bool& ok) const {
        const int TW = TB + 96;
        T w[15], g[15];                               // upper triangles (mpc_core.hpp::pit_block_inertia_tri has the arithmetic, compiled for the host by the tests)
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = a; b < 5; ++b) {
                const int u = a * (9 - a) / 2 + b;
                w[u] = T(sm[TW + 16 * a + 9 + b]);                                        // W[a][b] lives in lane 9 + b as wn[a] (the asymmetry of the accumulated tile is rounding)
                g[u] = rd_lane(Vp[a], LP_A ? b : (b < 2 ? 6 + b : 10 + b));
            }
        return pit_block_inertia_tri(w, g, ok);
    }
    // one combine step.  LP_A: the value function's P+ columns sit in lane set A (then the result's sit in B), else the other way round.
    // In: Vp / Wp / omp = value function at the segment's end, the element's tile in LDS at TB.  Out: the same registers = value function at the
    // segment's start; the eliminated tile and the old value function go to `save` for the forward pass; wpiv = min |pivot| so far.
    template <bool LP_A>
    __device__ __forceinline__ void combine(const int TB, const int save, const CombLane& cl, const int lm, T (&Vp)[6], T (&Wp)[3], T& omp, T& wpiv, int& inert, bool& iok) const {
        inert += pit_block_inertia<LP_A>(TB, Vp, iok);        // first: nothing of the combine is live yet
        const int TW = TB + 96;
        T Wt[5], WV[5], A[6], U[6], St[6], Ra[6];
#pragma unroll
        for (int i = 0; i < 5; ++i) Wt[i] = sm[TW + 16 * i + 9 + lm];                      // left factor W: column m in lane m
#pragma unroll
        for (int i = 0; i < 6; ++i) A[i] = sm[cl.bb + i];
#pragma unroll
        for (int i = 0; i < 6; ++i) { St[i] = sm[TB + 16 * i + 9 + lm]; Ra[i] = sm[cl.pb + 16 * i]; }
        MPC_DPP_BLOCK_CWV
#pragma unroll
        for (int i = 0; i < 5; ++i) A[i] += cl.sig * WV[i];
        T gj_pv, gj_r, gj_e, gj_na;
        if constexpr (LP_A) { MPC_DPP_BLOCK_CGJ_A } else { MPC_DPP_BLOCK_CGJ_B }
#pragma unroll
        for (int i = 0; i < 6; ++i) { U[i] = Vp[i]; Ra[i] *= cl.keep_p; }
        if constexpr (LP_A) { MPC_DPP_BLOCK_CU_A } else { MPC_DPP_BLOCK_CU_B }
        MPC_DPP_BLOCK_CRA
        MPC_DPP_BLOCK_CWN
        // for the forward pass: the value function at the segment's END (still in Vp) and the eliminated tile; every read of the hand-off tile is done
        pit_save(save + 50, cl.sp, Vp);
        pit_save(save, cl.sx, A);
#pragma unroll
        for (int i = 0; i < 6; ++i) Vp[i] = Ra[i] * cl.keep;
    }

#ifdef MPC_PIT_VERBOSE
#define PIT_DBG_W(tag) if (blockIdx.x == MPC_PIT_CHECK) { const double q00 = rd_lane(Wp[0], 9), q01 = rd_lane(Wp[0], 10), q02 = rd_lane(Wp[0], 11), q10 = rd_lane(Wp[1], 9), q11 = rd_lane(Wp[1], 10), q12 = rd_lane(Wp[1], 11), q20 = rd_lane(Wp[2], 9), q22 = rd_lane(Wp[2], 11), o0 = rd_lane(omp, 9); \
            if (lane == 0) printf("   [%s] Wp0 %.9e %.9e %.9e Wp1 %.9e %.9e %.9e Wp2 %.9e .. %.9e om0 %.9e\n", tag, q00, q01, q02, q10, q11, q12, q20, q22, o0); }
#else
#define PIT_DBG_W(tag)
#endif
    __device__ __forceinline__ int backward_pit(T delta, T dc, T& dd_out, T nu_out[3]) const {
        const int n = L.n, N = n - 1, Lm = N >> 2, rem = N - 4 * Lm;
        const T d = SCL(SC_D);
        const int ZC = W_ZC();
        const int ll = local_lane();
        const int c = ll & 15, row = ll >> 4;
        // columns: 0..5 P, 6 7 the u columns of Hhat, 8 p, 9..13 border (row 3: 9..11 = the fixed goal components), 14 the dt border column, 15 idle
        constexpr unsigned long long KIND = 1ull | (2ull << 3) | (3ull << 6) | (4ull << 15) | (5ull << 18) | (6ull << 21) | (7ull << 24);
        const int kind = c < 12 ? (int)((KIND >> (3 * c)) & 7) : 0;
        const int gb = kind < 3 ? ZC + (kind == 1 ? 4 : (kind == 2 ? 3 : 0)) : (kind < 7 ? W_STG() + 3 * (kind - 3) : W_CC());
        const int gsw = kind < 3 ? 0 : (kind < 7 ? NSTG : 3);
        const int gs = sw_step(gsw);
        constexpr unsigned long long rows[8] = {stage_add_row(0, EXT), stage_add_row(1, EXT), stage_add_row(2, EXT), stage_add_row(3, EXT),
                                                stage_add_row(4, EXT), stage_add_row(5, EXT), stage_add_row(6, EXT), stage_add_row(7, EXT)};
        const int sh = c < 12 ? 5 * c : 60;
        int slot1[8], as_[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { slot1[r] = (int)((rows[r] >> sh) & 31); as_[r] = sw_step(slot1[r] ? NSTG : 0); }
        const T ec = (c == 5 || (c >= 8 && c < 15)) ? T(1) : T(0);
        const T E3 = c == 6 ? T(1) : T(0), E4 = c == 7 ? T(1) : T(0);
        const T dA0 = c == 0 ? delta : T(0), dA1 = c == 1 ? delta : T(0), dA2 = c == 2 ? delta : T(0);
        const T dA6 = c == 6 ? delta : T(0), dA7 = c == 7 ? delta : T(0);
        const bool wrG = c < 14 && c != 6 && c != 7;                       // every row stores the gains of its own stages
        const int g0 = c < 6 ? c : (c == 8 ? 6 : 7 + (c - 9));
        SwCRef gp;
        SwCRef ap[8];
        SwRef kp;
        const int ks = sw_step(wrG ? NGAIN : 0);
        auto point_at = [&](int k) {                                      // running pointers at stage k (per lane)
            gp = sw(gb + k * gsw);
#pragma unroll
            for (int r = 0; r < 8; ++r) ap[r] = sw(slot1[r] ? W_STG() + RA - 1 + slot1[r] + k * NSTG : ZC);
            kp = sw(wrG ? W_GAIN() + g0 + k * NGAIN : W_VP());
        };
        T V[6], wn[5] = {T(0), T(0), T(0), T(0), T(0)}, om = T(0);
        {
            const int kf = n - 1;                                          // record of the final rate rows
            const T a3 = slot1[3] ? S_(RA - 1 + slot1[3], kf) : T(0), a4 = slot1[4] ? S_(RA - 1 + slot1[4], kf) : T(0), a5 = slot1[5] ? S_(RA - 1 + slot1[5], kf) : T(0);
            terminal_value(V, c, delta, d, a3, a4, a5);
        }
        T add_dd0 = T(0), add_qd0 = T(0);
        if (mintime()) add_qd0 += T(n - 1);
        if (dtf()) {
            const T dl = d - P.dt_lb, du = P.dt_ub - d;
            const T idl = fast_rcp(dl), idu = fast_rcp(du);
            add_dd0 = SCL(SC_PDL) * idl + SCL(SC_PDU) * idu + delta;
            add_qd0 += mu * idu - mu * idl;
        }
        // stage 0 belongs to row 0: no regularisation on x_0 (it is fixed), the dt-box / objective terms on row 5
        const T s05 = row == 0 ? (c == 5 ? add_dd0 : (c == 8 ? add_qd0 : T(0))) : T(0);
        const T dL0 = row == 0 ? T(0) : dA0, dL1 = row == 0 ? T(0) : dA1, dL2 = row == 0 ? T(0) : dA2;
        T worst = T(1);
        int negc = 0, negc_rem = 0;                                       // negative eigenvalues of the control pivots: this row's segment / the leftover stages all rows sweep together
        auto load_stage = [&](T (&g)[3], T (&a)[8]) {
            g[0] = sw_ld(gp, 0); g[1] = sw_ld(gp, 1); g[2] = sw_ld(gp, 2);
            gp -= gs;
#pragma unroll
            for (int r = 0; r < 8; ++r) { a[r] = sw_ld(ap[r]); ap[r] -= as_[r]; }
        };
        auto stage = [&](T dk0, T dk1, T dk2, T s5, T (&G)[3], T (&A)[8], T (&Gn)[3], T (&An)[8]) {
            load_stage(Gn, An);
            T t[6];
            MPC_DPP_BLOCK_T1
            T h[8];
            h[0] = (A[0] + dk0) + t[0]; h[1] = (A[1] + dk1) + t[1]; h[2] = (A[2] + dk2) + t[2];
            h[3] = A[3]; h[4] = A[4];
            h[5] = (A[5] + s5) + t[5];
            h[6] = (A[6] + dA6) + t[3]; h[7] = (A[7] + dA7) + t[4];
            MPC_DPP_BLOCK_H
            T R00, R01, R11;
            MPC_DPP_BLOCK_R
            const T r2 = R01 * R01;
            const T det = R00 * R11 - r2;
            worst = t_fmin(worst, t_abs(det) - T(1e-14) * (t_abs(R00 * R11) + r2));
            { const int s0 = sign_word(R00), s1 = sign_word(det); negc += (int)((unsigned)s0 >> 31) + (int)((unsigned)(s0 ^ s1) >> 31); }
            const T nid = -fast_rcp(det);
            const T nRi00 = R11 * nid, Ri01 = -(R01 * nid), nRi11 = R00 * nid;
            const T nK0 = nRi00 * h[6] + Ri01 * h[7], nK1 = Ri01 * h[6] + nRi11 * h[7];
            sw_st(kp, 0, nK0); sw_st(kp, NGH, nK1);
            kp -= ks;
            V[0] = h[0]; V[1] = h[1]; V[2] = h[2]; V[3] = h[3]; V[4] = h[4]; V[5] = h[5];
            MPC_DPP_BLOCK_V5
        };
        T Ga[3], Aa[8], Gb[3], Ab[8];
#ifdef MPC_PROFILE
        const long long tp0 = __builtin_readcyclecounter();
#endif
        // ---- the N mod 4 leftover stages at the end of the horizon: all rows together (k = N-1 .. 4 Lm)
        if (rem > 0) {
            point_at(N - 1);
            load_stage(Ga, Aa);
            stage(dA0, dA1, dA2, T(0), Ga, Aa, Gb, Ab);
            if (rem > 1) stage(dA0, dA1, dA2, T(0), Gb, Ab, Ga, Aa);
            if (rem > 2) stage(dA0, dA1, dA2, T(0), Ga, Aa, Gb, Ab);
        }
        negc_rem = negc; negc = 0;
        // ---- rows 0..2 start their segments from the identity border
        if (row < 3) {
#pragma unroll
            for (int i = 0; i < 6; ++i) V[i] = c == 9 + i ? T(1) : T(0);
#pragma unroll
            for (int a = 0; a < 5; ++a) wn[a] = T(0);
            om = T(0);
        }
        // ---- the four segments: row s sweeps k = s Lm + Lm - 1 .. s Lm
        point_at(row * Lm + Lm - 1);
        load_stage(Ga, Aa);
        int j = Lm;
        for (; j >= 3; j -= 2) {
#ifdef MPC_ASM_MARK
            asm volatile("; PIT_LOOP_BEGIN");
#endif
            stage(dA0, dA1, dA2, T(0), Ga, Aa, Gb, Ab);
            stage(dA0, dA1, dA2, T(0), Gb, Ab, Ga, Aa);
#ifdef MPC_ASM_MARK
            asm volatile("; PIT_LOOP_END");
#endif
        }
        if (j == 2) { stage(dA0, dA1, dA2, T(0), Ga, Aa, Gb, Ab); stage(dL0, dL1, dL2, s05, Gb, Ab, Ga, Aa); }
        else stage(dL0, dL1, dL2, s05, Ga, Aa, Gb, Ab);
        {
            const T w4 = t_fmin(t_fmin(rd_lane(worst, 0), rd_lane(worst, 16)), t_fmin(rd_lane(worst, 32), rd_lane(worst, 48)));
            if (!(w4 > T(0))) return 0;
        }
#ifdef MPC_PROFILE
        const long long tp1 = __builtin_readcyclecounter();
        prof_loop += tp1 - tp0;
#endif
        // ---- hand-off + combine.  Tile (words from TB): [i][lane] V rows 0..5 | [a][lane] wn rows 0..4 | [lane] om
        const int TB = L.DX;
        // (the sync() on either side is what makes the hand-off visible ACROSS lanes: without it the compiler treats the tile as per-thread memory and
        // lets the other rows' loads overtake this row's stores)
        auto put_tile = [&](int r) {
            sync();
            if (row == r) {
#pragma unroll
                for (int i = 0; i < 6; ++i) sm[TB + 16 * i + c] = V[i];
#pragma unroll
                for (int a = 0; a < 5; ++a) sm[TB + 96 + 16 * a + c] = wn[a];
                sm[TB + 176 + c] = om;
            }
            sync();
        };
        T Vp[6], Wp[3], omp, wpiv = T(1);
        int inert = __builtin_amdgcn_readlane(negc_rem, 0) + __builtin_amdgcn_readlane(negc, 0) + __builtin_amdgcn_readlane(negc, 16) + __builtin_amdgcn_readlane(negc, 32) + __builtin_amdgcn_readlane(negc, 48);
        bool iok = true;
#ifdef MPC_ASM_MARK
        asm volatile("; PIT_COMBINE_BEGIN");
#endif
        put_tile(3);
        {
            const int cz = (c < 6 || (c >= 8 && c < 12)) ? c : 15;         // everything else reads the idle lane's zeros
#pragma unroll
            for (int i = 0; i < 6; ++i) Vp[i] = sm[TB + 16 * i + cz];
#pragma unroll
            for (int a = 0; a < 3; ++a) Wp[a] = sm[TB + 96 + 16 * a + c];
            omp = sm[TB + 176 + c];
        }
        PIT_DBG_W("V3")
        const int lm = c < 6 ? c : 0;
        put_tile(2);
        combine<true>(TB, pit_tile(2), comb_lane(c, true, TB), lm, Vp, Wp, omp, wpiv, inert, iok);      // (the per-lane constants are recomputed per step: cheaper than keeping them)
        PIT_DBG_W("V2")
        put_tile(1);
        combine<false>(TB, pit_tile(1), comb_lane(c, false, TB), lm, Vp, Wp, omp, wpiv, inert, iok);
        PIT_DBG_W("V1")
        put_tile(0);
        combine<true>(TB, pit_tile(0), comb_lane(c, true, TB), lm, Vp, Wp, omp, wpiv, inert, iok);
        PIT_DBG_W("V0")
#ifdef MPC_ASM_MARK
        asm volatile("; PIT_COMBINE_END");
#endif
#ifdef MPC_PROFILE
        prof_setup += __builtin_readcyclecounter() - tp1;
#endif
        if (!(rd_lane(wpiv, 0) > T(1e-9)) || !iok) return 0;           // a pivot of I - W P+ (or of the inertia count) broke down: the caller repeats this factorisation with the serial sweep
        // ---- root: the value function at stage 0 has its P columns in lane set B (lane 15 = column 5)
        RicState<T> Vr;
        Vr.neg = inert;                                                    // segments' control pivots + the excess of the three combine blocks (pit_block_inertia)
        Vr.P[5][5] = rd_lane(Vp[5], 15);
        Vr.p[5] = rd_lane(Vp[5], 8);
        Vr.S[5][0] = rd_lane(Vp[5], 9); Vr.S[5][1] = rd_lane(Vp[5], 10); Vr.S[5][2] = rd_lane(Vp[5], 11);
        Vr.om[0] = rd_lane(omp, 9); Vr.om[1] = rd_lane(omp, 10); Vr.om[2] = rd_lane(omp, 11);
        {
            const T w00 = rd_lane(Wp[0], 9), w01 = rd_lane(Wp[0], 10), w02 = rd_lane(Wp[0], 11);
            const T w10 = rd_lane(Wp[1], 9), w11 = rd_lane(Wp[1], 10), w12 = rd_lane(Wp[1], 11);
            const T w20 = rd_lane(Wp[2], 9), w21 = rd_lane(Wp[2], 10), w22 = rd_lane(Wp[2], 11);
            Vr.W[0][0] = w00 - (fx(0) ? dc : T(0)); Vr.W[1][1] = w11 - (fx(1) ? dc : T(0)); Vr.W[2][2] = w22 - (fx(2) ? dc : T(0));
            Vr.W[0][1] = Vr.W[1][0] = T(0.5) * (w01 + w10);
            Vr.W[0][2] = Vr.W[2][0] = T(0.5) * (w02 + w20);
            Vr.W[1][2] = Vr.W[2][1] = T(0.5) * (w12 + w21);
        }
#ifdef MPC_PIT_VERBOSE
        if (blockIdx.x == MPC_PIT_CHECK && lane == 0)
            printf("  %s root: P55 %.10e p5 %.10e S5 %.10e %.10e %.10e om %.10e %.10e %.10e W %.10e %.10e %.10e %.10e %.10e %.10e\n", "pit   ", (double)Vr.P[5][5], (double)Vr.p[5],
                   (double)Vr.S[5][0], (double)Vr.S[5][1], (double)Vr.S[5][2], (double)Vr.om[0], (double)Vr.om[1], (double)Vr.om[2],
                   (double)Vr.W[0][0], (double)Vr.W[0][1], (double)Vr.W[0][2], (double)Vr.W[1][1], (double)Vr.W[1][2], (double)Vr.W[2][2]);
#endif
        return riccati_root(Vr, P, dd_out, nu_out);       // 1 good, -1 wrong inertia (no point in repeating the sweep), 0 breakdown
    }

    __device__ __forceinline__ void forward_pit(T dd, const T nu[3], T delta) const {
        const int n = L.n, N = n - 1, Lm = N >> 2;
        const int ll = local_lane();
        const int c = ll & 15, row = ll >> 4;
        // ---- boundary states (components 0..4; component 5 is dd) and costates, wave-uniform, left to right; they go to a small table behind the s = 0 tiles:
        //      TX + 5 s + i = state at b_{s+1}, TL + 5 s + i = costate at b_{s+1}
        const int TX = L.DX + 100, TL = L.DX + 115;
        {
            const CombLane cla = comb_lane(c, true, L.DX), clb = comb_lane(c, false, L.DX);
            // all six tiles first (one LDS round trip), then the six products back to back
            T XA0[5], VB0[5], XA1[5], VB1[5], XA2[5], VB2[5];
            pit_load(pit_tile(0), cla.sx, XA0); pit_load(pit_tile(0) + 50, cla.sp, VB0);
            pit_load(pit_tile(1), clb.sx, XA1); pit_load(pit_tile(1) + 50, clb.sp, VB1);
            pit_load(pit_tile(2), cla.sx, XA2); pit_load(pit_tile(2) + 50, cla.sp, VB2);
            T acc[5], xi[5];
            { const T (&M)[5] = XA0; MPC_DPP_BLOCK_BX0_B }
#pragma unroll
            for (int i = 0; i < 5; ++i) { xi[i] = acc[i]; sm[TX + i] = acc[i]; }
            { const T (&M)[5] = VB0; MPC_DPP_BLOCK_BX_A }
#pragma unroll
            for (int i = 0; i < 5; ++i) sm[TL + i] = acc[i];
            { const T (&M)[5] = XA1; MPC_DPP_BLOCK_BX_A }
#pragma unroll
            for (int i = 0; i < 5; ++i) { xi[i] = acc[i]; sm[TX + 5 + i] = acc[i]; }
            { const T (&M)[5] = VB1; MPC_DPP_BLOCK_BX_B }
#pragma unroll
            for (int i = 0; i < 5; ++i) sm[TL + 5 + i] = acc[i];
            { const T (&M)[5] = XA2; MPC_DPP_BLOCK_BX_B }
#pragma unroll
            for (int i = 0; i < 5; ++i) { xi[i] = acc[i]; sm[TX + 10 + i] = acc[i]; }
            { const T (&M)[5] = VB2; MPC_DPP_BLOCK_BX_A }
#pragma unroll
            for (int i = 0; i < 5; ++i) sm[TL + 10 + i] = acc[i];
        }
        // component c of the boundary state of this row's segment (row 0 starts at 0): read before the step arrays are written
        const T xi_row = (c < 5 && row > 0) ? sm[TX + 5 * (row - 1) + c] : T(0);
        // ---- lane-parallel: fold the border multiplier of the stage's own segment and dd into the affine terms (as forward_states does with nu)
        for (int k = lane; k < n - 1; k += kWave) {
            const int seg = (k >= Lm ? 1 : 0) + (k >= 2 * Lm ? 1 : 0) + (k >= 3 * Lm ? 1 : 0);
            T m5[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) m5[b] = seg < 3 ? sm[TL + 5 * seg + b] : (b < 3 ? nu[b] : T(0));
            for (int a = 0; a < 2; ++a)
                G_(NGH * a + 6, k) += G_(NGH * a + 7, k) * m5[0] + G_(NGH * a + 8, k) * m5[1] + G_(NGH * a + 9, k) * m5[2] + G_(NGH * a + 10, k) * m5[3] + G_(NGH * a + 11, k) * m5[4] +
                                      G_(NGH * a + 5, k) * dd;
            for (int i = 0; i < 3; ++i) CH_(i, k) = C_(i, k) + S_(3 + i, k) * dd;
        }
        if (lane == 0) { SCL(SC_DD) = dd; F(L.DX, 0, 0) = T(0); F(L.DX, 1, 0) = T(0); F(L.DX, 2, 0) = T(0); }
        sync();
        // ---- the four segments at once: row s carries (dx, du_prev) through its stages, starting from its boundary state
        {
            const int ZC = W_ZC(), k0 = row * Lm;
            int qw[8], qs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { qw[j] = ZC; qs[j] = 0; }
            if (c < 3) {
                qw[0] = W_CH() + c * L.NS; qs[0] = 1;
                qw[1 + c] = ZC + 4;
                if (c < 2) { qw[3] = W_STG() + c; qs[3] = NSTG; }
                qw[6] = W_STG() + 6 + c; qs[6] = NSTG;
                qw[7] = W_STG() + 9 + c; qs[7] = NSTG;
            } else if (c < 5) {
                const int a = c - 3;
                qw[0] = W_GAIN() + NGH * a + 6; qs[0] = NGAIN;
#pragma unroll
                for (int j = 0; j < 5; ++j) { qw[1 + j] = W_GAIN() + NGH * a + j; qs[1 + j] = NGAIN; }
            }
            SwCRef qp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { qp[j] = sw(qw[j] + qs[j] * k0); qs[j] = sw_step(qs[j]); }
            int os = c < 5 ? 1 : 0;
            LdsT* op = lds((c < 3 ? L.DX + c * L.NS + 1 : (c < 5 ? L.DU + (c - 3) * L.NS : L.VP + 12)) + os * k0);
            auto load_q = [&](T (&q)[8]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { q[j] = sw_ld(qp[j]); qp[j] += qs[j]; }
            };
            T xi = xi_row;
            auto stage = [&](T (&q)[8], T (&qn)[8]) {
                load_q(qn);
                T s = q[0], s2 = T(0), xn;
                MPC_DPP_BLOCK_FWD
                *op = xn; op += os;
                xi = xn;
            };
            T qa[8], qb[8];
#ifdef MPC_PROFILE
            const long long tf0 = __builtin_readcyclecounter();
#endif
            load_q(qa);
            int k = 0;
            for (; k + 1 < Lm; k += 2) { stage(qa, qb); stage(qb, qa); }
            if (k < Lm) { stage(qa, qb);
#pragma unroll
                          for (int j = 0; j < 8; ++j) qa[j] = qb[j]; }
            // leftover stages 4 Lm .. N-1: row 3 simply goes on (its pointers are there); the other rows compute along but write to a dummy word
            if (row < 3) { op = lds(L.VP + 12); os = 0; }
            for (k = 4 * Lm; k < N; ++k) { stage(qa, qb);
#pragma unroll
                                           for (int j = 0; j < 8; ++j) qa[j] = qb[j]; }
#ifdef MPC_PROFILE
            prof_fwd_loop += __builtin_readcyclecounter() - tf0;
#endif
        }
        multipliers(dd, nu, delta);
    }

    // ---------------------------------------------------------------- parallel post-processing of the step
    struct Fwd { T hdz, clam, dz2, dphi, a_p, a_d, dzmax, nunu; bool finite; };

    // fraction to the boundary: alpha = min(1, tau / max_i(-dval_i / val_i)).  The pass collects the largest ratio -dval / val (with the reciprocals of the slacks
    // it holds anyway; one reciprocal per multiplier) and divides ONCE after the wave reduction, instead of one IEEE division per row and bound.
    __device__ __forceinline__ void ftb_ratio(T ival, T dval, T& r) const { r = t_max(r, -dval * ival); }
    __device__ __forceinline__ T ftb_alpha(T r, T tau) const { return r > tau ? tau / r : T(1); }

    __device__ __forceinline__ Fwd post_pass(T dd, const T nu[3], T tau) const {
        const int n = L.n;
        const T d = SCL(SC_D);
        T hdz = T(0), clam = T(0), dz2 = T(0), dphi = T(0), r_p = T(0), r_d = T(0), dzmax = T(0);
        if (lane == 0) {
            if (dtf()) {
                T dl = d - P.dt_lb, du = P.dt_ub - d;
                T pl = SCL(SC_PDL), pu = SCL(SC_PDU);
                const T idl = t_rcp(dl), idu = t_rcp(du);
                T gb = -mu * idl + mu * idu;
                hdz += gb * dd; dphi += gb * dd;
                ftb_ratio(idl, dd, r_p); ftb_ratio(idu, -dd, r_p);
                ftb_ratio(t_rcp(pl), mu * idl - pl - (pl * idl) * dd, r_d);
                ftb_ratio(t_rcp(pu), mu * idu - pu + (pu * idu) * dd, r_d);
                dz2 += dd * dd; dzmax = t_max(dzmax, t_abs(dd));
            }
            if (mintime()) { hdz += T(n - 1) * dd; dphi += T(n - 1) * dd; }
            if (ball()) {
                const T jdz = ball_jdz(), s = SCL(SC_TS), y = SCL(SC_TY), res = SCL(SC_TG) + s;
                const T is = t_rcp(s);
                const T sig = y * is, ybar = mu * is + sig * res;
                const T ds = -res - jdz, dy = ybar + sig * jdz - y;
                hdz += ybar * jdz;
                dphi -= (mu * is) * ds;
                ftb_ratio(is, ds, r_p);
                ftb_ratio(t_rcp(y), dy, r_d);
            }
        }
        for (int k = lane; k < n; k += kWave) {
            if (k < n - 1) {
                for (int j = 0; j < 2; ++j) {
                    T u = F(L.U, j, k), du_ = F(L.DU, j, k);
                    T dl = u - P.u_lb[j], du = P.u_ub[j] - u;
                    T pl = F(L.PL, j, k), pu = F(L.PU, j, k);
                    const T idl = t_rcp(dl), idu = t_rcp(du);
                    T gbar = mu * idu - mu * idl + (quad() ? T(2) * P.R[j] * u * (intf() ? d : T(1)) : T(0));   // barrier (+ objective) gradient wrt u
                    if (costx() && quad()) gbar += T(2) * P.Ro * F(L.U, 1 - j, k) * (intf() ? d : T(1));
                    hdz += gbar * du_; dphi += gbar * du_;
                    ftb_ratio(idl, du_, r_p); ftb_ratio(idu, -du_, r_p);
                    ftb_ratio(t_rcp(pl), mu * idl - pl - (pl * idl) * du_, r_d);
                    ftb_ratio(t_rcp(pu), mu * idu - pu + (pu * idu) * du_, r_d);
                    dz2 += du_ * du_; dzmax = t_max(dzmax, t_abs(du_));
                }
                for (int i = 0; i < 3; ++i) {
                    clam += C_(i, k) * F(L.LAMN, i, k);        // (a non-finite multiplier makes this sum non-finite: tested below)
                }
                if (intf()) {       // d/ddt of the integral-form stage cost
                    const T xd0 = F(L.X, 0, k) - xf[0], xd1 = F(L.X, 1, k) - xf[1], xd2 = normalize_theta(F(L.X, 2, k) - xf[2]);
                    const T v = F(L.U, 0, k), w = F(L.U, 1, k);
                    T sc = P.Q[0] * xd0 * xd0 + P.Q[1] * xd1 * xd1 + P.Q[2] * xd2 * xd2 + P.R[0] * v * v + P.R[1] * w * w;
                    if (costx()) {
                        const T xd[3] = {xd0, xd1, xd2};
                        sc += offquad(P.Qo, xd) + T(2) * P.Ro * v * w;
                        if (k == 0 && P.trapz) sc -= T(0.5) * fullquad(P.Q, P.Qo, xd);
                    }
                    hdz += sc * dd; dphi += sc * dd;
                }
            }
            if (k >= 1) {
                T vg[3] = {T(0), T(0), T(0)};
                if (via() && k < n - 1) { T vv; via_terms(k, F(L.X, 0, k), F(L.X, 1, k), F(L.X, 2, k), vv, vg); }
                T gxo[3] = {T(0), T(0), T(0)};         // cost variants: what the off-diagonal weights / the trapezoid term add to the gradient wrt x_k
                if (costx()) {
                    if (k < n - 1) {
                        if (quad()) {
                            const T xd[3] = {F(L.X, 0, k) - xf[0], F(L.X, 1, k) - xf[1], normalize_theta(F(L.X, 2, k) - xf[2])};
                            offmul(P.Qo, xd, gxo);
                            for (int i = 0; i < 3; ++i) gxo[i] *= T(2) * (intf() ? d : T(1));
                        }
                    } else {
                        T xdT[3], y3[3]; xd_final(T(0), xdT);
                        if (hasqf()) { offmul(P.Qfo, xdT, y3); for (int i = 0; i < 3; ++i) gxo[i] += T(2) * y3[i]; }
                        if (P.trapz) {
                            offmul(P.Qo, xdT, y3);
                            for (int i = 0; i < 3; ++i) gxo[i] += d * (P.Q[i] * xdT[i] + y3[i]);
                            const T sc = T(0.5) * fullquad(P.Q, P.Qo, xdT);
                            hdz += sc * dd; dphi += sc * dd;
                        }
                    }
                }
                for (int i = 0; i < 3; ++i) {
                    if (k < n - 1 || !fx(i)) {
                        T dx = F(L.DX, i, k);
                        dz2 += dx * dx; dzmax = t_max(dzmax, t_abs(dx));
                        T g = vg[i];
                        if (quad() && k < n - 1) { T xd = F(L.X, i, k) - xf[i]; if (i == 2) xd = normalize_theta(xd); g = T(2) * P.Q[i] * xd * (intf() ? d : T(1)); }
                        else if (k == n - 1 && hasqf()) { T xd = F(L.X, i, k) - xf[i]; if (i == 2) xd = normalize_theta(xd); g += T(2) * P.Qf[i] * xd; }
                        g += gxo[i];
                        hdz += g * dx; dphi += g * dx;
                    }
                }
            }
            for (int q = 0; q < 4; ++q) {
                if (!row_on(k, q)) continue;
                T jdz = row_jdz(k, q, dd);
                T s = F(L.SR, q, k), y = F(L.YR, q, k);
                T res = row_val(L.U, d, k, q) + s;
                const T is = t_rcp(s);
                T sig = y * is;
                T ybar = mu * is + sig * res;
                T ds = -res - jdz;
                T dy = ybar + sig * jdz - y;
                hdz += ybar * jdz;
                dphi -= (mu * is) * ds;
                ftb_ratio(is, ds, r_p);
                ftb_ratio(t_rcp(y), dy, r_d);
            }
            if (nM() > 0 && k >= 1 && k < n - 1) {
                for (int m = 0; m < nM(); ++m) {
                    if (oi(m, k) < 0) continue;
                    const T jdz = obst_jdz(k, m);
                    const T s = F(L.OS, m, k), y = F(L.OY, m, k);
                    if (erho > T(0)) {      // restoration mode: elastic row; the step of e is kept for the trials and the accept pass
                        const T ee = OE_(0, m, k);
                        T ds, de, dy, ybar;
                        elastic_steps(s, y, OB_(0, m, k), ee, jdz, ds, de, dy, ybar);
                        OE_(1, m, k) = de;
                        hdz += ybar * jdz;
                        dphi += -(mu * t_rcp(s)) * ds - (mu * t_rcp(ee)) * de + erho * de;
                        ftb_ratio(t_rcp(s), ds, r_p); ftb_ratio(t_rcp(ee), de, r_p);
                        ftb_ratio(t_rcp(y), dy, r_d); ftb_ratio(t_rcp(erho - y), -dy, r_d);
                        continue;
                    }
                    const T res = OB_(0, m, k) + s;
                    const T is = t_rcp(s);
                    const T sig = y * is;
                    const T ybar = mu * is + sig * res;
                    const T ds = -res - jdz;
                    const T dy = ybar + sig * jdz - y;
                    hdz += ybar * jdz;
                    dphi -= (mu * is) * ds;
                    ftb_ratio(is, ds, r_p);
                    ftb_ratio(t_rcp(y), dy, r_d);
                }
            }
        }
        Fwd o;
        o.hdz = wave_sum(hdz); o.clam = wave_sum(clam); o.dz2 = wave_sum(dz2); o.dphi = wave_sum(dphi);
        o.a_p = ftb_alpha(wave_max(r_p), tau); o.a_d = ftb_alpha(wave_max(r_d), tau); o.dzmax = wave_max(dzmax);
        o.nunu = T(0);
        for (int i = 0; i < 3; ++i) if (fx(i)) o.nunu += nu[i] * nu[i];
        o.finite = t_finite(o.hdz) && t_finite(o.dz2) && t_finite(o.clam);
        return o;
    }

    // ---------------------------------------------------------------- trial point / acceptance (parallel)
    __device__ __forceinline__ void accept(T alpha, T a_d) const {
        const int n = L.n;
        const T kS = T(1e10);
        const T d_old = SCL(SC_D), dd = SCL(SC_DD), d_new = d_old + (dtf() ? alpha * dd : T(0));
        // phase 1: everything that reads the OLD point
        T sn[4], yn[4];
        // Chunks of 64 grid points are visited from the END of the horizon: a rate row k reads the OLD u_{k-1}, which belongs to the
        // previous chunk when k is a multiple of 64, so that chunk must not have been overwritten yet (n > 64: configs 3 and 5).
        for (int k = ((n - 1) / kWave) * kWave + lane; k >= 0; k -= kWave) {
            const bool act = k < n;
            if (act) {
            for (int q = 0; q < 4; ++q) {
                if (!row_on(k, q)) continue;
                T s = F(L.SR, q, k), y = F(L.YR, q, k);
                T res = row_val(L.U, d_old, k, q) + s;
                T jdz = row_jdz(k, q, dd);
                const T is = t_rcp(s);
                T sig = y * is;
                T ds = -res - jdz;
                T dy = mu * is + sig * res + sig * jdz - y;
                sn[q] = s + alpha * ds;
                T yv = y + a_d * dy;
                const T musn = mu * t_rcp(sn[q]);
                yn[q] = t_min(t_max(yv, musn * (T(1) / kS)), kS * musn);
            }
            if (nM() > 0 && k >= 1 && k < n - 1) {
                for (int m = 0; m < nM(); ++m) {
                    if (oi(m, k) < 0) continue;
                    const T jdz = obst_jdz(k, m);
                    const T s = F(L.OS, m, k), y = F(L.OY, m, k);
                    const T res = OB_(0, m, k) + s;
                    const T is = t_rcp(s);
                    const T sig = y * is;
                    if (erho > T(0)) {      // restoration mode: slack, elastic variable and multiplier of the elastic row; the same safeguards for e and its multiplier rho - y
                        const T ee = OE_(0, m, k);
                        T ds, de, dy, ybar;
                        elastic_steps(s, y, OB_(0, m, k), ee, jdz, ds, de, dy, ybar);
                        const T so = s + alpha * ds, eo = ee + alpha * de;
                        T yo = y + a_d * dy;
                        const T muso = mu * t_rcp(so), mueo = mu * t_rcp(eo);
                        yo = t_min(t_max(yo, muso * (T(1) / kS)), kS * muso);
                        yo = t_min(t_max(yo, erho - kS * mueo), erho - mueo * (T(1) / kS));
                        F(L.OS, m, k) = so; F(L.OY, m, k) = yo; OE_(0, m, k) = eo;
                        continue;
                    }
                    const T so = s + alpha * (-res - jdz);
                    T yo = y + a_d * (mu * is + sig * res + sig * jdz - y);
                    const T muso = mu * t_rcp(so);
                    yo = t_min(t_max(yo, muso * (T(1) / kS)), kS * muso);
                    F(L.OS, m, k) = so; F(L.OY, m, k) = yo;
                }
            }
            }
            sync();      // all lanes of this chunk have read their neighbours' old controls
            if (act) {
            for (int q = 0; q < 4; ++q) if (row_on(k, q)) { F(L.SR, q, k) = sn[q]; F(L.YR, q, k) = yn[q]; }
            if (k < n - 1) {
                for (int j = 0; j < 2; ++j) {
                    T u = F(L.U, j, k), du_ = F(L.DU, j, k);
                    T dl = u - P.u_lb[j], du = P.u_ub[j] - u;
                    T pl = F(L.PL, j, k), pu = F(L.PU, j, k);
                    const T idl = t_rcp(dl), idu = t_rcp(du);
                    T pln = pl + a_d * (mu * idl - pl - (pl * idl) * du_);
                    T pun = pu + a_d * (mu * idu - pu + (pu * idu) * du_);
                    T un = ut(j, k, alpha);
                    const T mdl = mu * t_rcp(un - P.u_lb[j]), mdu = mu * t_rcp(P.u_ub[j] - un);
                    F(L.PL, j, k) = t_min(t_max(pln, mdl * (T(1) / kS)), kS * mdl);
                    F(L.PU, j, k) = t_min(t_max(pun, mdu * (T(1) / kS)), kS * mdu);
                    F(L.U, j, k) = un;
                }
                for (int i = 0; i < 3; ++i) {
                    T lo = F(L.LAM, i, k);
                    F(L.LAM, i, k) = lo + alpha * (F(L.LAMN, i, k) - lo);
                }
            }
            for (int i = 0; i < 3; ++i) F(L.X, i, k) = xt(i, k, alpha);
            }
        }
        if (lane == 0) {
            if (dtf()) {
                T dl = d_old - P.dt_lb, du = P.dt_ub - d_old;
                T pl = SCL(SC_PDL), pu = SCL(SC_PDU);
                const T idl = t_rcp(dl), idu = t_rcp(du);
                T pln = pl + a_d * (mu * idl - pl - (pl * idl) * dd);
                T pun = pu + a_d * (mu * idu - pu + (pu * idu) * dd);
                const T mdl = mu * t_rcp(d_new - P.dt_lb), mdu = mu * t_rcp(P.dt_ub - d_new);
                SCL(SC_PDL) = t_min(t_max(pln, mdl * (T(1) / kS)), kS * mdl);
                SCL(SC_PDU) = t_min(t_max(pun, mdu * (T(1) / kS)), kS * mdu);
            }
            SCL(SC_D) = d_new;
            if (ball()) {        // from the caches of the OLD point (value, gradient) and the step
                const T jdz = ball_jdz(), s = SCL(SC_TS), y = SCL(SC_TY), res = SCL(SC_TG) + s, is = t_rcp(s), sig = y * is;
                const T so = s + alpha * (-res - jdz);
                T yo = y + a_d * (mu * is + sig * res + sig * jdz - y);
                const T muso = mu * t_rcp(so);
                yo = t_min(t_max(yo, muso * (T(1) / kS)), kS * muso);
                SCL(SC_TS) = so; SCL(SC_TY) = yo;
            }
        }
    }

    // ---------------------------------------------------------------- initial point (parallel)
    __device__ __forceinline__ void cold_start() const {
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

    // initial vertex values of candidate kind `kind` (mpc_candidate_kind; MPC_CAND_REFERENCE is cold_start() / the caller's guess):
    // positions on the straight line as in cold_start(); heading = direction of travel, turned by pi when the goal lies behind the start
    // pose (initializeSequences without xinit, full_discretization_grid_base_se2.cpp:136-190), + pi for the *_REVERSE kinds; the BLEND kinds
    // turn from the start heading into that direction over the first m grid points and into the goal heading over the last m.
    // Kinds 5..8 (HERMITE_FF / _RR / _FR / _RF): positions on the cubic Hermite curve between the two poses, end tangents along the headings scaled by
    // tscale * |goal - start| and signed by the driving direction at that end; heading = tangent direction (+ pi where the robot drives backwards:
    // the first half of the horizon takes the start's direction, the second half the goal's).
    __device__ __forceinline__ void seed_start(int kind, T tscale = T(2)) const {
        const int n = L.n;
        const T ddx = xf[0] - x0[0], ddy = xf[1] - x0[1];
        if (kind >= 5) {
            const T sg0 = (kind == 5 || kind == 7) ? T(1) : T(-1), sg1 = (kind == 5 || kind == 8) ? T(1) : T(-1);
            const T dd = sqrt(ddx * ddx + ddy * ddy);
            T s0_, c0_, s1_, c1_;
            t_sincos(x0[2], &s0_, &c0_);
            t_sincos(xf[2], &s1_, &c1_);
            const T m0x = sg0 * tscale * dd * c0_, m0y = sg0 * tscale * dd * s0_, m1x = sg1 * tscale * dd * c1_, m1y = sg1 * tscale * dd * s1_;
            for (int k = lane; k < n; k += kWave) {
                const T t = T(k) / T(n - 1), t2 = t * t, t3 = t2 * t;
                const T h00 = T(2) * t3 - T(3) * t2 + T(1), h10 = t3 - T(2) * t2 + t, h01 = T(-2) * t3 + T(3) * t2, h11 = t3 - t2;
                const T g00 = T(6) * t2 - T(6) * t, g10 = T(3) * t2 - T(4) * t + T(1), g01 = T(-6) * t2 + T(6) * t, g11 = T(3) * t2 - T(2) * t;
                T xk[3];
                if (k == 0) { xk[0] = x0[0]; xk[1] = x0[1]; xk[2] = x0[2]; }
                else if (k == n - 1) { xk[0] = xf[0]; xk[1] = xf[1]; xk[2] = xf[2]; }
                else {
                    xk[0] = h00 * x0[0] + h10 * m0x + h01 * xf[0] + h11 * m1x;
                    xk[1] = h00 * x0[1] + h10 * m0y + h01 * xf[1] + h11 * m1y;
                    const T tx = g00 * x0[0] + g10 * m0x + g01 * xf[0] + g11 * m1x, ty = g00 * x0[1] + g10 * m0y + g01 * xf[1] + g11 * m1y;
                    T th = t_atan2(ty, tx);
                    if ((2 * k < n - 1 ? sg0 : sg1) < T(0)) th = normalize_theta(th + T(3.14159265358979323846));
                    xk[2] = th;
                }
                for (int i = 0; i < 3; ++i) F(L.X, i, k) = xk[i];
                if (k < n - 1) { F(L.U, 0, k) = T(0); F(L.U, 1, k) = T(0); }
            }
            if (lane == 0) SCL(SC_D) = P.dt_ref;
            return;
        }
        T orient = t_atan2(ddy, ddx);
        T s0, c0;
        t_sincos(x0[2], &s0, &c0);
        if (ddx * c0 + ddy * s0 < T(0)) orient = normalize_theta(orient + T(3.14159265358979323846));
        if (kind == 2 || kind == 4) orient = normalize_theta(orient + T(3.14159265358979323846));
        const bool blend = kind >= 3;
        int m = P.cand_blend;
        if (m > (n - 1) / 2) m = (n - 1) / 2;
        const T d0 = normalize_theta(orient - x0[2]), df = normalize_theta(orient - xf[2]);
        for (int k = lane; k < n; k += kWave) {
            const T fr = T(k) / T(n - 1);
            T xk[3];
            if (k == 0) { xk[0] = x0[0]; xk[1] = x0[1]; xk[2] = x0[2]; }
            else if (k == n - 1) { xk[0] = xf[0]; xk[1] = xf[1]; xk[2] = xf[2]; }
            else {
                xk[0] = x0[0] + fr * ddx;
                xk[1] = x0[1] + fr * ddy;
                xk[2] = orient;
                if (blend && k < m) xk[2] = normalize_theta(x0[2] + (T(k) / T(m)) * d0);
                if (blend && n - 1 - k < m) xk[2] = normalize_theta(xf[2] + (T(n - 1 - k) / T(m)) * df);
            }
            for (int i = 0; i < 3; ++i) F(L.X, i, k) = xk[i];
            if (k < n - 1) { F(L.U, 0, k) = T(0); F(L.U, 1, k) = T(0); }
        }
        if (lane == 0) SCL(SC_D) = P.dt_ref;
    }

    // ---- multipliers kept in the handle between control cycles (dual_warm_start).  Block layout (doubles): [0] grid size, [1] pi_dt lower,
    //      [2] pi_dt upper, [3] terminal-ball multiplier, then the LDS word ranges LAM (3 NS), YR (4 NS), PL (2 NS), PU (2 NS) verbatim.
    __host__ __device__ static int dual_words(int ns) { return 4 + 11 * ns; }
    __device__ __forceinline__ void store_duals(double* blk) const {
        const int NS = L.NS;
        if (lane == 0) { blk[0] = double(L.n); blk[1] = double(SCL(SC_PDL)); blk[2] = double(SCL(SC_PDU)); blk[3] = ball() ? double(SCL(SC_TY)) : 0.0; }
        for (int e = lane; e < 3 * NS; e += kWave) blk[4 + e] = double(sm[L.LAM + e]);
        for (int e = lane; e < 4 * NS; e += kWave) blk[4 + 3 * NS + e] = double(sm[L.YR + e]);
        for (int e = lane; e < 2 * NS; e += kWave) { blk[4 + 7 * NS + e] = double(sm[L.PL + e]); blk[4 + 9 * NS + e] = double(sm[L.PU + e]); }
    }
    // every inequality multiplier max(previous, mu0 / slack) (the slacks and mu0 / slack were just set by init_point), lam as it was
    __device__ __forceinline__ void load_duals(const double* blk) const {
        const int n = L.n, NS = L.NS;
        for (int k = lane; k < n; k += kWave) {
            if (k < n - 1) {
                for (int i = 0; i < 3; ++i) F(L.LAM, i, k) = T(blk[4 + i * NS + k]);
                for (int j = 0; j < 2; ++j) {
                    F(L.PL, j, k) = t_max(F(L.PL, j, k), T(blk[4 + 7 * NS + j * NS + k]));
                    F(L.PU, j, k) = t_max(F(L.PU, j, k), T(blk[4 + 9 * NS + j * NS + k]));
                }
            }
            for (int q = 0; q < 4; ++q) if (row_on(k, q)) F(L.YR, q, k) = t_max(F(L.YR, q, k), T(blk[4 + 3 * NS + q * NS + k]));
        }
        if (lane == 0) {
            if (dtf()) { SCL(SC_PDL) = t_max(SCL(SC_PDL), T(blk[1])); SCL(SC_PDU) = t_max(SCL(SC_PDU), T(blk[2])); }
            if (ball()) SCL(SC_TY) = t_max(SCL(SC_TY), T(blk[3]));
        }
    }

    __device__ __forceinline__ void init_point() {
        const int n = L.n;
        if (lane == 0) {
            for (int i = 0; i < 3; ++i) {
                F(L.X, i, 0) = x0[i];
                if (fx(i)) F(L.X, i, n - 1) = xf[i];
            }
            if (!dtf()) SCL(SC_D) = P.dt_ref;
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
            sync();
            // ... and keep the seeded controls inside the control-rate rows, as the reference's u = 0 start is (every row but the first): increments clamped to
            // rate_seed_frac x the rate limits forward from u_prev, then backward from the final row (against u_ref = 0).  A seed that jumps violates the rows it
            // crosses: their slacks start at the 1e-2 floor with a residual and the fraction-to-boundary rule pins the first iterations.  Lane j walks control j.
            if (lane < 2 && ron(lane) && ron(2 + lane)) {
                const int j = lane;
                const T fr = Algo<T>::rate_seed_frac, lo = P.rate_lim[j] * d0 * fr, hi = P.rate_lim[2 + j] * d0 * fr;
                T prev = F(L.U, j, 0);
                if (row0_on) { const T up_ = j ? uprev[1] : uprev[0]; prev = t_min(t_max(prev, up_ + P.rate_lim[j] * dtprev * fr), up_ + P.rate_lim[2 + j] * dtprev * fr); F(L.U, j, 0) = prev; }
                for (int k = 1; k < n - 1; ++k) { prev = t_min(t_max(F(L.U, j, k), prev + lo), prev + hi); F(L.U, j, k) = prev; }
                T nxt = T(0);
                for (int k = n - 2; k >= 0; --k) { nxt = t_min(t_max(F(L.U, j, k), nxt - hi), nxt - lo); F(L.U, j, k) = nxt; }
            }
        }
        sync();
        for (int k = lane; k < n - 1; k += kWave)
            for (int j = 0; j < 2; ++j) F(L.U, j, k) = push_interior(F(L.U, j, k), P.u_lb[j], P.u_ub[j]);
        if (lane == 0 && dtf()) SCL(SC_D) = push_interior(SCL(SC_D), P.dt_lb, P.dt_ub);
        sync();
        if (nM() > 0) { rows_dropped = associate_obstacles(); sync(); }
        if (via()) { associate_via_points(); sync(); }
        const bool dual_ok = dual_in != nullptr && warm_guess && (int)dual_in[0] == n;       // same grid size as the solve that left the multipliers
        mu = dual_ok ? P.mu_init_dual : (warm_guess ? P.mu_init_warm : P.mu_init); rho = T(0); delta_last = T(0); fail0 = false;
        const T d = SCL(SC_D);
        for (int k = lane; k < n; k += kWave) {
            for (int q = 0; q < 4; ++q) {
                T s = T(1), y = T(0);
                if (row_on(k, q)) { s = t_max(-row_val(L.U, d, k, q), Algo<T>::slack_push); y = mu / s; }
                F(L.SR, q, k) = s; F(L.YR, q, k) = y;
            }
            if (nM() > 0) {
                const T px = F(L.X, 0, k), py = F(L.X, 1, k);
                for (int m = 0; m < nM(); ++m) {
                    T s = T(1), y = T(0), g, a3[3], hk, h3[3];
                    if (k >= 1 && k < n - 1) {
                        if (obst_row3(k, m, px, py, F(L.X, 2, k), g, a3, hk, h3, d)) { s = t_max(-g, Algo<T>::clearance_slack_push); y = mu / s; }
                    } else set_oi(m, k, -1);
                    F(L.OS, m, k) = s; F(L.OY, m, k) = y;
                }
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
            SCL(SC_PDL) = dtf() ? mu / (d - P.dt_lb) : T(0);
            SCL(SC_PDU) = dtf() ? mu / (P.dt_ub - d) : T(0);
            if (ball()) {
                T a[3];
                const T s = t_max(-ball_eval(T(0), a), Algo<T>::slack_push);
                SCL(SC_TS) = s; SCL(SC_TY) = mu / s;
            }
        }
        sync();
        if (dual_ok) { load_duals(dual_in); sync(); }
    }

    // ---------------------------------------------------------------- driver (all lanes, uniform control flow)
    // enters the restoration mode at the current point (the caches of kkt_pass hold the rows' values there); returns what the objective gains: rho x sum of e
    __device__ __forceinline__ T enter_restoration() {
        const int n = L.n;
        erho = Algo<T>::elastic_rho;
        T esum = T(0);
        for (int k = lane; k < n - 1; k += kWave) {
            if (k < 1) continue;
            for (int m = 0; m < nM(); ++m) {
                if (oi(m, k) < 0) continue;
                const T g = OB_(0, m, k);
                const T s = t_max(t_max(-g, Algo<T>::clearance_slack_push), F(L.OS, m, k));
                const T ee = t_max(g + s, mu / erho);
                const T y = t_max(t_min(F(L.OY, m, k), T(0.5) * erho), mu / s);
                F(L.OS, m, k) = s; F(L.OY, m, k) = y; OE_(0, m, k) = ee; OE_(1, m, k) = T(0);
                esum += ee;
            }
        }
        esum = wave_sum(esum);
        sync();
        return erho * esum;
    }

    __device__ __forceinline__ SolveStats<T> solve() {
        SolveStats<T> out;
        flags = (P.xf_fixed[0] ? 1 : 0) | (P.xf_fixed[1] ? 2 : 0) | (P.xf_fixed[2] ? 4 : 0) | (P.dt_free ? 8 : 0) | (P.objective == OBJ_QUADRATIC ? 16 : 0) |
                (P.has_Qf ? 32 : 0) | (P.rate_on[0] ? 64 : 0) | (P.rate_on[1] ? 128 : 0) | (P.rate_on[2] ? 256 : 0) | (P.rate_on[3] ? 512 : 0) | (P.ball ? 1024 : 0) | (P.via ? 2048 : 0) | ((P.n_obst > 0 && (P.footprint_kind == 2 || P.footprint_kind == 3 || P.footprint_kind == 4)) ? 4096 : 0) | (P.integral_form ? 8192 : 0) | (P.dyn_obst ? 16384 : 0) | (P.hess_mode ? 32768 : 0) | ((P.objective != OBJ_QUADRATIC || P.hybrid) ? 65536 : 0) | (P.costx ? 131072 : 0);
        flags = __builtin_amdgcn_readfirstlane(flags);
        nfix = (int)fx(0) + (int)fx(1) + (int)fx(2);
        row0_on = dtprev != T(0);
        if (iter_cap <= 0) iter_cap = P.max_iter;
        if (lane < 8) sm[L.ZC + lane] = lane == 4 ? T(1) : T(0);      // constant coefficient triples of the sweeps
        if constexpr (GS) { if (lane < 8) gw((unsigned)(GlobalStage::ZC + lane)) = lane == 4 ? T(1) : T(0); }      // ... and their copy where the sweeps' pointers live
        if (lane < 12) sm[L.ZI + lane] = lane == 6 ? T(1) : T(0);     // unit vectors / zeros of the partitioned sweep's combine step
        const bool pit = pit_enabled();
        init_point();
        T theta_c, fobj;
        eval_point(SCL(SC_D), theta_c, fobj);
        sync();
        int it = 0, status = ST_MAX_ITER, n_acc = 0;
        const long long t_start = P.max_ticks > 0 ? (long long)wall_clock64() : 0ll;      // mpc_config.max_time_us: this solve's own clock
        const T acc_tol = P.acc_tol;
        const int acc_it = P.acc_iter;
        T e0 = T(0), logs_cur = T(0), dc_mu = T(-1), dc_val = T(0);
        T last_alpha = T(0), last_ad = T(0);
        int jam_streak = 0;
        T jam_theta0 = T(0);
        bool endgame = false;
        const T mu_max = Algo<T>::mu_max_fact * mu;
        bool have_logs = false;
#ifdef MPC_PROFILE
        long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int nfac = 0, ntrial = 0;
        const long long t_begin = __builtin_readcyclecounter();
        const long long w_begin = wall_clock64();
#define MPC_TICK(i, stmt) { long long t0_ = __builtin_readcyclecounter(); stmt; tk[i] += __builtin_readcyclecounter() - t0_; }
#else
#define MPC_TICK(i, stmt) { stmt; }
#endif
        while (true) {
            Err er;
#ifdef MPC_ASM_MARK
            asm volatile("; KKT_BEGIN");
#endif
            MPC_TICK(0, er = kkt_pass());
#ifdef MPC_ASM_MARK
            asm volatile("; KKT_END");
#endif
            if constexpr (OBST) {
                // RESTORATION for clearance rows that jam (r05; what Ipopt leaves to its restoration phase -- src/controller.cpp:388-421 hands the NLP to Ipopt; restated in
                // the CPU restatements the tests check against; DESIGN.md section 3.3 has the derivation).  A row that starts violated pulls its slack to the boundary within a
                // few iterations; from then on the fraction-to-boundary rule admits steps of 1e-3 and the infeasibility stays where it is.  Detected as elastic_trigger
                // iterations in a row with a primal step limit below elastic_ap while the infeasibility has not fallen below elastic_prog x its value at the start of the
                // streak.  From there on the clearance rows are ELASTIC: g + s - e = 0, e >= 0, + rho e in the objective (the exact l1 penalty of the row's violation).
                // Every row is satisfied again at once -- e takes up the violation, the slack goes back to its start rule --, and rho pushes e to zero as the trajectory
                // moves out of the band.  The mode stays on until the solve ends; e counts as primal infeasibility (kkt_pass), so a solve can only end with e <= tol.
                if (nM() > 0 && erho == T(0) && jam_streak >= Algo<T>::elastic_trigger && er.theta >= Algo<T>::elastic_prog * jam_theta0) {
                    fobj += enter_restoration();
                    rho = T(0); have_logs = false; jam_streak = 0;
                    er = kkt_pass();
                }
            }
            e0 = err_value(er, T(0));
            // (t_max / t_min drop a NaN operand: the maxima inside e0 cannot carry one; the sums do -- ADVICE r03)
            if (!t_finite(e0) || !t_finite(er.theta) || !t_finite(er.sum_mult) || !t_finite(er.csum)) { status = ST_NUMERICAL; break; }
            if (e0 <= P.tol) { status = ST_CONVERGED; break; }
            // Ipopt's acceptable-level stop, counting half: acc_iter iterations in a row at the level acc_tol (mpc_config.acceptable_tol / _iter)
            n_acc = (acc_it > 0 && e0 <= acc_tol) ? n_acc + 1 : 0;
            if (acc_it > 0 && n_acc >= acc_it) { status = ST_CONVERGED; break; }
            if (it >= iter_cap) { status = ST_MAX_ITER; break; }
            if (P.max_ticks > 0 && (long long)wall_clock64() - t_start > P.max_ticks) { status = ST_TIME_LIMIT; break; }
            if (win_ptr) {      // hedged candidate: one L2 read per iteration (all lanes, same word)
                const int w = __builtin_amdgcn_readfirstlane(__hip_atomic_load(win_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (w < my_cand) { status = ST_SUPERSEDED; break; }
            }
            if (P.mu_strategy == 1 || endgame) {        // monotone Fiacco-McCormick (Ipopt's own default mu_strategy; the end game of the adaptive one)
                for (int guard = 0; guard < 50; ++guard) {
                    T emu = err_value(er, mu);
                    if (emu <= Algo<T>::kappa_eps * mu && mu > P.tol / T(10)) {
                        mu = t_max(P.tol / T(10), t_min(Algo<T>::kappa_mu * mu, t_pow(mu, Algo<T>::theta_mu)));
                        rho = T(0);
                    } else break;
                }
            } else if (it > 0) {
                // adaptive (the default; what corbo's SolverIpopt is believed to set; the first iteration keeps the start value): mu = sigma x the average complementarity, sigma from the step lengths the
                // LAST iteration achieved -- Mehrotra's (mu_aff / mu)^3 read off the step that was actually taken, no second solve --, never below
                // min(mu, mu_err_floor x E_0) (a barrier far below the optimality error is what stalls the non-convex instances), inside [tol / 10, mu_max_fact x mu_0]
                const T avg = er.csum * inv_cnt_bmult;
                const T a_ = T(1) - t_min(last_alpha, last_ad);
                const T sig = t_min(t_max(a_ * a_ * a_, Algo<T>::sigma_min), T(1));
                T mu_new = t_min(t_max(sig * avg, P.tol / T(10)), mu_max);
                mu_new = t_max(mu_new, t_min(mu, Algo<T>::mu_err_floor * e0));
                if (mu_new <= P.tol) { mu_new = P.tol; endgame = true; }      // end game: from mu = tol on the monotone rule takes over (tol -> tol / 10 once the barrier problem is solved to kappa_eps mu): a solve stops at a point of the central path, as with the monotone strategy
                if (mu_new != mu) { mu = mu_new; rho = T(0); }
            }
#ifdef MPC_ASM_MARK
            asm volatile("; BARRIER_BEGIN");
#endif
            MPC_TICK(1, stage_barrier_terms(); sync());
#ifdef MPC_ASM_MARK
            asm volatile("; BARRIER_END");
#endif
            const T tau = t_max(Algo<T>::tau_min, T(1) - mu);
            if (mu != dc_mu) { dc_mu = mu; dc_val = nfix > 0 ? Algo<T>::delta_c * t_pow(mu, Algo<T>::kappa_c) : T(0); }   // pow() only when mu moved
            const T dc = dc_val;
            T delta = (fail0 && delta_last > T(0)) ? t_max(Algo<T>::delta_min, Algo<T>::kappa_minus * delta_last) : T(0);
            const bool started_zero = delta == T(0);
            bool ok = false;
            Fwd fw;
            T dd = T(0), nu[3] = {T(0), T(0), T(0)}, curv = T(0);
            for (int ntry = 0; ntry <= 40; ++ntry) {
                bool good, used_pit = false;
#ifdef MPC_PIT_CHECK     // developer aid: both sweeps on the same factorisation; blocks below MPC_PIT_CHECK report every factorisation on which the two differ
                if (pit && (int)blockIdx.x < MPC_PIT_CHECK) {
                    T dd1 = T(0), nu1[3] = {T(0), T(0), T(0)}, dd2 = T(0), nu2[3] = {T(0), T(0), T(0)};
                    const int g1c = backward_pit(delta, dc, dd1, nu1); sync();
                    const bool g1 = g1c > 0;
                    if (g1) { forward_pit(dd1, nu1, delta); sync(); }
                    T keepx = T(0), keepu = T(0), keepl = T(0);
                    const int kk = lane < L.n - 1 ? lane : 0;
                    keepx = F(L.DX, 2, kk); keepu = F(L.DU, 1, kk); keepl = F(L.LAMN, 2, kk);
                    sync();
                    const bool g2 = backward_dpp(delta, dc, dd2, nu2); sync();
                    if (g2) { forward_states(dd2, nu2, delta); sync(); }
                    const T ex = wave_max(t_abs(keepx - F(L.DX, 2, kk))), eu = wave_max(t_abs(keepu - F(L.DU, 1, kk))), el = wave_max(t_abs(keepl - F(L.LAMN, 2, kk)));
                    const T sx = wave_max(t_abs(F(L.DX, 2, kk))), su = wave_max(t_abs(F(L.DU, 1, kk))), sl = wave_max(t_abs(F(L.LAMN, 2, kk)));
                    if (lane == 0 && (g1 != g2 || ex > T(1e-7) * (sx + T(1e-3)) || eu > T(1e-7) * (su + T(1e-3)) || el > T(1e-7) * (sl + T(1e-3))))
                        printf("blk %d it %d try %d delta %.2e mu %.1e: pit ok %d (code %d) serial ok %d | dd %.9e vs %.9e | max diff dx %.2e (of %.2e) du %.2e (of %.2e) lam %.2e (of %.2e)\n", (int)blockIdx.x, it, ntry, (double)delta, (double)mu,
                               (int)g1, g1c, (int)g2, (double)dd1, (double)dd2, (double)ex, (double)sx, (double)eu, (double)su, (double)el, (double)sl);
                    sync();
                }
#endif
                if (pit && mu > pit_floor()) {        // partitioned sweep; a broken-down combine pivot (or a singular stage pivot) falls back to the serial sweep; below pit_floor() the end game takes the serial sweeps
                    int gp_;
                    MPC_TICK(2, gp_ = backward_pit(delta, dc, dd, nu); sync());
                    good = used_pit = gp_ > 0;
                    if (gp_ == 0) { MPC_TICK(2, good = backward_dpp(delta, dc, dd, nu); sync()); }
                } else { MPC_TICK(2, good = backward_dpp(delta, dc, dd, nu); sync()); }
#ifdef MPC_PROFILE
                ++nfac;
#endif
#ifdef MPC_NANCHECK      // developer aid: non-finite words per field of the LDS record after each sweep (lane 0 prints)
                if (blockIdx.x == MPC_NANCHECK && lane == 0) {
                    const int offs[] = {L.X, L.U, L.LAM, L.LAMN, L.SR, L.YR, L.PL, L.PU, L.DX, L.DU, L.CC, L.TRIG, L.GAIN, L.STG, L.SC, L.VP, L.ZC, L.total};
                    const char* nm[] = {"X", "U", "LAM", "LAMN", "SR", "YR", "PL", "PU", "DX", "DU", "CC", "TRIG", "GAIN", "STG", "SC", "VP", "ZC"};
                    printf("it %d try %d good %d delta %g dd %g nu %g %g %g mu %g |", it, ntry, (int)good, (double)delta, (double)dd, (double)nu[0], (double)nu[1], (double)nu[2], (double)mu);
                    for (int f = 0; f < 17; ++f) {
                        int cnt = 0, first = -1;
                        for (int q = offs[f]; q < offs[f + 1]; ++q) if (!t_finite(sm[q])) { ++cnt; if (first < 0) first = q - offs[f]; }
                        if (cnt) printf(" %s:%d@%d", nm[f], cnt, first);
                    }
                    printf("\n");
                }
#endif
                if (good) {
                    if (used_pit) { MPC_TICK(3, forward_pit(dd, nu, delta); sync()); }
                    else { MPC_TICK(3, forward_states(dd, nu, delta); sync()); }
#ifdef MPC_NANCHECK
                    if (blockIdx.x == MPC_NANCHECK) {
                        for (int k = lane; k < L.n; k += kWave) {
                            bool f = t_finite(F(L.DX, 0, k)) && t_finite(F(L.DX, 1, k)) && t_finite(F(L.DX, 2, k));
                            if (k < L.n - 1) f = f && t_finite(F(L.DU, 0, k)) && t_finite(F(L.DU, 1, k)) && t_finite(F(L.LAMN, 0, k)) && t_finite(F(L.LAMN, 1, k)) && t_finite(F(L.LAMN, 2, k));
                            if (!f) printf("it %d try %d fwd nonfinite at k %d: dx %g %g %g du %g %g lamn %g %g %g\n", it, ntry, k, (double)F(L.DX, 0, k), (double)F(L.DX, 1, k), (double)F(L.DX, 2, k), (double)F(L.DU, 0, k), (double)F(L.DU, 1, k), (double)F(L.LAMN, 0, k), (double)F(L.LAMN, 1, k), (double)F(L.LAMN, 2, k));
                        }
                    }
#endif
#ifdef MPC_ASM_MARK
                    asm volatile("; POST_BEGIN");
#endif
                    MPC_TICK(4, fw = post_pass(dd, nu, tau));
#ifdef MPC_ASM_MARK
                    asm volatile("; POST_END");
#endif
                    good = fw.finite;
                    if (good) {      // the backward sweep has checked the inertia of this factorisation (mpc_core.hpp::riccati_root); the curvature only feeds the penalty update
                        curv = -fw.hdz + fw.clam - dc * fw.nunu;
                        ok = true; break;
                    }
                }
                if (delta == T(0)) delta = (delta_last == T(0)) ? Algo<T>::delta_first : t_max(Algo<T>::delta_min, Algo<T>::kappa_minus * delta_last);
                else delta *= (delta_last == T(0)) ? Algo<T>::kappa_plus_first : Algo<T>::kappa_plus;
                if (delta > Algo<T>::delta_max) break;
            }
            if (!ok) { status = ST_LINSOLVE; break; }
            if constexpr (OBST) {      // the restoration trigger's streak
                if (nM() > 0 && erho == T(0)) {
                    if (fw.a_p < Algo<T>::elastic_ap && er.rp > T(1e-3)) { if (jam_streak == 0) jam_theta0 = er.theta; ++jam_streak; }
                    else jam_streak = 0;
                }
            }
            if (delta > T(0)) delta_last = delta;
            if (started_zero) fail0 = delta > T(0);
            const T theta = er.theta;
            if (theta > T(0)) {
                T sigma = curv > T(0) ? T(1) : T(0);
                T rho_trial = (fw.dphi + T(0.5) * sigma * curv) / ((T(1) - Algo<T>::rho_frac) * theta);
                if (rho < rho_trial) rho = rho_trial + T(1);
            }
            T phi0;
            // sum of the barrier logs at the current point: it is the value of the last accepted trial (same point, mu-independent)
            if (!have_logs) { MPC_TICK(5, logs_cur = barrier_logs(SCL(SC_D), T(0), false, dd)); have_logs = true; }
            phi0 = fobj - mu * logs_cur + rho * theta;
            const T Dm = fw.dphi - rho * theta;
            const T theta_rows = theta - theta_c;
            T alpha = fw.a_p;
            bool accepted = false;
            const bool fast_trials = trial_fast_ok();
            TrialRegs tregs;
            if (fast_trials) { MPC_TICK(5, trial_setup(tregs, dd)); }
            T th_t = T(0), f_t = T(0), lg_t = T(0);
            for (int ls = 0; ls < Algo<T>::max_ls; ++ls) {
                if (ls > 0) alpha *= T(0.5);
                T phit, tht;
                const T d_t = SCL(SC_D) + (dtf() ? alpha * SCL(SC_DD) : T(0));
#ifdef MPC_ASM_MARK
                asm volatile("; TRIAL_BEGIN");
#endif
                if (fast_trials) {
                    MPC_TICK(6, trial_eval(tregs, alpha, d_t, th_t, f_t, lg_t);
                             tht = th_t + (T(1) - alpha) * theta_rows;
                             phit = f_t - mu * lg_t + rho * tht; sync());
                } else {
                    MPC_TICK(6, eval_point(d_t, th_t, f_t, alpha, true);
                             tht = th_t + (T(1) - alpha) * theta_rows;
                             lg_t = barrier_logs(d_t, alpha, true, dd);
                             phit = f_t - mu * lg_t + rho * tht; sync());
                }
#ifdef MPC_ASM_MARK
                asm volatile("; TRIAL_END");
#endif
#ifdef MPC_PROFILE
                ++ntrial;
#endif
                if (t_finite(phit) && phit - phi0 - Algo<T>::ls_eps * t_abs(phi0) <= Algo<T>::eta_armijo * alpha * Dm) { accepted = true; break; }
            }
#ifdef MPC_NANCHECK
            if (blockIdx.x == MPC_NANCHECK && lane == 0 && !accepted)
                printf("   ls FAILED: it %d phi0 %.12e Dm %.6e rho %.6e mu %g theta %.6e theta_c %.6e fobj %.9f logs_cur %.9f | last alpha %g f_t %.9f th_t %.6e lg_t %.9f dzmax %g a_p %g\n",
                       it, (double)phi0, (double)Dm, (double)rho, (double)mu, (double)theta, (double)theta_c, (double)fobj, (double)logs_cur, (double)alpha, (double)f_t, (double)th_t, (double)lg_t, (double)fw.dzmax, (double)fw.a_p);
#endif
            // Ipopt's acceptable-level stop, refused-step half (tested before the line-search failure: Ipopt answers a failed line search at an
            // acceptable point with success): every trial step refused (or only one below 1e-6 of the fraction-to-boundary step accepted) at a point
            // at the acceptable level: the solve ends here, nothing is moved (the next iteration would compute the same step and refuse it again)
            if (acc_tol > T(0) && (!accepted || alpha < T(1e-6) * fw.a_p) && e0 <= acc_tol) { status = ST_CONVERGED; break; }
            if (!accepted && alpha * fw.dzmax < T(1e-14)) { status = ST_LINESEARCH; break; }
#ifdef MPC_NANCHECK
            if (blockIdx.x == MPC_NANCHECK && lane == 0)
                printf("   ls: it %d f %.6f -> %.6f theta_c %.3e alpha %.4f a_p %.4f a_d %.4f rho %.3e Dm %.4e hdz %.6e clam %.6e dz2 %.6e dphi %.6e\n", it, (double)fobj, (double)f_t, (double)th_t, (double)alpha, (double)fw.a_p, (double)fw.a_d, (double)rho, (double)Dm, (double)fw.hdz, (double)fw.clam, (double)fw.dz2, (double)fw.dphi);
#endif
#ifdef MPC_ASM_MARK
            asm volatile("; ACCEPT_BEGIN");
#endif
            MPC_TICK(7, accept(alpha, fw.a_d); sync());
#ifdef MPC_ASM_MARK
            asm volatile("; ACCEPT_END");
#endif
            theta_c = th_t; fobj = f_t; logs_cur = lg_t;
            last_alpha = alpha; last_ad = fw.a_d;
            ++it;
        }
#ifdef MPC_PROFILE
        if (lane == 0 && blockIdx.x < 4096) {
            long long* o = g_mpc_prof[blockIdx.x];
            o[0] = __builtin_readcyclecounter() - t_begin; o[1] = wall_clock64() - w_begin; o[2] = it; o[3] = nfac; o[4] = ntrial;
            for (int i = 0; i < 8; ++i) o[5 + i] = tk[i];
            o[13] = prof_loop; o[14] = prof_setup; o[15] = prof_fwd_loop;
        }
#endif
        out.status = status; out.iters = it; out.kkt_error = e0; out.objective = fobj;
        return out;
    }
};

}  // namespace mpc

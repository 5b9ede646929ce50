// mpc_wave_layout.hpp -- LDS / global-memory layout of one planner instance's working set and the wavefront reductions of the wave kernel (device + host).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "mpc_core.hpp"

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
// the little else their running pointers touch (the constant triples, the residuals c_k, the folded residuals c^_k) -- sits in ONE block of global memory per
// workgroup.  The LDS record shrinks from 97 to 34 words per grid point (n = 120 in fp64: 95 KB -> 33 KB, four workgroups per CU instead of one); the block is
// written and re-read by the same CU within one interior-point iteration (L2 / Infinity-Cache resident).
// Layout inside the block.  The vector-memory path of a CU charges an instruction per DISTINCT cache line it touches, and four resident waves share it
// (scripts/ubench/vmem_lines.hip).  A lane-parallel pass has lane = stage: stage-major records put every lane of a load or store into a line of its own (64 per
// instruction); a sweep has lane = entry: component-major rows put every lane into a row of its own (12 per instruction, 48 in the partitioned sweeps).  The stage
// records therefore live in TILES of four stages, interleaved word by word: entry e of stage k is word 4 e + (k mod 4) of tile k / 4.  A pass touches 16 lines per
// instruction (four lanes share 32 bytes), a sweep three or four, reused for four stages.  A stage's tile slot carries, behind the record, the copy of c_k
// and the constants 0 0 0 1 0 0 (the sweeps' constant coefficient triples are entries of the slot like everything else they read: every running pointer of a lane
// moves through the tiles the same way).  One guard tile in front of stage 0 and the tiles behind stage n - 1 take the prefetches that run past either end.
// The gains stay STAGE-major, 24 adjacent words per stage: the sweeps WRITE them (as rows a stage stored into 20 lines).  Word offsets:
struct GlobalStage {
    static constexpr int ZC = 0;          // 8 words: constants 0 0 0 0 1 0 0 0 (single words read with stride 0 by the forward sweeps)
    static constexpr int GAIN = 32;       // NGAIN (ns + 1) words, stage-major; the record behind the last stage is the idle lanes' store target
    static constexpr int kGuard = 4;      // stages in front of stage 0 (one tile)
    // entries of a stage's slot: [0, nstg) the record | c_k (3) | 0 0 0 1 0 0 | padding to a multiple of four  (c^_k, the forward sweeps' constant term, is in LDS in both forms)
    __host__ __device__ static constexpr int nt(int nstg) { return ((nstg + 9 + 3) / 4) * 4; }
    __host__ __device__ static constexpr int tiles(int ns) { return (ns + kGuard + 4 + 3) / 4; }                      // stages -4 .. ns + 3
    __host__ __device__ static constexpr int TILE(int ns) { return GAIN + ((NGAIN * (ns + 1) + 15) / 16) * 16; }
    __host__ __device__ static constexpr int tile_k(int nstg, int k) { return ((k + kGuard) >> 2) * (4 * nt(nstg)) + ((k + kGuard) & 3); }      // + 4 e: entry e of stage k, relative to TILE
    __host__ __device__ static constexpr int OBC(int ns, int nstg) { return TILE(ns) + tiles(ns) * 4 * nt(nstg); }   // 4 M NS words [OG | OAX | OAY | OHK][m][k]: the clearance rows' cached value,
                                                                                                     // gradient and curvature (touched by the lane-parallel passes only)
    __host__ __device__ static constexpr int OEL(int ns, int nstg, int M) { return OBC(ns, nstg) + 4 * M * ns; }   // 2 M NS words [OE | ODE][m][k]: the elastic variables of the clearance rows and
                                                                                                     // their steps (restoration mode, IpmWave::solve)
    __host__ __device__ static constexpr int words(int ns, int nstg, int M) { return ((OEL(ns, nstg, M) + 2 * M * ns + 15) / 16) * 16; }     // (128-byte multiple in fp64)
    static constexpr int kPrefetchPad = 256;      // words behind the LAST block of the allocation (slack for reads past a block's last row; values never used)
    // a layout that keeps its factorisation data in LDS still has a block when it has clearance rows: the elastic arrays alone (touched by the lane-parallel passes only, and
    // only in the restoration mode: not worth 2 M words of LDS per grid point)
    static constexpr int OEL_ONLY = 16;
    __host__ __device__ static constexpr int words_elastic_only(int ns, int M) { return ((OEL_ONLY + 2 * M * ns + 15) / 16) * 16; }
};

// (compile-time checks of the block's layout: the tiles start on a cache line and hold a whole number of lines, the regions follow each other without overlap, the
//  spare gain record, the guard tile and the tiles behind the last stage exist, the obstacle arrays fit)
constexpr bool global_stage_ok(int ns, int nstg, int M) {
    using G = GlobalStage;
    return G::nt(nstg) % 4 == 0 && G::nt(nstg) >= nstg + 9 && G::TILE(ns) % 16 == 0 && G::TILE(ns) >= G::GAIN + NGAIN * (ns + 1) && (4 * G::nt(nstg)) % 16 == 0 &&
           G::tile_k(nstg, -G::kGuard) == 0 && G::tile_k(nstg, ns + 3) + 4 * (G::nt(nstg) - 1) < G::tiles(ns) * 4 * G::nt(nstg) &&
           G::OBC(ns, nstg) == G::TILE(ns) + G::tiles(ns) * 4 * G::nt(nstg) && G::OEL(ns, nstg, M) == G::OBC(ns, nstg) + 4 * M * ns &&
           G::words(ns, nstg, M) >= G::OEL(ns, nstg, M) + 2 * M * ns && G::words(ns, nstg, M) % 16 == 0;
}
static_assert(global_stage_ok(3, NSTG_BASE, 0) && global_stage_ok(15, NSTG_BASE, 4) && global_stage_ok(16, NSTG_BASE, 4) && global_stage_ok(17, NSTG_BASE, 0) && global_stage_ok(50, NSTG_BASE, 0) &&
              global_stage_ok(80, NSTG_BASE, 4) && global_stage_ok(120, NSTG_BASE, 0) && global_stage_ok(127, NSTG_EXT, 8) && global_stage_ok(128, NSTG_EXT, 8) && global_stage_ok(590, NSTG_BASE, 0),
              "GlobalStage: a region overlaps its neighbour or a tile does not start on a cache line");

struct WaveLayout {
    int n, NS;
    int NTR;                                      // trig-cache words per stage (3, or 4 for the bicycle / front-wheel car; +2 for Crank-Nicolson)
    int X, U, LAM, LAMN, SR, YR, PL, PU, DX, DU, CC, TRIG, GAIN, STG, SC, VP, ZC, ZI, total;
    int M, O, V;                                  // clearance rows per grid point, obstacles, vertices per obstacle
    int OS, OY, OI, OG, OAX, OAY, OHK;            // per-row slack, multiplier, obstacle index, cached g, gradient, curvature
    int GV, GNV, GR, GC;                          // obstacle geometry: vertices, vertex counts, radii, centroids
    int GE;                                       // edge table (V >= 2): 3 words per edge (b - a, 1 / |b - a|^2), computed once per solve by load_obstacles
    int OAT, OHXT, OHYT, OHTT;                    // third-variable parts of the clearance rows: heading (footprints that turn with the pose) or
                                                  // dt (dynamic obstacles); MT = M when either is configured, else 0 words
    int OAD, OHXD, OHYD, OHDD, OHTD;              // dt parts when BOTH apply (dynamic obstacles + a turning footprint): gradient, hess [x dt, y dt, dt dt, theta dt]
    int GVEL;                                     // obstacle velocities (dynamic obstacles; 2 * OD words)
    int NV, VIA, VIDX;                            // via-points: capacity, poses (x, y, theta), attached grid point (-1 = skipped)
    int GSW;                                      // > 0: the workgroup has a block of GSW words of GLOBAL memory (GlobalStage): the elastic arrays of the clearance rows, and with GSF the factorisation data
    int GSF;                                      // 1: the factorisation data (GAIN, STG) and the clearance rows' caches live in that block instead of LDS (IpmWave<..., GS = true>)
    int OEB;                                      // word offset of the elastic arrays [OE | ODE] inside the block
    int MT, MD;                                   // rows per grid point that carry a third-variable part (heading or dt) / a second one (dt next to the heading)
    int OXB;                                      // GSF: word offset of the rows' third-variable caches [OAT | OHXT | OHYT | OHTT][MT][NS] + [OAD | OHXD | OHYD | OHDD | OHTD][MD][NS] inside the block
                                                  // (r06: written by kkt_pass, read by lane-parallel passes only -- like OG / OAX / OAY / OHK, they leave LDS in the global form)
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
        L.GSW = gs ? GlobalStage::words(n, nstg, M) + (((4 * MT + 5 * MD) * n + 15) / 16) * 16 : (M > 0 ? GlobalStage::words_elastic_only(n, M) : 0);
        L.GSF = gs ? 1 : 0;
        L.OEB = gs ? GlobalStage::OEL(n, nstg, M) : GlobalStage::OEL_ONLY;
        L.MT = MT; L.MD = MD;
        L.OXB = gs ? GlobalStage::words(n, nstg, M) : 0;
        L.SC = o; o += 16;    // scalars: D, DT, DD, PDL, PDU | terminal-ball row: slack, multiplier, cached value and gradient
        L.VP = o; o += 16;    // dummy store targets of the idle lanes in the sweeps
        L.ZC = o; o += 8;     // constants 0 0 0 0 1 0 0 0 (coefficient triples of the constant columns)
        L.ZI = o; o += 12;    // constants 0 0 0 0 0 0 1 0 0 0 0 0: the unit vector e_c (6 words) starts at ZI + 6 - c, six zeros at ZI (partitioned sweep)
        L.M = M; L.O = O; L.V = V;
        L.OS = take(M); L.OY = take(M);
        L.OI = o; o += (M * n * 2 + tsize - 1) / tsize;      // uint16 per row and grid point (0xffff = no row): a quarter of a T word each -- what lets BASELINE configs[2] (n = 80, 16 polygons) keep TWO workgroups per CU
        L.OG = take(gs ? 0 : M); L.OAX = take(gs ? 0 : M); L.OAY = take(gs ? 0 : M); L.OHK = take(gs ? 0 : M);      // (gs: the cached row values / gradients / curvatures live in the global block too)
        L.GV = o; o += 2 * O * V; L.GNV = o; o += O; L.GR = o; o += O; L.GC = o; o += 2 * O;
        L.GE = o; o += V >= 2 ? 3 * O * V : 0;
        L.NV = NV; L.VIA = o; o += 3 * NV; L.VIDX = o; o += NV;
        L.OAT = take(gs ? 0 : MT); L.OHXT = take(gs ? 0 : MT); L.OHYT = take(gs ? 0 : MT); L.OHTT = take(gs ? 0 : MT);
        L.OAD = take(gs ? 0 : MD); L.OHXD = take(gs ? 0 : MD); L.OHYD = take(gs ? 0 : MD); L.OHDD = take(gs ? 0 : MD); L.OHTD = take(gs ? 0 : MD);
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
    static constexpr int M = 0, O = 0, V = 1, NV = 0, GSW = 0, GSF = 0, OEB = 0, MT = 0, MD = 0, OXB = 0;
    static constexpr int OS = c().OS, OY = c().OY, OI = c().OI, OG = c().OG, OAX = c().OAX, OAY = c().OAY, OHK = c().OHK, GV = c().GV, GNV = c().GNV, GR = c().GR, GC = c().GC, GE = c().GE,
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

}  // namespace mpc

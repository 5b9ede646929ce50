// Micro-benchmark: what a wave pays for the vector-memory loads of a serial sweep when ONE or FOUR one-wave workgroups share a compute unit (the regime of the solve kernel's global
// form).  Every wave streams through its own block of global memory in "stages": per stage K loads of 8 bytes per lane, one stage ahead of ~90 dependent fp64 FMAs
// that consume them.  Patterns (what the 12 active lanes of each 16-lane row read in one load instruction; the four rows of the wave read the same addresses, as in the serial sweep):
//   rows     : lane c reads row (12 k + c) mod 42 at column `stage`   -- component-major rows, 12 distinct cache lines per instruction, 42 per stage, each reused for 16 stages
//   records  : lane c reads word (12 k + c) mod 42 of the stage's record -- stage-major records, 1-2 distinct lines per instruction, 3 new lines every stage
//   rows4    : as rows, but every 16-lane row of the wave at its own quarter of the columns (the partitioned sweep): 48 distinct lines per instruction
//   none     : no loads (the arithmetic alone)
//   tile4    : records of FOUR stages interleaved word by word (word e of stage s at (s / 4) * 4 * 42 + 4 e + s % 4): 3-4 distinct lines per instruction, reused for 4 stages
// and the lane-parallel side of the same layouts (`pass`): lane = stage, 39 words per stage written (or read) by 39 instructions, two rounds of 64 stages, ~500 ticks of
// arithmetic per round: rows = 4 lines per instruction, tile4 = 16, records = 64.
// build: hipcc --offload-arch=gfx950 -O3 vmem_lines.hip -o vmem_lines ; run: ./vmem_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int K = 11, NROW = 42, PITCH = 144, NST = 120;         // per wave: NROW rows of PITCH words (rows) or NST records of NROW words (records) -- the 39 + 3 words per stage the backward sweep reads
constexpr int BLOCK_WORDS = NROW * PITCH;                        // 6048 words = 47 KB per wave

template <int MODE> __global__ __launch_bounds__(64) void k(const double* __restrict__ mem, double* out, long long* ticks, int sweeps) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x, c = lane & 15, row = lane >> 4;
    const double* blk = mem + (size_t)blockIdx.x * BLOCK_WORDS;
    if (lane == 0) sm[0] = 0.0;
    __syncthreads();
    double acc = 1.0, q[K], qn[K];
    auto load = [&](int st, double (&v)[K]) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int cc = c < 12 ? c : 0;
            size_t w;
            const int e = (12 * j + cc) % NROW;                  // which of the stage's 42 words this lane reads in load j
            if (MODE == 0) w = (size_t)e * PITCH + st;
            else if (MODE == 1) w = (size_t)st * NROW + e;
            else if (MODE == 4) w = (size_t)(st >> 2) * (4 * NROW) + 4 * e + (st & 3);
            else w = (size_t)e * PITCH + (st % (NST / 4)) + row * (NST / 4);
            v[j] = MODE == 3 ? 1.0 : blk[w];
        }
    };
    long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < sweeps; ++s) {
        load(NST - 1, q);
        for (int st = NST - 1; st >= 0; --st) {
            load(st > 0 ? st - 1 : 0, qn);
            double x = acc;
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < K; ++j) x = __builtin_fma(x, 1.0000001, q[j] * 1e-12);      // 88 dependent FMAs (+ 88 muls the compiler may fold into them)
            acc = x;
#pragma unroll
            for (int j = 0; j < K; ++j) q[j] = qn[j];
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * 64 + lane] = acc;
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int LAYOUT, bool STORE> __global__ __launch_bounds__(64) void kp(double* __restrict__ mem, double* out, long long* ticks, int reps) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    double* blk = mem + (size_t)blockIdx.x * BLOCK_WORDS;
    if (lane == 0) sm[0] = 0.0;
    __syncthreads();
    double acc = 1.0 + lane;
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r)
        for (int base = 0; base < 128; base += 64) {
            const int st = base + lane;
            if (st >= NST) continue;
            double v[39];
#pragma unroll
            for (int e = 0; e < 39; ++e) {
                const size_t w = LAYOUT == 0 ? (size_t)e * PITCH + st : (LAYOUT == 1 ? (size_t)st * NROW + e : (size_t)(st >> 2) * (4 * NROW) + 4 * e + (st & 3));
                if (STORE) blk[w] = acc + e; else v[e] = blk[w];
            }
            double x = acc;
            if (!STORE) {
#pragma unroll
                for (int e = 0; e < 39; ++e) x += v[e];
            }
#pragma unroll
            for (int j = 0; j < 100; ++j) x = __builtin_fma(x, 1.0000001, 1e-12);
            acc = x;
            __syncthreads();
        }
    long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * 64 + lane] = acc;
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double* mem; double* out; long long* ticks;
    const int maxwg = cus * 4;
    hipMalloc(&mem, (size_t)maxwg * BLOCK_WORDS * 8); hipMemset(mem, 0, (size_t)maxwg * BLOCK_WORDS * 8);
    hipMalloc(&out, (size_t)maxwg * 64 * 8); hipMalloc(&ticks, (size_t)maxwg * 8);
    const char* names[5] = {"rows (12 lines / instruction, reused 16 stages)", "records (1-2 lines / instruction, new every stage)", "rows, four segments (48 lines / instruction)", "no loads",
                            "tile4 (3-4 lines / instruction, reused 4 stages)"};
    for (int mode = 0; mode < 5; ++mode)
        for (int w = 1; w <= 4; w *= 2) {
            const int grid = cus * w;
            const size_t lds = w == 1 ? 100 * 1024 : (w == 2 ? 70 * 1024 : 36 * 1024);      // the dynamic LDS request sets how many workgroups share a CU
            const int sweeps = 20;
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), lds, 0, mem, out, ticks, sweeps);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), lds, 0, mem, out, ticks, sweeps);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), lds, 0, mem, out, ticks, sweeps);
                else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), lds, 0, mem, out, ticks, sweeps);
                else hipLaunchKernelGGL(k<4>, dim3(grid), dim3(64), lds, 0, mem, out, ticks, sweeps);
                hipDeviceSynchronize();
            }
            std::vector<long long> h(grid); hipMemcpy(h.data(), ticks, grid * 8, hipMemcpyDeviceToHost);
            double s = 0; for (long long v : h) s += (double)v;
            printf("%-52s %d workgroup(s) per CU: %7.1f ticks per stage (%d loads, ~%d FMAs)\n", names[mode], w, s / grid / sweeps / NST, K, 8 * K);
        }
    const char* lay[3] = {"rows (4 lines / instruction)", "records (64 lines / instruction)", "tile4 (16 lines / instruction)"};
    for (int st = 0; st < 2; ++st)
        for (int l = 0; l < 3; ++l)
            for (int w = 1; w <= 4; w *= 4) {
                const int grid = cus * w;
                const size_t lds = w == 1 ? 100 * 1024 : 36 * 1024;
                const int reps = 50;
                for (int rep = 0; rep < 2; ++rep) {
                    if (st == 0) { if (l == 0) hipLaunchKernelGGL((kp<0, false>), dim3(grid), dim3(64), lds, 0, mem, out, ticks, reps); else if (l == 1) hipLaunchKernelGGL((kp<1, false>), dim3(grid), dim3(64), lds, 0, mem, out, ticks, reps); else hipLaunchKernelGGL((kp<2, false>), dim3(grid), dim3(64), lds, 0, mem, out, ticks, reps); }
                    else { if (l == 0) hipLaunchKernelGGL((kp<0, true>), dim3(grid), dim3(64), lds, 0, mem, out, ticks, reps); else if (l == 1) hipLaunchKernelGGL((kp<1, true>), dim3(grid), dim3(64), lds, 0, mem, out, ticks, reps); else hipLaunchKernelGGL((kp<2, true>), dim3(grid), dim3(64), lds, 0, mem, out, ticks, reps); }
                    hipDeviceSynchronize();
                }
                std::vector<long long> h(grid); hipMemcpy(h.data(), ticks, grid * 8, hipMemcpyDeviceToHost);
                double s = 0; for (long long v : h) s += (double)v;
                printf("pass, %-6s %-34s %d workgroup(s) per CU: %8.0f ticks per pass over 120 stages (39 words each)\n", st ? "stores" : "loads", lay[l], w, s / grid / reps);
            }
    return 0;
}

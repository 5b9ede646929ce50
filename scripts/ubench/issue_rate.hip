// Micro-benchmark: instruction issue cost for ONE wave per SIMD on gfx950 (the regime the solve kernel runs in).
// build: hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate ; run: ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int MODE> __global__ __launch_bounds__(64) void k(double* out, long long* ticks, int iters) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    double a0 = out[lane], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1.0000001, c = 1e-9;
    for (int i = lane; i < 4096; i += 64) sm[i] = i;
    __syncthreads();
    int addr_b = 0, addr_d = lane * 8, addr_c = (lane * 17 % 64) * 8 + (lane & 3) * 4096;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // 64 independent-ish fp64 FMAs (8 chains)
            asm volatile(REP16("v_fmac_f64 %0, %8, %9\n v_fmac_f64 %1, %8, %9\n v_fmac_f64 %2, %8, %9\n v_fmac_f64 %3, %8, %9\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (MODE == 1) {   // 64 dependent fp64 FMAs
            asm volatile(REP64("v_fmac_f64 %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
        } else if (MODE == 2) {   // 64 fp64 DPP FMAs, 4 chains
            asm volatile(REP16("v_fmac_f64_dpp %0, %4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %4, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        } else if (MODE == 3) {   // 32 (s_nop 1 + fp64 FMA), 4 chains
            asm volatile(REP16("s_nop 1\n v_fmac_f64 %0, %4, %5\n s_nop 1\n v_fmac_f64 %1, %4, %5\n s_nop 1\n v_fmac_f64 %2, %4, %5\n s_nop 1\n v_fmac_f64 %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        } else if (MODE == 4) {   // 64 independent fp32 FMAs
            float f0 = a0, f1 = a1, f2 = a2, f3 = a3, fb = 1.0001f, fc = 1e-5f;
            asm volatile(REP16("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fb), "v"(fc));
            a0 += f0 + f1 + f2 + f3;
        } else if (MODE == 5) {   // 64 broadcast ds_read_b64 (all lanes same address), one wait at the end
            double r;
            asm volatile(REP64("ds_read_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r) : "v"(addr_b));
            a0 += r;
        } else if (MODE == 6) {   // 64 conflict-free ds_read_b64 (lane-linear)
            double r;
            asm volatile(REP64("ds_read_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r) : "v"(addr_d));
            a0 += r;
        } else if (MODE == 7) {   // 64 scattered ds_read_b64
            double r;
            asm volatile(REP64("ds_read_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r) : "v"(addr_c));
            a0 += r;
        } else if (MODE == 8) {   // LDS round trip latency: 16 x (write, wait, read, wait)
            double r = a0;
            asm volatile(REP16("ds_write_b64 %1, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(r) : "v"(addr_d));
            a0 = r;
        } else if (MODE == 9) {   // 64 dependent fp64 adds
            asm volatile(REP64("v_add_f64 %0, %0, %1\n") : "+v"(a0) : "v"(c));
        } else if (MODE == 10) {  // 64 v_mov_b32 (cheap VALU)
            int x = lane, y;
            asm volatile(REP64("v_mov_b32 %0, %1\n") : "=v"(y) : "v"(x));
            a0 += y;
        } else if (MODE == 11) {  // 64 x s_nop 0
            asm volatile(REP64("s_nop 0\n"));
        } else if (MODE == 12) {  // v_rcp_f64 dependent chain x16
            asm volatile(REP16("v_rcp_f64 %0, %0\n") : "+v"(a0));
        } else if (MODE == 13) {  // readlane pairs x32
            int x = lane, s;
            asm volatile(REP64("v_readlane_b32 %0, %1, 3\n") : "=s"(s) : "v"(x));
            a0 += s;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, int per_iter, int blocks, size_t lds) {
    double* out; long long* ticks;
    hipMalloc(&out, blocks * 64 * sizeof(double)); hipMalloc(&ticks, blocks * sizeof(long long));
    hipMemset(out, 0, blocks * 64 * sizeof(double));
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64, lds>>>(out, ticks, 10);
    hipEventRecord(e0);
    k<MODE><<<blocks, 64, lds>>>(out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), ticks, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto t : h) mean += t; mean /= blocks;
    printf("%-34s blocks=%5d  ticks/instr %7.2f   ns/instr %7.3f  (ticks/ns %.3f)\n", name, blocks, mean / iters / per_iter,
           ms * 1e6 / iters / per_iter, mean / (ms * 1e6));
    hipFree(out); hipFree(ticks);
}

int main() {
    for (int blocks : {1, 1024}) {
        const size_t lds = 39 * 1024;     // same LDS footprint as the solve kernel: 4 workgroups per CU, one wave per SIMD
        run<0>("fp64 fma, 8 chains", 64, blocks, lds);
        run<1>("fp64 fma, dependent", 64, blocks, lds);
        run<9>("fp64 add, dependent", 64, blocks, lds);
        run<2>("fp64 fma dpp newbcast, 4 chains", 64, blocks, lds);
        run<3>("s_nop 1 + fp64 fma (per pair)", 64, blocks, lds);
        run<4>("fp32 fma, 4 chains", 64, blocks, lds);
        run<10>("v_mov_b32", 64, blocks, lds);
        run<11>("s_nop 0", 64, blocks, lds);
        run<12>("v_rcp_f64 dependent", 16, blocks, lds);
        run<13>("v_readlane_b32", 64, blocks, lds);
        run<5>("ds_read_b64 broadcast", 64, blocks, lds);
        run<6>("ds_read_b64 lane-linear", 64, blocks, lds);
        run<7>("ds_read_b64 scattered", 64, blocks, lds);
        run<8>("LDS write+read round trip", 16, blocks, lds);
    }
    return 0;
}

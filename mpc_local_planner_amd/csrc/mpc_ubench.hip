// mpc_ubench.hip -- MEASUREMENT AID of bench.py, not part of the C ABI of include/mpc_hip.h (own library: csrc/libmpc_ubench.so).
//
// Full-occupancy v_fma_f64 / v_fma_f32 rate of the device bench.py runs on: the denominator of `roofline.fp64_valu` next to the 78.6 TFLOP/s of the data sheet
// (MI355X_MICROARCH.md does not state a vector fp64 peak; SURVEY.md 8d asks for this measurement).  Every lane carries NACC independent accumulator chains
// (a = a * b + c), 8 waves per SIMD are resident, nothing touches memory inside the timed loop.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int NACC = 8, UNROLL = 8;
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T>
__global__ __launch_bounds__(256) void fma_chain_kernel(T* out, int iters, T b, T c) {
    T a[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) a[j] = T(threadIdx.x + j) * T(1e-3);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) a[j] = fma_t(a[j], b, c);
    }
    T s = T(0);
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += a[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename T>
int run(int device, int iters, double* tflops, double* ms_out) {
    if (hipSetDevice(device) != hipSuccess) return -2;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -4;
    const int blocks = prop.multiProcessorCount * 8, threads = 256;       // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    T* out = nullptr;
    if (hipMalloc(&out, sizeof(T) * (size_t)blocks * threads) != hipSuccess) return -3;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipFree(out); return -4; }
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {          // first repetition = warm-up (clock ramp, code load)
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(fma_chain_kernel<T>, dim3(blocks), dim3(threads), 0, 0, out, iters, T(1.0000001), T(1e-9));
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { (void)hipFree(out); return -4; }
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(out);
    const double flops = 2.0 * NACC * UNROLL * (double)iters * (double)blocks * threads;
    *tflops = flops / (best * 1e-3) / 1e12;
    if (ms_out) *ms_out = best;
    return 0;
}
}  // namespace

extern "C" {
// best of three timed launches; `iters` loop trips of 64 FMAs per lane (20000 -> ~10 ms on an MI355X)
int mpc_ubench_fma_f64(int device, int iters, double* tflops, double* ms) { return run<double>(device, iters, tflops, ms); }
int mpc_ubench_fma_f32(int device, int iters, double* tflops, double* ms) { return run<float>(device, iters, tflops, ms); }
}

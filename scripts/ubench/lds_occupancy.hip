// developer probe: how many 64-lane workgroups with D bytes of dynamic LDS fit on one CU of this GPU (hipOccupancyMaxActiveBlocksPerMultiprocessor)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(double* o) { extern __shared__ double sm[]; sm[threadIdx.x] = 1.0; __syncthreads(); o[threadIdx.x] = sm[63 - threadIdx.x]; }
int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int last = -1;
    for (int d = 30000; d <= 163840; d += 64) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64, d) != hipSuccess) { printf("query failed at %d\n", d); break; }
        if (nb != last) { printf("dynamic LDS %6d B -> %d workgroups per CU\n", d, nb); last = nb; }
    }
    return 0;
}
// measured on MI355X (gfx950, 160 KB LDS per CU), 64-lane workgroups: 5 workgroups per CU up to 32 752 B of dynamic LDS, 4 up to 40 944 B, 3 up to
// 54 576 B, 2 up to 81 904 B, 1 above.  The solve kernel's record: 38.5 KB at n = 50 (4 per CU), 80.5 KB for config 3 (n = 80, 16 hexagons, 4 rows: 2 per CU).

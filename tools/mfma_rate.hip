// tools/mfma_rate.hip -- issue rate of the matrix-core instructions the prefill kernels use, per SIMD: N back-to-back MFMAs on NACC independent accumulators,
// 1 or 2 waves per SIMD, cycles from s_memtime.   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int   i32x4  __attribute__((ext_vector_type(4)));
typedef int   i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int   i32x4c __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND, int NACC>
__global__ void __launch_bounds__(512) k(unsigned long long * out, int iters, int seed) {
    i32x4 a, b; f16x8 ha, hb;
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 77 + i + seed; b[i] = threadIdx.x * 31 + 3 * i + seed; }
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16) (float) ((threadIdx.x + i) & 7); hb[i] = (_Float16) (float) ((threadIdx.x * 3 + i) & 7); }
    i32x16 acc[NACC]; f32x16 facc[NACC]; i32x4c acc4[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) { acc[n][e] = 0; facc[n][e] = 0.0f; }
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) acc4[n][e] = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            if (KIND == 0) acc[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[n], 0, 0, 0);
            if (KIND == 1) facc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, facc[n], 0, 0, 0);
            if (KIND == 2) acc4[n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc4[n], 0, 0, 0);
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0; float f = 0;
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) { s += acc[n][e]; f += facc[n][e]; }
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) s += acc4[n][e];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (unsigned long long) (s + (int) f); }
}
template <int KIND, int NACC> int run(const char * name, int threads, int grid) {
    unsigned long long * d; CK(hipMalloc(&d, grid * 16));
    const int iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<KIND, NACC><<<grid, threads>>>(d, iters, 1); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k<KIND, NACC><<<grid, threads>>>(d, iters, 2); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    const double per_simd = (double) iters * NACC * (threads / 256);          // MFMAs issued on one SIMD (threads / 256 waves per SIMD)
    const double mac = KIND == 2 ? 16.0 * 16 * 64 : (KIND == 0 ? 32.0 * 32 * 32 : 32.0 * 32 * 16);
    printf("%-28s %d acc, %d waves/SIMD, grid %4d: %7.1f memtime ticks / MFMA / SIMD, wall %.3f ms -> %.0f T(FL)OP/s (%.2f GHz-equivalent at 32 cyc)\n", name, NACC, threads / 256, grid,
           (double) h[0] / per_simd, ms, 2.0 * mac * per_simd * 4 * grid / (ms * 1e-3) / 1e12, per_simd * 32 / (ms * 1e-3) / 1e9);
    CK(hipFree(d)); return 0;
}
int main() {
    run<0, 4>("i32_32x32x32_i8", 256, 256); run<0, 4>("i32_32x32x32_i8", 512, 256); run<0, 2>("i32_32x32x32_i8", 512, 256); run<0, 1>("i32_32x32x32_i8", 256, 256);
    run<1, 4>("f32_32x32x16_f16", 256, 256); run<1, 4>("f32_32x32x16_f16", 512, 256);
    run<2, 4>("i32_16x16x64_i8", 256, 256); run<2, 4>("i32_16x16x64_i8", 512, 256); run<2, 8>("i32_16x16x64_i8", 256, 256);
    run<0, 4>("i32_32x32x32_i8 (1 CU)", 256, 1); run<1, 4>("f32_32x32x16_f16 (1 CU)", 256, 1);
    return 0;
}

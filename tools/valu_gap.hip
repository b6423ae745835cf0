// tools/valu_gap.hip -- what N filler instructions of one kind cost between two v_mfma_i32_32x32x32_i8 of ONE wave per SIMD (in-order issue: 32 cycles per MFMA is
// the floor; whatever the fillers add on top is what they cost beside the matrix core).  hipcc --offload-arch=gfx950 -O3 tools/valu_gap.hip -o tools/bin/valu_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
template <int KIND, int N>
__global__ void __launch_bounds__(256) k(unsigned long long * out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    i32x4 a, b; i32x16 acc[4];
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 77 + i; b[i] = threadIdx.x * 31 + 3 * i; }
    for (int n = 0; n < 4; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0;
    uint32_t v0 = threadIdx.x, v1 = threadIdx.x * 3 + 1, v2 = 7, v3 = 9, v4 = 11, v5 = 13, v6 = 15, v7 = 17;
    float f0 = 1.0f, f1 = 2.0f, f2 = 3.0f, f3 = 0.5f, f4 = 1.5f, f5 = 2.5f, f6 = 3.5f, f7 = 4.5f;
    const uint32_t la = (threadIdx.x & 63) * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            acc[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[n], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < N; ++r) {
                uint32_t & x = r % 4 == 0 ? v0 : r % 4 == 1 ? v2 : r % 4 == 2 ? v4 : v6;          // four independent chains
                uint32_t & y = r % 4 == 0 ? v1 : r % 4 == 1 ? v3 : r % 4 == 2 ? v5 : v7;
                float & fx = r % 4 == 0 ? f0 : r % 4 == 1 ? f2 : r % 4 == 2 ? f4 : f6;
                float & fy = r % 4 == 0 ? f1 : r % 4 == 1 ? f3 : r % 4 == 2 ? f5 : f7;
                if (KIND == 0) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (KIND == 1) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(y));
                if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fx) : "v"(fy));
                if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double *) &fx) : "v"(*(double *) &fy));
                if (KIND == 4) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(fx) : "v"(y));
                if (KIND == 5) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(y));
                if (KIND == 6) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
                if (KIND == 7) { i32x4 d; asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(la)); asm volatile("" :: "v"(d)); }
                if (KIND == 8) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(y));
                if (KIND == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *) &fx) : "v"(*(double *) &fy));
                if (KIND == 10) asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(x));
                if (KIND == 11) asm volatile("v_mad_u32_u16 %0, %0, %1, %1" : "+v"(x) : "v"(y));
            }
        }
        if (KIND == 7) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int n = 0; n < 4; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
    s += v0 + v2 + v4 + v6 + (int) (f0 + f2 + f4 + f6) + lds[threadIdx.x];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = s; }
}
template <int KIND, int N> int run(const char * name) {
    unsigned long long * d; CK(hipMalloc(&d, 256 * 16));
    const int iters = 500;
    k<KIND, N><<<256, 256>>>(d, iters); CK(hipDeviceSynchronize());
    k<KIND, N><<<256, 256>>>(d, iters); CK(hipDeviceSynchronize());
    unsigned long long h[2]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("%-18s x %d per gap: %6.1f cycles per MFMA\n", name, N, (double) h[0] / (iters * 4.0));
    CK(hipFree(d)); return 0;
}
#define ALL(K, name) run<K, 0>(name); run<K, 2>(name); run<K, 4>(name); run<K, 6>(name); run<K, 8>(name);
int main() {
    ALL(0, "v_and_b32") ALL(1, "v_pk_mul_lo_u16") ALL(2, "v_fma_f32") ALL(3, "v_pk_fma_f32") ALL(9, "v_pk_mul_f32") ALL(4, "v_cvt_f32_i32") ALL(5, "v_lshl_add_u32")
    ALL(6, "v_perm_b32") ALL(8, "v_mul_u32_u24") ALL(10, "v_lshrrev_b32") ALL(11, "v_mad_u32_u16") ALL(7, "ds_read_b128")
    return 0;
}

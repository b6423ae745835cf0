// tools/lds_align.hip -- measurement: cost of ds_read_b128 / b64 / b32 by address alignment (the Q6_K super-block is 210 bytes: 2-byte aligned in an LDS ring)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
extern __shared__ char lds[];
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 __attribute__((aligned(2))) u32x4_a2;
typedef u32x2 __attribute__((aligned(2))) u32x2_a2;
typedef uint32_t __attribute__((aligned(2))) u32_a2;
template <int W> __global__ void __launch_bounds__(256) k(int stride, int mis, int iters, unsigned long long * cyc, uint32_t * out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 256) ((uint32_t *) lds)[i] = i;
    __syncthreads();
    const char * p = lds + wave * 16384 + lane * stride + mis;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const char * q = p + ((i + u) & 3) * 16;
            if (W == 16) { const u32x4 v = *(const volatile u32x4_a2 *) q; acc ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
            if (W == 8)  { const u32x2 v = *(const volatile u32x2_a2 *) q; acc ^= v[0] ^ v[1]; }
            if (W == 4)  { acc ^= *(const volatile u32_a2 *) q; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    unsigned long long * cyc; uint32_t * out;
    CHECK(hipMalloc(&cyc, 8 * 256)); CHECK(hipMalloc(&out, 4 * 256 * 256));
    const int iters = 256;
    auto run = [&](auto kern, int w, int stride, int mis) {
        kern<<<1, 256, 65536>>>(stride, mis, iters, cyc, out);
        unsigned long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        printf("  ds_read_b%-3d lane stride %3d B, misalignment %2d B: %6.1f cycles per wave-instruction (4 waves on the CU)\n", w * 8, stride, mis, (double) c / (iters * 8));
    };
    for (int mis : { 0, 8, 4, 2, 6 }) run(k<16>, 16, 16, mis);
    for (int mis : { 0, 4, 2 }) run(k<16>, 16, 210 / 4 * 4 == 0 ? 16 : 52, mis);        // a 52-byte lane stride (a quarter of a Q6_K block, roughly)
    for (int mis : { 0, 4, 2 }) run(k<8>, 8, 8, mis);
    for (int mis : { 0, 2 }) run(k<4>, 4, 4, mis);
    return 0;
}

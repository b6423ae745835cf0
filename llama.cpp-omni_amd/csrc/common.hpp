// common.hpp -- shared host/device helpers for the MI355X (gfx950, wave64) ggml backend.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "ggml_abi.h"

#define MI_WAVE 64
#ifdef __HIPCC__
#define MI_HD __host__ __device__
#else
#define MI_HD
#endif

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            fprintf(stderr, "[mi355x] HIP error %d (%s) at %s:%d: %s\n", (int) _e,                \
                    hipGetErrorString(_e), __FILE__, __LINE__, #expr);                           \
            abort();                                                                             \
        }                                                                                        \
    } while (0)

#ifdef __HIPCC__
// ---------------------------------------------------------------- device helpers
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ float h2f(uint16_t h) {       // IEEE half -> float, exact
    _Float16 v; __builtin_memcpy(&v, &h, 2); return (float) v;
}
static __device__ __forceinline__ uint16_t f2h(float f) {       // float -> IEEE half, round-to-nearest-even
    _Float16 v = (_Float16) f; uint16_t h; __builtin_memcpy(&h, &v, 2); return h;
}

// streamed-once weight load: 16 B, non-temporal (does not displace L2/MALL-resident activations)
static __device__ __forceinline__ u32x4 ld_nt16(const void * p) {
    return __builtin_nontemporal_load((const u32x4 *) p);
}
static __device__ __forceinline__ uint32_t ld_nt4(const void * p) {
    return __builtin_nontemporal_load((const uint32_t *) p);
}

// 4 x int8 dot product with int32 accumulate (v_dot4_i32_i8)
static __device__ __forceinline__ int dot4(uint32_t a, uint32_t b, int c) {
    return __builtin_amdgcn_sdot4((int) a, (int) b, c, false);
}

template <typename T>
static __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum over 256/512/1024 threads through LDS (scratch holds one value per wave)
template <typename T>
static __device__ __forceinline__ T block_sum(T v, T * scratch) {
    v = wave_sum(v);
    const int nw = (blockDim.x + 63) >> 6;
    if (nw == 1) return v;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) scratch[w] = v;
    __syncthreads();
    T r = scratch[0];
    for (int i = 1; i < nw; ++i) r += scratch[i];
    return r;
}
static __device__ __forceinline__ float block_max(float v, float * scratch) {
    v = wave_max(v);
    const int nw = (blockDim.x + 63) >> 6;
    if (nw == 1) return v;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}
#endif // __HIPCC__

// ---------------------------------------------------------------- activation scratch layout ("q8 image")
// A row of K f32 activations quantised like the reference's block_q8_K stream
// (ggml-quants.c:2555-2592), but stored split so that every part is 16-B aligned for LDS staging:
//   [ qs : K int8 ][ bsums : K/16 int16 ][ d : K/256 float ][ pad to 16 B ]
MI_HD static inline size_t q8k_image_bytes(int64_t K) {
    size_t n = (size_t) K + (size_t) K / 8 + (size_t) K / 64;
    return (n + 15) & ~(size_t) 15;
}
// Q8_0 image (activations for Q8_0 weights, per-32 scale; reference x86 path arch/x86/quants.c quantize_row_q8_0):
//   [ qs : K int8 ][ d : K/32 float (already widened from the f16 the reference stores) ][ pad ]
MI_HD static inline size_t q80_image_bytes(int64_t K) {
    size_t n = (size_t) K + (size_t) K / 32 * 4;
    return (n + 15) & ~(size_t) 15;
}

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
// f32 wave64 sum on the DPP network (no LDS round trips): quad butterflies, row mirrors, then the two row broadcasts;
// the total lands in lane 63 and is returned wave-uniform through v_readlane.  ~8 VALU instructions instead of six
// dependent ds_bpermute (each ~100+ cycles), which matters at the end of every mat-vec row group.
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
static __device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_f32<0xB1, 0xf>(v);        // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E, 0xf>(v);        // quad_perm [2,3,0,1]
    v += dpp_f32<0x141, 0xf>(v);       // row_half_mirror
    v += dpp_f32<0x140, 0xf>(v);       // row_mirror        -> every lane holds its 16-lane row total
    v += dpp_f32<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    v += dpp_f32<0x143, 0xc>(v);       // row_bcast:31 into rows 2 and 3 -> lanes 48..63 hold the wave total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
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
// ------------------------------------------------------------------------------------------------
// One 256-element Q8_K block held by one wave (lane l owns elements 4l..4l+3), written into the image.
//   reference: quantize_row_q8_K_ref, ggml-quants.c:2555-2592 (x86 `quantize_row_q8_K` forwards to it,
//   ggml-cpu/arch/x86/quants.c:493-495):
//     amax/max  : first element (lowest index) with the largest |x|   (strict '>' scan)
//     iscale    = -127.f / max
//     q[j]      = min(127, nearest_int(iscale * x[j]))   -- nearest_int == round-half-even (:444-449)
//     bsums[g]  = sum of 16 consecutive q
//     d         = 1 / iscale           (amax == 0 -> d = 0, q = 0)
// ------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void q8k_block_from_regs(const f32x4 v, int lane, int8_t * qs, int16_t * bs, float * ds) {
    // (|x|, index) arg-max with lowest-index tie-break == the reference's sequential strict-'>' scan
    float amax = fabsf(v[0]); float mval = v[0]; int idx = 4 * lane;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const float a = fabsf(v[i]);
        if (a > amax) { amax = a; mval = v[i]; idx = 4 * lane + i; }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float a2 = __shfl_xor(amax, o, 64);
        const float m2 = __shfl_xor(mval, o, 64);
        const int   i2 = __shfl_xor(idx, o, 64);
        if (a2 > amax || (a2 == amax && i2 < idx)) { amax = a2; mval = m2; idx = i2; }
    }
    if (amax == 0.0f) {                  // all-zero block (also catches -0.0f)
        *(uint32_t *) (qs + 4 * lane) = 0u;
        if ((lane & 3) == 0) bs[lane >> 2] = 0;
        if (lane == 0) *ds = 0.0f;
        return;
    }
    const float iscale = -127.0f / mval;
    int q[4]; int s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float p = iscale * v[i];               // one rounding, like the C source (no FMA with the magic add)
        int r = (int) __builtin_rintf(p);            // round-half-even == nearest_int()
        r = r > 127 ? 127 : r;
        q[i] = r; s += r;
    }
    *(uint32_t *) (qs + 4 * lane) = (uint32_t) (q[0] & 0xff) | ((uint32_t) (q[1] & 0xff) << 8) |
                                    ((uint32_t) (q[2] & 0xff) << 16) | ((uint32_t) (q[3] & 0xff) << 24);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if ((lane & 3) == 0) bs[lane >> 2] = (int16_t) s;
    if (lane == 0) *ds = 1.0f / iscale;
}

// What the reference's Q8_K quantisation makes of one 256-block held by a wave (lane l owns elements 4l..4l+3): d * q per element, d = 1 / iscale, q as
// quantize_row_q8_K_ref computes it (ggml-quants.c:2555-2592; q8k_block_from_regs above) -- the value ggml_vec_dot_q*_K_q8_K multiplies the weights with.
// The prefill GEMMs that run on F16 images of K-quant weights take these values (rounded to f16) as their activations instead of the raw f32 rows, so the only
// difference to the reference's integer arithmetic left is f16 rounding of the two factors, not the reference's own 8-bit quantisation noise.
static __device__ __forceinline__ float wave_max_pos_f32(float v) {     // wave64 max of non-negative values on the DPP network (masked-out rows read 0)
    v = fmaxf(v, dpp_f32<0xB1, 0xf>(v));
    v = fmaxf(v, dpp_f32<0x4E, 0xf>(v));
    v = fmaxf(v, dpp_f32<0x141, 0xf>(v));
    v = fmaxf(v, dpp_f32<0x140, 0xf>(v));
    v = fmaxf(v, dpp_f32<0x142, 0xa>(v));
    v = fmaxf(v, dpp_f32<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
static __device__ __forceinline__ f32x4 q8k_requant4(const f32x4 v, int lane) {
    float am = fabsf(v[0]); float mv = v[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) { const float a = fabsf(v[i]); if (a > am) { am = a; mv = v[i]; } }
    const float amax = wave_max_pos_f32(am);
    if (amax == 0.0f) return f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    const unsigned long long hit = __ballot(am == amax);               // the FIRST element with the largest |x| decides the sign of the scale (strict '>' scan)
    const float maxv = __shfl(mv, (int) __builtin_ctzll(hit), 64);
    const float iscale = -127.0f / maxv, d = 1.0f / iscale;
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) { int r = (int) __builtin_rintf(iscale * v[i]); r = r > 127 ? 127 : r; o[i] = d * (float) r; }
    return o;
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

// mv_dev.hpp -- device helpers shared by the batch-1 decode mat-vec families (mmv1.hip, mmv2.hip): DPP-network wave reductions, the LDS
// image of the Q8_K activation row, the in-kernel activation prologue (RMS norm + Q8_K quantiser), matrix descriptors of a launch.
#pragma once
#include "../kernels.hpp"

namespace mi {

extern __shared__ __attribute__((aligned(16))) char mv1_lds[];

// ------------------------------------------------------------------------------------------------ DPP-network wave reductions
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }

// max of non-negative values (lanes outside a row_bcast's row mask contribute the identity 0)
// (non-negative IEEE floats order like their bit patterns, and an unsigned max with identity 0 folds into one v_max_u32_dpp per step)
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ uint32_t dpp_u32(uint32_t v) { return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, CTRL, ROW_MASK, 0xf, false); }
static __device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
template <int CTRL> static __device__ __forceinline__ uint32_t dpp_u32q(uint32_t v) { return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, CTRL, 0xf, 0xf, true); }
static __device__ __forceinline__ float wave_max_pos(float f) {
    uint32_t v = (uint32_t) __float_as_int(f);
    v = umax32(v, dpp_u32<0xB1, 0xf>(v));
    v = umax32(v, dpp_u32<0x4E, 0xf>(v));
    v = umax32(v, dpp_u32<0x141, 0xf>(v));
    v = umax32(v, dpp_u32<0x140, 0xf>(v));
    v = umax32(v, dpp_u32<0x142, 0xa>(v));
    v = umax32(v, dpp_u32<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane((int) v, 63));
}
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ double dpp_f64(double v) {
    const int lo = dpp_i32<CTRL, ROW_MASK>(__double2loint(v)), hi = dpp_i32<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<0xB1, 0xf>(v);
    v += dpp_f64<0x4E, 0xf>(v);
    v += dpp_f64<0x141, 0xf>(v);
    v += dpp_f64<0x140, 0xf>(v);
    v += dpp_f64<0x142, 0xa>(v);
    v += dpp_f64<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// LDS image of the activation row (private to this file; nb = K / 256 blocks):
//   [ qs : nb x 272 B ]  int8 quants of block b at b * 272 (16 B of padding per block: the four lanes of a weight block read 64-B pieces of
//                        one activation block, sixteen weight blocks per wave -- a 256-B stride would put them all on the same banks)
//   [ bs : nb x 16 B  ]  int16 sums of the 8 sub-blocks of 32 (= bsums[2s] + bsums[2s+1] of block_q8_K): the Q4_K body's `mins` term
//   [ b16: nb x 32 B  ]  the 16 bsums of block_q8_K, permuted so that a Q6_K lane reads its four as one 8-byte piece:
//                        position 8n + 4hf + m holds bsums[8n + 2m + hf] (mv1_b16_pos)
//   [ d  : nb x 4 B   ]  f32 scale
static __device__ __forceinline__ int mv1_img_bs(int nb)  { return nb * 272; }
static __device__ __forceinline__ int mv1_img_b16(int nb) { return nb * 288; }
static __device__ __forceinline__ int mv1_img_d(int nb)   { return nb * 320; }
static __device__ __forceinline__ int mv1_b16_pos(int s)  { return (s & 8) | ((s & 1) << 2) | ((s >> 1) & 3); }
MI_HD static inline size_t mv1_image_bytes(int64_t K) { return (size_t) (K / 256) * 324 + 16; }

// One 256-element Q8_K block held by one wave (lane l owns elements 4l..4l+3) -> image parts.  Same results as q8k_block_from_regs
// (common.hpp; reference quantize_row_q8_K_ref): the first element with the largest |x| is found with a DPP max, a ballot and a
// v_readlane instead of an 18-step shuffle tournament.
static __device__ __forceinline__ void q8k_block_fast(const f32x4 v, int lane, int8_t * qs, int16_t * bs32, int16_t * b16, float * ds) {
    const float a0 = fabsf(v[0]), a1 = fabsf(v[1]), a2 = fabsf(v[2]), a3 = fabsf(v[3]);
    const float a  = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
    const float amax = wave_max_pos(a);
    if (amax == 0.0f) {                                                  // all-zero block (wave-uniform branch)
        *(uint32_t *) (qs + 4 * lane) = 0u;
        if ((lane & 7) == 0) bs32[lane >> 3] = 0;
        if ((lane & 3) == 0) b16[lane >> 2] = 0;
        if (lane == 0) *ds = 0.0f;
        return;
    }
    const float mc = a0 == a ? v[0] : (a1 == a ? v[1] : (a2 == a ? v[2] : v[3]));       // first element of this lane with the lane's largest |x|
    const unsigned long long bal = __ballot(a == amax);
    const int first = __builtin_ctzll(bal);                              // lowest lane holding the block's largest |x|
    const float mval = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mc), first));
    const float iscale = -127.0f / mval;
    int q[4]; int s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float p = iscale * v[i];
        int r = (int) __builtin_rintf(p);                                // round-half-even == nearest_int()
        r = r > 127 ? 127 : r;
        q[i] = r; s += r;
    }
    *(uint32_t *) (qs + 4 * lane) = (uint32_t) (q[0] & 0xff) | ((uint32_t) (q[1] & 0xff) << 8) | ((uint32_t) (q[2] & 0xff) << 16) | ((uint32_t) (q[3] & 0xff) << 24);
    s += dpp_i32<0xB1, 0xf>(s);
    s += dpp_i32<0x4E, 0xf>(s);                                          // sum of 16 (the reference's bsums entry) in every lane of the quad
    if ((lane & 3) == 0) b16[mv1_b16_pos(lane >> 2)] = (int16_t) s;
    s += dpp_i32<0x141, 0xf>(s);                                         // row_half_mirror: + the neighbouring quad -> sum of 32
    if ((lane & 7) == 0) bs32[lane >> 3] = (int16_t) s;
    if (lane == 0) *ds = 1.0f / iscale;
}

// ------------------------------------------------------------------------------------------------ activation prologue
// Build the Q8_K image of the activation row in LDS (layout: common.hpp q8k_image_bytes).  NW waves; wave w owns blocks w, w + NW, ...
//   img != null : copy a ready-made image
//   nw  != null : y = (x * (1 / sqrtf(mean(x^2) + eps))) * nw, image of y          (RMS_NORM + MUL + from_float)
//   else        : image of x                                                      (from_float only)
// Two halves, because a wave's memory operations return IN ORDER (s_waitcnt vmcnt counts oldest-first): mv1_act_issue requests the
// wave's share of the row (and of the norm weights) BEFORE the first weight stage is requested, so the row arrives after one memory
// latency instead of behind the wave's own 8 KB of weight loads; mv1_act_finish reduces / quantises while the weights are in flight.
// XB = blocks per wave held in registers: the launcher guarantees K / 256 <= XB * NW.
struct mv1_src { const float * x; const float * nw; float eps; const char * img; };
template <int XB> struct mv1_act_regs { f32x4 x[XB], w[XB]; };

template <int NW, int XB>
static __device__ __forceinline__ void mv1_act_issue(const mv1_src s, int K, mv1_act_regs<XB> & r) {
    if (s.img) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // exact-bounds buffer descriptors instead of `block < nb ? load : 0`: a conditional load costs a branch AND a full s_waitcnt before the
    // next request, i.e. one serial memory round trip per condition in front of the weight stream; out-of-range elements read as zero
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *) s.x, (short) 0, K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *) s.nw, (short) 0, s.nw ? K * 4 : 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < XB; ++c) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, (uint32_t) ((wave + c * NW) * 1024 + 16 * lane), 0, 0);
        r.x[c] = f32x4{ __uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]) };
    }
#pragma unroll
    for (int c = 0; c < XB; ++c) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wr, (uint32_t) ((wave + c * NW) * 1024 + 16 * lane), 0, 0);
        r.w[c] = f32x4{ __uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]) };
    }
}
// called by every wave of the workgroup; contains workgroup barriers; the image is complete when it returns
template <int NW, int XB>
static __device__ __forceinline__ void mv1_act_finish(const mv1_src s, int K, const mv1_act_regs<XB> & r, char * im, double * red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nb = K >> 8;
    if (s.img) {                                        // common.hpp layout [qs : K][bsums : K/16 int16][d : K/256 f32] -> this file's layout
        for (int i = threadIdx.x; i < nb * 16; i += 64 * NW) *(u32x4 *) (im + (i >> 4) * 272 + (i & 15) * 16) = ((const u32x4 *) s.img)[i];
        for (int i = threadIdx.x; i < nb * 8; i += 64 * NW) {
            const uint32_t p = *(const uint32_t *) (s.img + K + i * 4);
            *(int16_t *) (im + mv1_img_bs(nb) + i * 2) = (int16_t) ((int) (int16_t) (p & 0xffff) + (int) (int16_t) (p >> 16));
            const int b = i >> 3, s0 = (i & 7) * 2;
            *(int16_t *) (im + mv1_img_b16(nb) + b * 32 + mv1_b16_pos(s0) * 2)     = (int16_t) (p & 0xffff);
            *(int16_t *) (im + mv1_img_b16(nb) + b * 32 + mv1_b16_pos(s0 + 1) * 2) = (int16_t) (p >> 16);
        }
        for (int i = threadIdx.x; i < nb; i += 64 * NW) *(float *) (im + mv1_img_d(nb) + i * 4) = *(const float *) (s.img + K + (K >> 3) + i * 4);
        __syncthreads();
        return;
    }
    float scale = 1.0f;
    if (s.nw) {
        double ss = 0.0;
#pragma unroll
        for (int c = 0; c < XB; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) ss += (double) (r.x[c][i] * r.x[c][i]);
        ss = wave_sum_f64(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[w];
        const float mean = (float) (tot / (double) K);
        scale = 1.0f / sqrtf(mean + s.eps);
    }
#pragma unroll
    for (int c = 0; c < XB; ++c) {
        const int ib = wave + c * NW;
        if (ib < nb) {
            f32x4 y = r.x[c];
            if (s.nw) {
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = (r.x[c][i] * scale) * r.w[c][i];
            }
            q8k_block_fast(y, lane, (int8_t *) im + ib * 272, (int16_t *) (im + mv1_img_bs(nb)) + ib * 8, (int16_t *) (im + mv1_img_b16(nb)) + ib * 16, (float *) (im + mv1_img_d(nb)) + ib);
        }
    }
    __syncthreads();
}

// 24-bit integer multiply / multiply-add: with both operands visibly sign-extended from 24 bits hipcc selects v_mul_i32_i24 / v_mad_i32_i24
// (full rate; v_mul_lo_u32 / v_mad_u64_u32 are quarter rate) and drops the extension itself.  Operands must fit 24 bits signed.
static __device__ __forceinline__ int sx24(int a) { return (a << 8) >> 8; }
static __device__ __forceinline__ int mul24(int a, int b) { return sx24(a) * sx24(b); }
static __device__ __forceinline__ int mad24(int a, int b, int c) { return sx24(a) * sx24(b) + c; }

static __device__ __forceinline__ float mv1_silu(float x) { return x / (1.0f + expf(-x)); }    // ggml_silu_f32, vec.h:958

// ------------------------------------------------------------------------------------------------ matrices of a launch
struct mv1_mat { const char * W; size_t w_rs; char * dst; const char * resid; int nrows; int type; int wave_end; };   // waves [prev.wave_end, wave_end)
struct mv1_dev { mv1_mat m[3]; int nmat; const char * W1; mv1_src src; int K; };

// buffer descriptor of a whole matrix (raw buffer, byte-addressed, bounds = the matrix; built from wave-uniform values only)
typedef __amdgpu_buffer_rsrc_t mv1_rsrc;
#define MV1_KILL 0xF0000000u                            // lane offset that no TAIL matrix's descriptor covers
static __device__ __forceinline__ mv1_rsrc mv1_make_rsrc(const char * p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) p, (short) 0, (int) (bytes > 0xfffffffful ? 0xffffffffu : (uint32_t) bytes), 0x00020000);
}

// row groups [g0, g1) of one matrix for wave lw of nw: contiguous ranges (a wave streams one contiguous piece of the matrix)
static __device__ __forceinline__ void mv1_range(int ngrp, int lw, int nw, int & g0, int & g1) {
    const int q = ngrp / nw, r = ngrp % nw;
    g0 = lw * q + (lw < r ? lw : r);
    g1 = g0 + q + (lw < r ? 1 : 0);
}

} // namespace mi

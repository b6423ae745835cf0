// mmvq.hip -- weight-streaming mat-vec kernels for Q8_0 / F16 / F32 weights (gfx950, wave64); the K-quants live in mmvk.hip.
//
// Computes what the reference CPU backend computes in ggml_compute_forward_mul_mat
// (ggml-cpu/ggml-cpu.c:1210-1402) for ne11 <= 8: every output is one `vec_dot` of a quantised weight
// row with the activation row quantised to the weight type's vec_dot_type:
//     Q4_K x Q8_K : ggml_vec_dot_q4_K_q8_K  (ggml-cpu/quants.c:550-623)
//     Q6_K x Q8_K : ggml_vec_dot_q6_K_q8_K  (ggml-cpu/quants.c:705-758)
//     Q8_0 x Q8_0 : ggml_vec_dot_q8_0_q8_0  (ggml-cpu/quants.c:305-333)
//     F16  x F16  : ggml_vec_dot_f16        (ggml-cpu/vec.cpp)
// Integer sub-block sums are exact and identical to the oracle; only the order of the f32 additions
// across blocks differs (64-lane tree instead of an 8-lane SIMD accumulator).
//
// Design (HBM-bound, 8 TB/s): the weight matrix is read exactly once with 16-B non-temporal loads
// straight into VGPRs (no LDS round trip for single-use data); the small activation image is staged once
// per workgroup in LDS; the next step's weight loads are issued before the current step is consumed so
// every wave keeps >= 2*ROWS KiB in flight; 4-way int8 dot products (v_dot4_i32_i8); wave64 butterfly at
// the end of each row group.  No bounds branches around loads: addresses are clamped and the contribution
// of out-of-range lanes is zeroed, so the compiler keeps all loads of a step in one clause.
#include "../kernels.hpp"

namespace mi {

extern __shared__ __attribute__((aligned(16))) char mmv_lds[];

// stage `ncols` activation images (each `bytes`, multiple of 16) into LDS
static __device__ __forceinline__ void stage_act(const char * act, size_t act_cs, int ncols, size_t bytes) {
    const int n16 = (int) (bytes >> 4);
    for (int c = 0; c < ncols; ++c) {
        const u32x4 * s = (const u32x4 *) (act + c * act_cs);
        u32x4 *       d = (u32x4 *) (mmv_lds + c * bytes);
        for (int i = threadIdx.x; i < n16; i += blockDim.x) d[i] = s[i];
    }
}

// =================================================================================================
// Q8_0 : 34-B block {f16 d, int8 qs[32]} (ggml-common.h:219-224).  8 lanes per block (4 bytes each);
// a wave covers 8 blocks (272 B) per step.  Activation image: qs[K] int8 + per-32 f32 scale.
//   sumf += sumi * (d_x * d_y)   (ggml-cpu/quants.c:318-327)
// =================================================================================================
template <int NCOLS, int ROWS>
__global__ void __launch_bounds__(256) k_mmv_q80(const char * __restrict__ W, size_t w_rs, const char * __restrict__ act, size_t act_cs,
                                                char * __restrict__ dst, size_t dst_cs, int K, int nrows) {
    // four lanes per 34-byte block (8 quants = one hardware-unaligned 8-byte load each; blocks are only 2-byte aligned), 16 blocks per
    // wave step, U steps per stage; the loads of the next stage are issued before the current one is consumed, the first ones before
    // the activation images are staged
    typedef u32x2 __attribute__((aligned(2))) u32x2a2;
    constexpr int U = NCOLS <= 2 ? 4 : 2;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 2, lp = lane & 3;
    const int nb  = K >> 5;
    const int nit = (nb + 16 * U - 1) / (16 * U);
    const size_t img = q80_image_bytes(K);
    const int wave   = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int ngrp   = (nrows + ROWS - 1) / ROWS;

    u32x2 q[U][ROWS]; uint32_t dw[U][ROWS];
    auto issue = [&](int grp, int it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int ib = (it * U + u) * 16 + g; ib = ib < nb ? ib : nb - 1;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                int row = grp * ROWS + r; row = row < nrows ? row : nrows - 1;
                const char * bp = W + (size_t) row * w_rs + (size_t) ib * 34;
                dw[u][r] = *(const uint16_t *) bp;
                q[u][r]  = *(const u32x2a2 *) (bp + 2 + 8 * lp);
            }
        }
    };
    int grp = wave, it = 0;
    if (grp < ngrp) issue(grp, 0);
    stage_act(act, act_cs, NCOLS, img);
    __syncthreads();
    if (grp >= ngrp) return;

    float acc[ROWS][NCOLS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) acc[r][c] = 0.0f;
    while (true) {
        u32x2 cq[U][ROWS]; uint32_t cd[U][ROWS];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { cq[u][r] = q[u][r]; cd[u][r] = dw[u][r]; }
        const int cgrp = grp, cit = it;
        ++it;
        if (it == nit) { it = 0; grp += nwaves; }
        const bool more = grp < ngrp;
        if (more) issue(grp, it);

#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int  ib    = (cit * U + u) * 16 + g;
            const bool valid = ib < nb;
            const int  ibc   = valid ? ib : nb - 1;
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                const char * im = mmv_lds + c * img;
                const u32x2 a  = *(const u32x2 *) (im + ibc * 32 + 8 * lp);
                const float yd = *(const float *) (im + K + ibc * 4);
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const bool rv = valid && (cgrp * ROWS + r) < nrows;
                    const float t = (float) dot4(cq[u][r][0], a[0], dot4(cq[u][r][1], a[1], 0)) * (h2f((uint16_t) cd[u][r]) * yd);
                    acc[r][c] += rv ? t : 0.0f;
                }
            }
        }
        if (cit == nit - 1) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int row = cgrp * ROWS + r;
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
                    const float s = wave_sum(acc[r][c]);
                    if (lane == 0 && row < nrows) *(float *) (dst + c * dst_cs + (size_t) row * 4) = s;
                    acc[r][c] = 0.0f;
                }
            }
        }
        if (!more) break;
    }
}

// =================================================================================================
// Q4_0 / Q5_0 weights x Q8_0 activations.  reference: ggml_vec_dot_q4_0_q8_0 / _q5_0_q8_0 (ggml-cpu/quants.c:115-149, 219-262):
//   sumi = sum_j ((x.qs[j] & 0xF) [| fifth bit] - OFF) * y.qs[j] + ((x.qs[j] >> 4) [| fifth bit] - OFF) * y.qs[j + 16],  OFF = 8 / 16
//   sumf += sumi * d_x * d_y
// Four lanes per 32-weight block (4 bytes of nibbles = weights 4l..4l+3 and 16+4l..19+4l each), 16 blocks per wave step; the offset is
// taken out of the dot products (sum q*y - OFF * sum y), all in exact integers.
// =================================================================================================
template <int NCOLS, int ROWS, bool Q5>
__global__ void __launch_bounds__(256) k_mmv_q40(const char * __restrict__ W, size_t w_rs, const char * __restrict__ act, size_t act_cs,
                                                char * __restrict__ dst, size_t dst_cs, int K, int nrows) {
    // two lanes per block: lane half hf owns nibble bytes 8hf .. 8hf+7 = weights 8hf..8hf+7 (low nibbles) and 16+8hf..23+8hf (high);
    // 32 blocks per wave step, U steps per stage, next stage requested before the current one is consumed
    typedef u32x2 __attribute__((aligned(2))) u32x2a2;
    typedef uint32_t __attribute__((aligned(2))) u32a2;
    constexpr int BS = Q5 ? 22 : 18, QOFF = Q5 ? 6 : 2, OFF = Q5 ? 16 : 8;
    constexpr int U = NCOLS <= 2 ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 1, hf = lane & 1;
    const int nb  = K >> 5;
    const int nit = (nb + 32 * U - 1) / (32 * U);
    const size_t img = q80_image_bytes(K);
    const int wave   = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int ngrp   = (nrows + ROWS - 1) / ROWS;

    u32x2 q[U][ROWS]; uint32_t dw[U][ROWS], qh[Q5 ? U : 1][Q5 ? ROWS : 1];
    auto issue = [&](int grp, int it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int ib = (it * U + u) * 32 + g; ib = ib < nb ? ib : nb - 1;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                int row = grp * ROWS + r; row = row < nrows ? row : nrows - 1;
                const char * bp = W + (size_t) row * w_rs + (size_t) ib * BS;
                dw[u][r] = *(const uint16_t *) bp;
                q[u][r]  = *(const u32x2a2 *) (bp + QOFF + 8 * hf);
                if (Q5) qh[u][r] = *(const u32a2 *) (bp + 2);
            }
        }
    };
    // fifth bits of 4 consecutive weights (bits b .. b+3 of qh) spread into bit 4 of the four bytes of a word
    auto spread = [](uint32_t bits) { return ((bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14) | ((bits & 8u) << 21)) << 4; };
    int grp = wave, it = 0;
    if (grp < ngrp) issue(grp, 0);
    stage_act(act, act_cs, NCOLS, img);
    __syncthreads();
    if (grp >= ngrp) return;

    float acc[ROWS][NCOLS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) acc[r][c] = 0.0f;
    while (true) {
        u32x2 cq[U][ROWS]; uint32_t cd[U][ROWS], ch[Q5 ? U : 1][Q5 ? ROWS : 1];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { cq[u][r] = q[u][r]; cd[u][r] = dw[u][r]; if (Q5) ch[u][r] = qh[u][r]; }
        const int cgrp = grp, cit = it;
        ++it;
        if (it == nit) { it = 0; grp += nwaves; }
        const bool more = grp < ngrp;
        if (more) issue(grp, it);

#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int  ib    = (cit * U + u) * 32 + g;
            const bool valid = ib < nb;
            const int  ibc   = valid ? ib : nb - 1;
            uint32_t lo[ROWS][2], hi[ROWS][2]; float dx[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                dx[r] = h2f((uint16_t) cd[u][r]);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    lo[r][k] = cq[u][r][k] & 0x0f0f0f0fu; hi[r][k] = (cq[u][r][k] >> 4) & 0x0f0f0f0fu;
                    if (Q5) { lo[r][k] |= spread((ch[u][r] >> (8 * hf + 4 * k)) & 0xfu); hi[r][k] |= spread((ch[u][r] >> (16 + 8 * hf + 4 * k)) & 0xfu); }
                }
            }
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                const char * im = mmv_lds + c * img;
                const u32x2 a0 = *(const u32x2 *) (im + ibc * 32 + 8 * hf);
                const u32x2 a1 = *(const u32x2 *) (im + ibc * 32 + 16 + 8 * hf);
                const float yd = *(const float *) (im + K + ibc * 4);
                const int ysum = dot4(0x01010101u, a0[0], dot4(0x01010101u, a0[1], dot4(0x01010101u, a1[0], dot4(0x01010101u, a1[1], 0))));
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const bool rv = valid && (cgrp * ROWS + r) < nrows;
                    const int isum = dot4(lo[r][0], a0[0], dot4(lo[r][1], a0[1], dot4(hi[r][0], a1[0], dot4(hi[r][1], a1[1], 0)))) - OFF * ysum;
                    const float t = (float) isum * (dx[r] * yd);
                    acc[r][c] += rv ? t : 0.0f;
                }
            }
        }
        if (cit == nit - 1) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int row = cgrp * ROWS + r;
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
                    const float s = wave_sum(acc[r][c]);
                    if (lane == 0 && row < nrows) *(float *) (dst + c * dst_cs + (size_t) row * 4) = s;
                    acc[r][c] = 0.0f;
                }
            }
        }
        if (!more) break;
    }
}

// =================================================================================================
// F16 / F32 weights: each lane consumes 16 B (8 halfs / 4 floats) per step; activations (f16 rows for F16
// weights, as the reference rounds src1 to the F16 vec_dot_type; f32 rows for F32 weights) live in LDS.
// =================================================================================================
// blockIdx.y walks the broadcast batch (attention without FLASH_ATTN_EXT: one K / V^T matrix per KV head, one activation per query
// head): batch b = i13 * ne12 + i12 reads W + (i12 / r2) * w_nb2 + (i13 / r3) * w_nb3 -- one launch instead of one per head.
struct mmv_batch { int ne12, r2, r3; size_t w_nb2, w_nb3, act_bs, dst_nb2, dst_nb3; };

template <int NCOLS, int ROWS, bool WF16>
__global__ void __launch_bounds__(256) k_mmv_f(const char * __restrict__ W, size_t w_rs, const char * __restrict__ act, size_t act_cs,
                                              char * __restrict__ dst, size_t dst_cs, int K, int nrows, const mmv_batch bt) {
    {
        const int b = blockIdx.y, i12 = b % bt.ne12, i13 = b / bt.ne12;
        W   += (size_t) (i12 / bt.r2) * bt.w_nb2 + (size_t) (i13 / bt.r3) * bt.w_nb3;
        act += (size_t) b * bt.act_bs;
        dst += (size_t) i12 * bt.dst_nb2 + (size_t) i13 * bt.dst_nb3;
    }
    constexpr int EPL = WF16 ? 8 : 4;                 // elements per lane per step
    const int lane = threadIdx.x & 63;
    const int nstep = (K + 64 * EPL - 1) / (64 * EPL);
    const size_t arow = ((size_t) K * (WF16 ? 2 : 4) + 15) & ~(size_t) 15;
    const int wave   = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int ngrp   = (nrows + ROWS - 1) / ROWS;

    // stage activations (row bytes may be unaligned to 16: copy by 2/4-byte elements)
    for (int c = 0; c < NCOLS; ++c) {
        if (WF16) {
            const uint16_t * s = (const uint16_t *) (act + c * act_cs); uint16_t * d = (uint16_t *) (mmv_lds + c * arow);
            for (int i = threadIdx.x; i < K; i += blockDim.x) d[i] = s[i];
        } else {
            const float * s = (const float *) (act + c * act_cs); float * d = (float *) (mmv_lds + c * arow);
            for (int i = threadIdx.x; i < K; i += blockDim.x) d[i] = s[i];
        }
    }
    __syncthreads();

    const bool vec_ok = (K % EPL == 0) && (w_rs % 16 == 0) && (((uintptr_t) W & 15) == 0);
    if (vec_ok && K % (64 * EPL) == 0) {
        // whole 16-byte steps only (every model matrix): U steps per stage, the next stage's loads issued before the current one is
        // multiplied, activations read from LDS as one 16-byte vector per step.  Same multiply-add order as the general loop below.
        constexpr int U = NCOLS <= 2 ? 4 : 2;
        const int nstage = (nstep + U - 1) / U;
        u32x4 v[U][ROWS];
        auto issue = [&](int grp, int st) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int sidx = st * U + u; sidx = sidx < nstep ? sidx : nstep - 1;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    int row = grp * ROWS + r; row = row < nrows ? row : nrows - 1;
                    v[u][r] = ld_nt16(W + (size_t) row * w_rs + ((size_t) sidx * 64 + lane) * 16);
                }
            }
        };
        int grp = wave, st = 0;
        if (grp >= ngrp) return;
        issue(grp, 0);
        float acc[ROWS][NCOLS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) acc[r][c] = 0.0f;
        while (true) {
            u32x4 cv[U][ROWS];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < ROWS; ++r) cv[u][r] = v[u][r];
            const int cgrp = grp, cst = st;
            ++st;
            if (st == nstage) { st = 0; grp += nwaves; }
            const bool more = grp < ngrp;
            if (more) issue(grp, st);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sidx = cst * U + u;
                if (sidx >= nstep) continue;
                float w[ROWS][EPL];
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    if (WF16) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { w[r][2 * k] = h2f((uint16_t) (cv[u][r][k] & 0xffff)); w[r][2 * k + 1] = h2f((uint16_t) (cv[u][r][k] >> 16)); }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const uint32_t bits = cv[u][r][k]; w[r][k] = __uint_as_float(bits); }
                    }
                }
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
                    const u32x4 xv = *(const u32x4 *) (mmv_lds + c * arow + ((size_t) sidx * 64 + lane) * 16);
                    float x[EPL];
                    if (WF16) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { x[2 * k] = h2f((uint16_t) (xv[k] & 0xffff)); x[2 * k + 1] = h2f((uint16_t) (xv[k] >> 16)); }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const uint32_t bits = xv[k]; x[k] = __uint_as_float(bits); }
                    }
#pragma unroll
                    for (int r = 0; r < ROWS; ++r)
#pragma unroll
                        for (int k = 0; k < EPL; ++k) acc[r][c] = fmaf(w[r][k], x[k], acc[r][c]);
                }
            }
            if (cst == nstage - 1) {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int row = cgrp * ROWS + r;
#pragma unroll
                    for (int c = 0; c < NCOLS; ++c) {
                        const float sum = wave_sum(acc[r][c]);
                        if (lane == 0 && row < nrows) *(float *) (dst + c * dst_cs + (size_t) row * 4) = sum;
                        acc[r][c] = 0.0f;
                    }
                }
            }
            if (!more) break;
        }
        return;
    }
    for (int grp = wave; grp < ngrp; grp += nwaves) {
        float acc[ROWS][NCOLS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) acc[r][c] = 0.0f;
        for (int s = 0; s < nstep; ++s) {
            const int e0 = (s * 64 + lane) * EPL;
            float w[ROWS][EPL];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                int row = grp * ROWS + r; row = row < nrows ? row : nrows - 1;
                const char * rp = W + (size_t) row * w_rs;
                if (vec_ok && e0 + EPL <= K) {
                    const u32x4 v = ld_nt16(rp + (size_t) e0 * (WF16 ? 2 : 4));
                    if (WF16) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { w[r][2 * k] = h2f((uint16_t) (v[k] & 0xffff)); w[r][2 * k + 1] = h2f((uint16_t) (v[k] >> 16)); }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const uint32_t bits = v[k]; w[r][k] = __uint_as_float(bits); }   // (bit_cast on a vector element mis-selects element 0)
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < EPL; ++k) {
                        const int e = e0 + k;
                        w[r][k] = e < K ? (WF16 ? h2f(((const uint16_t *) rp)[e]) : ((const float *) rp)[e]) : 0.0f;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                float x[EPL];
#pragma unroll
                for (int k = 0; k < EPL; ++k) {
                    const int e = e0 + k;
                    x[k] = e < K ? (WF16 ? h2f(((const uint16_t *) (mmv_lds + c * arow))[e]) : ((const float *) (mmv_lds + c * arow))[e]) : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
#pragma unroll
                    for (int k = 0; k < EPL; ++k) acc[r][c] = fmaf(w[r][k], x[k], acc[r][c]);
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int row = grp * ROWS + r;
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                const float s = wave_sum(acc[r][c]);
                if (lane == 0 && row < nrows) *(float *) (dst + c * dst_cs + (size_t) row * 4) = s;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ launch
static int grid_for(int64_t nrows, int rows_per_wave) {
    const int64_t ngrp = (nrows + rows_per_wave - 1) / rows_per_wave;
    int64_t g = (ngrp + 3) / 4;
    const int64_t cap = 256 * 8;          // 256 CUs x up to 8 resident workgroups
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int) g;
}

// LDS budget: a workgroup may own up to 160 KiB; split the columns when the images would not fit.
static const size_t MMV_LDS_MAX = 152 * 1024;

template <typename F>
static void split_cols(const mmv_args & a, size_t bytes_per_col, F && launch) {
    int maxc = (int) (MMV_LDS_MAX / bytes_per_col);
    if (maxc < 1) { fprintf(stderr, "[mi355x] mmv: K=%lld too large for LDS staging\n", (long long) a.K); abort(); }
    if (maxc > MI_MMVQ_MAX_COLS) maxc = MI_MMVQ_MAX_COLS;
    for (int c0 = 0; c0 < a.ncols; c0 += maxc) {
        mmv_args s = a;
        s.ncols = a.ncols - c0 < maxc ? a.ncols - c0 : maxc;
        s.act   = (const char *) a.act + (size_t) c0 * a.act_cs;
        s.dst   = (float *) ((char *) a.dst + (size_t) c0 * a.dst_cs);
        launch(s);
    }
}

typedef void (*mmv_kernel_t)(const char *, size_t, const char *, size_t, char *, size_t, int, int);

static void launch_mmv(mmv_kernel_t k, int rows_per_wave, size_t lds, const mmv_args & a, hipStream_t st) {
    if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    k<<<dim3(grid_for(a.nrows, rows_per_wave)), dim3(256), lds, st>>>((const char *) a.W, a.w_rs, (const char *) a.act, a.act_cs,
                                                                     (char *) a.dst, a.dst_cs, (int) a.K, (int) a.nrows);
}

// tuning knob (decode, ncols == 1): MI355X_MMV_CFG = "<rows><u>" e.g. "22" (default), "12", "21", "41"
static void mmv_kquant_single(int type, const mmv_args & a0, hipStream_t st) {
    if (a0.nrows == 0 || a0.ncols == 0) return;
    split_cols(a0, q8k_image_bytes(a0.K), [&](const mmv_args & a) {
        mmv_multi_args m;
        m.nmat = 1; m.act = a.act; m.act_cs = a.act_cs; m.K = a.K; m.ncols = a.ncols;
        m.m[0] = { a.W, a.w_rs, a.dst, a.dst_cs, nullptr, 0, a.nrows, type };
        mmv_kquant_multi(m, st);
    });
}
void mmv_q4_K(const mmv_args & a, hipStream_t st) { mmv_kquant_single(GGML_TYPE_Q4_K, a, st); }
void mmv_q6_K(const mmv_args & a, hipStream_t st) { mmv_kquant_single(GGML_TYPE_Q6_K, a, st); }
void mmv_q5_K(const mmv_args & a, hipStream_t st) { mmv_kquant_single(GGML_TYPE_Q5_K, a, st); }

void mmv_q8_0(const mmv_args & a0, hipStream_t st) {
    if (a0.nrows == 0 || a0.ncols == 0) return;
    const size_t ib = q80_image_bytes(a0.K);
    split_cols(a0, ib, [&](const mmv_args & a) {
        mmv_kernel_t k = nullptr; int rows = 2;
        switch (a.ncols) {
            case 1: k = k_mmv_q80<1, 2>; break;
            case 2: k = k_mmv_q80<2, 2>; break;
            case 3: k = k_mmv_q80<3, 2>; break;
            case 4: k = k_mmv_q80<4, 2>; break;
            case 5: k = k_mmv_q80<5, 1>; rows = 1; break;
            case 6: k = k_mmv_q80<6, 1>; rows = 1; break;
            case 7: k = k_mmv_q80<7, 1>; rows = 1; break;
            case 8: k = k_mmv_q80<8, 1>; rows = 1; break;
            default: abort();
        }
        launch_mmv(k, rows, ib * a.ncols, a, st);
    });
}

template <bool Q5>
static void mmv_q40_t(const mmv_args & a0, hipStream_t st) {
    if (a0.nrows == 0 || a0.ncols == 0) return;
    const size_t ib = q80_image_bytes(a0.K);
    split_cols(a0, ib, [&](const mmv_args & a) {
        mmv_kernel_t k = nullptr; int rows = 2;
        switch (a.ncols) {
            case 1: k = k_mmv_q40<1, 2, Q5>; break;
            case 2: k = k_mmv_q40<2, 2, Q5>; break;
            case 3: k = k_mmv_q40<3, 2, Q5>; break;
            case 4: k = k_mmv_q40<4, 2, Q5>; break;
            case 5: k = k_mmv_q40<5, 1, Q5>; rows = 1; break;
            case 6: k = k_mmv_q40<6, 1, Q5>; rows = 1; break;
            case 7: k = k_mmv_q40<7, 1, Q5>; rows = 1; break;
            case 8: k = k_mmv_q40<8, 1, Q5>; rows = 1; break;
            default: abort();
        }
        launch_mmv(k, rows, ib * a.ncols, a, st);
    });
}
void mmv_q4_0(const mmv_args & a, hipStream_t st) { mmv_q40_t<false>(a, st); }
void mmv_q5_0(const mmv_args & a, hipStream_t st) { mmv_q40_t<true>(a, st); }

#define MMVF_LAUNCH(NC, ROWS, WF16)                                                                                    \
    do {                                                                                                               \
        const size_t ldsb = arow * (NC);                                                                               \
        if (ldsb > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *) k_mmv_f<NC, ROWS, WF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));                              \
        k_mmv_f<NC, ROWS, WF16><<<dim3(grid_for(a.nrows, ROWS), (unsigned) a.nbatch), dim3(256), ldsb, st>>>(          \
            (const char *) a.W, a.w_rs, (const char *) a.act, a.act_cs, (char *) a.dst, a.dst_cs, (int) a.K, (int) a.nrows, bt); \
    } while (0)

template <bool WF16>
static void mmv_float(const mmv_args & a0, hipStream_t st) {
    if (a0.nrows == 0 || a0.ncols == 0) return;
    const size_t arow = ((size_t) a0.K * (WF16 ? 2 : 4) + 15) & ~(size_t) 15;
    const mmv_batch bt = { a0.nbatch > 1 ? a0.ne12 : 1, a0.nbatch > 1 ? a0.r2 : 1, a0.nbatch > 1 ? a0.r3 : 1, a0.w_nb2, a0.w_nb3, a0.act_bs, a0.dst_nb2, a0.dst_nb3 };
    split_cols(a0, arow, [&](const mmv_args & a) {
        switch (a.ncols) {
            case 1: MMVF_LAUNCH(1, 2, WF16); break;
            case 2: MMVF_LAUNCH(2, 2, WF16); break;
            case 3: MMVF_LAUNCH(3, 1, WF16); break;
            case 4: MMVF_LAUNCH(4, 1, WF16); break;
            case 5: MMVF_LAUNCH(5, 1, WF16); break;
            case 6: MMVF_LAUNCH(6, 1, WF16); break;
            case 7: MMVF_LAUNCH(7, 1, WF16); break;
            case 8: MMVF_LAUNCH(8, 1, WF16); break;
            default: abort();
        }
    });
}
void mmv_f16(const mmv_args & a, hipStream_t st) { mmv_float<true>(a, st); }
void mmv_f32(const mmv_args & a, hipStream_t st) { mmv_float<false>(a, st); }

} // namespace mi

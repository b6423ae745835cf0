// mmvq.hip -- weight-streaming mat-vec kernels (decode hot loop) for gfx950 / wave64.
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
// Q4_K : 144-B super-block = 16-B header {d, dmin, 12 B of 6-bit scales/mins} + 128 B of nibbles
// (ggml-common.h:295-305).  8 lanes per super-block: lane (j = lp>>1, h = lp&1) owns qs[32j+16h .. +16),
// i.e. 16 low nibbles of sub-block 2j and 16 high nibbles of sub-block 2j+1 (dequantize_row_q4_K,
// ggml-quants.c:1352-1374).  A wave covers 8 super-blocks (1152 contiguous bytes) per step.
// =================================================================================================
template <int NCOLS, int ROWS, int U>
__global__ void __launch_bounds__(256) k_mmv_q4k(const char * __restrict__ W, size_t w_rs, const char * __restrict__ act, size_t act_cs,
                                                char * __restrict__ dst, size_t dst_cs, int K, int nrows) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 3, lp = lane & 7, j = lp >> 1, h = lp & 1;
    const int nb  = K >> 8;
    const int nit = (nb + 8 * U - 1) / (8 * U);          // pipeline stages per row group (U steps of 8 super-blocks each)
    const size_t img = q8k_image_bytes(K);

    const int wave   = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int ngrp   = (nrows + ROWS - 1) / ROWS;

    // ---- software pipeline state: loads for stage (grp, it)
    u32x4 hdr[U][ROWS], qs[U][ROWS];
    auto issue = [&](int grp, int it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int ib = (it * U + u) * 8 + g; ib = ib < nb ? ib : nb - 1;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                int row = grp * ROWS + r; row = row < nrows ? row : nrows - 1;
                const char * bp = W + (size_t) row * w_rs + (size_t) ib * 144;
                hdr[u][r] = ld_nt16(bp);
                qs[u][r]  = ld_nt16(bp + 16 + lp * 16);
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

    const int sh = (j & 1) * 16;
    while (true) {
        // take ownership of the loaded stage, then immediately issue the next one
        u32x4 chdr[U][ROWS], cqs[U][ROWS];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { chdr[u][r] = hdr[u][r]; cqs[u][r] = qs[u][r]; }
        const int cgrp = grp, cit = it;
        ++it;
        if (it == nit) { it = 0; grp += nwaves; }
        const bool more = grp < ngrp;
        if (more) issue(grp, it);

#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int  ib    = (cit * U + u) * 8 + g;
            const bool valid = ib < nb;
            const int  ibc   = valid ? ib : nb - 1;

            // activation pieces for this lane (LDS)
            u32x4 alo[NCOLS], ahi[NCOLS]; int bs0[NCOLS], bs1[NCOLS]; float yd[NCOLS];
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                const char * im = mmv_lds + c * img;
                alo[c] = *(const u32x4 *) (im + ibc * 256 + 64 * j + 16 * h);
                ahi[c] = *(const u32x4 *) (im + ibc * 256 + 64 * j + 16 * h + 32);
                const u32x2 b = *(const u32x2 *) (im + K + (ibc * 16 + 4 * j) * 2);       // bsums[4j .. 4j+3]
                bs0[c] = (int) (int16_t) (h ? (b[0] >> 16) : b[0]);                       // bsums[4j + h]
                bs1[c] = (int) (int16_t) (h ? (b[1] >> 16) : b[1]);                       // bsums[4j + 2 + h]
                yd[c]  = *(const float *) (im + K + (K >> 3) + ibc * 4);
            }

#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const uint32_t u0 = chdr[u][r][1], u1 = chdr[u][r][2], u2 = chdr[u][r][3];
                // 6-bit scale/min unpack, same bit surgery as ggml-cpu/quants.c:592-597 / get_scale_min_k4
                const uint32_t s_lo = u0 & 0x3f3f3f3fu;
                const uint32_t s_hi = (u2 & 0x0f0f0f0fu) | (((u0 >> 6) & 0x03030303u) << 4);
                const uint32_t m_lo = u1 & 0x3f3f3f3fu;
                const uint32_t m_hi = ((u2 >> 4) & 0x0f0f0f0fu) | (((u1 >> 6) & 0x03030303u) << 4);
                const uint32_t sw = (j < 2 ? s_lo : s_hi) >> sh;
                const uint32_t mw = (j < 2 ? m_lo : m_hi) >> sh;
                const int sc0 = sw & 0xff, sc1 = (sw >> 8) & 0xff;
                const int mn0 = mw & 0xff, mn1 = (mw >> 8) & 0xff;
                const float dx   = h2f((uint16_t) (chdr[u][r][0] & 0xffff));
                const float dmin = h2f((uint16_t) (chdr[u][r][0] >> 16));

                uint32_t lo[4], hi[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { lo[k] = cqs[u][r][k] & 0x0f0f0f0fu; hi[k] = (cqs[u][r][k] >> 4) & 0x0f0f0f0fu; }

                const bool rv = valid && (cgrp * ROWS + r) < nrows;
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
                    int dl = 0, dh = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { dl = dot4(lo[k], alo[c][k], dl); dh = dot4(hi[k], ahi[c][k], dh); }
                    const int isum = sc0 * dl + sc1 * dh;
                    const int msum = mn0 * bs0[c] + mn1 * bs1[c];
                    const float t = (dx * yd[c]) * (float) isum - (dmin * yd[c]) * (float) msum;
                    acc[r][c] += rv ? t : 0.0f;
                }
            }
        }

        if (cit == nit - 1) {             // row group finished: butterfly and store
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
// Q6_K : 210-B super-block {ql[128], qh[64], int8 scales[16], f16 d} (ggml-common.h:330-335); only 2-byte
// aligned, so pieces are fetched with (hardware-)unaligned 8-B loads.  8 lanes per super-block:
// lane (n = lp>>2, tp = lp&3) owns l in [8tp, 8tp+8) of the 128-half n:
//   ql[64n+l], ql[64n+32+l], qh[32n+l]  ->  4 x 8 six-bit weights at y[128n + {0,32,64,96} + l]
// (dequantize_row_q6_K, ggml-quants.c:1762-1791).  A wave covers 8 super-blocks (1680 B) per step.
// =================================================================================================
static __device__ __forceinline__ u32x2 ld_u8x8(const char * p) {      // 2-B aligned 8-byte load
    typedef uint32_t __attribute__((aligned(2))) u32a2;
    u32x2 v; v[0] = __builtin_nontemporal_load((const u32a2 *) p); v[1] = __builtin_nontemporal_load((const u32a2 *) (p + 4));
    return v;
}
// bytes in [0,63] -> signed bytes (w - 32), SWAR without inter-byte borrow
static __device__ __forceinline__ uint32_t sub32(uint32_t w) { return ((w | 0x80808080u) - 0x20202020u) ^ 0x80808080u; }

template <int NCOLS, int ROWS, int U>
__global__ void __launch_bounds__(256) k_mmv_q6k(const char * __restrict__ W, size_t w_rs, const char * __restrict__ act, size_t act_cs,
                                                char * __restrict__ dst, size_t dst_cs, int K, int nrows) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 3, lp = lane & 7, n = lp >> 2, tp = lp & 3;
    const int nb  = K >> 8;
    const int nit = (nb + 8 * U - 1) / (8 * U);
    const size_t img = q8k_image_bytes(K);

    const int wave   = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int ngrp   = (nrows + ROWS - 1) / ROWS;

    u32x2 qla[U][ROWS], qlb[U][ROWS], qh[U][ROWS], scw[U][ROWS]; uint32_t dw[U][ROWS];
    auto issue = [&](int grp, int it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int ib = (it * U + u) * 8 + g; ib = ib < nb ? ib : nb - 1;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                int row = grp * ROWS + r; row = row < nrows ? row : nrows - 1;
                const char * bp = W + (size_t) row * w_rs + (size_t) ib * 210;
                qla[u][r] = ld_u8x8(bp + 64 * n + 8 * tp);
                qlb[u][r] = ld_u8x8(bp + 64 * n + 32 + 8 * tp);
                qh[u][r]  = ld_u8x8(bp + 128 + 32 * n + 8 * tp);
                scw[u][r] = ld_u8x8(bp + 192 + 8 * n);
                dw[u][r]  = __builtin_nontemporal_load((const uint16_t *) (bp + 208));
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

    const int is = tp >> 1;                 // l/16 for l in [8tp, 8tp+8)
    while (true) {
        u32x2 cqla[U][ROWS], cqlb[U][ROWS], cqh[U][ROWS], cscw[U][ROWS]; uint32_t cdw[U][ROWS];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { cqla[u][r] = qla[u][r]; cqlb[u][r] = qlb[u][r]; cqh[u][r] = qh[u][r]; cscw[u][r] = scw[u][r]; cdw[u][r] = dw[u][r]; }
        const int cgrp = grp, cit = it;
        ++it;
        if (it == nit) { it = 0; grp += nwaves; }
        const bool more = grp < ngrp;
        if (more) issue(grp, it);

#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int  ib    = (cit * U + u) * 8 + g;
            const bool valid = ib < nb;
            const int  ibc   = valid ? ib : nb - 1;

            u32x2 a[NCOLS][4]; float yd[NCOLS];
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                const char * im = mmv_lds + c * img;
#pragma unroll
                for (int k = 0; k < 4; ++k) a[c][k] = *(const u32x2 *) (im + ibc * 256 + 128 * n + 32 * k + 8 * tp);
                yd[c] = *(const float *) (im + K + (K >> 3) + ibc * 4);
            }

#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                // scales[8n + is + {0,2,4,6}] (signed)
                const uint32_t s01 = cscw[u][r][0] >> (8 * is), s23 = cscw[u][r][1] >> (8 * is);
                const int sc0 = (int8_t) (s01 & 0xff), sc1 = (int8_t) ((s01 >> 16) & 0xff);
                const int sc2 = (int8_t) (s23 & 0xff), sc3 = (int8_t) ((s23 >> 16) & 0xff);
                const float dx = h2f((uint16_t) cdw[u][r]);
                uint32_t w[4][2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const uint32_t la = cqla[u][r][e], lb = cqlb[u][r][e], hh = cqh[u][r][e];
                    w[0][e] = sub32((la & 0x0f0f0f0fu)        | ((hh << 4) & 0x30303030u));
                    w[1][e] = sub32((lb & 0x0f0f0f0fu)        | ((hh << 2) & 0x30303030u));
                    w[2][e] = sub32(((la >> 4) & 0x0f0f0f0fu) | (hh & 0x30303030u));
                    w[3][e] = sub32(((lb >> 4) & 0x0f0f0f0fu) | ((hh >> 2) & 0x30303030u));
                }
                const bool rv = valid && (cgrp * ROWS + r) < nrows;
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
                    const int d0 = dot4(w[0][1], a[c][0][1], dot4(w[0][0], a[c][0][0], 0));
                    const int d1 = dot4(w[1][1], a[c][1][1], dot4(w[1][0], a[c][1][0], 0));
                    const int d2 = dot4(w[2][1], a[c][2][1], dot4(w[2][0], a[c][2][0], 0));
                    const int d3 = dot4(w[3][1], a[c][3][1], dot4(w[3][0], a[c][3][0], 0));
                    const int isum = sc0 * d0 + sc1 * d1 + sc2 * d2 + sc3 * d3;
                    const float t = (dx * yd[c]) * (float) isum;
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
// Q8_0 : 34-B block {f16 d, int8 qs[32]} (ggml-common.h:219-224).  8 lanes per block (4 bytes each);
// a wave covers 8 blocks (272 B) per step.  Activation image: qs[K] int8 + per-32 f32 scale.
//   sumf += sumi * (d_x * d_y)   (ggml-cpu/quants.c:318-327)
// =================================================================================================
template <int NCOLS, int ROWS>
__global__ void __launch_bounds__(256) k_mmv_q80(const char * __restrict__ W, size_t w_rs, const char * __restrict__ act, size_t act_cs,
                                                char * __restrict__ dst, size_t dst_cs, int K, int nrows) {
    typedef uint32_t __attribute__((aligned(2))) u32a2;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 3, lp = lane & 7;
    const int nb  = K >> 5;
    const int nit = (nb + 7) >> 3;
    const size_t img = q80_image_bytes(K);
    const int wave   = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int ngrp   = (nrows + ROWS - 1) / ROWS;

    stage_act(act, act_cs, NCOLS, img);
    __syncthreads();

    for (int grp = wave; grp < ngrp; grp += nwaves) {
        float acc[ROWS][NCOLS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) acc[r][c] = 0.0f;
#pragma unroll 4
        for (int it = 0; it < nit; ++it) {
            const int  ib    = it * 8 + g;
            const bool valid = ib < nb;
            const int  ibc   = valid ? ib : nb - 1;
            uint32_t q[ROWS]; uint32_t dw[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                int row = grp * ROWS + r; row = row < nrows ? row : nrows - 1;
                const char * bp = W + (size_t) row * w_rs + (size_t) ibc * 34;
                dw[r] = *(const uint16_t *) bp;
                q[r]  = __builtin_nontemporal_load((const u32a2 *) (bp + 2 + 4 * lp));
            }
#pragma unroll
            for (int c = 0; c < NCOLS; ++c) {
                const char * im = mmv_lds + c * img;
                const uint32_t a  = *(const uint32_t *) (im + ibc * 32 + 4 * lp);
                const float    yd = *(const float *) (im + K + ibc * 4);
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const bool rv = valid && (grp * ROWS + r) < nrows;
                    const float t = (float) dot4(q[r], a, 0) * (h2f((uint16_t) dw[r]) * yd);
                    acc[r][c] += rv ? t : 0.0f;
                }
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

// =================================================================================================
// F16 / F32 weights: each lane consumes 16 B (8 halfs / 4 floats) per step; activations (f16 rows for F16
// weights, as the reference rounds src1 to the F16 vec_dot_type; f32 rows for F32 weights) live in LDS.
// =================================================================================================
template <int NCOLS, int ROWS, bool WF16>
__global__ void __launch_bounds__(256) k_mmv_f(const char * __restrict__ W, size_t w_rs, const char * __restrict__ act, size_t act_cs,
                                              char * __restrict__ dst, size_t dst_cs, int K, int nrows) {
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
static int mmv_cfg() {
    static int cfg = -1;
    if (cfg < 0) { const char * e = getenv("MI355X_MMV_CFG"); cfg = e ? atoi(e) : 0; }
    return cfg;
}

#define MMV_TABLE(KERNEL)                                                                                              \
    static mmv_kernel_t KERNEL##_pick(int ncols, int nstep, int * rows) {                                              \
        if (ncols == 1) {                                                                                              \
            switch (mmv_cfg()) {                                                                                       \
                case 12: *rows = 1; return KERNEL<1, 1, 2>;                                                            \
                case 14: *rows = 1; return KERNEL<1, 1, 4>;                                                            \
                case 21: *rows = 2; return KERNEL<1, 2, 1>;                                                            \
                case 41: *rows = 4; return KERNEL<1, 4, 1>;                                                            \
                case 42: *rows = 4; return KERNEL<1, 4, 2>;                                                            \
                case 22: *rows = 2; return KERNEL<1, 2, 2>;                                                            \
                default: break;                                                                                        \
            }                                                                                                          \
            *rows = 2;                                                                                                 \
            return nstep >= 2 ? KERNEL<1, 2, 2> : KERNEL<1, 2, 1>;                                                     \
        }                                                                                                              \
        switch (ncols) {                                                                                               \
            case 2: *rows = 2; return KERNEL<2, 2, 1>;                                                                 \
            case 3: *rows = 2; return KERNEL<3, 2, 1>;                                                                 \
            case 4: *rows = 2; return KERNEL<4, 2, 1>;                                                                 \
            case 5: *rows = 1; return KERNEL<5, 1, 1>;                                                                 \
            case 6: *rows = 1; return KERNEL<6, 1, 1>;                                                                 \
            case 7: *rows = 1; return KERNEL<7, 1, 1>;                                                                 \
            case 8: *rows = 1; return KERNEL<8, 1, 1>;                                                                 \
            default: fprintf(stderr, "[mi355x] mmv: ncols=%d out of range\n", ncols); abort();                         \
        }                                                                                                              \
    }
MMV_TABLE(k_mmv_q4k)
MMV_TABLE(k_mmv_q6k)

void mmv_q4_K(const mmv_args & a0, hipStream_t st) {
    if (a0.nrows == 0 || a0.ncols == 0) return;
    const size_t ib = q8k_image_bytes(a0.K);
    split_cols(a0, ib, [&](const mmv_args & a) {
        int rows; mmv_kernel_t k = k_mmv_q4k_pick(a.ncols, (int) ((a.K / 256 + 7) / 8), &rows);
        launch_mmv(k, rows, ib * a.ncols, a, st);
    });
}

void mmv_q6_K(const mmv_args & a0, hipStream_t st) {
    if (a0.nrows == 0 || a0.ncols == 0) return;
    const size_t ib = q8k_image_bytes(a0.K);
    split_cols(a0, ib, [&](const mmv_args & a) {
        int rows; mmv_kernel_t k = k_mmv_q6k_pick(a.ncols, (int) ((a.K / 256 + 7) / 8), &rows);
        launch_mmv(k, rows, ib * a.ncols, a, st);
    });
}

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

#define MMVF_LAUNCH(NC, ROWS, WF16)                                                                                    \
    do {                                                                                                               \
        const size_t ldsb = arow * (NC);                                                                               \
        if (ldsb > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *) k_mmv_f<NC, ROWS, WF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsb));                              \
        k_mmv_f<NC, ROWS, WF16><<<dim3(grid_for(a.nrows, ROWS)), dim3(256), ldsb, st>>>(                               \
            (const char *) a.W, a.w_rs, (const char *) a.act, a.act_cs, (char *) a.dst, a.dst_cs, (int) a.K, (int) a.nrows); \
    } while (0)

template <bool WF16>
static void mmv_float(const mmv_args & a0, hipStream_t st) {
    if (a0.nrows == 0 || a0.ncols == 0) return;
    const size_t arow = ((size_t) a0.K * (WF16 ? 2 : 4) + 15) & ~(size_t) 15;
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

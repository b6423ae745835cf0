// mmvk.hip -- K-quant weight-streaming mat-vec kernels (the decode hot loop) for gfx950 / wave64.
//
// What is computed (reference: ggml_compute_forward_mul_mat, ggml-cpu/ggml-cpu.c:1210-1402, ne11 <= 8):
//     dst[col][row] = vec_dot(W[row, :], Q8_K(act[col, :]))
//     Q4_K x Q8_K : ggml_vec_dot_q4_K_q8_K  (ggml-cpu/quants.c:550-623)
//     Q6_K x Q8_K : ggml_vec_dot_q6_K_q8_K  (ggml-cpu/quants.c:705-758)
// Integer sub-block sums are exact and identical to the oracle; only the order of the f32 additions across
// super-blocks differs (64-lane butterfly instead of an 8-lane SIMD accumulator).
//
// Launch shapes
//   k_mmv_multi : up to 3 matrices that share the activation vector (wq/wk/wv, or a single matrix) in ONE launch;
//                 waves are split between the matrices in proportion to their bytes; Q4_K and Q6_K may be mixed
//                 (Q4_K_M stores attn_v / ffn_down as Q6_K on half of the layers).  Optional epilogue:
//                 dst = W.x + resid (the graph's following residual ADD).
//   k_mmv_pair  : ffn_gate + ffn_up rows processed together, epilogue silu(gate)*up (the graph's GLU node),
//                 so neither intermediate vector is written.
//
// HBM design (bound: 8 TB/s): every weight byte is read exactly once with non-temporal vector loads straight
// into VGPRs (no LDS round trip for single-use data); the small activation image is staged once per workgroup in
// LDS; the next pipeline stage's loads are issued before the current one is consumed; 4-way int8 dot products
// (v_dot4_i32_i8); no bounds branches around loads (addresses are clamped, contributions of out-of-range lanes
// are zeroed) so each stage's loads stay in one clause.
#include "../kernels.hpp"

#ifndef MI_Q4K_NT
#define MI_Q4K_NT 0        // measured: plain loads beat non-temporal (the 16-B header and the nibbles of a block share 128-B lines across two instructions)
#endif
#ifndef MI_Q6K_NT
#define MI_Q6K_NT 0        // Q6_K pieces of one block are spread over 5 instructions: keep the lines in L1 between them
#endif
#ifndef MI_Q6K_X3
#define MI_Q6K_X3 0        // 1: aligned 12-byte loads + v_alignbyte; 0: hardware-unaligned 8-byte loads
#endif

namespace mi {

extern __shared__ __attribute__((aligned(16))) char mmv_lds[];

template <typename T> static __device__ __forceinline__ T ld_w(const T * p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }

static __device__ __forceinline__ void stage_act_k(const char * act, size_t act_cs, int ncols, size_t bytes) {
    const int n16 = (int) (bytes >> 4);
    for (int c = 0; c < ncols; ++c) {
        const u32x4 * s = (const u32x4 *) (act + c * act_cs);
        u32x4 *       d = (u32x4 *) (mmv_lds + c * bytes);
        for (int i = threadIdx.x; i < n16; i += blockDim.x) d[i] = s[i];
    }
}

// Activation source of a launch: ready-made Q8_K images, or (NORM kernels, one column, K <= 4096) the f32 row they are to be
// made from -- then EVERY workgroup builds the image of rms_norm(x) * w itself, in LDS.  That replaces a whole dependent launch
// (k_rms_norm_mul_quant, ~5 us of pure latency per use) by 32 KB of L2 reads per workgroup.  Ordering matters: the row and the
// norm weights are requested FIRST (memory returns in order), then the first stage of weight blocks; the reduction and the
// quantisation then run while those weight loads are in flight.  The arithmetic is the stand-alone kernel's (sum of squares in
// double, (x*scale)*w, q8k_block_from_regs); every workgroup produces the same bits (the order depends on the thread layout only).
struct act_norm { const char * x; size_t x_cs; const float * w; float eps; };
constexpr int NORM_MAXB = 4;                                       // 256-element blocks per wave: K <= 4 waves * 4 * 256
struct norm_regs { f32x4 x[NORM_MAXB], w[NORM_MAXB]; };

static __device__ __forceinline__ void norm_prefetch(const act_norm nr, int K, norm_regs & r) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nb = K >> 8;
    const float * xr = (const float *) nr.x;
#pragma unroll
    for (int b = 0; b < NORM_MAXB; ++b) {
        const int ib = wave + 4 * b;
        if (ib < nb) { r.x[b] = *(const f32x4 *) (xr + ib * 256 + 4 * lane); r.w[b] = *(const f32x4 *) (nr.w + ib * 256 + 4 * lane); }
        else { r.x[b] = f32x4{0, 0, 0, 0}; r.w[b] = f32x4{0, 0, 0, 0}; }
    }
}
static __device__ __forceinline__ void norm_finish(const act_norm nr, int K, const norm_regs & r) {
    __shared__ double nred[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nb = K >> 8;
    double ss = 0.0;
#pragma unroll
    for (int b = 0; b < NORM_MAXB; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) ss += (double) (r.x[b][i] * r.x[b][i]);
    ss = block_sum<double>(ss, nred);
    const float mean  = (float) (ss / (double) K);
    const float scale = 1.0f / sqrtf(mean + nr.eps);
    char * im = mmv_lds;
#pragma unroll
    for (int b = 0; b < NORM_MAXB; ++b) {
        const int ib = wave + 4 * b;
        if (ib >= nb) break;
        f32x4 y;
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = (r.x[b][i] * scale) * r.w[b][i];
        q8k_block_from_regs(y, lane, (int8_t *) im + ib * 256, (int16_t *) (im + K) + ib * 16, (float *) (im + K + (K >> 3)) + ib);
    }
}
bool mmv_norm_ok(int64_t K, int ncols) { return ncols == 1 && K % 256 == 0 && K / 256 <= 4 * NORM_MAXB; }

static __device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }    // ggml_silu_f32, vec.h:958

// epilogue shared by all bodies
struct mmv_out {
    char *       dst;    size_t dst_cs;      // f32 output, column stride
    const char * resid;  size_t resid_cs;    // optional residual (added after the dot product), may be null
};

// =================================================================================================
// Q4_K : 144-B super-block = 16-B header {d, dmin, 12 B of 6-bit scales/mins} + 128 B of nibbles
// (ggml-common.h:295-305).  8 lanes per super-block: lane (j = lp>>1, h = lp&1) owns qs[32j+16h .. +16),
// i.e. 16 low nibbles of sub-block 2j and 16 high nibbles of sub-block 2j+1 (dequantize_row_q4_K,
// ggml-quants.c:1352-1374).  A wave covers 8 super-blocks (1152 contiguous bytes) per step, U steps per stage.
// PAIR: the two "rows" of a group are row r of W0 (gate) and row r of W1 (up); epilogue silu(g)*u.
// =================================================================================================
// Q5: the block is a block_q5_K (ggml-common.h:313-323: d, dmin, scales[12], qh[32], qs[128] = 176 B) -- the same 6-bit scales / mins and
// nibble layout as Q4_K plus one high bit per weight (bit 2j of qh[l] for sub-block 2j, bit 2j+1 for 2j+1: dequantize_row_q5_K,
// ggml-quants.c:1554-1580; ggml_vec_dot_q5_K_q8_K, ggml-cpu/quants.c): the 5-bit values go through the same dot4 against the Q8_K bytes
template <int NCOLS, int ROWS, int U, bool PAIR, bool NORM, bool Q5 = false>
static __device__ __forceinline__ void q4k_body(const char * __restrict__ W0, const char * __restrict__ W1, size_t w_rs, const mmv_out o,
                                                const char * __restrict__ act, size_t act_cs, int K, int nrows, int wave, int nwaves, const act_norm nr) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 3, lp = lane & 7, j = lp >> 1, h = lp & 1;
    const int nb  = K >> 8;
    const int nit = (nb + 8 * U - 1) / (8 * U);
    const size_t img = q8k_image_bytes(K);
    const int ngrp = PAIR ? nrows : (nrows + ROWS - 1) / ROWS;

    constexpr int BS = Q5 ? 176 : 144, QOFF = Q5 ? 48 : 16;
    u32x4 hdr[U][ROWS], qs[U][ROWS], qhb[Q5 ? U : 1][Q5 ? ROWS : 1];
    auto issue = [&](int grp, int it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int ib = (it * U + u) * 8 + g; ib = ib < nb ? ib : nb - 1;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                int row = PAIR ? grp : grp * ROWS + r; row = row < nrows ? row : nrows - 1;
                const char * bp = ((PAIR && r == 1) ? W1 : W0) + (size_t) row * w_rs + (size_t) ib * BS;
                hdr[u][r] = ld_w((const u32x4 *) bp, MI_Q4K_NT);
                qs[u][r]  = ld_w((const u32x4 *) (bp + QOFF + lp * 16), MI_Q4K_NT);
                if (Q5) qhb[u][r] = ld_w((const u32x4 *) (bp + 16 + (lp & 1) * 16), MI_Q4K_NT);      // qh[16h .. 16h+15]
            }
        }
    };

    int grp = wave, it = 0;
    norm_regs nrg;
    if (NORM) norm_prefetch(nr, K, nrg);           // the row to normalise is requested before anything else
    if (grp < ngrp) issue(grp, 0);                 // first weight loads are in flight while the activation image is staged / built
    if (NORM) norm_finish(nr, K, nrg); else stage_act_k(act, act_cs, NCOLS, img);
    __syncthreads();
    if (grp >= ngrp) return;

    float acc[ROWS][NCOLS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) acc[r][c] = 0.0f;

    const int sh = (j & 1) * 16;
    while (true) {
        u32x4 chdr[U][ROWS], cqs[U][ROWS], cqh[Q5 ? U : 1][Q5 ? ROWS : 1];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { chdr[u][r] = hdr[u][r]; cqs[u][r] = qs[u][r]; if (Q5) cqh[u][r] = qhb[u][r]; }
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
                for (int k = 0; k < 4; ++k) {
                    lo[k] = cqs[u][r][k] & 0x0f0f0f0fu; hi[k] = (cqs[u][r][k] >> 4) & 0x0f0f0f0fu;
                    if (Q5) { lo[k] |= ((cqh[u][r][k] >> (2 * j)) & 0x01010101u) << 4; hi[k] |= ((cqh[u][r][k] >> (2 * j + 1)) & 0x01010101u) << 4; }
                }
                const bool rv = valid && (PAIR ? cgrp : cgrp * ROWS + r) < nrows;
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

        if (cit == nit - 1) {             // row group finished: butterfly, epilogue, store
            if (PAIR) {
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
                    const float gsum = wave_sum_f32(acc[0][c]), usum = wave_sum_f32(acc[1][c]);
                    if (lane == 0) *(float *) (o.dst + c * o.dst_cs + (size_t) cgrp * 4) = silu_f(gsum) * usum;
                    acc[0][c] = 0.0f; acc[1][c] = 0.0f;
                }
            } else {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int row = cgrp * ROWS + r;
#pragma unroll
                    for (int c = 0; c < NCOLS; ++c) {
                        float s = wave_sum_f32(acc[r][c]);
                        if (lane == 0 && row < nrows) {
                            if (o.resid) s += *(const float *) (o.resid + c * o.resid_cs + (size_t) row * 4);
                            *(float *) (o.dst + c * o.dst_cs + (size_t) row * 4) = s;
                        }
                        acc[r][c] = 0.0f;
                    }
                }
            }
        }
        if (!more) break;
    }
}

// =================================================================================================
// Q6_K : 210-B super-block {ql[128], qh[64], int8 scales[16], f16 d} (ggml-common.h:330-335): only 2-byte aligned.
// 8 lanes per super-block: lane (n = lp>>2, tp = lp&3) owns l in [8tp, 8tp+8) of the 128-half n:
//   ql[64n+l], ql[64n+32+l], qh[32n+l]  ->  4 x 8 six-bit weights at y[128n + {0,32,64,96} + l]
// (dequantize_row_q6_K, ggml-quants.c:1762-1791).  A wave covers 8 super-blocks (1680 B) per step.
// Every 8-byte piece is fetched as an ALIGNED 12-byte load and funnel-shifted (v_alignbyte_b32) by the block's
// 0- or 2-byte phase; the over-read stays inside the same 210-byte block ("Q6_K loads").
// =================================================================================================
static __device__ __forceinline__ u32x2 ld_piece8(const char * p) {
#if MI_Q6K_X3
    const uintptr_t a = (uintptr_t) p;
    const unsigned  s = (unsigned) (a & 3);                              // 0 or 2
    const uint32_t * q = (const uint32_t *) (a - s);
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    const u32x3 v = ld_w((const u32x3 *) q, MI_Q6K_NT);                   // global_load_dwordx3, 4-byte aligned
    u32x2 r;
    r[0] = __builtin_amdgcn_alignbyte(v[1], v[0], s);
    r[1] = __builtin_amdgcn_alignbyte(v[2], v[1], s);
    return r;
#else
    typedef uint32_t __attribute__((aligned(2))) u32a2;                  // 2-byte aligned: the hardware handles the phase
    u32x2 r; r[0] = ld_w((const u32a2 *) p, MI_Q6K_NT); r[1] = ld_w((const u32a2 *) (p + 4), MI_Q6K_NT);
    return r;
#endif
}
// same, but never touches a byte beyond p+8 (phase 0) / p+10 (phase 2): used for the scales piece, whose 12-byte
// form would cross the end of the super-block (and, for the last block, of the tensor)
static __device__ __forceinline__ u32x2 ld_piece8_tail(const char * p) {
#if MI_Q6K_X3
    const uintptr_t a = (uintptr_t) p;
    const unsigned  s = (unsigned) (a & 3);
    const uint32_t * q = (const uint32_t *) (a - s);
    const u32x2    v01 = ld_w((const u32x2 *) q, MI_Q6K_NT);
    const uint32_t v2  = ld_w(q + (s ? 2 : 1), MI_Q6K_NT);
    u32x2 r;
    r[0] = __builtin_amdgcn_alignbyte(v01[1], v01[0], s);
    r[1] = __builtin_amdgcn_alignbyte(v2, v01[1], s);
    return r;
#else
    return ld_piece8(p);
#endif
}
// bytes in [0,63] -> signed bytes (w - 32), SWAR without inter-byte borrow
static __device__ __forceinline__ uint32_t sub32(uint32_t w) { return ((w | 0x80808080u) - 0x20202020u) ^ 0x80808080u; }

template <int NCOLS, int ROWS, int U, bool PAIR, bool NORM>
static __device__ __forceinline__ void q6k_body(const char * __restrict__ W0, const char * __restrict__ W1, size_t w_rs, const mmv_out o,
                                                const char * __restrict__ act, size_t act_cs, int K, int nrows, int wave, int nwaves, const act_norm nr) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 3, lp = lane & 7, n = lp >> 2, tp = lp & 3;
    const int nb  = K >> 8;
    const int nit = (nb + 8 * U - 1) / (8 * U);
    const size_t img = q8k_image_bytes(K);
    const int ngrp = PAIR ? nrows : (nrows + ROWS - 1) / ROWS;

    u32x2 qla[U][ROWS], qlb[U][ROWS], qh[U][ROWS], sc8[U][ROWS]; uint32_t dw[U][ROWS];
    auto issue2 = [&](int grp, int it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int ib = (it * U + u) * 8 + g; ib = ib < nb ? ib : nb - 1;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                int row = PAIR ? grp : grp * ROWS + r; row = row < nrows ? row : nrows - 1;
                const char * bp = ((PAIR && r == 1) ? W1 : W0) + (size_t) row * w_rs + (size_t) ib * 210;
                qla[u][r] = ld_piece8(bp + 64 * n + 8 * tp);
                qlb[u][r] = ld_piece8(bp + 64 * n + 32 + 8 * tp);
                qh[u][r]  = ld_piece8(bp + 128 + 32 * n + 8 * tp);
                sc8[u][r] = ld_piece8_tail(bp + 192 + 8 * n);              // scales[8n .. 8n+7]
                dw[u][r]  = ld_w((const uint16_t *) (bp + 208), MI_Q6K_NT);
            }
        }
    };

    int grp = wave, it = 0;
    norm_regs nrg;
    if (NORM) norm_prefetch(nr, K, nrg);
    if (grp < ngrp) issue2(grp, 0);
    if (NORM) norm_finish(nr, K, nrg); else stage_act_k(act, act_cs, NCOLS, img);
    __syncthreads();
    if (grp >= ngrp) return;

    float acc[ROWS][NCOLS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) acc[r][c] = 0.0f;

    const int is = tp >> 1;                 // l/16 for l in [8tp, 8tp+8)
    while (true) {
        u32x2 cqla[U][ROWS], cqlb[U][ROWS], cqh[U][ROWS], csc[U][ROWS]; uint32_t cdw[U][ROWS];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { cqla[u][r] = qla[u][r]; cqlb[u][r] = qlb[u][r]; cqh[u][r] = qh[u][r]; csc[u][r] = sc8[u][r]; cdw[u][r] = dw[u][r]; }
        const int cgrp = grp, cit = it;
        ++it;
        if (it == nit) { it = 0; grp += nwaves; }
        const bool more = grp < ngrp;
        if (more) issue2(grp, it);

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
                const uint32_t s01 = csc[u][r][0] >> (8 * is), s23 = csc[u][r][1] >> (8 * is);
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
                const bool rv = valid && (PAIR ? cgrp : cgrp * ROWS + r) < nrows;
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
            if (PAIR) {
#pragma unroll
                for (int c = 0; c < NCOLS; ++c) {
                    const float gsum = wave_sum_f32(acc[0][c]), usum = wave_sum_f32(acc[1][c]);
                    if (lane == 0) *(float *) (o.dst + c * o.dst_cs + (size_t) cgrp * 4) = silu_f(gsum) * usum;
                    acc[0][c] = 0.0f; acc[1][c] = 0.0f;
                }
            } else {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int row = cgrp * ROWS + r;
#pragma unroll
                    for (int c = 0; c < NCOLS; ++c) {
                        float s = wave_sum_f32(acc[r][c]);
                        if (lane == 0 && row < nrows) {
                            if (o.resid) s += *(const float *) (o.resid + c * o.resid_cs + (size_t) row * 4);
                            *(float *) (o.dst + c * o.dst_cs + (size_t) row * 4) = s;
                        }
                        acc[r][c] = 0.0f;
                    }
                }
            }
        }
        if (!more) break;
    }
}

// =================================================================================================
// kernels
// =================================================================================================
struct mmv_mat_dev { const char * W; size_t w_rs; mmv_out o; int nrows; int type; int wave_end; };   // waves [prev.wave_end, wave_end) work on this matrix
struct mmv_multi_dev { mmv_mat_dev m[3]; int nmat; const char * act; size_t act_cs; int K; act_norm nr; };

// TM: bit0 = Q4_K bodies compiled in, bit1 = Q6_K, bit2 = Q5_K
template <int NCOLS, int ROWS, int U, int TM, bool NORM>
__global__ void __launch_bounds__(256) k_mmv_multi(const mmv_multi_dev a) {
    // (each body stages the activation image itself, after issuing its first weight loads; every wave of the workgroup
    // runs exactly one body, so the single __syncthreads inside is met by all of them)
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    int mi_ = 0, w0 = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) if (i + 1 < a.nmat && wave >= a.m[i].wave_end) { mi_ = i + 1; w0 = a.m[i].wave_end; }
    // (static selection so the descriptor stays in SGPRs)
    const mmv_mat_dev M = mi_ == 0 ? a.m[0] : (mi_ == 1 ? a.m[1] : a.m[2]);
    const int lw = wave - w0, nw = M.wave_end - w0;
    if (TM == 1)      q4k_body<NCOLS, ROWS, U, false, NORM>(M.W, nullptr, M.w_rs, M.o, a.act, a.act_cs, a.K, M.nrows, lw, nw, a.nr);
    else if (TM == 2) q6k_body<NCOLS, ROWS, U, false, NORM>(M.W, nullptr, M.w_rs, M.o, a.act, a.act_cs, a.K, M.nrows, lw, nw, a.nr);
    else if (TM == 4) q4k_body<NCOLS, ROWS, U, false, NORM, true>(M.W, nullptr, M.w_rs, M.o, a.act, a.act_cs, a.K, M.nrows, lw, nw, a.nr);
    else if ((TM & 1) && M.type == GGML_TYPE_Q4_K) q4k_body<NCOLS, ROWS, U, false, NORM>(M.W, nullptr, M.w_rs, M.o, a.act, a.act_cs, a.K, M.nrows, lw, nw, a.nr);
    else if ((TM & 4) && M.type == GGML_TYPE_Q5_K) q4k_body<NCOLS, ROWS, U, false, NORM, true>(M.W, nullptr, M.w_rs, M.o, a.act, a.act_cs, a.K, M.nrows, lw, nw, a.nr);
    else                                           q6k_body<NCOLS, ROWS, U, false, NORM>(M.W, nullptr, M.w_rs, M.o, a.act, a.act_cs, a.K, M.nrows, lw, nw, a.nr);
}

template <int NCOLS, int U, int TYPE, bool NORM>
__global__ void __launch_bounds__(256) k_mmv_pair(const char * __restrict__ Wg, const char * __restrict__ Wu, size_t w_rs, const char * __restrict__ act, size_t act_cs,
                                                 char * __restrict__ dst, size_t dst_cs, int K, int nrows, const act_norm nr) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const mmv_out o = { dst, dst_cs, nullptr, 0 };
    if (TYPE == GGML_TYPE_Q4_K)      q4k_body<NCOLS, 2, U, true, NORM>(Wg, Wu, w_rs, o, act, act_cs, K, nrows, wave, nwaves, nr);
    else if (TYPE == GGML_TYPE_Q5_K) q4k_body<NCOLS, 2, U, true, NORM, true>(Wg, Wu, w_rs, o, act, act_cs, K, nrows, wave, nwaves, nr);
    else                             q6k_body<NCOLS, 2, U, true, NORM>(Wg, Wu, w_rs, o, act, act_cs, K, nrows, wave, nwaves, nr);
}

// ------------------------------------------------------------------------------------------------ launch
static const size_t MMVK_LDS_MAX = 152 * 1024;

// workgroups per launch are capped at (resident workgroups per CU) x 256 CUs = 1024 (occupancy 4 waves/SIMD), so that a big matrix
// is ONE resident wave of workgroups that grid-stride over the row groups: 12288 gate/up rows = exactly 3 per wave.  Measured on
// the ffn_gate+ffn_up launch: 1024 -> 13.4 us, 2048 -> 14.2 us (1.5 rows per wave: uneven tail), 3072 -> 15.2 us.
// MI355X_MMV_WGS overrides for tuning
static int mmv_grid_cap() {
    static int cap = 0;
    if (!cap) { const char * e = getenv("MI355X_MMV_WGS"); cap = e ? atoi(e) : 1024; if (cap < 1) cap = 1024; }
    return cap;
}

template <int NCOLS, int ROWS, int U, bool NORM = false>
static void launch_multi_tm(const mmv_multi_dev & d, int tm, int grid, size_t lds, hipStream_t st) {
    auto go = [&](auto kern) {
        if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        kern<<<dim3(grid), dim3(256), lds, st>>>(d);
    };
    if (tm == 1) go(k_mmv_multi<NCOLS, ROWS, U, 1, NORM>); else if (tm == 2) go(k_mmv_multi<NCOLS, ROWS, U, 2, NORM>); else if (tm == 3) go(k_mmv_multi<NCOLS, ROWS, U, 3, NORM>);
    else if (tm == 4) go(k_mmv_multi<NCOLS, ROWS, U, 4, NORM>); else if (tm == 6) go(k_mmv_multi<NCOLS, ROWS, U, 6, NORM>); else go(k_mmv_multi<NCOLS, ROWS, U, 7, NORM>);
}

void mmv_kquant_multi(const mmv_multi_args & a, hipStream_t st) {
    if (a.nmat == 0 || a.ncols == 0) return;
    if (a.norm.x && a.ncols != 1) { fprintf(stderr, "[mi355x] mmv_kquant_multi: in-kernel norm is a one-column path\n"); abort(); }
    const size_t img = q8k_image_bytes(a.K);
    if (img * a.ncols > MMVK_LDS_MAX) { fprintf(stderr, "[mi355x] mmv_kquant_multi: activation images exceed LDS (K=%lld, ncols=%d)\n", (long long) a.K, a.ncols); abort(); }
    const int rows_pw = 2;
    // grid: enough waves that every row group is resident at once when the matrices are small, capped at 8 WG/CU
    double bytes[3]; double total = 0; int64_t groups = 0;
    int tm = 0;
    for (int i = 0; i < a.nmat; ++i) {
        bytes[i] = (double) a.m[i].nrows * (double) (a.m[i].type == GGML_TYPE_Q4_K ? 144 : a.m[i].type == GGML_TYPE_Q5_K ? 176 : 210) * (double) (a.K / 256);
        total += bytes[i];
        groups += (a.m[i].nrows + rows_pw - 1) / rows_pw;
        tm |= a.m[i].type == GGML_TYPE_Q4_K ? 1 : a.m[i].type == GGML_TYPE_Q5_K ? 4 : 2;
    }
    int64_t grid = (groups + 3) / 4;
    const int cap = a.norm.x ? (mmv_grid_cap() < 1024 ? mmv_grid_cap() : 1024) : mmv_grid_cap();   // in-kernel norm: fewer, longer workgroups
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    const int nwaves = (int) grid * 4;
    mmv_multi_dev d;
    d.nmat = a.nmat; d.act = (const char *) a.act; d.act_cs = a.act_cs; d.K = (int) a.K;
    d.nr = { (const char *) a.norm.x, a.norm.x_cs, a.norm.w, a.norm.eps };
    int acc_w = 0; double acc_b = 0;
    for (int i = 0; i < 3; ++i) {
        if (i >= a.nmat) { d.m[i] = d.m[0]; d.m[i].wave_end = nwaves; continue; }
        acc_b += bytes[i];
        int end = i == a.nmat - 1 ? nwaves : (int) (nwaves * (acc_b / total) + 0.5);
        if (end <= acc_w) end = acc_w + 1;                                   // every matrix gets at least one wave
        if (end > nwaves - (a.nmat - 1 - i)) end = nwaves - (a.nmat - 1 - i);
        d.m[i].W = (const char *) a.m[i].W; d.m[i].w_rs = a.m[i].w_rs; d.m[i].nrows = (int) a.m[i].nrows; d.m[i].type = a.m[i].type;
        d.m[i].o = { (char *) a.m[i].dst, a.m[i].dst_cs, (const char *) a.m[i].resid, a.m[i].resid_cs };
        d.m[i].wave_end = end; acc_w = end;
    }
    const int nstep = (int) ((a.K / 256 + 7) / 8);
    const size_t lds = img * a.ncols;
    const bool u2 = nstep >= 2 && a.ncols == 1;
#define MM_GO(NC, R, UU) launch_multi_tm<NC, R, UU>(d, tm, (int) grid, lds, st)
    switch (a.ncols) {
        case 1:
            if (a.norm.x) {
                if (!mmv_norm_ok(a.K, 1)) { fprintf(stderr, "[mi355x] mmv_kquant_multi: in-kernel norm needs one column and K <= %d\n", 4 * NORM_MAXB * 256); abort(); }
                if (u2) launch_multi_tm<1, 2, 2, true>(d, tm, (int) grid, lds, st); else launch_multi_tm<1, 2, 1, true>(d, tm, (int) grid, lds, st);
            } else if (u2) MM_GO(1, 2, 2); else MM_GO(1, 2, 1);
            break;
        case 2: MM_GO(2, 2, 1); break;
        case 3: MM_GO(3, 2, 1); break;
        case 4: MM_GO(4, 2, 1); break;
        case 5: MM_GO(5, 2, 1); break;
        case 6: MM_GO(6, 2, 1); break;
        case 7: MM_GO(7, 2, 1); break;
        case 8: MM_GO(8, 2, 1); break;
        default: fprintf(stderr, "[mi355x] mmv_kquant_multi: ncols=%d out of range\n", a.ncols); abort();
    }
#undef MM_GO
}

void mmv_kquant_pair_swiglu(int type, const void * Wg, const void * Wu, size_t w_rs, const void * act, size_t act_cs, float * dst, size_t dst_cs,
                            int64_t K, int64_t nrows, int ncols, hipStream_t st, const mmv_norm * norm) {
    if (nrows == 0 || ncols == 0) return;
    const size_t lds = q8k_image_bytes(K) * ncols;
    if (norm && norm->x && ncols != 1) { fprintf(stderr, "[mi355x] mmv_kquant_pair_swiglu: in-kernel norm is a one-column path\n"); abort(); }
    const act_norm nr = norm && norm->x ? act_norm{ (const char *) norm->x, norm->x_cs, norm->w, norm->eps } : act_norm{ nullptr, 0, nullptr, 0.0f };
    const int cap = nr.x ? (mmv_grid_cap() < 1024 ? mmv_grid_cap() : 1024) : mmv_grid_cap();
    int64_t grid = (nrows + 3) / 4; if (grid > cap) grid = cap;
    const bool u2 = (K / 256 + 7) / 8 >= 2 && ncols == 1;
#define MP_GO(NC, UU) MP_GO2(NC, UU, false)
#define MP_GO2(NC, UU, NRM)                                                                                            \
    do {                                                                                                               \
        if (type == GGML_TYPE_Q4_K) k_mmv_pair<NC, UU, GGML_TYPE_Q4_K, NRM><<<dim3((unsigned) grid), dim3(256), lds, st>>>(  \
            (const char *) Wg, (const char *) Wu, w_rs, (const char *) act, act_cs, (char *) dst, dst_cs, (int) K, (int) nrows, nr);  \
        else if (type == GGML_TYPE_Q5_K) k_mmv_pair<NC, UU, GGML_TYPE_Q5_K, NRM><<<dim3((unsigned) grid), dim3(256), lds, st>>>(  \
            (const char *) Wg, (const char *) Wu, w_rs, (const char *) act, act_cs, (char *) dst, dst_cs, (int) K, (int) nrows, nr);  \
        else k_mmv_pair<NC, UU, GGML_TYPE_Q6_K, NRM><<<dim3((unsigned) grid), dim3(256), lds, st>>>(                    \
            (const char *) Wg, (const char *) Wu, w_rs, (const char *) act, act_cs, (char *) dst, dst_cs, (int) K, (int) nrows, nr);  \
    } while (0)
    switch (ncols) {
        case 1:
            if (nr.x) {
                if (!mmv_norm_ok(K, 1)) { fprintf(stderr, "[mi355x] mmv_kquant_pair_swiglu: in-kernel norm needs one column and K <= %d\n", 4 * NORM_MAXB * 256); abort(); }
                if (u2) MP_GO2(1, 2, true); else MP_GO2(1, 1, true);
            } else if (u2) MP_GO(1, 2); else MP_GO(1, 1);
            break;
        case 2: MP_GO(2, 1); break;
        case 3: MP_GO(3, 1); break;
        case 4: MP_GO(4, 1); break;
        case 5: MP_GO(5, 1); break;
        case 6: MP_GO(6, 1); break;
        case 7: MP_GO(7, 1); break;
        case 8: MP_GO(8, 1); break;
        default: fprintf(stderr, "[mi355x] mmv_kquant_pair_swiglu: ncols=%d out of range (1..8)\n", ncols); abort();
    }
#undef MP_GO
#undef MP_GO2
}

} // namespace mi

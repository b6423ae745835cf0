// mmv1.hip -- the batch-1 decode mat-vec of K-quant weights (Q4_K / Q6_K x one activation column), gfx950 / wave64.
//
// What is computed (reference: ggml_compute_forward_mul_mat, ggml-cpu/ggml-cpu.c:1210-1402, ne11 == 1):
//     dst[row] = vec_dot(W[row, :], Q8_K(act))         act = x                        (src1 as the graph hands it over)
//                                                       or  rms_norm(x) * w            (the RMS_NORM + MUL nodes in front of src1)
//     Q8_K(.)  : quantize_row_q8_K_ref, ggml-quants.c:2555-2592
//     rms_norm : ggml_compute_forward_rms_norm_f32, ggml-cpu/ops.cpp:3517-3566 (sum of squares in double)
//     vec_dot  : ggml_vec_dot_q4_K_q8_K / _q6_K_q8_K, ggml-cpu/quants.c:550-623 / 705-758 (integer sub-block sums exact)
// Epilogues: + resid (the graph's residual ADD), or silu(gate) * up for the ffn_gate / ffn_up pair (the graph's GLU node).
//
// Why a second family next to mmvk.hip: a decode step is ~180 dependent launches, and what a launch costs is its serial latency chain,
// not its bytes (tools/launch_bench.hip: a 9.4 MB stream behind a dependent 16 KB read is 4.0 us, 56.6 MB 10.4 us -- 2.7 us + bytes /
// 7.4 TB/s).  So this kernel
//   * takes the f32 activation row itself: no stand-alone norm / quantise launch in front of it (3 of the 8 launches of a layer).  Every
//     workgroup builds the Q8_K image in LDS with DPP-network reductions (no ds_bpermute chains: ~80 VALU per 256-block) while its
//     first weight loads are already in flight;
//   * keeps DEPTH stages of weight loads in flight per wave (VGPRs are the largest prefetch buffer of a CU: 512 KB);
//   * hands contiguous row ranges to waves.
#include "../kernels.hpp"
#include "mv_dev.hpp"

namespace mi {

// ================================================================================================= Q4_K
// 144-B super-block = 16-B header {d, dmin, 12 B of 6-bit scales/mins} + 128 B of nibbles (ggml-common.h:295-305).  FOUR lanes per
// super-block: lane q owns qs[32q .. 32q+32) = all 32 low nibbles of sub-block 2q and all 32 high nibbles of sub-block 2q+1
// (dequantize_row_q4_K, ggml-quants.c:1352-1374), so one decode of (scale, min) x 2 serves 64 weights.  A wave covers 16 super-blocks
// (2304 contiguous bytes = one row of K = 4096) per step; R rows per task (rows R*t .. R*t+R-1 of W, or -- PAIR -- row t of the gate and
// of the up matrix); DEPTH stages of loads in flight.  The kernel is VALU-issue-bound before it is HBM-bound (rocprofv3 SQ_INSTS_VALU),
// so everything per-lane that can be a kernel-lifetime constant is one: global loads are `scalar base + lane offset`, LDS reads are
// `lane base + immediate`, the 6-bit fields are picked with v_perm_b32 under a lane-constant selector, 24-bit multiplies.
// Requires K % 256 == 0 (TAIL unless K % 4096 == 0), 16-byte aligned rows; R == 2 without PAIR requires an even row count (launcher-checked).
template <int R, int DEPTH, bool PAIR, bool NT, bool TAIL, typename PRO>
static __device__ __forceinline__ void mv1_q4k(const char * __restrict__ W0, const char * __restrict__ W1, size_t w_rs, char * __restrict__ dst, const char * __restrict__ resid,
                                               int K, int nrows, int lw, int nw, PRO & pro) {
    constexpr int NBUF = DEPTH + 1;
    const int lane = threadIdx.x & 63;
    const int blk = lane >> 2, q = lane & 3;
    const int nb  = K >> 8;
    // steps per row.  TAIL (K % 4096 != 0: other model widths -- 2560, 3584, 5120, 14336 ...): the last step of a row covers fewer than 16 super-blocks;
    // the lanes past the row still load (into the next row, or zeros past the matrix) and multiply, but with d = dmin = yd = 0
    const int nit = TAIL ? (nb + 15) >> 4 : nb >> 4;
    const int ntask = PAIR ? nrows : nrows / R;
    int g0, g1; mv1_range(ntask, lw, nw, g0, g1);

    // in-flight bytes per VGPR decide the HBM rate (Little: ~3.5 us loaded latency x 20 GB/s per CU = 70 KB per CU just to break even), so the
    // 16-byte header is fetched ONCE per block -- lane q takes dword q -- and broadcast inside the quad on the DPP network when it is used
#ifdef MV1_DENSE_LOADS      // measurement only (wrong results): every cache line requested by exactly one instruction
    const uint32_t voff_h = 2048u + 4u * (uint32_t) lane, voff_q = 16u * (uint32_t) lane;
#define MV1_QB_OFF 1024u
#else
    const uint32_t voff_h = (uint32_t) blk * 144u + 4u * (uint32_t) q, voff_q = (uint32_t) blk * 144u + 16u + 32u * (uint32_t) q;
#define MV1_QB_OFF 16u
#endif
    uint32_t hq[NBUF][R]; u32x4 qa[NBUF][R], qb[NBUF][R];
    // buffer loads: 128-bit descriptor of the whole matrix + wave-uniform byte offset (SGPR) + lane-constant offset (one VGPR) + immediate --
    // no per-load address arithmetic, no 64-bit pointer pairs in VGPRs
    const mv1_rsrc rs0 = mv1_make_rsrc(W0, (size_t) nrows * w_rs), rs1 = mv1_make_rsrc(PAIR ? W1 : W0, (size_t) nrows * w_rs);
    const uint32_t rs32 = (uint32_t) w_rs;
    // TAIL: the lanes past the row in its last step request nothing -- their offset is pushed out of the descriptor's range (reads as zero, no
    // memory traffic; mmv1_ok keeps TAIL matrices below MV1_KILL bytes) -- so the launch still moves exactly the matrix
    const uint32_t kill = TAIL && (nit - 1) * 16 + blk >= nb ? MV1_KILL : 0u;
    auto issue = [&](int task, int it, int bf) {
        const uint32_t kl = TAIL && it == nit - 1 ? kill : 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = PAIR ? task : task * R + r;
            const uint32_t so = (uint32_t) row * rs32 + (uint32_t) it * 2304u;                           // wave-uniform
            const mv1_rsrc rs = (PAIR && r == 1) ? rs1 : rs0;
            hq[bf][r] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff_h | kl, so, NT ? 2 : 0);
            qa[bf][r] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_q | kl, so, NT ? 2 : 0);
            qb[bf][r] = __builtin_amdgcn_raw_buffer_load_b128(rs, (voff_q + MV1_QB_OFF) | kl, so, NT ? 2 : 0);
        }
    };
#ifndef MV1_KO
#define MV1_KO 0
#endif
    if (!(MV1_KO & 1)) pro.issue();                     // the activation row is requested before the first weight stage
    if (MV1_KO & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // measurement: the row has ARRIVED before the first weight request
    if (MV1_KO & 32) __builtin_amdgcn_s_barrier();                         // measurement: every wave's row request is out before the first weight request
    int ig = g0, iit = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (ig < g1) { issue(ig, iit, d); if (++iit == nit) { iit = 0; ++ig; } }

    if (!(MV1_KO & 1)) pro.finish();                    // builds the activation image in LDS, ends with a workgroup barrier
    if (g0 >= g1) return;

    // lane-constant LDS addresses of step 0; a step advances them by 16 blocks
    const char * im = mv1_lds;
    const char * la = im + blk * 272 + 64 * q;
    const char * lb = im + mv1_img_bs(nb) + blk * 16 + 4 * q;
    const char * ld = im + mv1_img_d(nb) + blk * 4;
    // byte selector for v_perm_b32(hi, lo, sel): bytes (2(q&1), 2(q&1)+1) of `lo` (sub-blocks 0..3) or of `hi` (4..7), upper half zero
    const uint32_t sel = 0x0c0c0000u | (uint32_t) ((q < 2 ? 0 : 4) + 2 * (q & 1)) | ((uint32_t) ((q < 2 ? 0 : 4) + 2 * (q & 1) + 1) << 8);
    float acc[R], accm[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { acc[r] = 0.0f; accm[r] = 0.0f; }
    int cg = g0, cit = 0;
    for (;;) {
#pragma unroll
        for (int ph = 0; ph < NBUF; ++ph) {
            if (ig < g1) { issue(ig, iit, (ph + DEPTH) % NBUF); if (++iit == nit) { iit = 0; ++ig; } }
            if (MV1_KO & 2) {
#pragma unroll
                for (int r = 0; r < R; ++r) { const u32x4 Q = qa[ph][r], P = qb[ph][r]; acc[r] += __int_as_float((hq[ph][r] ^ Q[0] ^ Q[1] ^ Q[2] ^ Q[3] ^ P[0] ^ P[1] ^ P[2] ^ P[3]) & 0x3fffffff); }
            } else {
                const int so = cit * 16;                                             // first block of this step (wave-uniform)
                const u32x4 a0 = *(const u32x4 *) (la + so * 272), a1 = *(const u32x4 *) (la + so * 272 + 16);      // activations of sub-block 2q
                const u32x4 a2 = *(const u32x4 *) (la + so * 272 + 32), a3 = *(const u32x4 *) (la + so * 272 + 48); // ... of sub-block 2q+1
                const uint32_t bsw = *(const uint32_t *) (lb + so * 16);                                           // their two sums of 32
                const int bs0 = (int) (int16_t) (bsw & 0xffff), bs1 = (int) (int16_t) (bsw >> 16);
                const bool live = !TAIL || so + blk < nb;
                const float yd = live ? *(const float *) (ld + so * 4) : 0.0f;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const u32x4 Q = qa[ph][r], P = qb[ph][r];
                    const uint32_t hw = hq[ph][r];
                    const uint32_t H[4] = { dpp_u32q<0x00>(hw), dpp_u32q<0x55>(hw), dpp_u32q<0xAA>(hw), dpp_u32q<0xFF>(hw) };       // quad_perm broadcasts
                    // 6-bit scale / min unpack for all eight sub-blocks at once (get_scale_min_k4, ggml-quants.c:703-710), then this lane's pair
                    const uint32_t s_lo = H[1] & 0x3f3f3f3fu, s_hi = (H[3] & 0x0f0f0f0fu) | ((H[1] >> 2) & 0x30303030u);
                    const uint32_t m_lo = H[2] & 0x3f3f3f3fu, m_hi = ((H[3] >> 4) & 0x0f0f0f0fu) | ((H[2] >> 2) & 0x30303030u);
                    const uint32_t sw = __builtin_amdgcn_perm(s_hi, s_lo, sel), mw = __builtin_amdgcn_perm(m_hi, m_lo, sel);
                    const int sc0 = sw & 0xff, sc1 = sw >> 8;
                    const int mn0 = mw & 0xff, mn1 = mw >> 8;
                    const float dx   = live ? h2f((uint16_t) (H[0] & 0xffff)) : 0.0f;
                    const float dmin = live ? h2f((uint16_t) (H[0] >> 16)) : 0.0f;
                    int dl = 0, dh = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        dl = dot4(Q[k] & 0x0f0f0f0fu, a0[k], dl); dh = dot4((Q[k] >> 4) & 0x0f0f0f0fu, a2[k], dh);
                        dl = dot4(P[k] & 0x0f0f0f0fu, a1[k], dl); dh = dot4((P[k] >> 4) & 0x0f0f0f0fu, a3[k], dh);
                    }
                    // |dl|, |dh| <= 32 * 15 * 128 and the scales are 6-bit: the products fit 24-bit multiplies
                    const int isum = mad24(sc0, dl, mul24(sc1, dh));
                    const int msum = mad24(mn0, bs0, mul24(mn1, bs1));
                    acc[r]  = fmaf(dx * yd, (float) isum, acc[r]);
                    accm[r] = fmaf(dmin * yd, (float) msum, accm[r]);
                }
            }
            if (cit == nit - 1) {                          // task finished: butterfly, epilogue, store
                if (PAIR) {
                    const float gsum = wave_sum_f32(acc[0] - accm[0]), usum = wave_sum_f32(acc[R - 1] - accm[R - 1]);
                    if (lane == 0) *(float *) (dst + (size_t) cg * 4) = mv1_silu(gsum) * usum;
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int row = cg * R + r;
                        float s = wave_sum_f32(acc[r] - accm[r]);
                        if (lane == 0) {
                            if (resid) s += *(const float *) (resid + (size_t) row * 4);
                            *(float *) (dst + (size_t) row * 4) = s;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) { acc[r] = 0.0f; accm[r] = 0.0f; }
            }
            if (++cit == nit) { cit = 0; ++cg; }
            if (cg >= g1) return;
        }
    }
}

// ================================================================================================= Q6_K
// 210-B super-block {ql[128], qh[64], int8 scales[16], f16 d} (ggml-common.h:330-335): only 2-byte aligned.  FOUR lanes per super-block:
// lane (n, hf) owns l in [16hf, 16hf+16) of the 128-half n: ql[64n+l], ql[64n+32+l], qh[32n+l] -> 4 x 16 six-bit weights at
// y[128n + 32m + l], m = 0..3, all with scale index 8n + hf + 2m (dequantize_row_q6_K, ggml-quants.c:1762-1791).  A wave covers 16
// super-blocks (3360 contiguous bytes = one row of K = 4096) per step.  The pieces are hardware-unaligned 16-byte loads.  The -32 offset of
// the 6-bit values is not applied per weight (3 VALU per dword) but per sub-block of 16 through the activation's bsums:
// sum((q - 32) * a) = sum(q * a) - 32 * bsum -- the same integers.
typedef u32x4 __attribute__((aligned(2))) u32x4_a2;
typedef uint32_t __attribute__((aligned(2))) u32_a2;

template <int R, int DEPTH, bool PAIR, bool TAIL, typename PRO>
static __device__ __forceinline__ void mv1_q6k(const char * __restrict__ W0, const char * __restrict__ W1, size_t w_rs, char * __restrict__ dst, const char * __restrict__ resid,
                                               int K, int nrows, int lw, int nw, PRO & pro) {
    constexpr int NBUF = DEPTH + 1;
    const int lane = threadIdx.x & 63;
    const int blk = lane >> 2, n = (lane >> 1) & 1, hf = lane & 1;
    const int nb  = K >> 8;
    const int nit = TAIL ? (nb + 15) >> 4 : nb >> 4;       // (TAIL: see mv1_q4k)
    const int ntask = PAIR ? nrows : nrows / R;
    int g0, g1; mv1_range(ntask, lw, nw, g0, g1);

    const uint32_t vb = (uint32_t) blk * 210u;
    const uint32_t voff_l = vb + 64u * n + 16u * hf, voff_h = vb + 128u + 32u * n + 16u * hf;
    u32x4 qla[NBUF][R], qlb[NBUF][R], qh[NBUF][R]; uint32_t sc[NBUF][R], dw[NBUF][R];     // sc: lane (n, hf) fetches scale dword 2n + hf
    const mv1_rsrc rs0 = mv1_make_rsrc(W0, (size_t) nrows * w_rs), rs1 = mv1_make_rsrc(PAIR ? W1 : W0, (size_t) nrows * w_rs);
    const uint32_t rs32 = (uint32_t) w_rs;
    const uint32_t voff_s = vb + 192u + 4u * (uint32_t) (lane & 3);
    const uint32_t kill = TAIL && (nit - 1) * 16 + blk >= nb ? MV1_KILL : 0u;                      // (see mv1_q4k)
    auto issue = [&](int task, int it, int bf) {
        const uint32_t kl = TAIL && it == nit - 1 ? kill : 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = PAIR ? task : task * R + r;
            const uint32_t so = (uint32_t) row * rs32 + (uint32_t) it * 3360u;                           // wave-uniform
            const mv1_rsrc rs = (PAIR && r == 1) ? rs1 : rs0;
            qla[bf][r] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_l | kl, so, 0);
            qlb[bf][r] = __builtin_amdgcn_raw_buffer_load_b128(rs, (voff_l + 32u) | kl, so, 0);
            qh[bf][r]  = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_h | kl, so, 0);
            sc[bf][r]  = __builtin_amdgcn_raw_buffer_load_b32(rs, voff_s | kl, so, 0);
            dw[bf][r]  = __builtin_amdgcn_raw_buffer_load_b16(rs, (vb + 208u) | kl, so, 0);
        }
    };
    if (!(MV1_KO & 1)) pro.issue();
    int ig = g0, iit = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (ig < g1) { issue(ig, iit, d); if (++iit == nit) { iit = 0; ++ig; } }

    if (!(MV1_KO & 1)) pro.finish();
    if (g0 >= g1) return;

    const char * im = mv1_lds;
    const char * la = im + blk * 272 + 128 * n + 16 * hf;
    const char * lb = im + mv1_img_b16(nb) + blk * 32 + (8 * n + 4 * hf) * 2;
    const char * ld = im + mv1_img_d(nb) + blk * 4;
    // scales[8n + hf + 2m], m = 0..3: bytes (hf, hf+2) of scale dwords 2n and 2n+1
    const uint32_t sel = (uint32_t) hf | ((uint32_t) (hf + 2) << 8) | ((uint32_t) (4 + hf) << 16) | ((uint32_t) (6 + hf) << 24);
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0f;
    int cg = g0, cit = 0;
    for (;;) {
#pragma unroll
        for (int ph = 0; ph < NBUF; ++ph) {
            if (ig < g1) { issue(ig, iit, (ph + DEPTH) % NBUF); if (++iit == nit) { iit = 0; ++ig; } }
            {
                const int so = cit * 16;
                u32x4 a[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = *(const u32x4 *) (la + so * 272 + 32 * m);
                const u32x2 bq = *(const u32x2 *) (lb + so * 32);
                const int bs[4] = { (int) (int16_t) (bq[0] & 0xffff), (int) (int16_t) (bq[0] >> 16), (int) (int16_t) (bq[1] & 0xffff), (int) (int16_t) (bq[1] >> 16) };
                const bool live = !TAIL || so + blk < nb;
                const float yd = live ? *(const float *) (ld + so * 4) : 0.0f;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t sq = sc[ph][r];                                  // quad_perm [0,0,2,2] / [1,1,3,3]: scale dwords 2n and 2n+1
                    const uint32_t scw = __builtin_amdgcn_perm(dpp_u32q<0xF5>(sq), dpp_u32q<0xA0>(sq), sel);
                    const float dx = live ? h2f((uint16_t) dw[ph][r]) : 0.0f;
                    int d[4] = { 0, 0, 0, 0 };
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t A = qla[ph][r][k], B = qlb[ph][r][k], H = qh[ph][r][k];
                        d[0] = dot4((A & 0x0f0f0f0fu)        | ((H << 4) & 0x30303030u), a[0][k], d[0]);
                        d[1] = dot4((B & 0x0f0f0f0fu)        | ((H << 2) & 0x30303030u), a[1][k], d[1]);
                        d[2] = dot4(((A >> 4) & 0x0f0f0f0fu) | (H & 0x30303030u),        a[2][k], d[2]);
                        d[3] = dot4(((B >> 4) & 0x0f0f0f0fu) | ((H >> 2) & 0x30303030u), a[3][k], d[3]);
                    }
                    int isum = 0;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int scm = (int) (int8_t) ((scw >> (8 * m)) & 0xff);
                        // |d - 32 bs| <= 16 * 63 * 128 + 32 * 2048: 24-bit multiplies
                        isum = mad24(scm, mad24(bs[m], -32, d[m]), isum);
                    }
                    acc[r] = fmaf(dx * yd, (float) isum, acc[r]);
                }
            }
            if (cit == nit - 1) {
                if (PAIR) {
                    const float gsum = wave_sum_f32(acc[0]), usum = wave_sum_f32(acc[R - 1]);
                    if (lane == 0) *(float *) (dst + (size_t) cg * 4) = mv1_silu(gsum) * usum;
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int row = cg * R + r;
                        float s = wave_sum_f32(acc[r]);
                        if (lane == 0) {
                            if (resid) s += *(const float *) (resid + (size_t) row * 4);
                            *(float *) (dst + (size_t) row * 4) = s;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = 0.0f;
            }
            if (++cit == nit) { cit = 0; ++cg; }
            if (cg >= g1) return;
        }
    }
}

// ================================================================================================= kernel
// TM: bit0 = Q4_K body compiled in, bit1 = Q6_K.  PAIR: m[0] = gate, W1 = up (same type / shape), epilogue silu(g) * u.
template <int NW, int XB> struct mv1_pro {
    const mv1_src s; int K; double * red; mv1_act_regs<XB> r;
    __device__ __forceinline__ void issue()  { mv1_act_issue<NW, XB>(s, K, r); }
    __device__ __forceinline__ void finish() { mv1_act_finish<NW, XB>(s, K, r, mv1_lds, red); }
};

template <int NW, int XB, int R, int DEPTH, int TM, bool PAIR, bool NT, bool TAIL = false>
__global__ void __launch_bounds__(64 * NW) k_mv1(const mv1_dev a) {
    __shared__ double red[NW];
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * NW + (threadIdx.x >> 6));
    int mi_ = 0, w0 = 0;
    if (!PAIR) {
#pragma unroll
        for (int i = 0; i < 2; ++i) if (i + 1 < a.nmat && wave >= a.m[i].wave_end) { mi_ = i + 1; w0 = a.m[i].wave_end; }
    }
    const mv1_mat M = mi_ == 0 ? a.m[0] : (mi_ == 1 ? a.m[1] : a.m[2]);
    const int lw = wave - w0, nw = M.wave_end - w0;
    mv1_pro<NW, XB> pro = { a.src, a.K, red, {} };
    if (TM == 1 || ((TM & 1) && M.type == GGML_TYPE_Q4_K)) mv1_q4k<R, DEPTH, PAIR, NT, TAIL>(M.W, a.W1, M.w_rs, M.dst, M.resid, a.K, M.nrows, lw, nw, pro);
    else                                                    mv1_q6k<R, DEPTH, PAIR, TAIL>(M.W, a.W1, M.w_rs, M.dst, M.resid, a.K, M.nrows, lw, nw, pro);
}

} // namespace mi

// ------------------------------------------------------------------------------------------------ host side
namespace mi {

// measured on MI355X (tools/mmv_lab.hip, replayed hipGraph, rotating weights): 4096 waves, one stage of loads in flight per wave; the
// gate / up pair as 512 workgroups of 8 waves with two rows per task, everything else as 256 workgroups of 16 waves (ONE image build per CU)
// with one row per task
static const int MV1_WAVES = 4096;

static int g_mv2 = -1;
void mmv2_enable(bool on) { g_mv2 = on ? 1 : 0; }
bool mmv2_enabled();
static bool mv2_on() { if (g_mv2 < 0) { const char * e = getenv("MI355X_MV2"); g_mv2 = e ? (atoi(e) != 0) : 1; } return g_mv2 != 0; }

bool mmv1_ok(const mv1_args & a) {
    if (a.nmat >= 1 && (a.m[0].type == GGML_TYPE_Q8_0 || a.m[0].type == GGML_TYPE_F16))                                 // the Q8_0 / F16 twins (mmv1q.hip), or the engine for Q8_0 at the 8B widths
        return mmv1q_ok(a) || (a.m[0].type == GGML_TYPE_Q8_0 && mv2_on() && mmv2_ok(a));
    if (a.nmat < 1 || a.nmat > 3 || a.K <= 0 || a.K % 256 != 0 || a.K > 16384) return false;              // (K % 4096 != 0: the TAIL instances)
    if (a.W_up && a.nmat != 1) return false;
    for (int i = 0; i < a.nmat; ++i) {
        const mmv_mat & m = a.m[i];
        if (m.type != GGML_TYPE_Q4_K && m.type != GGML_TYPE_Q6_K) return false;
        if (m.nrows <= 0 || (uint64_t) m.nrows * m.w_rs > (a.K % 4096 ? (uint64_t) MV1_KILL : 0xffffffffull)) return false;
        if (m.type == GGML_TYPE_Q4_K && (m.w_rs % 16 != 0 || ((uintptr_t) m.W & 15) != 0)) return false;
        if (m.type == GGML_TYPE_Q6_K && (m.w_rs % 2 != 0 || ((uintptr_t) m.W & 1) != 0)) return false;
        if (((uintptr_t) m.dst & 3) != 0 || ((uintptr_t) m.resid & 3) != 0) return false;
    }
    if (a.W_up && (((uintptr_t) a.W_up & 15) != 0 || a.m[0].resid)) return false;
    if (a.img) return ((uintptr_t) a.img & 15) == 0;
    return a.x && ((uintptr_t) a.x & 15) == 0 && ((uintptr_t) a.norm_w & 15) == 0;
}

template <int NW, int XB, int R, bool PAIR, bool TAIL = false>
static void mv1_go(const mv1_dev & d, int tm, int grid, hipStream_t st) {
    const size_t lds = mv1_image_bytes(d.K);
    if (tm == 1)      k_mv1<NW, XB, R, 1, 1, PAIR, false, TAIL><<<dim3(grid), dim3(64 * NW), lds, st>>>(d);
    else if (tm == 2) k_mv1<NW, XB, R, 1, 2, PAIR, false, TAIL><<<dim3(grid), dim3(64 * NW), lds, st>>>(d);
    else if (!PAIR)   k_mv1<NW, XB, R, 1, 3, false, false, TAIL><<<dim3(grid), dim3(64 * NW), lds, st>>>(d);
    else { fprintf(stderr, "[mi355x] mmv1: a gate / up pair of mixed types\n"); abort(); }
}


bool mmv2_enabled() { return mv2_on(); }

void mmv1(const mv1_args & a, hipStream_t st) {
    if (a.nmat >= 1 && (a.m[0].type == GGML_TYPE_Q8_0 || a.m[0].type == GGML_TYPE_F16)) {
        if (a.m[0].type == GGML_TYPE_Q8_0 && mv2_on() && mmv2_ok(a)) { mmv2(a, st); return; }   // the 8B widths (K = 4096 / 12288, any row count): the LDS-DMA engine
        mmv1q(a, st); return;
    }
    if (mv2_on() && mmv2_ok(a)) { mmv2(a, st); return; }
    if (a.parts) { fprintf(stderr, "[mi355x] mmv1: attention partial states on a launch the LDS-DMA engine refuses\n"); abort(); }
    if (!mmv1_ok(a)) { fprintf(stderr, "[mi355x] mmv1: unsupported arguments (K=%lld)\n", (long long) a.K); abort(); }
    const bool pair = a.W_up != nullptr;
    const int nw_wg = pair ? 8 : 16;
    double bytes[3], total = 0; int64_t tasks = 0; int tm = 0;
    for (int i = 0; i < a.nmat; ++i) {
        bytes[i] = (double) a.m[i].nrows * (double) (a.m[i].type == GGML_TYPE_Q4_K ? 144 : 210) * (double) (a.K / 256);
        total += bytes[i]; tasks += a.m[i].nrows;
        tm |= a.m[i].type == GGML_TYPE_Q4_K ? 1 : 2;
    }
    int64_t grid = (tasks + nw_wg - 1) / nw_wg;
    if (grid > MV1_WAVES / nw_wg) grid = MV1_WAVES / nw_wg;
    const int nwaves = (int) grid * nw_wg;
    mv1_dev d;
    d.nmat = a.nmat; d.K = (int) a.K; d.W1 = (const char *) a.W_up;
    d.src = { a.img ? nullptr : a.x, a.img ? nullptr : a.norm_w, a.eps, (const char *) a.img };
    int acc_w = 0; double acc_b = 0;
    for (int i = 0; i < 3; ++i) {
        if (i >= a.nmat) { d.m[i] = d.m[0]; d.m[i].wave_end = nwaves; continue; }
        acc_b += bytes[i];
        int end = i == a.nmat - 1 ? nwaves : (int) (nwaves * (acc_b / total) + 0.5);
        if (end <= acc_w) end = acc_w + 1;                                   // every matrix gets at least one wave
        if (end > nwaves - (a.nmat - 1 - i)) end = nwaves - (a.nmat - 1 - i);
        d.m[i] = { (const char *) a.m[i].W, a.m[i].w_rs, (char *) a.m[i].dst, (const char *) a.m[i].resid, (int) a.m[i].nrows, a.m[i].type, end };
        acc_w = end;
    }
    const int nb = (int) (a.K / 256);
    if (nb % 16 != 0 || nb > 48) {                           // other widths: the TAIL instances (XB: image blocks per wave, rounded up to what is compiled)
        if (pair) { if (nb <= 16) mv1_go<8, 2, 2, true, true>(d, tm, (int) grid, st); else if (nb <= 32) mv1_go<8, 4, 2, true, true>(d, tm, (int) grid, st); else mv1_go<8, 8, 2, true, true>(d, tm, (int) grid, st); }
        else      { if (nb <= 16) mv1_go<16, 1, 1, false, true>(d, tm, (int) grid, st); else if (nb <= 32) mv1_go<16, 2, 1, false, true>(d, tm, (int) grid, st); else mv1_go<16, 4, 1, false, true>(d, tm, (int) grid, st); }
        return;
    }
    if (pair) { if (nb <= 16) mv1_go<8, 2, 2, true>(d, tm, (int) grid, st); else mv1_go<8, 6, 2, true>(d, tm, (int) grid, st); }
    else      { if (nb <= 16) mv1_go<16, 1, 1, false>(d, tm, (int) grid, st); else mv1_go<16, 3, 1, false>(d, tm, (int) grid, st); }
}

} // namespace mi

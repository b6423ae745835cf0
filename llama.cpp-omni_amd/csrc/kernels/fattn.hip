// fattn.hip -- FLASH_ATTN_EXT for gfx950 (wave64), decode / short-query form.
//
// reference: ggml_compute_forward_flash_attn_ext_f16, ggml-cpu/ops.cpp:7912-8148
//   per (query row, head):  Q -> f16 ; s_ic = (K_ic . Q) * scale (+softcap) + slope*mask_ic ; online softmax ;
//   V accumulated with expf(s - M) ; result / S ; written permuted as dst[DV, n_head, n_q, n_seq].
// Differences allowed by the reference's own tolerance (NMSE 5e-4, tests/test-backend-ops.cpp:5085):
// the CPU accumulates V in f16 when V is f16 (:8069-8083); this kernel keeps f32 accumulators.
//
// One workgroup (4 waves) owns R query vectors that share one KV head (the GQA group x a few query rows), so K and V
// are streamed once per group, not once per head.  The KV range is cut into 16-row granules dealt round-robin to the
// waves.  K and V go from HBM/L2 straight into registers -- no LDS staging:
//   scores : lane (r = lane>>2, dq = lane&3) holds dims [dq*D/4, (dq+1)*D/4) of granule row r (D/32 x 16-B loads, four lanes
//            cover one contiguous row), dots them with the R queries (f32 in LDS, broadcast reads) and the four partial sums
//            are folded with two DPP butterflies;
//   P.V    : lane owns D/64 output dims; the 16 V rows of the granule are fetched by 16 coalesced row loads issued
//            BEFORE the score arithmetic, so their latency hides under it.
// Granules whose mask is -inf for every query vector are skipped without touching K/V: a 256-padded KV view costs nothing
// at small depth.  The four waves' (M, S, acc) partials merge through LDS.  Optional epilogue: emit the Q8_K image of the
// output row (what the following wo MUL_MAT would otherwise quantise in its own launch).
#include "../kernels.hpp"
#include "fattn_dev.hpp"

namespace mi {

extern __shared__ __attribute__((aligned(16))) char fa_lds[];

template <int D, int R, int NW, bool PRE>
__global__ void __launch_bounds__(64 * NW) k_fattn_dec(const fa_dev a) {
    constexpr int DPL = D / 64;               // output dims per lane
    constexpr int GR  = 16;                   // KV rows per granule
    constexpr int KCH = D / 32;               // 16-B K chunks per lane (a quarter row)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane >> 2, dq = lane & 3;

    // ---- which query vectors does this workgroup own
    const int ngrp_h = (a.gq + a.hpw - 1) / a.hpw;
    const int nqb    = (a.nq + a.qpw - 1) / a.qpw;
    int b = (int) blockIdx.x;                                              // 32-bit index math: 64-bit div/mod are ~100-instruction sequences
    const int sp  = b % a.nsplit; b /= a.nsplit;                           // which slice of the KV range (long contexts, see launch_fa)
    const int qb  = b % nqb;    b /= nqb;
    const int hc  = b % ngrp_h; b /= ngrp_h;
    const int ikv = b % a.nhkv; const int is3 = b / a.nhkv;
    int r_q[R], r_h[R]; bool r_ok[R]; const uint16_t * mrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int hq = r % a.hpw, qq = r / a.hpw;
        r_q[r] = qb * a.qpw + qq;
        r_h[r] = ikv * a.gq + hc * a.hpw + hq;
        r_ok[r] = qq < a.qpw && r_q[r] < a.nq && (hc * a.hpw + hq) < a.gq;
        mrow[r] = (a.mask && r_ok[r]) ? (const uint16_t *) (a.mask + r_q[r] * a.mnb1 + (r_h[r] % (int) a.mne2) * a.mnb2 + (is3 % (int) a.mne3) * a.mnb3) : nullptr;
    }

    // ---- LDS carve-up
    float * qf   = (float *) fa_lds;                                                 // [R][D] queries (rounded through f16)
    float * pl   = (float *) (fa_lds + R * D * 4) + wave * (GR * R);                 // per-wave P[16][R]
    float * comb = (float *) (fa_lds + R * D * 4 + NW * GR * R * 4);                 // [NW][R][D+2] merge area, later [R][D] finals

    if (PRE) {
        // one token, one sequence, the whole GQA group in this workgroup (launcher-checked): tasks 0 = k head (norm, rope, store),
        // 1 = v head (store), 2 + r = q head r (norm, rope, into LDS); one wave per task
        const fa_pre & P = a.pre;
        const float posf = (float) P.pos[0];
        for (int task = wave; task < R + 2; task += NW) {
            if (task == 1) {
                const int64_t row = P.idx_is64 ? *(const int64_t *) P.vidx : (int64_t) *(const int32_t *) P.vidx;
                uint16_t * vr = (uint16_t *) (P.vcache + row * P.vc_rs) + ikv * D;
                const float * xv = (const float *) (P.vraw + ikv * P.v_hs);
                for (int e = lane; e < D; e += 64) vr[e] = f2h(xv[e]);
            } else {
                const bool isk = task == 0;
                const int  r   = task - 2;
                const int  hq  = ikv * a.gq + hc * a.hpw + (isk ? 0 : r);
                const char * xr = isk ? P.kraw + ikv * P.k_hs : P.qraw + hq * P.q_hs;
                float r0[1], r1[1]; int e0[1], e1[1]; bool act[1];
                norm_rope_wave<1>(xr, isk ? P.kw : P.qw, D, P.eps, posf, P.ff, P.rd, lane, r0, r1, e0, e1, act);
                if (act[0]) {
                    if (isk) {
                        const int64_t row = P.idx_is64 ? *(const int64_t *) P.kidx : (int64_t) *(const int32_t *) P.kidx;
                        uint16_t * kr = (uint16_t *) (P.kcache + row * P.kc_rs) + ikv * D;
                        kr[e0[0]] = f2h(r0[0]); kr[e1[0]] = f2h(r1[0]);
                    } else {
                        qf[r * D + e0[0]] = h2f(f2h(r0[0])); qf[r * D + e1[0]] = h2f(f2h(r1[0]));   // q_to_vec_dot rounding (ops.cpp:8040)
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            for (int d = threadIdx.x; d < D; d += 64 * NW) {
                float v = 0.0f;
                if (r_ok[r]) v = *(const float *) (a.q + d * 4 + r_q[r] * a.qnb1 + r_h[r] * a.qnb2 + is3 * a.qnb3);
                qf[r * D + d] = h2f(f2h(v));                                          // q_to_vec_dot: f32 -> f16 (ops.cpp:8040)
            }
        }
    }
    __syncthreads();                                       // (drains the k / v cache stores of the pre-stage before any wave reads the cache)

    float slope[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t h = (uint32_t) r_h[r];
        slope[r] = a.max_bias > 0.0f ? (h < a.n_head_log2 ? powf(a.m0, (float) (h + 1)) : powf(a.m1, (float) (2 * (h - a.n_head_log2) + 1))) : 1.0f;
    }

    float M[R], S[R], acc[R][DPL];
#pragma unroll
    for (int r = 0; r < R; ++r) { M[r] = -INFINITY; S[r] = 0.0f;
#pragma unroll
        for (int e = 0; e < DPL; ++e) acc[r][e] = 0.0f; }

    const char * kbase = a.k + ikv * a.knb2 + is3 * a.knb3;
    const char * vbase = a.v + ikv * a.vnb2 + is3 * a.vnb3;
    const int ngran = (a.nkv + GR - 1) / GR;

    const int gps = (ngran + a.nsplit - 1) / a.nsplit, g_lo = sp * gps, g_hi = g_lo + gps < ngran ? g_lo + gps : ngran;
    // Two granules in flight per wave: the mask row and the K / V rows of granule g + NW are requested before granule g is reduced, so a
    // wave with several granules (long contexts, KV split) pays the memory latency once instead of once per granule.
    struct granule { float mv[R]; u32x4 kk[KCH]; uint32_t vv[GR]; bool live; };
    auto fetch = [&](const int gi, granule & G) {
        const int  kv    = gi * GR + r16;
        const bool kv_ok = kv < a.nkv;
        bool any_live = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float m = 0.0f;
            if (mrow[r] && kv_ok) m = slope[r] * h2f(mrow[r][kv]);
            if (!kv_ok || !r_ok[r]) m = -INFINITY;
            G.mv[r] = m;
            any_live |= (m != -INFINITY);
        }
        G.live = __any(any_live);
        if (!G.live) return;                                          // whole granule masked for every query vector

        // ---- issue every load of the granule up front: a quarter K row per lane, then the 16 V rows (D/64 dims per lane)
        const int64_t kvc = kv_ok ? kv : a.nkv - 1;
#pragma unroll
        for (int c = 0; c < KCH; ++c) G.kk[c] = *(const u32x4 *) (kbase + kvc * a.knb1 + dq * (D / 2) + c * 16);
#pragma unroll
        for (int j = 0; j < GR; ++j) {
            int vr = gi * GR + j; vr = vr < a.nkv ? vr : a.nkv - 1;
            if (DPL == 2) G.vv[j] = *(const uint32_t *) (vbase + vr * a.vnb1 + lane * 4);
            else          G.vv[j] = *(const uint16_t *) (vbase + vr * a.vnb1 + lane * 2);
        }
    };
    auto reduce = [&](const granule & G) {
        // ---- scores
        float s[R];
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] = 0.0f;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            float kf[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { kf[2 * e] = h2f((uint16_t) (G.kk[c][e] & 0xffff)); kf[2 * e + 1] = h2f((uint16_t) (G.kk[c][e] >> 16)); }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const f32x4 q0 = *(const f32x4 *) (qf + r * D + dq * (D / 4) + c * 8);
                const f32x4 q1 = *(const f32x4 *) (qf + r * D + dq * (D / 4) + c * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[r] = fmaf(kf[e], q0[e], s[r]); s[r] = fmaf(kf[4 + e], q1[e], s[r]); }
            }
        }
        // ---- fold the four dim-quarters, scale / softcap / mask, online softmax across the granule
        float ms[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t = s[r];
            t += __shfl_xor(t, 1, 64);
            t += __shfl_xor(t, 2, 64);
            float v = t * a.scale;
            if (a.logit_softcap != 0.0f) v = a.logit_softcap * tanhf(v);
            v += G.mv[r];
            if (G.mv[r] == -INFINITY) v = -INFINITY;
            float tmax = v;                                             // max over the 16 rows (every row is replicated in its 4 dq lanes)
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
            const float Mn   = fmaxf(M[r], tmax);
            const float p    = (v == -INFINITY) ? 0.0f : expf(v - Mn);
            ms[r] = (M[r] == -INFINITY) ? 0.0f : expf(M[r] - Mn);
            if (Mn == -INFINITY) ms[r] = 1.0f;                         // nothing seen yet and nothing live: keep zeros
            float psum = p;
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) psum += __shfl_xor(psum, o, 64);
            S[r] = S[r] * ms[r] + psum;
            M[r] = Mn;
            if (dq == 0) pl[r16 * R + r] = p;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- P.V
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < DPL; ++e) acc[r][e] *= ms[r];
#pragma unroll
        for (int j = 0; j < GR; ++j) {
            float vf[DPL];
            if (DPL == 2) { vf[0] = h2f((uint16_t) (G.vv[j] & 0xffff)); vf[DPL - 1] = h2f((uint16_t) (G.vv[j] >> 16)); }
            else          { vf[0] = h2f((uint16_t) G.vv[j]); }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float p = pl[j * R + r];
                // masked cells are SKIPPED by the reference (ops.cpp:8047-8050), never multiplied: an uninitialised cache cell
                // holding inf/NaN must not leak in through 0 * x
#pragma unroll
                for (int e = 0; e < DPL; ++e) acc[r][e] = p != 0.0f ? fmaf(p, vf[e], acc[r][e]) : acc[r][e];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    constexpr bool PIPE = R * DPL <= 8;                  // the second granule's registers fit (<= 256 VGPRs, no scratch)
    if (PIPE) {
        granule A, B;
        int gi = g_lo + wave;
        if (gi < g_hi) {
            fetch(gi, A);
            for (;;) {
                int gn = gi + NW;
                const bool has_b = gn < g_hi;
                if (has_b) fetch(gn, B);
                if (A.live) reduce(A);
                if (!has_b) break;
                gi = gn; gn = gi + NW;
                const bool has_a = gn < g_hi;
                if (has_a) fetch(gn, A);
                if (B.live) reduce(B);
                if (!has_a) break;
                gi = gn;
            }
        }
    } else {
        for (int gi = g_lo + wave; gi < g_hi; gi += NW) {
            granule A;
            fetch(gi, A);
            if (A.live) reduce(A);
        }
    }

    // ---- merge the four waves' partial (M, S, acc)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float * cw = comb + (wave * R + r) * (D + 2);
#pragma unroll
        for (int e = 0; e < DPL; ++e) cw[lane * DPL + e] = acc[r][e];
        if (lane == 0) { cw[D] = M[r]; cw[D + 1] = S[r]; }
    }
    __syncthreads();
    float o[R][DPL];                                                   // finals of the query vectors this wave owns (r % NW == wave)
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[r][e] = 0.0f;
        if ((r % NW) != wave || !r_ok[r]) continue;
        float Mx = -INFINITY;
        for (int w = 0; w < NW; ++w) Mx = fmaxf(Mx, comb[(w * R + r) * (D + 2) + D]);
        float St = 0.0f;
        for (int w = 0; w < NW; ++w) {
            const float * cw = comb + (w * R + r) * (D + 2);
            const float Mw = cw[D];
            const float f  = (Mw == -INFINITY) ? 0.0f : expf(Mw - Mx);
            St += cw[D + 1] * f;
#pragma unroll
            for (int e = 0; e < DPL; ++e) o[r][e] += cw[lane * DPL + e] * f;
        }
        if (a.nsplit > 1) {                                           // partial state of this KV slice; k_fattn_merge finishes the row
            float * pr = a.part + ((((int64_t) is3 * a.nq + r_q[r]) * a.nh + r_h[r]) * a.nsplit + sp) * (D + 2);
#pragma unroll
            for (int e = 0; e < DPL; ++e) pr[lane * DPL + e] = o[r][e];
            if (lane == 0) { pr[D] = Mx; pr[D + 1] = St; }
            continue;
        }
        if (a.sinks) {                                                // ops.cpp:8116-8130
            const float sk = a.sinks[r_h[r]];
            if (sk > Mx) { const float f = expf(Mx - sk); St = St * f + 1.0f;
#pragma unroll
                for (int e = 0; e < DPL; ++e) o[r][e] *= f; }
            else St += expf(sk - Mx);
        }
        const float inv = St == 0.0f ? 0.0f : 1.0f / St;
        float * out = (float *) (a.dst + r_h[r] * a.dnb1 + r_q[r] * a.dnb2 + is3 * a.dnb3);
#pragma unroll
        for (int e = 0; e < DPL; ++e) { o[r][e] *= inv; out[lane * DPL + e] = o[r][e]; }
    }

    // ---- optional epilogue: Q8_K image of the output row (consumed by the following wo mat-vec)
    if (a.img) {
        __syncthreads();                                              // everyone is done reading the merge area
        float * fin = comb;                                           // [R][D]
#pragma unroll
        for (int r = 0; r < R; ++r)
            if ((r % NW) == wave && r_ok[r])
#pragma unroll
                for (int e = 0; e < DPL; ++e) fin[r * D + lane * DPL + e] = o[r][e];
        __syncthreads();
        const int nblk = a.qpw * a.hpw * D / 256;                     // launcher guarantees hpw*D % 256 == 0 and full head groups
        const int64_t Kimg = a.nh * D;
        for (int bq = wave; bq < nblk; bq += NW) {
            const int per_q = a.hpw * D / 256;
            const int qq = bq / per_q, bb = bq % per_q;
            const int64_t qrow = qb * a.qpw + qq;
            if (qrow >= a.nq) continue;
            const f32x4 v = *(const f32x4 *) (fin + (qq * a.hpw) * D + bb * 256 + 4 * lane);
            char * im = a.img + (is3 * a.nq + qrow) * a.img_bytes;
            const int64_t ib = ((ikv * a.gq + hc * a.hpw) * D) / 256 + bb;
            q8k_block_from_regs(v, lane, (int8_t *) im + ib * 256, (int16_t *) (im + Kimg) + ib * 16, (float *) (im + Kimg + Kimg / 8) + ib);
        }
    }
}

// second pass of the split decode (flash-decoding): one workgroup per (sequence, query row, 256-element block of the output row).  Its
// four waves each fold every fourth partial state of their heads (all loads of a wave independent: two round trips, not nsplit),
// park the result in LDS, and wave 0 folds those four, applies the sinks, normalises, stores -- and emits the Q8_K block of the row's
// image for wo
template <int D>
__global__ void __launch_bounds__(256) k_fattn_merge(const float * __restrict__ part, int nsplit, int nq, int nh, int nblk, const float * __restrict__ sinks,
                                                    char * __restrict__ dst, int64_t dnb1, int64_t dnb2, int64_t dnb3, char * __restrict__ img, size_t img_bytes) {
    __shared__ __attribute__((aligned(16))) float park[4][64][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int b = blockIdx.x;
    const int ib = b % nblk; b /= nblk;
    const int qrow = b % nq, is3 = b / nq;
    const int n = nh * D;
    const int e0 = ib * 256 + 4 * lane;
    const int head = e0 / D, d = e0 % D;
    f32x4 v = { 0.0f, 0.0f, 0.0f, 0.0f };
    float Mx = -INFINITY, St = 0.0f;
    if (e0 < n) {
        const float * base = part + (((int64_t) is3 * nq + qrow) * nh + head) * nsplit * (D + 2);
        constexpr int B = 16;                                         // partial states in flight per wave: clamped indices, no branches
        for (int s0 = wave; s0 < nsplit; s0 += 4 * B) {
            float Ms[B], Ss[B]; f32x4 p[B];
#pragma unroll
            for (int i = 0; i < B; ++i) {
                const int s = s0 + 4 * i, sc = s < nsplit ? s : s0;
                Ms[i] = base[sc * (D + 2) + D]; Ss[i] = base[sc * (D + 2) + D + 1];
                p[i]  = *(const f32x4 *) (base + sc * (D + 2) + d);
                if (s >= nsplit) Ms[i] = -INFINITY;
            }
            float Mn = Mx;
#pragma unroll
            for (int i = 0; i < B; ++i) Mn = fmaxf(Mn, Ms[i]);
            const float f0 = Mx == -INFINITY ? 0.0f : expf(Mx - Mn);
            St *= f0; v *= f0; Mx = Mn;
#pragma unroll
            for (int i = 0; i < B; ++i) {
                const float f = Ms[i] == -INFINITY ? 0.0f : expf(Ms[i] - Mx);
                St += Ss[i] * f;
                v += p[i] * f;
            }
        }
    }
    float * pk = &park[wave][lane][0];
    *(f32x4 *) pk = v; pk[4] = Mx; pk[5] = St;
    __syncthreads();
    if (wave != 0) return;
    if (e0 < n) {
        float Mw[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) { Mw[w] = park[w][lane][4]; Mx = fmaxf(Mx, Mw[w]); }
        v = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f }; St = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = Mw[w] == -INFINITY ? 0.0f : expf(Mw[w] - Mx);
            St += park[w][lane][5] * f;
            v += *(const f32x4 *) &park[w][lane][0] * f;
        }
        if (sinks) {                                                  // ops.cpp:8116-8130
            const float sk = sinks[head];
            if (sk > Mx) { const float f = expf(Mx - sk); St = St * f + 1.0f; v *= f; }
            else St += expf(sk - Mx);
        }
        v *= St == 0.0f ? 0.0f : 1.0f / St;
        *(f32x4 *) (dst + head * dnb1 + qrow * dnb2 + is3 * dnb3 + d * 4) = v;
    }
    if (img) {                                                        // (launcher: n % 256 == 0)
        char * im = img + ((int64_t) is3 * nq + qrow) * img_bytes;
        q8k_block_from_regs(v, lane, (int8_t *) im + ib * 256, (int16_t *) (im + n) + ib * 16, (float *) (im + n + n / 8) + ib);
    }
}

// long contexts: the streaming decode kernel has one workgroup per KV-head group, which at depth 32k is 8 workgroups reading 4 GB per
// token; cut the KV range into slices of >= 256 rows so that a few hundred workgroups share it (flash-decoding) and merge afterwards
static int fa_decode_nsplit(const fattn_args & f) {
    static const bool off = getenv("MI355X_FA_NO_DECODE_SPLIT") != nullptr;
    const int64_t nkv = f.k.ne[1];
    if (off || nkv < 1024 || f.pre) return 1;
    int64_t s = nkv / 256;
    const int64_t groups = f.k.ne[2] * f.q.ne[3] * f.q.ne[1];        // workgroups without a split (at least)
    while (s > 1 && s * groups > 1024) s /= 2;
    return (int) (s < 1 ? 1 : (s > 64 ? 64 : s));
}

// A few query tokens whose (token, head) pairs fit 32-column tiles go through the matrix cores (k_fattn_gqa, fattn_mma.hip): one token,
// one token of each of a few sequences, a short draft.  The streaming kernel above remains for the shapes that do not fit (more than
// 32 heads per KV head, per-head masks) and as the cross-check ("fattn_gqa" option off).
static bool g_gqa_enabled = true;
void fattn_set_gqa(bool on) { g_gqa_enabled = on; }
struct gqa_plan { bool ok; int qpw, nsplit, nw; };
static gqa_plan fa_gqa_plan(const fattn_args & f) {
    static const bool off = getenv("MI355X_FA_NO_DECODE_MMA") != nullptr;
    static const int  direct_tiles = getenv("MI355X_FA_GQA_DIRECT_TILES") ? atoi(getenv("MI355X_FA_GQA_DIRECT_TILES")) : 8;
    gqa_plan p = { false, 1, 1, 4 };
    const int64_t D = f.q.ne[0], nq = f.q.ne[1], nkv = f.k.ne[1], gq = f.k.ne[2] > 0 ? f.q.ne[2] / f.k.ne[2] : 0;
    static const int64_t max_nq = getenv("MI355X_FA_GQA_MAX_NQ") ? atoll(getenv("MI355X_FA_GQA_MAX_NQ")) : 32;   // more tokens: the prefill kernel
    if (off || !g_gqa_enabled || (D != 64 && D != 128) || f.v.ne[0] != D || nq < 1 || nq > max_nq || gq < 1 || gq > 32 || nkv < 1) return p;
    if (nq > 8 && (f.out16 || !fattn_mma_ok(nkv))) return p;
    if (f.q.ne[2] != gq * f.k.ne[2]) return p;
    if (f.mask && f.mask->ne[2] != 1) return p;                       // (a per-head mask would need per-lane mask rows)
    // one token over a shallow cache stays on the streaming kernel: it skips the dead granules of a padded cache view before touching
    // K / V and measures ~3 us per layer faster there (tg128: 398 vs 381 tok/s); everything else is faster here
    static const int stream_tiles = getenv("MI355X_FA_STREAM_TILES") ? atoi(getenv("MI355X_FA_STREAM_TILES")) : 8;
    if (nq == 1 && (nkv + 31) / 32 <= stream_tiles) return p;
    p.ok  = true;
    p.qpw = (int) (nq < 32 / gq ? nq : 32 / gq);
    const int64_t groups = f.k.ne[2] * f.q.ne[3] * ((nq + p.qpw - 1) / p.qpw);
    const int64_t ntile  = (nkv + 31) / 32;
    if (ntile <= direct_tiles) {                                      // shallow: one workgroup per group finishes the rows itself
        static const int force_nw = getenv("MI355X_FA_GQA_NW") ? atoi(getenv("MI355X_FA_GQA_NW")) : 0;
        p.nsplit = 1; p.nw = force_nw ? force_nw : (f.pre ? 8 : 4);   // pre-stage: gq + 2 wave tasks in one round
    } else {                                                          // deep: slices of >= 4 tiles, two workgroups per CU, merge pass
        static const int deep_nw  = getenv("MI355X_FA_GQA_DEEP_NW") ? atoi(getenv("MI355X_FA_GQA_DEEP_NW")) : 4;
        static const int deep_wgs = getenv("MI355X_FA_GQA_WGS") ? atoi(getenv("MI355X_FA_GQA_WGS")) : 512;
        int64_t s = deep_wgs / groups;
        if (s > (ntile + deep_nw - 1) / deep_nw) s = (ntile + deep_nw - 1) / deep_nw;
        p.nsplit = (int) (s < 2 ? 2 : s); p.nw = deep_nw;
    }
    return p;
}

template <int D, int R, int NW>
static size_t fa_lds_bytes() { return (size_t) R * D * 4 + NW * 16 * R * 4 + NW * R * (D + 2) * 4; }

static bool fa_use_mma(const fattn_args & f) {
    if (f.v_transposed) return true;                                  // (fattn_sm_prefill_ok checked the shape)
    if (!(f.q.ne[1] > 8 && !f.img && (f.q.ne[0] == 64 || f.q.ne[0] == 128) && fattn_mma_ok(f.k.ne[1]))) return false;
    fattn_args g = f; g.pre = nullptr; g.out16 = nullptr;
    return f.out16 != nullptr || !fa_gqa_plan(g).ok;              // 9 .. 32 tokens: the decode-shape kernel, unless the f16 rows of a GEMM consumer are wanted
}
bool fattn_uses_mma(const fattn_args & f) { return fa_use_mma(f); }
bool fattn_sm_prefill_ok(const fattn_args & f) {
    static const bool off = getenv("MI355X_NO_ATTN_SM_PREFILL") != nullptr;
    const int64_t D = f.q.ne[0];
    // more than 32 rows: prompt / encoder batches.  A parallel-decode step of up to 32 sequences keeps the node-by-node arithmetic (normalised probabilities rounded to f16 --
    // the reference's own order), which keeps its greedy ids identical to the CPU backend's even on near-ties (tests/test_llama_dropin.py)
    return !off && f.v_transposed && (D == 64 || D == 128) && f.q.ne[1] > 32 && f.kv_type == GGML_TYPE_F16 && !f.img && !f.pre && fattn_mma_ok(f.k.ne[1]) && f.max_bias == 0.0f &&
           f.v.ne[0] == f.k.ne[1] && f.v.ne[1] == D && f.v.nb[0] == 2 && f.k.nb[0] == 2 && f.k.ne[2] > 0 && f.q.ne[2] % f.k.ne[2] == 0 && f.v.ne[2] == f.k.ne[2];
}
size_t fattn_map_bytes(int64_t nq, int64_t nkv, int64_t mne2, int64_t mne3);
size_t fattn_map_bytes_host(int64_t nq, int64_t nkv) { return fattn_map_bytes(nq, nkv, 1, 1); }
size_t fattn_scratch_bytes(const fattn_args & f) {
    if (!fa_use_mma(f)) {                                             // decode kernels: partial rows of the KV split
        fattn_args g = f; g.pre = nullptr;
        const gqa_plan gp = fa_gqa_plan(g);
        int ns = gp.ok ? gp.nsplit : fa_decode_nsplit(g);
        if (f.q.ne[1] == 1 && f.q.ne[3] == 1 && f.k.ne[1] > 256 && f.k.ne[1] <= 8192) { const int n1 = fattn_one_nsplit(f); if (n1 > ns) ns = n1; }   // (the one-token kernel's slices)
        return ns > 1 ? (size_t) (f.q.ne[1] * f.q.ne[3] * f.q.ne[2]) * (size_t) ns * (size_t) (f.q.ne[0] + 2) * 4 : 0;
    }
    if (!f.mask) return 0;
    return fattn_map_bytes(f.q.ne[1], f.k.ne[1], f.mask->ne[2], f.mask->ne[3]);
}

// choose R (query vectors per workgroup): cover the GQA group first, then extra query rows
static void fa_split(const fa_dev & a, int & R, int & hpw, int & qpw) {
    if (a.gq >= 8) R = 8; else if (a.gq >= 4) R = (a.nq > 1 ? 8 : 4); else if (a.gq >= 2) R = (a.nq > 2 ? 8 : (a.nq > 1 ? 4 : 2)); else R = (a.nq >= 8 ? 8 : (a.nq >= 4 ? 4 : (a.nq >= 2 ? 2 : 1)));
    hpw = a.gq < R ? a.gq : R;
    qpw = R / hpw; if (qpw < 1) qpw = 1;
}

// the q/k/v pre-stage needs: the streaming decode kernel, one token of one sequence, the whole GQA group of a KV head in ONE
// workgroup (a second workgroup would read the cache row the first one is still writing), head size <= 128
bool fattn_pre_ok(const fattn_args & f) {
    const int D = (int) f.q.ne[0];
    if ((D != 64 && D != 128) || f.q.ne[1] != 1 || f.q.ne[3] != 1 || f.k.ne[3] != 1) return false;
    if (fa_gqa_plan(f).ok) return true;                               // (with a KV split only the slice that owns the new cache row stores it)
    fa_dev a; a.nq = 1; a.gq = (int) (f.q.ne[2] / f.k.ne[2]);
    int R, hpw, qpw; fa_split(a, R, hpw, qpw);
    fattn_args g = f; g.pre = nullptr;
    if (fa_decode_nsplit(g) > 1) return false;                        // another workgroup would read the cache row this one is still writing
    return hpw == a.gq && qpw == 1 && R == hpw;
}

bool fattn_can_emit_image(const fattn_args & f) {
    const int D = (int) f.q.ne[0];
    if (D != 64 && D != 128) return false;
    fa_dev a; a.nq = f.q.ne[1]; a.gq = (int) (f.q.ne[2] / f.k.ne[2]);
    const gqa_plan gp = fa_gqa_plan(f);                               // finishing workgroup: whole 256-element blocks of the row; else the merge pass
    if (gp.ok) return (f.q.ne[2] * D) % 256 == 0 && (gp.nsplit > 1 || (a.gq * D) % 256 == 0);
    int R, hpw, qpw; fa_split(a, R, hpw, qpw);
    return (hpw * D) % 256 == 0 && a.gq % hpw == 0 && (f.q.ne[2] * D) % 256 == 0;
}

template <int D>
static void launch_fa(const fa_dev & a0, hipStream_t st) {
    fa_dev a = a0;
    int R; fa_split(a, R, a.hpw, a.qpw);
    const int64_t ngrp_h = (a.gq + a.hpw - 1) / a.hpw;
    const int64_t nqb    = (a.nq + a.qpw - 1) / a.qpw;
    const int64_t nblk   = nqb * ngrp_h * a.nhkv * a.ns * a.nsplit;
    dim3 grid((unsigned) nblk);
    // waves per workgroup: one 16-row granule per wave for short contexts (pure latency), 4 waves when there are many workgroups anyway
    const bool wide = a.nkv > 64 && nblk <= 1024;
#define FA_GO3(RR, NWW, PP)                                                                                            \
    do {                                                                                                               \
        const size_t lds = fa_lds_bytes<D, RR, NWW>();                                                                 \
        if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *) k_fattn_dec<D, RR, NWW, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
        k_fattn_dec<D, RR, NWW, PP><<<grid, dim3(64 * NWW), lds, st>>>(a);                                             \
    } while (0)
#define FA_GO2(RR, NWW) do { if (a.pre.qraw) FA_GO3(RR, NWW, true); else FA_GO3(RR, NWW, false); } while (0)
#define FA_GO(RR) do { if (wide) FA_GO2(RR, 8); else FA_GO2(RR, 4); } while (0)     // (16 waves would cap VGPRs at 128 and spill)
    switch (R) { case 1: FA_GO(1); break; case 2: FA_GO(2); break; case 4: FA_GO(4); break; default: FA_GO(8); break; }
#undef FA_GO
#undef FA_GO2
#undef FA_GO3
}

void flash_attn_ext_f16(const fattn_args & f, hipStream_t st) {
    fa_dev a;
    a.q = (const char *) f.q.p; a.k = (const char *) f.k.p; a.v = (const char *) f.v.p;
    a.mask = f.mask ? (const char *) f.mask->p : nullptr; a.sinks = f.sinks; a.dst = (char *) f.dst.p;
    a.nq = f.q.ne[1]; a.nh = f.q.ne[2]; a.ns = f.q.ne[3]; a.nhkv = f.k.ne[2]; a.nkv = f.k.ne[1];
    if (a.nq * a.nh * a.ns == 0) return;
    a.qnb1 = f.q.nb[1]; a.qnb2 = f.q.nb[2]; a.qnb3 = f.q.nb[3];
    a.knb1 = f.k.nb[1]; a.knb2 = f.k.nb[2]; a.knb3 = f.k.nb[3];
    a.vnb1 = f.v.nb[1]; a.vnb2 = f.v.nb[2]; a.vnb3 = f.v.nb[3];
    a.vt = 0;
    if (f.v_transposed) { const uintptr_t al = (uintptr_t) f.v.p | f.v.nb[1] | f.v.nb[2] | f.v.nb[3]; a.vt = al % 16 == 0 ? 16 : (al % 4 == 0 ? 4 : 2); }
    if (f.mask) { a.mnb1 = f.mask->nb[1]; a.mnb2 = f.mask->nb[2]; a.mnb3 = f.mask->nb[3]; a.mne2 = f.mask->ne[2]; a.mne3 = f.mask->ne[3]; }
    else { a.mnb1 = a.mnb2 = a.mnb3 = 0; a.mne2 = a.mne3 = 1; }
    a.dnb1 = f.dst.nb[1]; a.dnb2 = f.dst.nb[2]; a.dnb3 = f.dst.nb[3];
    a.scale = f.scale; a.max_bias = f.max_bias; a.logit_softcap = f.logit_softcap;
    if (a.logit_softcap != 0.0f) a.scale /= a.logit_softcap;                       // ops.cpp:7985-7987
    a.n_head_log2 = 1u << (uint32_t) floorf(log2f((float) a.nh));
    a.m0 = powf(2.0f, -(a.max_bias) / a.n_head_log2);
    a.m1 = powf(2.0f, -(a.max_bias / 2.0f) / a.n_head_log2);
    a.gq = (int) (a.nh / a.nhkv);
    a.hpw = a.qpw = 1;
    a.img = (char *) f.img; a.img_bytes = f.img ? q8k_image_bytes(a.nh * f.q.ne[0]) : 0;
    a.pre.qraw = nullptr;
    a.out16 = (char *) f.out16; a.out16_rs = (int64_t) f.out16_rs; a.write_f32 = f.write_f32 ? 1 : 0;
    if (f.pre) {
        if (!fattn_pre_ok(f)) { fprintf(stderr, "[mi355x] flash_attn: q/k/v pre-stage requested for an unsupported shape\n"); abort(); }
        const fattn_pre & p = *f.pre;
        a.pre = { (const char *) p.qraw, p.q_hs, (const char *) p.kraw, p.k_hs, (const char *) p.vraw, p.v_hs, p.qw, p.kw, p.pos, p.ff, p.eps, make_rope_dev(p.rp),
                  (char *) p.kcache, p.kc_rs, (char *) p.vcache, p.vc_rs, (const char *) p.kidx, (const char *) p.vidx, p.idx_is64 };
    }
    if (!f.v_transposed && ((f.q.ne[0] != 64 && f.q.ne[0] != 128) || f.v.ne[0] != f.q.ne[0] || f.kv_type != GGML_TYPE_F16)) {     // other head sizes / cache types: the generic kernel (no pre-stage, no images)
        if (f.pre || f.img || f.out16 || !fattn_any_ok(f.q.ne[0], f.v.ne[0])) { fprintf(stderr, "[mi355x] flash_attn: head size %d / %d with a fused stage\n", (int) f.q.ne[0], (int) f.v.ne[0]); abort(); }
        a.nsplit = 1; a.part = nullptr; a.tile_map = nullptr; a.map_nqb = 0;
        flash_attn_ext_any(a, (int) f.q.ne[0], (int) f.v.ne[0], f.kv_type, st);
        return;
    }
    if (f.pre && f.rope_tab && f.gs_parts && fattn_gs_ok(f)) {       // one token, <= 256 rows, the consumer folds the slices: one workgroup per (KV head, slice)
        a.nsplit = 1; a.part = nullptr; a.tile_map = nullptr; a.map_nqb = 0;
        flash_attn_gs(a, (int) f.q.ne[0], f.rope_tab, f.gs_parts, st);
        return;
    }
    if (f.pre && f.rope_tab && fattn_one_ok(f)) {                    // one token, up to 4096 cache rows: the latency-optimised one-token kernel
        const int ns1 = fattn_one_nsplit(f);
        const size_t need = (size_t) a.nh * (size_t) ns1 * (size_t) (f.q.ne[0] + 2) * 4;
        if (ns1 == 1 || (f.scratch && f.scratch_bytes >= need)) {
            static const bool no_inl = getenv("MI355X_FA_ONE_MERGE_LAUNCH") != nullptr;
            a.nsplit = ns1; a.part = ns1 > 1 ? (float *) f.scratch : nullptr; a.tile_map = nullptr; a.map_nqb = 0;
            a.cnt = (ns1 > 1 && !no_inl && a.nh <= 1024) ? f.counters : nullptr;
            flash_attn_one(a, (int) f.q.ne[0], f.rope_tab, st);
            if (ns1 > 1 && !a.cnt) {                                 // fold the slices' partial rows, apply the sinks, normalise
                const int nblk = (int) ((a.nh * f.q.ne[0] + 255) / 256);
                if (f.q.ne[0] == 64) k_fattn_merge<64><<<dim3((unsigned) nblk), dim3(256), 0, st>>>(a.part, ns1, 1, a.nh, nblk, a.sinks, a.dst, a.dnb1, a.dnb2, a.dnb3, nullptr, 0);
                else                 k_fattn_merge<128><<<dim3((unsigned) nblk), dim3(256), 0, st>>>(a.part, ns1, 1, a.nh, nblk, a.sinks, a.dst, a.dnb1, a.dnb2, a.dnb3, nullptr, 0);
            }
            return;
        }
    }
    // batches of query rows go to the matrix-core kernel (fattn_mma.hip); single / few rows stay on the streaming decode kernel
    if (fa_use_mma(f)) {
        a.tile_map = nullptr; a.map_nqb = (a.nq + 31) / 32;
        if (f.mask) {
            if (!f.scratch || f.scratch_bytes < fattn_scratch_bytes(f)) { fprintf(stderr, "[mi355x] flash_attn: mask tile map scratch missing\n"); abort(); }
            if (!f.map_valid) fattn_mask_map(a, (uint8_t *) f.scratch, st);
            a.tile_map = (const uint8_t *) f.scratch;
        }
        flash_attn_ext_mma(a, (int) f.q.ne[0], st);
        return;
    }
    a.nsplit = 1; a.part = nullptr; a.tile_map = nullptr; a.map_nqb = 0;
    const gqa_plan gp = fa_gqa_plan(f);
    const int nsplit = gp.ok ? gp.nsplit : fa_decode_nsplit(f);
    char * img_final = nullptr;
    if (nsplit > 1 && f.scratch && f.scratch_bytes >= fattn_scratch_bytes(f) && (!a.img || (a.nh * f.q.ne[0]) % 256 == 0)) {
        a.nsplit = nsplit; a.part = (float *) f.scratch;
        img_final = a.img; a.img = nullptr;                           // the image is emitted by the merge pass
    }
    if (gp.ok) {
        if (a.nsplit == 1 && a.img && (a.gq * f.q.ne[0]) % 256 != 0) { fprintf(stderr, "[mi355x] flash_attn: image epilogue needs the KV split workspace for this shape\n"); abort(); }
        a.qpw = gp.qpw;
        flash_attn_ext_gqa(a, (int) f.q.ne[0], gp.nw, st);
    } else switch ((int) f.q.ne[0]) {
        case 64:  launch_fa<64>(a, st); break;
        case 128: launch_fa<128>(a, st); break;
        default: fprintf(stderr, "[mi355x] flash_attn: unsupported head size %d\n", (int) f.q.ne[0]); abort();
    }
    if (a.nsplit > 1) {
        const int nblk = (int) ((a.nh * f.q.ne[0] + 255) / 256);
        const dim3 grid((unsigned) (a.nq * a.ns * nblk));
        if (f.q.ne[0] == 64) k_fattn_merge<64><<<grid, dim3(256), 0, st>>>(a.part, a.nsplit, a.nq, a.nh, nblk, a.sinks, a.dst, a.dnb1, a.dnb2, a.dnb3, img_final, a.img_bytes);
        else                 k_fattn_merge<128><<<grid, dim3(256), 0, st>>>(a.part, a.nsplit, a.nq, a.nh, nblk, a.sinks, a.dst, a.dnb1, a.dnb2, a.dnb3, img_final, a.img_bytes);
    }
}

} // namespace mi

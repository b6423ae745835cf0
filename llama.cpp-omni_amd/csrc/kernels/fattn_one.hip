// fattn_one.hip -- FLASH_ATTN_EXT of ONE token of ONE sequence over a shallow cache view (<= 256 rows: what llama-bench tg128 and the
// first 256 tokens of every chat see), with the layer's q chain, k chain + cache store and v cache store folded in (the decode pre-stage,
// fattn_dev.hpp fa_pre).  gfx950 / wave64.
//
// reference: ggml_compute_forward_flash_attn_ext_f16, ggml-cpu/ops.cpp:7912-8148 (scores, soft-max, V accumulation, sinks), the q / k chains
// RMS_NORM -> MUL -> ROPE of llm_build_qwen3 (src/llama-model.cpp:9331-9349: ops.cpp:3517-3566, :5534-5720) and llama_kv_cache::cpy_k / cpy_v.
//
// Why a third decode-attention kernel: at this shape the op moves < 1 MB, so its cost is its serial latency chain.  k_fattn_dec (fattn.hip)
// runs it on 8 workgroups as: raw loads -> norm (ds_bpermute butterflies) -> rope (sequential theta products, sincosf) -> cache store -> barrier ->
// mask -> K / V loads -> scores (bpermute) -> ... -> merge -> Q8_K image: 14 us per layer, a quarter of the decode step.  Here
//   * one workgroup per QUERY head (32 instead of 8), each recomputing the (cheap) k / v head of its group in LDS, so nothing waits for a
//     cache row another workgroup stores; only the first head of a group writes the cache;
//   * EVERY load of the kernel -- raw q / k / v, norm weights, (cos, sin) table, mask, the 256 K rows and the 256 V rows -- is requested in the
//     first few hundred cycles; K and V sit in registers (4 waves x 1 per SIMD: 512 VGPRs each) when the scores / weights are ready;
//   * reductions on the DPP network; soft-max in two passes over scores parked in LDS (no running rescale);
//   * V is split by output dims across the waves (no cross-wave merge of partial rows), rows past the last visible one are skipped;
//   * (cos, sin) of the token come from a table computed once per graph (k_rope_table), not per layer and head.
// The output is the plain f32 row; the following wo mat-vec quantises it in its own prologue (mmv1.hip).
#include "../kernels.hpp"
#include "fattn_dev.hpp"

namespace mi {

template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ float dppf_old(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
static __device__ __forceinline__ float wave_max_f32(float v) {                // any sign; lanes outside a row_bcast's mask contribute -inf
    v = fmaxf(v, dppf_old<0xB1, 0xf>(-INFINITY, v));
    v = fmaxf(v, dppf_old<0x4E, 0xf>(-INFINITY, v));
    v = fmaxf(v, dppf_old<0x141, 0xf>(-INFINITY, v));
    v = fmaxf(v, dppf_old<0x140, 0xf>(-INFINITY, v));
    v = fmaxf(v, dppf_old<0x142, 0xa>(-INFINITY, v));
    v = fmaxf(v, dppf_old<0x143, 0xc>(-INFINITY, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ double dpp_f64o(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double wave_sum_f64o(double v) {
    v += dpp_f64o<0xB1, 0xf>(v); v += dpp_f64o<0x4E, 0xf>(v); v += dpp_f64o<0x141, 0xf>(v); v += dpp_f64o<0x140, 0xf>(v);
    v += dpp_f64o<0x142, 0xa>(v); v += dpp_f64o<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

constexpr int FA1_NKV = 256;                      // cache rows a workgroup holds in registers

template <int D>
__global__ void __launch_bounds__(256) k_fattn_one(const fa_dev a, const float * __restrict__ tab) {
    constexpr int KCH = D / 32;                   // 16-B K chunks per lane (a quarter row)
    constexpr int NG  = FA1_NKV / 16 / 4;         // 16-row granules per wave
    constexpr int DPW = D / 4;                    // output dims per wave
    constexpr int LPR = DPW / 2;                  // lanes per V row piece (2 dims per lane)
    constexpr int RPI = 64 / LPR;                 // V rows per load instruction
    constexpr int NJ  = FA1_NKV / RPI;            // V loads per lane
    constexpr int HALF = D / 2;
    __shared__ __attribute__((aligned(16))) float qf[D], kc[D], vc[D], sc[FA1_NKV], pl[4][FA1_NKV];

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = blockIdx.x, ikv = h / a.gq;
    const fa_pre & P = a.pre;
    const int nkv = a.nkv < FA1_NKV ? a.nkv : FA1_NKV;

    // ---------------------------------------------------------------- 1. request everything
    const int64_t krow = P.idx_is64 ? *(const int64_t *) P.kidx : (int64_t) *(const int32_t *) P.kidx;     // cache rows of the new token (scalar loads)
    const int64_t vrow = P.idx_is64 ? *(const int64_t *) P.vidx : (int64_t) *(const int32_t *) P.vidx;
    // q head (wave 0) / k head (wave 1): rotation pair ip = lane
    const bool neox = P.rd.mode & GGML_ROPE_TYPE_NEOX;
    const bool act  = lane < HALF;
    const int  e0 = neox ? lane : 2 * lane, e1 = neox ? lane + HALF : 2 * lane + 1;
    float x0 = 0, x1 = 0, w0 = 0, w1 = 0, tc = 1, ts = 0, xv[D / 64];
    if (wave < 2 && act) {
        const char * xr = wave == 0 ? P.qraw + h * P.q_hs : P.kraw + ikv * P.k_hs;
        const float * w = wave == 0 ? P.qw : P.kw;
        x0 = *(const float *) (xr + e0 * 4); x1 = *(const float *) (xr + e1 * 4);
        w0 = w[e0]; w1 = w[e1];
        tc = tab[2 * lane]; ts = tab[2 * lane + 1];
    }
    if (wave == 2) {
#pragma unroll
        for (int i = 0; i < D / 64; ++i) xv[i] = *(const float *) (P.vraw + ikv * P.v_hs + (lane + 64 * i) * 4);
    }
    // mask row of this head: lane owns rows lane + 64 i
    const uint16_t * mrow = a.mask ? (const uint16_t *) (a.mask + (h % (int) a.mne2) * a.mnb2) : nullptr;
    uint16_t mraw[FA1_NKV / 64];
#pragma unroll
    for (int i = 0; i < FA1_NKV / 64; ++i) { const int kv = lane + 64 * i; mraw[i] = (mrow && kv < nkv) ? mrow[kv] : (uint16_t) 0; }
    // K: granule g = wave + 4 i, row g * 16 + (lane >> 2), lane quarter dq = lane & 3
    const int r16 = lane >> 2, dq = lane & 3;
    const char * kbase = a.k + ikv * a.knb2, * vbase = a.v + ikv * a.vnb2;
    u32x4 kk[NG][KCH];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        int row = (wave + 4 * i) * 16 + r16; row = row < nkv ? row : nkv - 1;
#pragma unroll
        for (int c = 0; c < KCH; ++c) kk[i][c] = *(const u32x4 *) (kbase + (int64_t) row * a.knb1 + dq * (D / 2) + c * 16);
    }
    // V: this wave's DPW output dims of every row; lane (rs, dp): row j * RPI + rs, dims wave * DPW + 2 dp, +1
    const int rs = lane / LPR, dp = lane % LPR;
    uint32_t vv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int row = j * RPI + rs; row = row < nkv ? row : nkv - 1;
        vv[j] = *(const uint32_t *) (vbase + (int64_t) row * a.vnb1 + (wave * DPW + 2 * dp) * 2);
    }

    // ---------------------------------------------------------------- 2. q chain, k chain + store, v store (norm_rope_dev.hpp arithmetic)
    if (wave < 2) {
        double ss = (double) (x0 * x0) + (double) (x1 * x1);
        ss = wave_sum_f64o(ss);
        const float mean  = (float) (ss / (double) D);
        const float scale = 1.0f / sqrtf(mean + P.eps);
        const float v0 = (x0 * scale) * w0, v1 = (x1 * scale) * w1;
        const float r0 = v0 * tc - v1 * ts, r1 = v0 * ts + v1 * tc;
        if (act) {
            const uint16_t h0 = f2h(r0), h1 = f2h(r1);
            if (wave == 0) { qf[e0] = h2f(h0); qf[e1] = h2f(h1); }                 // q_to_vec_dot rounding (ops.cpp:8040)
            else {
                kc[e0] = h2f(h0); kc[e1] = h2f(h1);
                if (h % a.gq == 0) { uint16_t * kr = (uint16_t *) (P.kcache + krow * P.kc_rs) + ikv * D; kr[e0] = h0; kr[e1] = h1; }
            }
        }
    } else if (wave == 2) {
#pragma unroll
        for (int i = 0; i < D / 64; ++i) {
            const uint16_t hv = f2h(xv[i]);
            vc[lane + 64 * i] = h2f(hv);
            if (h % a.gq == 0) ((uint16_t *) (P.vcache + vrow * P.vc_rs) + ikv * D)[lane + 64 * i] = hv;
        }
    }
    // mask values + liveness (every wave computes the same)
    const float slope = a.max_bias > 0.0f ? ((uint32_t) h < a.n_head_log2 ? powf(a.m0, (float) (h + 1)) : powf(a.m1, (float) (2 * (h - (int) a.n_head_log2) + 1))) : 1.0f;
    float mv[FA1_NKV / 64]; int n_live = 0;
#pragma unroll
    for (int i = 0; i < FA1_NKV / 64; ++i) {
        const int kv = lane + 64 * i;
        float m = mrow ? slope * h2f(mraw[i]) : 0.0f;
        if (kv >= nkv) m = -INFINITY;
        mv[i] = m;
        const unsigned long long live = __ballot(m != -INFINITY);
        if (live) n_live = 64 * i + 64 - __builtin_clzll(live);                    // 1 + the last visible row
    }
    __syncthreads();

    // ---------------------------------------------------------------- 3. scores
    {
        float qr[D / 4];
#pragma unroll
        for (int c = 0; c < D / 16; ++c) { const f32x4 t = *(const f32x4 *) (qf + dq * (D / 4) + 4 * c); qr[4 * c] = t[0]; qr[4 * c + 1] = t[1]; qr[4 * c + 2] = t[2]; qr[4 * c + 3] = t[3]; }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int g = wave + 4 * i;
            if (g * 16 >= n_live) continue;                                          // (wave-uniform) nothing visible in this granule
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < KCH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s = fmaf(h2f((uint16_t) (kk[i][c][e] & 0xffff)), qr[c * 8 + 2 * e], s);
                    s = fmaf(h2f((uint16_t) (kk[i][c][e] >> 16)), qr[c * 8 + 2 * e + 1], s);
                }
            s += dppf_old<0xB1, 0xf>(0.0f, s);                                       // fold the four dim-quarters (quad butterflies)
            s += dppf_old<0x4E, 0xf>(0.0f, s);
            const int row = g * 16 + r16;
            if (dq == 0 && row < nkv && row != (int) krow) sc[row] = s;              // (the new token's row is not in the cache yet: below)
        }
        if (wave == 3) {                                                             // score of the new token itself, from the k head in LDS
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < D / 64; ++i) s = fmaf(kc[lane + 64 * i], qf[lane + 64 * i], s);
            s = wave_sum_f32(s);
            if (lane == 0 && krow >= 0 && krow < nkv) sc[krow] = s;
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- 4. soft-max weights (every wave, own copy, rows regrouped for the V pass)
    float S = 0.0f, M;
    {
        float sv[FA1_NKV / 64];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < FA1_NKV / 64; ++i) {
            const int kv = lane + 64 * i;
            float v = kv < n_live ? sc[kv] * a.scale : 0.0f;
            if (a.logit_softcap != 0.0f) v = a.logit_softcap * tanhf(v);
            v += mv[i];
            if (mv[i] == -INFINITY) v = -INFINITY;
            sv[i] = v; mx = fmaxf(mx, v);
        }
        M = wave_max_f32(mx);
#pragma unroll
        for (int i = 0; i < FA1_NKV / 64; ++i) {
            const int kv = lane + 64 * i;
            const float p = sv[i] == -INFINITY ? 0.0f : expf(sv[i] - M);
            S += p;
            pl[wave][(kv % RPI) * NJ + kv / RPI] = p;                               // transposed: the V pass reads four consecutive j at once
        }
        S = wave_sum_f32(S);
    }
    // the new token's V row is in LDS, not in the registers: take its weight out of the table
    float pcur = 0.0f;
    const bool vin = vrow >= 0 && vrow < nkv;
    if (vin) { pcur = pl[wave][((int) vrow % RPI) * NJ + (int) vrow / RPI]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (vin && lane == 0) pl[wave][((int) vrow % RPI) * NJ + (int) vrow / RPI] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---------------------------------------------------------------- 5. P.V for this wave's dims
    float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
    for (int j4 = 0; j4 < NJ / 4; ++j4) {
        if (j4 * 4 * RPI >= n_live) continue;                                        // (wave-uniform)
        const f32x4 p4 = *(const f32x4 *) (&pl[wave][rs * NJ + 4 * j4]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float p = p4[t];
            // masked cells are SKIPPED by the reference (ops.cpp:8047-8050), never multiplied: an uninitialised cache cell holding inf / NaN
            // must not leak in through 0 * x
            const uint32_t w = p != 0.0f ? vv[j4 * 4 + t] : 0u;
            acc0 = fmaf(p, h2f((uint16_t) (w & 0xffff)), acc0);
            acc1 = fmaf(p, h2f((uint16_t) (w >> 16)), acc1);
        }
    }
    if (rs == 0) { acc0 = fmaf(pcur, vc[wave * DPW + 2 * dp], acc0); acc1 = fmaf(pcur, vc[wave * DPW + 2 * dp + 1], acc1); }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) { acc0 += __shfl_xor(acc0, o, 64); acc1 += __shfl_xor(acc1, o, 64); }
    if (a.sinks) {                                                                   // ops.cpp:8116-8130
        const float sk = a.sinks[h];
        if (sk > M) { const float f = expf(M - sk); S = S * f + 1.0f; acc0 *= f; acc1 *= f; }
        else S += expf(sk - M);
    }
    const float inv = S == 0.0f ? 0.0f : 1.0f / S;
    if (rs == 0) {
        float * out = (float *) (a.dst + h * a.dnb1) + wave * DPW + 2 * dp;
        out[0] = acc0 * inv; out[1] = acc1 * inv;
    }
}

// one token of one sequence, q / k / v pre-stage, <= 256 cache rows, f16 mask shared by the heads of a token (or per head), D 64 / 128
bool fattn_one_ok(const fattn_args & f) {
    static const bool off = getenv("MI355X_FA_NO_ONE") != nullptr;
    const int64_t D = f.q.ne[0];
    if (off || !f.pre || (D != 64 && D != 128) || f.v.ne[0] != D || f.q.ne[1] != 1 || f.q.ne[3] != 1 || f.k.ne[3] != 1) return false;
    if (f.k.ne[1] < 1 || f.k.ne[1] > FA1_NKV || f.k.ne[2] < 1 || f.q.ne[2] % f.k.ne[2] != 0) return false;
    if (f.k.nb[1] % 16 != 0 || f.v.nb[1] % 4 != 0 || ((uintptr_t) f.k.p & 15) != 0 || ((uintptr_t) f.v.p & 3) != 0 || f.k.nb[2] % 16 != 0 || f.v.nb[2] % 4 != 0) return false;
    if (f.dst.nb[0] != 4 || f.dst.nb[1] % 8 != 0 || ((uintptr_t) f.dst.p & 7) != 0) return false;
    if (f.img || f.out16) return false;
    return true;
}

void flash_attn_one(const fa_dev & a, int D, const float * rope_tab, hipStream_t st) {
    if (!rope_tab) { fprintf(stderr, "[mi355x] flash_attn_one: the (cos, sin) table of the token is missing\n"); abort(); }
    const dim3 grid((unsigned) a.nh);
    if (D == 64) k_fattn_one<64><<<grid, dim3(256), 0, st>>>(a, rope_tab);
    else         k_fattn_one<128><<<grid, dim3(256), 0, st>>>(a, rope_tab);
}

} // namespace mi

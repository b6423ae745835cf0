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
#include <type_traits>

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

#ifdef FA1_TRACE           // measurement builds (tools/fa1_lab.hip): per-wave s_memrealtime stamps (100 MHz) of the stages of k_fattn_one, written once at the end
__device__ unsigned long long * fa1_trace_buf = nullptr;
#define FA1_STAMP_DECL uint32_t tr_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }
#define FA1_STAMP(i) do { tr_[i] = (uint32_t) __builtin_amdgcn_s_memrealtime(); asm volatile("" : "+s"(tr_[i]) :: "memory"); } while (0)
#define FA1_STAMP_FLUSH do { if (fa1_trace_buf) { const int l_ = threadIdx.x & 63; uint32_t v_ = 0; _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) if (l_ == i_) v_ = tr_[i_]; \
    if (l_ < 8) fa1_trace_buf[(size_t) (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + l_] = v_; } } while (0)
#else
#define FA1_STAMP_DECL
#define FA1_STAMP(i) do { } while (0)
#define FA1_STAMP_FLUSH do { } while (0)
#endif
constexpr int FA1_NKV = 256;                      // cache rows a workgroup holds in registers
constexpr int FA1_MAX_SPLIT = 32;                 // 256-row slices per head handled by this kernel + k_fattn_merge (deeper: k_fattn_gqa's MFMA tiles)

// compact argument block (one scalar-load burst): the general fa_dev is ~450 bytes and reading it piecemeal costs round trips
struct fa1_dev {
    const char * qraw, * kraw, * vraw; const float * qw, * kw, * tab;        // pre-stage inputs
    char * kcache, * vcache; const char * kidx, * vidx;                      // cache tables + row index of the new token (i64 or i32)
    const char * k, * v, * mask; const float * sinks; char * dst;
    int q_hs, k_hs, v_hs, kc_rs, vc_rs, knb1, knb2, vnb1, vnb2, mnb2, mne2, dnb1;
    int nkv, gq, neox, n_head_log2;
    int n_head, nkvh_log2;                                                   // query heads of the launch; log2 of the KV head count when it is a power of two (else -1): the head index without a division
    int vidx_st, vidx_n;                                                     // soft-max path: bytes per v scatter index, number of indices
    int has_norm;                                                            // q / k chains start with RMS_NORM * w (Qwen3) or are ROPE only (llama architecture)
    int nsplit; float * part;                                                // KV range cut into 256-row slices, one workgroup per (head, slice): partial (O, M, S) rows
    unsigned * cnt;                                                          // per-head arrival counters (zero between launches); null: k_fattn_merge folds the slices
    float eps, scale, max_bias, logit_softcap, m0, m1;
};

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t fa1_rsrc(const void * p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *) p, (short) 0, bytes, 0x00020000);
}

// The first fourteen kernel-argument dwords -- what the first vector loads need: the raw q / k / v pointers, their head strides, the group size and the split count --
// are plain scalar arguments in front of the block, so that they arrive pre-loaded in SGPRs (Makefile: -amdgpu-kernarg-preload-count for this object) and the
// raw loads are issued before the scalar loads of the rest of the block have returned.
#define FA1_LEAD_PARAMS const char * qraw_, const char * kraw_, const char * vraw_, int q_hs_, int k_hs_, int v_hs_, int gq_, int nsplit_, int n_head_, int nkvh_log2_, int neox_
#define FA1_LEAD_ARGS(a) (a).qraw, (a).kraw, (a).vraw, (a).q_hs, (a).k_hs, (a).v_hs, (a).gq, (a).nsplit, (a).n_head, (a).nkvh_log2, (a).neox
// head index of a workgroup: XCD-aware order (see k_fattn_one); one slice -- every cache view up to 256 rows -- has no division in front of its loads
#define FA1_HEAD_INDEX \
    int bh = (int) blockIdx.x, sp = 0; \
    if (nsplit_ != 1) { sp = bh / n_head_; bh -= sp * n_head_; } \
    const int row0 = sp * FA1_NKV; \
    int ikv, hq; \
    if (nkvh_log2_ >= 0) { ikv = bh & ((1 << nkvh_log2_) - 1); hq = bh >> nkvh_log2_; } \
    else { const int nkvh_ = n_head_ / gq_; hq = bh / nkvh_; ikv = bh - hq * nkvh_; } \
    const int h = ikv * gq_ + hq;
template <int D>
__global__ void __launch_bounds__(256) k_fattn_one(FA1_LEAD_PARAMS, const fa1_dev a) {
    constexpr int KCH = D / 32;                   // 16-B K chunks per lane (a quarter row)
    constexpr int NG  = FA1_NKV / 16 / 4;         // 16-row granules per wave
    constexpr int DPW = D / 4;                    // output dims per wave
    constexpr int LPR = DPW / 2;                  // lanes per V row piece (2 dims per lane)
    constexpr int RPI = 64 / LPR;                 // V rows per load instruction
    constexpr int NJ  = FA1_NKV / RPI;            // V loads per lane
    constexpr int HALF = D / 2;
    __shared__ __attribute__((aligned(16))) float qf[D], kc[D], vc[D], sc[FA1_NKV], pl[4][FA1_NKV];

    FA1_STAMP_DECL; FA1_STAMP(0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware head order: workgroups go round-robin over the 8 XCDs (each with its own L2), so the gq heads that share one K / V head are
    // given workgroup ids that are congruent modulo the number of KV heads -- with 8 KV heads one XCD fetches each K / V head once
    // (rocprofv3 FETCH_SIZE: 4.4 MB -> ~1.1 MB per launch at n_kv 256)
    // deeper caches (257 .. 4096 rows): grid = heads x slices, slice sp takes rows [256 sp, 256 sp + 256) and leaves the partial state of its rows
    // (unnormalised O, running max M, sum S) for k_fattn_merge -- flash-decoding with this kernel's latency chain instead of the MFMA kernel's
    FA1_HEAD_INDEX
    const int nkv = a.nkv - row0 < FA1_NKV ? a.nkv - row0 : FA1_NKV;              // rows of this slice

    // ---------------------------------------------------------------- 1. request everything: one burst of vector loads, no wait in between
    // (buffer loads with exact bounds instead of branches or clamps: an out-of-range element reads as zero.  The row indices are fetched
    //  as VECTOR loads too: a scalar load of them would sit in front of every later kernarg read -- one more serial round trip.)
    const bool act = lane < HALF;
    const int  e0 = neox_ ? lane : 2 * lane, e1 = neox_ ? lane + HALF : 2 * lane + 1;
    const __amdgpu_buffer_rsrc_t xrs = fa1_rsrc(wave == 0 ? qraw_ + h * q_hs_ : (wave == 1 ? kraw_ + ikv * k_hs_ : vraw_ + ikv * v_hs_), wave < 3 ? D * 4 : 0);
    // waves 0 / 1: rotation pair `lane` of the q / k head (elements e0, e1); wave 2: elements lane, lane + 64 of the v head
    const uint32_t xo0 = wave < 2 ? (act ? e0 * 4 : D * 4) : lane * 4, xo1 = wave < 2 ? (act ? e1 * 4 : D * 4) : (lane + 64) * 4;
    const float x0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, xo0, 0, 0)), x1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, xo1, 0, 0));
    __builtin_amdgcn_sched_barrier(0);                             // (everything above needs only the pre-loaded arguments: the raw rows are requested before the first wait on the block's scalar loads)
    int wave_b = wave; asm volatile("" : "+s"(wave_b));            // (an opaque copy: keeps the selects below out of the control flow of the select above, which must not wait for the block)
    const __amdgpu_buffer_rsrc_t wrs = fa1_rsrc(wave_b == 0 ? a.qw : a.kw, (wave_b < 2 && a.has_norm) ? D * 4 : 0);
    const __amdgpu_buffer_rsrc_t trs = fa1_rsrc(a.tab, wave_b < 2 ? D * 4 : 0);
    const float w0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, xo0, 0, 0)), w1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, xo1, 0, 0));
    const u32x2 tcs = __builtin_amdgcn_raw_buffer_load_b64(trs, act ? lane * 8 : D * 4, 0, 0);
    // mask row of this head: lane owns rows lane + 64 i
    const __amdgpu_buffer_rsrc_t mrs = fa1_rsrc(a.mask ? a.mask + (h % a.mne2) * a.mnb2 + row0 * 2 : a.mask, a.mask ? nkv * 2 : 0);
    uint16_t mraw[FA1_NKV / 64];
#pragma unroll
    for (int i = 0; i < FA1_NKV / 64; ++i) mraw[i] = __builtin_amdgcn_raw_buffer_load_b16(mrs, (lane + 64 * i) * 2, 0, 0);
    const int krow_g = (int) __builtin_amdgcn_raw_buffer_load_b32(fa1_rsrc(a.kidx, 4), 0, 0, 0);   // (little-endian low half of an i64 index)
    const int vrow_g = (int) __builtin_amdgcn_raw_buffer_load_b32(fa1_rsrc(a.vidx, 4), 0, 0, 0);
    const int krow = krow_g - row0, vrow = vrow_g - row0;                                          // the new token's row relative to this slice
    const bool owner = a.nsplit == 1 || (krow_g >= row0 && krow_g < row0 + FA1_NKV) || (sp == 0 && (krow_g < 0 || krow_g >= a.nkv));   // the slice that stores the new cache rows
    // K: granule g = wave + 4 i, row g * 16 + (lane >> 2), lane quarter dq = lane & 3
    const int r16 = lane >> 2, dq = lane & 3;
    const __amdgpu_buffer_rsrc_t krs = fa1_rsrc(a.k + ikv * a.knb2 + (int64_t) row0 * a.knb1, (nkv - 1) * a.knb1 + D * 2);
    const __amdgpu_buffer_rsrc_t vrs = fa1_rsrc(a.v + ikv * a.vnb2 + (int64_t) row0 * a.vnb1, (nkv - 1) * a.vnb1 + D * 2);
    const uint32_t kvo = (uint32_t) r16 * (uint32_t) a.knb1 + (uint32_t) dq * (D / 2);
    u32x4 kk[NG][KCH];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const uint32_t so = (uint32_t) ((wave + 4 * i) * 16) * (uint32_t) a.knb1;                      // wave-uniform
#pragma unroll
        for (int c = 0; c < KCH; ++c) kk[i][c] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvo + 16u * c, so, 0);
    }
    // V: this wave's DPW output dims of every row; lane (rs, dp): row j * RPI + rs, dims wave * DPW + 2 dp, +1
    const int rs = lane / LPR, dp = lane % LPR;
    const uint32_t vvo = (uint32_t) rs * (uint32_t) a.vnb1 + (uint32_t) (wave * DPW + 2 * dp) * 2u;
    uint32_t vv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) vv[j] = __builtin_amdgcn_raw_buffer_load_b32(vrs, vvo, (uint32_t) (j * RPI) * (uint32_t) a.vnb1, 0);

    FA1_STAMP(1);
    // ---------------------------------------------------------------- 2. q chain, k chain + store, v store (norm_rope_dev.hpp arithmetic)
    if (wave < 2) {
        const float tc = __uint_as_float(tcs[0]), ts = __uint_as_float(tcs[1]);
        double ss = (double) (x0 * x0) + (double) (x1 * x1);
        ss = wave_sum_f64o(ss);
        const float mean  = (float) (ss * (1.0 / D));                                // D is a power of two: the same double as ss / D
        const float scale = 1.0f / sqrtf(mean + a.eps);
        const float v0 = a.has_norm ? (x0 * scale) * w0 : x0, v1 = a.has_norm ? (x1 * scale) * w1 : x1;      // (llama-architecture chains: ROPE only)
        const float r0 = v0 * tc - v1 * ts, r1 = v0 * ts + v1 * tc;
        if (act) {
            const uint16_t h0 = f2h(r0), h1 = f2h(r1);
            if (wave == 0) { qf[e0] = h2f(h0); qf[e1] = h2f(h1); }                 // q_to_vec_dot rounding (ops.cpp:8040)
            else {
                kc[e0] = h2f(h0); kc[e1] = h2f(h1);
                if (hq == 0 && owner) { uint16_t * kr = (uint16_t *) (a.kcache + (int64_t) krow_g * a.kc_rs) + ikv * D; kr[e0] = h0; kr[e1] = h1; }
            }
        }
    } else if (wave == 2) {
        const uint16_t hv0 = f2h(x0), hv1 = f2h(x1);
        vc[lane] = h2f(hv0);
        if (D > 64) vc[lane + 64] = h2f(hv1);
        if (hq == 0 && owner) {
            uint16_t * vr = (uint16_t *) (a.vcache + (int64_t) vrow_g * a.vc_rs) + ikv * D;
            vr[lane] = hv0;
            if (D > 64) vr[lane + 64] = hv1;
        }
    }
    FA1_STAMP(2);
    // mask values + liveness (every wave computes the same)
    const float slope = a.max_bias > 0.0f ? (h < a.n_head_log2 ? powf(a.m0, (float) (h + 1)) : powf(a.m1, (float) (2 * (h - a.n_head_log2) + 1))) : 1.0f;
    float mv[FA1_NKV / 64]; int n_live = 0;
#pragma unroll
    for (int i = 0; i < FA1_NKV / 64; ++i) {
        const int kv = lane + 64 * i;
        float m = a.mask ? slope * h2f(mraw[i]) : 0.0f;
        if (kv >= nkv) m = -INFINITY;
        mv[i] = m;
        const unsigned long long live = __ballot(m != -INFINITY);
        if (live) n_live = 64 * i + 64 - __builtin_clzll(live);                    // 1 + the last visible row
    }
    FA1_STAMP(3);
    __syncthreads();
    FA1_STAMP(4);

    // ---------------------------------------------------------------- 3. scores
    {
        // q is already rounded to f16 (q_to_vec_dot): keep this lane's quarter as packed halves and use v_dot2_f32_f16 (f32 accumulate)
        f16x2 qh[D / 8];
#pragma unroll
        for (int c = 0; c < D / 16; ++c) {
            const f32x4 t = *(const f32x4 *) (qf + dq * (D / 4) + 4 * c);
            qh[2 * c] = f16x2{ (_Float16) t[0], (_Float16) t[1] }; qh[2 * c + 1] = f16x2{ (_Float16) t[2], (_Float16) t[3] };
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int g = wave + 4 * i;
            if (g * 16 >= n_live) continue;                                          // (wave-uniform) nothing visible in this granule
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < KCH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f16x2 kh; const uint32_t kw = kk[i][c][e]; __builtin_memcpy(&kh, &kw, 4);
                    s = __builtin_amdgcn_fdot2(kh, qh[c * 4 + e], s, false);
                }
            s += dppf_old<0xB1, 0xf>(0.0f, s);                                       // fold the four dim-quarters (quad butterflies)
            s += dppf_old<0x4E, 0xf>(0.0f, s);
            const int row = g * 16 + r16;
            if (dq == 0 && row < nkv && row != krow) sc[row] = s;              // (the new token's row is not in the cache yet: below)
        }
        if (wave == 3) {                                                             // score of the new token itself, from the k head in LDS
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < D / 64; ++i) s = fmaf(kc[lane + 64 * i], qf[lane + 64 * i], s);
            s = wave_sum_f32(s);
            if (lane == 0 && krow >= 0 && krow < nkv) sc[krow] = s;
        }
    }
    FA1_STAMP(5);
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
    if (vin) { pcur = pl[wave][(vrow % RPI) * NJ + vrow / RPI]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (vin && lane == 0) pl[wave][(vrow % RPI) * NJ + vrow / RPI] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    FA1_STAMP(6);
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
    if (a.sinks && a.nsplit == 1) {                                                  // ops.cpp:8116-8130 (with a KV split: in the merge pass)
        const float sk = a.sinks[h];
        if (sk > M) { const float f = expf(M - sk); S = S * f + 1.0f; acc0 *= f; acc1 *= f; }
        else S += expf(sk - M);
    }
    if (a.nsplit > 1) {                                                              // partial state of this slice
        // written through to the coherence point (sc1): the slices of a head sit on one XCD by construction (same workgroup id modulo 8), but
        // that is placement, not a guarantee -- the merge below must be right wherever the last arriver runs
        const __amdgpu_buffer_rsrc_t prs = fa1_rsrc(a.part + (int64_t) h * a.nsplit * (D + 2), a.nsplit * (D + 2) * 4);
        const uint32_t po = (uint32_t) sp * (D + 2) * 4u;
        if (rs == 0) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc0), prs, po + (uint32_t) (wave * DPW + 2 * dp) * 4u, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc1), prs, po + (uint32_t) (wave * DPW + 2 * dp + 1) * 4u, 0, 16);
        }
        if (threadIdx.x == 0) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(M), prs, po + (uint32_t) D * 4u, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(S), prs, po + (uint32_t) (D + 1) * 4u, 0, 16);
        }
        if (!a.cnt) return;                                                          // k_fattn_merge finishes the rows
        // last arriver folds the head's slices (flash-decoding without the second launch): every wave drains its stores, one lane takes a ticket
        __shared__ unsigned last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) last = __hip_atomic_fetch_add(a.cnt + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned) (a.nsplit - 1) ? 1u : 0u;
        __syncthreads();
        if (!last) return;
        if (threadIdx.x < D) {                                                       // thread d: output dim d of head h
            const int d = threadIdx.x;
            float Mx = -INFINITY, St = 0.0f, o = 0.0f;
            // every slice's state requested at once (clamped indices, no branches); 16-wide for the common depths, 32-wide beyond
            auto fold = [&](auto WIDTH) {
                constexpr int W = decltype(WIDTH)::value;
                float Ms[W], Ss[W], os[W];
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const uint32_t so = (uint32_t) ((i < a.nsplit ? i : 0) * (D + 2)) * 4u;
                    Ms[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prs, so + (uint32_t) D * 4u, 0, 16));
                    Ss[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prs, so + (uint32_t) (D + 1) * 4u, 0, 16));
                    os[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prs, so + (uint32_t) d * 4u, 0, 16));
                }
#pragma unroll
                for (int i = 0; i < W; ++i) { if (i >= a.nsplit) Ms[i] = -INFINITY; Mx = fmaxf(Mx, Ms[i]); }
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const float f = Ms[i] == -INFINITY ? 0.0f : expf(Ms[i] - Mx);
                    St += Ss[i] * f; o += os[i] * f;
                }
            };
            if (a.nsplit <= 16) fold(std::integral_constant<int, 16>()); else fold(std::integral_constant<int, FA1_MAX_SPLIT>());
            if (a.sinks) {                                                           // ops.cpp:8116-8130
                const float sk = a.sinks[h];
                if (sk > Mx) { const float f = expf(Mx - sk); St = St * f + 1.0f; o *= f; }
                else St += expf(sk - Mx);
            }
            ((float *) (a.dst + h * a.dnb1))[d] = St == 0.0f ? 0.0f : o * (1.0f / St);
        }
        if (threadIdx.x == 0) __hip_atomic_store(a.cnt + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
        return;
    }
    const float inv = S == 0.0f ? 0.0f : 1.0f / S;
    if (rs == 0) {
        float * out = (float *) (a.dst + h * a.dnb1) + wave * DPW + 2 * dp;
        out[0] = acc0 * inv; out[1] = acc1 * inv;
    }
    FA1_STAMP(7);
    FA1_STAMP_FLUSH;
}

// ================================================================================================= group-slice form (round 6)
// The same one-token step over <= 256 cache rows with ONE workgroup per (KV head, 64-row slice of the view) instead of one per query head:
// k_fattn_one's 32 workgroups each pull the WHOLE K / V head of their group (128 KB through one CU's vector-memory path: ~80 poorly coalesced
// load instructions per wave, "every load requested" 2.0 us after the launch starts, the last V row used at 4.5 -- tools/fa1_lab.hip); here a
// workgroup pulls the 64 rows of its slice once for the gq = 4 query heads of the group (16 + 16 KB as 32 LDS-DMA instructions of 1 KiB, four rows
// of 256 contiguous bytes each) and leaves, per head, the UNNORMALISED partial output of its rows with their running maximum and sum:
//     parts[slice][h * D + d] = sum_rows exp(s_row - M) v_row[d]      ms[slice][h] = (M, S)
// The slices are merged by the prologue of the wo mat-vec launch that consumes them (mmv2.hip, PARTS: out = sum_s f_s O_s / sum_s f_s S_s,
// f_s = exp(M_s - max M)), which every one of its workgroups runs anyway in front of the Q8_K quantiser -- no second launch, no arrival counters.
// Arithmetic per row as k_fattn_one (f16-rounded q / k, v_dot2_f32_f16 scores, f32 soft-max and V accumulation), reference ops.cpp:7912-8148.
// Restrictions (fattn_gs_ok): D = 128, gq = 4, one sequence, f16 mask shared by the heads or per head, no sinks / ALiBi / soft-cap, contiguous heads.
constexpr int FGS_NSL = 4;                        // slices of the 256-row view
constexpr int FGS_RS  = FA1_NKV / FGS_NSL;        // rows per slice
constexpr int FGS_W   = 8;                        // waves per workgroup
typedef const __attribute__((address_space(4))) char * fgs_kp;
template <typename T> static __device__ __forceinline__ T fgs_ld(fgs_kp kp, size_t off) {      // a volatile scalar load of one kernel-argument field
    typedef const volatile __attribute__((address_space(4))) T * P;
    return *(P) (kp + off);
}
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t fa1_rsrc_u(const void * p, int bytes) {      // wave-uniform by construction (an inline-asm "s" operand must be provably scalar)
    const uint64_t a = (uint64_t) (uintptr_t) p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) a), hi = __builtin_amdgcn_readfirstlane((uint32_t) (a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *) (uintptr_t) (((uint64_t) hi << 32) | lo), (short) 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
#define FGS_LEAD_PARAMS const char * qraw_, const float * qw_, const char * k_, int tab_off16_ /* (cos, sin) table - qraw, x 16 B */, int kidx_off8_ /* row index of the new token - qraw, x 8 B */, uint32_t kv_off16_ /* k | v raw rows: signed 16-bit offsets from qraw, x 16 B */, int mask_off16_ /* mask - qraw, x 16 B; 0: none */, int kw_off_, int v_off16_, float eps_, uint32_t pk_      /* 14 dwords, no padding: pre-loaded */
template <int D>
__global__ void __launch_bounds__(64 * FGS_W) k_fattn_gs(FGS_LEAD_PARAMS, const fa1_dev a) {
    static_assert(D == 128, "one 256-byte f16 row per (cache row, KV head)");
    constexpr int GQ = 4, HALF = D / 2, RS = FGS_RS, RPW = RS / FGS_W;      // rows per wave in the score / P.V passes
    __shared__ __attribute__((aligned(16))) char kt[RS * D * 2], vt[RS * D * 2];
    __shared__ __attribute__((aligned(16))) float qf[GQ][D], kc[D], vc[D], pl[GQ][RS], pcur_s[GQ], ms_s[GQ][2];
    __shared__ __attribute__((aligned(16))) uint16_t q16[GQ][D];
    FA1_STAMP_DECL; FA1_STAMP(0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // pk_: [7:0] KV heads, [8] neox, [9] norm, [18:10] rows of the view (<= 256), [30:19] v's row index relative to k's, x 8 B (signed), [31] FAR
    const int nkvh = (int) (pk_ & 0xffu), neox = (int) ((pk_ >> 8) & 1u), has_norm = (int) ((pk_ >> 9) & 1u), nkv_all = (int) ((pk_ >> 10) & 0x1ffu);
    // FAR (bit 31, round 6): some operand lies beyond the reach of its pre-loaded offset (the graph's buffers can be tens of GB apart in a 288 GB address space -- after a model
    // with resident F16 images was loaded between two of them, for one): every such pointer is then taken from the argument block instead, at the price of its ~0.85 us
    const bool far_ = (pk_ >> 31) != 0u;
    const fgs_kp kp0 = (fgs_kp) __builtin_amdgcn_kernarg_segment_ptr() + 56;
    const float * tab_ = (const float *) (qraw_ + (int64_t) tab_off16_ * 16);
    const char * kraw_p = qraw_ + (int) (int16_t) (kv_off16_ & 0xffffu) * 16, * vraw_p = qraw_ + (int) (int16_t) (kv_off16_ >> 16) * 16;
    const char * vview_p = k_ + (int64_t) v_off16_ * 16, * kw_p = (const char *) qw_ + kw_off_;
    const char * mask_p = mask_off16_ ? qraw_ + (int64_t) mask_off16_ * 16 : nullptr;
    const char * kidx_p = qraw_ + (int64_t) kidx_off8_ * 8, * vidx_p = kidx_p + (int64_t) ((int32_t) (pk_ << 1) >> 20) * 8;      // ([30:19]: v's index relative to k's, signed, x 8 B)
    if (far_) {
        tab_ = fgs_ld<const float *>(kp0, offsetof(fa1_dev, tab)); kraw_p = fgs_ld<const char *>(kp0, offsetof(fa1_dev, kraw)); vraw_p = fgs_ld<const char *>(kp0, offsetof(fa1_dev, vraw));
        vview_p = fgs_ld<const char *>(kp0, offsetof(fa1_dev, v)); kw_p = (const char *) fgs_ld<const float *>(kp0, offsetof(fa1_dev, kw)); mask_p = fgs_ld<const char *>(kp0, offsetof(fa1_dev, mask));
        kidx_p = fgs_ld<const char *>(kp0, offsetof(fa1_dev, kidx)); vidx_p = fgs_ld<const char *>(kp0, offsetof(fa1_dev, vidx));
    }
    const int knb1_ = nkvh * D * 2;                               // cache rows hold the KV heads back to back (fattn_gs_ok); eps rides in the pre-loaded scalars instead: the
                                                                  // chains then need nothing of the argument block (its scalar loads come back ~0.85 us into the launch)
    const int sp = __builtin_amdgcn_readfirstlane((int) blockIdx.x / nkvh), g = (int) blockIdx.x - sp * nkvh;        // workgroups of one KV head are congruent modulo the KV head count: one XCD's L2 serves the group
    const int row0 = sp * RS;
    const int nkv = nkv_all - row0 < RS ? (nkv_all - row0 > 0 ? nkv_all - row0 : 0) : RS;      // rows of this slice inside the view
    // ---------------------------------------------------------------- 1. K / V slice by LDS-DMA (pre-loaded arguments only), then the raw rows
    // lane -> (row of the instruction's four, 16-byte chunk of the row); exact bounds: rows past the view read nothing
    {
        const uint32_t vo = (uint32_t) (lane >> 4) * (uint32_t) knb1_ + (uint32_t) (lane & 15) * 16u;
        const int nb = nkv > 0 ? (nkv - 1) * knb1_ + D * 2 : 0;
        const __amdgpu_buffer_rsrc_t krs = fa1_rsrc_u(k_ + g * (D * 2) + (int64_t) row0 * knb1_, nb);
        const __amdgpu_buffer_rsrc_t vrs = fa1_rsrc_u(vview_p + g * (D * 2) + (int64_t) row0 * knb1_, nb);
        const uint32_t kl = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) void *) kt, vl = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) void *) vt;
#pragma unroll
        for (int i = 0; i < RS / 4 / FGS_W; ++i) {                        // instruction j of 16: rows 4 j .. 4 j + 3
            const int j = wave + FGS_W * i;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(__builtin_amdgcn_readfirstlane(kl + (uint32_t) j * 1024u)), "v"(vo), "s"(krs), "s"(__builtin_amdgcn_readfirstlane((uint32_t) (4 * j) * (uint32_t) knb1_)) : "memory", "m0");
        }
#pragma unroll
        for (int i = 0; i < RS / 4 / FGS_W; ++i) {
            const int j = wave + FGS_W * i;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(__builtin_amdgcn_readfirstlane(vl + (uint32_t) j * 1024u)), "v"(vo), "s"(vrs), "s"(__builtin_amdgcn_readfirstlane((uint32_t) (4 * j) * (uint32_t) knb1_)) : "memory", "m0");
        }
    }
    FA1_STAMP(4);
    // waves 0 .. 3: the q chain of head 4 g + wave; wave 4: the k chain; wave 5: the v head (elements lane, lane + 64)
    const int  e0 = neox ? lane : 2 * lane, e1 = neox ? lane + HALF : 2 * lane + 1;
    const char * xb = wave < GQ ? qraw_ + (g * GQ + wave) * (D * 4) : (wave == GQ ? kraw_p + g * (D * 4) : vraw_p + g * (D * 4));
    const __amdgpu_buffer_rsrc_t xrs = fa1_rsrc(xb, wave <= GQ + 1 ? D * 4 : 0);
    const uint32_t xo0 = wave <= GQ ? e0 * 4 : lane * 4, xo1 = wave <= GQ ? e1 * 4 : (lane + 64) * 4;
    const float x0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, xo0, 0, 0)), x1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, xo1, 0, 0));
    const __amdgpu_buffer_rsrc_t wrs = fa1_rsrc(wave < GQ ? (const char *) qw_ : kw_p, (wave <= GQ && has_norm) ? D * 4 : 0);
    const __amdgpu_buffer_rsrc_t trs = fa1_rsrc(tab_, wave <= GQ ? D * 4 : 0);
    const float w0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, xo0, 0, 0)), w1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, xo1, 0, 0));
    const u32x2 tcs = __builtin_amdgcn_raw_buffer_load_b64(trs, lane * 8, 0, 0);
    // the mask row (one f16 row shared by the heads: fattn_gs_ok), lane = row of the slice, exact bounds (rows past the view are set to -inf in the soft-max): its address
    // is pre-loaded too -- behind the argument block it would be requested 0.9 us into the launch and the soft-max would wait for it
    const char * a_mask = mask_p;
    const __amdgpu_buffer_rsrc_t mrs = fa1_rsrc(a_mask ? a_mask + row0 * 2 : a_mask, (a_mask && wave < GQ) ? nkv * 2 : 0);
    const uint16_t mraw = __builtin_amdgcn_raw_buffer_load_b16(mrs, lane * 2, 0, 0);
    // ... and the new token's row index (k's and v's: scalar loads -- constant address space + a uniform address; the indices were written before the launch)
    typedef const volatile __attribute__((address_space(4))) int * fgs_cint;
    const int krow_u = *(fgs_cint) (uintptr_t) kidx_p, vrow_u = *(fgs_cint) (uintptr_t) vidx_p;
    __builtin_amdgcn_sched_barrier(0);                             // (everything above needs only the pre-loaded arguments)
    // Every field of the argument block the kernel uses is REQUESTED here (volatile scalar loads through the kernarg segment pointer: issued in program order, waited for
    // at the first use) and none is used before barrier 1: the block's loads come back ~0.85 us into the launch, and the chains -- raw rows, norm weights, (cos, sin), eps:
    // all behind pre-loaded scalars -- do not wait for them.  (Left to the compiler each field would be loaded where it is first used: one round trip per phase.)
    const fgs_kp kp = (fgs_kp) __builtin_amdgcn_kernarg_segment_ptr() + 56;             // `a` behind the 14 pre-loaded dwords
#define FGS_ARG(T, f) fgs_ld<T>(kp, offsetof(fa1_dev, f))
    const int a_nh = FGS_ARG(int, n_head), a_kc_rs = FGS_ARG(int, kc_rs), a_vc_rs = FGS_ARG(int, vc_rs);
    const float a_scale = FGS_ARG(float, scale);
    float * a_part = FGS_ARG(float *, part); char * a_kcache = FGS_ARG(char *, kcache), * a_vcache = FGS_ARG(char *, vcache);
#undef FGS_ARG
    (void) a;
    __builtin_amdgcn_sched_barrier(0);
    FA1_STAMP(1);
    // ---------------------------------------------------------------- 2. q chains, k chain, v head (norm_rope_dev.hpp arithmetic); the cache stores wait for the row index (end of the kernel)
    uint16_t hs0 = 0, hs1 = 0;
    if (wave <= GQ) {
        const float tc = __uint_as_float(tcs[0]), ts = __uint_as_float(tcs[1]);
        double ss = (double) (x0 * x0) + (double) (x1 * x1);
        ss = wave_sum_f64o(ss);
        const float mean  = (float) (ss * (1.0 / D));
        const float scale = 1.0f / sqrtf(mean + eps_);
        const float v0 = has_norm ? (x0 * scale) * w0 : x0, v1 = has_norm ? (x1 * scale) * w1 : x1;
        const float r0 = v0 * tc - v1 * ts, r1 = v0 * ts + v1 * tc;
        hs0 = f2h(r0); hs1 = f2h(r1);
        if (wave < GQ) { qf[wave][e0] = h2f(hs0); qf[wave][e1] = h2f(hs1); q16[wave][e0] = hs0; q16[wave][e1] = hs1; }      // q_to_vec_dot rounding (ops.cpp:8040)
        else           { kc[e0] = h2f(hs0); kc[e1] = h2f(hs1); }
    } else if (wave == GQ + 1) {
        hs0 = f2h(x0); hs1 = f2h(x1);
        vc[lane] = h2f(hs0); vc[lane + 64] = h2f(hs1);
    }
    FA1_STAMP(2);
    // The K / V instructions were the wave's FIRST vector-memory requests and return in order: a wave that has used its raw rows (waves 0 .. 5) has its part of the
    // tiles in LDS; the two waves without a chain wait for theirs here.  (hipcc drains the queue -- vmcnt(0) -- in front of the first LDS read behind an inline-asm
    // LDS-DMA anyway: the mask row, requested ~0.8 us earlier, has arrived by then.)
    if (wave > GQ + 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    FA1_STAMP(3);
    const int krow = krow_u - row0, vrow = vrow_u - row0;          // the new token's row relative to this slice (its cache row is being written: taken from LDS instead)
    // ---------------------------------------------------------------- 3. + 4. scores and soft-max of head hh = wave < 4, lane = ROW of the slice: no cross-lane sums, the
    // scores never leave the registers.  Lane r walks its row's sixteen 16-byte chunks in the order c ^ (r & 15): the 16 lanes one ds_read_b128 pass serves
    // ({0-3, 12-15, 20-27}, ... MI355X_MICROARCH.md "LDS") then hit sixteen different bank quads although the rows are 256 bytes apart; q's chunks (f16, LDS) follow the same order
    if (wave < GQ) {
        float snew = 0.0f;                                          // score of the new token itself, from the k head in LDS
#pragma unroll
        for (int i = 0; i < D / 64; ++i) snew = fmaf(kc[lane + 64 * i], qf[wave][lane + 64 * i], snew);
        snew = wave_sum_f32(snew);
        const char * kr = kt + lane * (D * 2);
        const char * qr = (const char *) &q16[wave][0];
        const uint32_t x16 = (uint32_t) (lane & 15) * 16u;
        // every chunk of the row and of q requested first (32 ds_read_b128), four independent partial sums: one wave per SIMD runs this phase, nothing else hides a latency
        u32x4 kk[16], qq[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { const uint32_t o = (uint32_t) (c * 16) ^ x16; kk[c] = *(const u32x4 *) (kr + o); qq[c] = *(const u32x4 *) (qr + o); }
        float s4[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) { f16x2 kh, qh; const uint32_t kw = kk[c][e], qw2 = qq[c][e]; __builtin_memcpy(&kh, &kw, 4); __builtin_memcpy(&qh, &qw2, 4); s4[e] = __builtin_amdgcn_fdot2(kh, qh, s4[e], false); }
        const float sv = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        float m = a_mask ? h2f(mraw) : 0.0f;
        if (lane >= nkv) m = -INFINITY;
        float v = m == -INFINITY ? -INFINITY : (lane == krow ? snew : sv) * a_scale + m;
        const float M = wave_max_f32(v);
        float p = v == -INFINITY ? 0.0f : expf(v - M);
        const float S = wave_sum_f32(p);
        float pc = 0.0f;
        if (vrow >= 0 && vrow < nkv) { pc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), vrow & 63)); if (lane == vrow) p = 0.0f; }     // the new token's V row is in LDS, not in the tile
        pl[wave][lane] = p;
        if (lane == 0) { pcur_s[wave] = pc; ms_s[wave][0] = M; ms_s[wave][1] = S; }
    }
    __syncthreads();
    FA1_STAMP(5);
    float (* part)[GQ][D] = (float (*)[GQ][D]) kt;                 // the waves' partial rows [FGS_W][GQ][D] take the K tile's place (its last reader is behind the barrier above)
    static_assert(sizeof(float) * FGS_W * GQ * D <= sizeof kt, "partials alias the K tile");
    // ---------------------------------------------------------------- 5. P.V: wave w takes rows 8 w .. 8 w + 7 for the four heads, TWO rows per step: lanes 0 .. 31 row 2 t, lanes 32 .. 63
    // row 2 t + 1, four output dims per lane (ds_read_b64, 8 bytes x 32 lanes = the row)
    {
        const int half = lane >> 5, l5 = lane & 31;
        float acc[GQ][4];
#pragma unroll
        for (int hh = 0; hh < GQ; ++hh)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[hh][e] = 0.0f;
#pragma unroll
        for (int t = 0; t < RPW / 2; ++t) {
            const int row = wave * RPW + 2 * t + half;
            u32x2 w = *(const u32x2 *) (vt + row * (D * 2) + 8 * l5);
            float p[GQ];
#pragma unroll
            for (int hh = 0; hh < GQ; ++hh) p[hh] = pl[hh][row];
            // masked cells are SKIPPED by the reference (ops.cpp:8047-8050), never multiplied: an uninitialised cache cell holding inf / NaN must not leak in through
            // 0 * x.  A row every head of the group weights with zero is dropped whole (a cell that is live for one head is initialised, and 0 * finite adds nothing)
            const bool any = (p[0] != 0.0f) | (p[1] != 0.0f) | (p[2] != 0.0f) | (p[3] != 0.0f);
            if (!any) { w[0] = 0u; w[1] = 0u; }
            const float v0 = h2f((uint16_t) (w[0] & 0xffff)), v1 = h2f((uint16_t) (w[0] >> 16)), v2 = h2f((uint16_t) (w[1] & 0xffff)), v3 = h2f((uint16_t) (w[1] >> 16));
#pragma unroll
            for (int hh = 0; hh < GQ; ++hh) {
                acc[hh][0] = fmaf(p[hh], v0, acc[hh][0]); acc[hh][1] = fmaf(p[hh], v1, acc[hh][1]);
                acc[hh][2] = fmaf(p[hh], v2, acc[hh][2]); acc[hh][3] = fmaf(p[hh], v3, acc[hh][3]);
            }
        }
        // the two rows of a step fold inside the wave (its LDS operations execute in order): the lower half stores, the upper half adds
        if (half == 0) {
#pragma unroll
            for (int hh = 0; hh < GQ; ++hh) *(f32x4 *) (&part[wave][hh][4 * l5]) = f32x4{ acc[hh][0], acc[hh][1], acc[hh][2], acc[hh][3] };
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (half == 1) {
#pragma unroll
            for (int hh = 0; hh < GQ; ++hh) {
                const f32x4 t = *(const f32x4 *) (&part[wave][hh][4 * l5]);
                *(f32x4 *) (&part[wave][hh][4 * l5]) = f32x4{ t[0] + acc[hh][0], t[1] + acc[hh][1], t[2] + acc[hh][2], t[3] + acc[hh][3] };
            }
        }
    }
    __syncthreads();
    FA1_STAMP(6);
    // ---------------------------------------------------------------- 6. fold the waves' row groups (fixed order), add the new token's own term, store the slice's partial state
    {
        const int hh = threadIdx.x >> 7, d = threadIdx.x & (D - 1);          // 512 threads = 4 heads x 128 dims
        float o = 0.0f;
#pragma unroll
        for (int w = 0; w < FGS_W; ++w) o += part[w][hh][d];
        o = fmaf(pcur_s[hh], vc[d], o);
        const int h = g * GQ + hh;
        a_part[(size_t) sp * (a_nh * D) + h * D + d] = o;
        if (d < 2) a_part[(size_t) FGS_NSL * (a_nh * D) + ((size_t) sp * a_nh + h) * 2 + d] = ms_s[hh][d];
    }
    // the slice that holds the new token's row (or slice 0 when the index points outside the view) stores the new cache rows
    const bool owner = (krow_u >= row0 && krow_u < row0 + RS) || (sp == 0 && (krow_u < 0 || krow_u >= nkv_all));
    if (owner) {
        if (wave == GQ)          { uint16_t * kr = (uint16_t *) (a_kcache + (int64_t) krow_u * a_kc_rs) + g * D; kr[e0] = hs0; kr[e1] = hs1; }
        else if (wave == GQ + 1) { uint16_t * vr = (uint16_t *) (a_vcache + (int64_t) vrow_u * a_vc_rs) + g * D; vr[lane] = hs0; vr[lane + 64] = hs1; }
    }
    FA1_STAMP(7);
    FA1_STAMP_FLUSH;
}

// ================================================================================================= flash-attention OFF (llama-bench's default)
// The same one-token step as the reference's soft-max path emits it (src/llama-graph.cpp:1362-1420, build_attn_mha without flash_attn):
//     kq  = MUL_MAT(k f16 [D, n_kv, HK], q f32)          q rounded to f16 (vec_dot_type of f16), f32 accumulation
//     p   = SOFT_MAX_EXT(kq, mask f32, scale)            ops.cpp:5072-5182
//     kqv = MUL_MAT(v^T f16 [n_kv, D, HK], p)            p rounded to f16 AFTER normalisation, f32 accumulation
//     out = CONT(PERMUTE(kqv))                           [D * H]
// with the V cache TRANSPOSED ([n_ctx cells contiguous, D * HK], llama-kv-cache.cpp:1091-1109: the v store is a SET_ROWS scatter of single
// elements, index d_global * n_ctx + cell) and the same q / k / v pre-stage as k_fattn_one.  As separate launches this is 8 kernels per layer
// (norm_rope, scatter, f32->f16 x2, batched mat-vec x2, soft-max, cont: 436 launches and 356 tok/s per decode step against 181 / 509 with
// flash-attention on); here it is one.
template <int D>
__global__ void __launch_bounds__(256) k_attn_one_sm(FA1_LEAD_PARAMS, const fa1_dev a) {
    constexpr int KCH = D / 32, NG = FA1_NKV / 16 / 4, DPW = D / 4, HALF = D / 2;
    constexpr int LPD = 64 / DPW;                 // lanes per output dim (2 at D = 128, 4 at D = 64): each takes FA1_NKV / LPD consecutive cells
    constexpr int CPL = FA1_NKV / LPD;            // cells per lane
    constexpr int NV  = CPL / 8;                  // 16-B V loads per lane
    __shared__ __attribute__((aligned(16))) float qf[D], kc[D], vc[D], sc[FA1_NKV], pl[4][FA1_NKV];

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // deeper caches: grid = heads x slices of 256 cells, each slice leaves (O, M, S) of its cells and the last arriver of a head folds them (as
    // k_fattn_one does).  With one slice the arithmetic is the reference's to the letter (p normalised, then rounded to f16); with several the
    // weights stay f32 relative to the slice's own maximum -- the f16 rounding of p is the only thing not reproduced (~1e-10 NMSE on the output).
    FA1_HEAD_INDEX
    const int nkv = a.nkv - row0 < FA1_NKV ? a.nkv - row0 : FA1_NKV;

    // ---------------------------------------------------------------- 1. request everything
    const bool act = lane < HALF;
    const int  e0 = neox_ ? lane : 2 * lane, e1 = neox_ ? lane + HALF : 2 * lane + 1;
    const __amdgpu_buffer_rsrc_t xrs = fa1_rsrc(wave == 0 ? qraw_ + h * q_hs_ : (wave == 1 ? kraw_ + ikv * k_hs_ : vraw_ + ikv * v_hs_), wave < 3 ? D * 4 : 0);
    const uint32_t xo0 = wave < 2 ? (act ? e0 * 4 : D * 4) : lane * 4, xo1 = wave < 2 ? (act ? e1 * 4 : D * 4) : (lane + 64) * 4;
    const float x0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, xo0, 0, 0)), x1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, xo1, 0, 0));
    __builtin_amdgcn_sched_barrier(0);                             // (everything above needs only the pre-loaded arguments: the raw rows are requested before the first wait on the block's scalar loads)
    int wave_b = wave; asm volatile("" : "+s"(wave_b));            // (an opaque copy: keeps the selects below out of the control flow of the select above, which must not wait for the block)
    const __amdgpu_buffer_rsrc_t wrs = fa1_rsrc(wave_b == 0 ? a.qw : a.kw, (wave_b < 2 && a.has_norm) ? D * 4 : 0);
    const __amdgpu_buffer_rsrc_t trs = fa1_rsrc(a.tab, wave_b < 2 ? D * 4 : 0);
    const float w0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, xo0, 0, 0)), w1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, xo1, 0, 0));
    const u32x2 tcs = __builtin_amdgcn_raw_buffer_load_b64(trs, act ? lane * 8 : D * 4, 0, 0);
    const __amdgpu_buffer_rsrc_t mrs = fa1_rsrc(a.mask ? a.mask + (h % a.mne2) * a.mnb2 + row0 * 4 : a.mask, a.mask ? nkv * 4 : 0);          // f32 mask row
    uint32_t mraw[FA1_NKV / 64];
#pragma unroll
    for (int i = 0; i < FA1_NKV / 64; ++i) mraw[i] = __builtin_amdgcn_raw_buffer_load_b32(mrs, (lane + 64 * i) * 4, 0, 0);
    const int krow_g = (int) __builtin_amdgcn_raw_buffer_load_b32(fa1_rsrc(a.kidx, 4), 0, 0, 0);
    // v scatter indices: element e of the v row goes to vcache[vidx[e]] (e = ikv * D + d); element 0's index is the cell itself
    const __amdgpu_buffer_rsrc_t irs = fa1_rsrc(a.vidx, a.vidx_n * a.vidx_st);
    const int vrow_g = (int) __builtin_amdgcn_raw_buffer_load_b32(irs, 0, 0, 0);
    const int krow = krow_g - row0, vrow = vrow_g - row0;                                          // the new token's cell relative to this slice
    const bool owner = a.nsplit == 1 || (krow_g >= row0 && krow_g < row0 + FA1_NKV) || (sp == 0 && (krow_g < 0 || krow_g >= a.nkv));   // the slice that stores the new cache rows
    const int vi0 = (int) __builtin_amdgcn_raw_buffer_load_b32(irs, (uint32_t) (ikv * D + lane) * (uint32_t) a.vidx_st, 0, 0);
    const int vi1 = (int) __builtin_amdgcn_raw_buffer_load_b32(irs, (uint32_t) (ikv * D + lane + 64) * (uint32_t) a.vidx_st, 0, 0);
    const int r16 = lane >> 2, dq = lane & 3;
    const __amdgpu_buffer_rsrc_t krs = fa1_rsrc(a.k + ikv * a.knb2 + (int64_t) row0 * a.knb1, (nkv - 1) * a.knb1 + D * 2);
    const uint32_t kvo = (uint32_t) r16 * (uint32_t) a.knb1 + (uint32_t) dq * (D / 2);
    u32x4 kk[NG][KCH];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const uint32_t so = (uint32_t) ((wave + 4 * i) * 16) * (uint32_t) a.knb1;
#pragma unroll
        for (int c = 0; c < KCH; ++c) kk[i][c] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvo + 16u * c, so, 0);
    }
    // V^T: lane (dl, part): output dim wave * DPW + dl, cells 8 (LPD j + part) .. + 7 for j < NV: the LPD lanes of a dim read adjacent 16-byte
    // pieces, so an instruction takes LPD x 16 contiguous bytes per dim row and four consecutive instructions use up each 128-byte line
    const int dl = lane / LPD, part = lane % LPD, dmine = wave * DPW + dl;
    const __amdgpu_buffer_rsrc_t vrs = fa1_rsrc(a.v + ikv * a.vnb2 + (int64_t) row0 * 2, (D - 1) * a.vnb1 + nkv * 2);
    u32x4 vv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) vv[j] = __builtin_amdgcn_raw_buffer_load_b128(vrs, (uint32_t) dmine * (uint32_t) a.vnb1 + (uint32_t) (LPD * j + part) * 16u, 0, 0);

    // ---------------------------------------------------------------- 2. q chain, k chain + store, v scatter
    if (wave < 2) {
        const float tc = __uint_as_float(tcs[0]), ts = __uint_as_float(tcs[1]);
        double ss = (double) (x0 * x0) + (double) (x1 * x1);
        ss = wave_sum_f64o(ss);
        const float mean  = (float) (ss * (1.0 / D));
        const float scale = 1.0f / sqrtf(mean + a.eps);
        const float v0 = a.has_norm ? (x0 * scale) * w0 : x0, v1 = a.has_norm ? (x1 * scale) * w1 : x1;      // (llama-architecture chains: ROPE only)
        const float r0 = v0 * tc - v1 * ts, r1 = v0 * ts + v1 * tc;
        if (act) {
            const uint16_t h0 = f2h(r0), h1 = f2h(r1);
            if (wave == 0) { qf[e0] = h2f(h0); qf[e1] = h2f(h1); }                 // MUL_MAT(k f16, q): q rounded to f16
            else {
                kc[e0] = h2f(h0); kc[e1] = h2f(h1);
                if (hq == 0 && owner) { uint16_t * kr = (uint16_t *) (a.kcache + (int64_t) krow_g * a.kc_rs) + ikv * D; kr[e0] = h0; kr[e1] = h1; }
            }
        }
    } else if (wave == 2) {
        const uint16_t hv0 = f2h(x0), hv1 = f2h(x1);
        vc[lane] = h2f(hv0);
        if (D > 64) vc[lane + 64] = h2f(hv1);
        if (hq == 0 && owner) {
            *(uint16_t *) (a.vcache + (int64_t) vi0 * 2) = hv0;
            if (D > 64) *(uint16_t *) (a.vcache + (int64_t) vi1 * 2) = hv1;
        }
    }
    float mv[FA1_NKV / 64]; int n_live = 0;
#pragma unroll
    for (int i = 0; i < FA1_NKV / 64; ++i) {
        const int kv = lane + 64 * i;
        float m = a.mask ? __uint_as_float(mraw[i]) : 0.0f;
        if (kv >= nkv) m = -INFINITY;
        mv[i] = m;
        const unsigned long long live = __ballot(m != -INFINITY);
        if (live) n_live = 64 * i + 64 - __builtin_clzll(live);
    }
    __syncthreads();

    // ---------------------------------------------------------------- 3. scores
    {
        f16x2 qh[D / 8];
#pragma unroll
        for (int c = 0; c < D / 16; ++c) {
            const f32x4 t = *(const f32x4 *) (qf + dq * (D / 4) + 4 * c);
            qh[2 * c] = f16x2{ (_Float16) t[0], (_Float16) t[1] }; qh[2 * c + 1] = f16x2{ (_Float16) t[2], (_Float16) t[3] };
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int g = wave + 4 * i;
            if (g * 16 >= n_live) continue;
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < KCH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f16x2 kh; const uint32_t kw = kk[i][c][e]; __builtin_memcpy(&kh, &kw, 4);
                    s = __builtin_amdgcn_fdot2(kh, qh[c * 4 + e], s, false);
                }
            s += dppf_old<0xB1, 0xf>(0.0f, s);
            s += dppf_old<0x4E, 0xf>(0.0f, s);
            const int row = g * 16 + r16;
            if (dq == 0 && row < nkv && row != krow) sc[row] = s;
        }
        if (wave == 3) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < D / 64; ++i) s = fmaf(kc[lane + 64 * i], qf[lane + 64 * i], s);
            s = wave_sum_f32(s);
            if (lane == 0 && krow >= 0 && krow < nkv) sc[krow] = s;
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- 4. soft-max (ops.cpp:5072-5182): p = exp(s * scale + mask - max) / sum, then f16 (the KQV product's vec_dot_type)
    float M, S;
    {
        float sv[FA1_NKV / 64], pe[FA1_NKV / 64];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < FA1_NKV / 64; ++i) {
            const int kv = lane + 64 * i;
            float v = kv < n_live ? sc[kv] * a.scale : 0.0f;
            v += mv[i];
            if (mv[i] == -INFINITY) v = -INFINITY;
            sv[i] = v; mx = fmaxf(mx, v);
        }
        M = wave_max_f32(mx);
        S = 0.0f;
#pragma unroll
        for (int i = 0; i < FA1_NKV / 64; ++i) { pe[i] = sv[i] == -INFINITY ? 0.0f : expf(sv[i] - M); S += pe[i]; }
        S = wave_sum_f32(S);
        const float inv = S == 0.0f ? 0.0f : 1.0f / S;
#pragma unroll
        for (int i = 0; i < FA1_NKV / 64; ++i) pl[wave][lane + 64 * i] = a.nsplit == 1 ? h2f(f2h(pe[i] * inv)) : pe[i];
    }
    float pcur = 0.0f;
    const bool vin = vrow >= 0 && vrow < nkv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (vin) pcur = pl[wave][vrow];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (vin && lane == 0) pl[wave][vrow] = 0.0f;                       // the new token's V values come from LDS (they are not in the cache view yet)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---------------------------------------------------------------- 5. P . V^T for this lane's dim and cell range
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c0 = (LPD * j + part) * 8;
        if (c0 >= n_live) continue;                                      // (everything past the last visible cell has p == 0)
        const f32x4 pa = *(const f32x4 *) (&pl[wave][c0]), pb = *(const f32x4 *) (&pl[wave][c0 + 4]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t w = vv[j][t];
            const float p0 = t < 2 ? pa[2 * t] : pb[2 * t - 4], p1 = t < 2 ? pa[2 * t + 1] : pb[2 * t - 3];
            // masked cells are never multiplied (an uninitialised cell may hold inf / NaN): p == 0 -> skip
            acc = fmaf(p0, p0 != 0.0f ? h2f((uint16_t) (w & 0xffff)) : 0.0f, acc);
            acc = fmaf(p1, p1 != 0.0f ? h2f((uint16_t) (w >> 16)) : 0.0f, acc);
        }
    }
#pragma unroll
    for (int o = 1; o < LPD; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (part == 0) acc = fmaf(pcur, vc[dmine], acc);
    if (a.nsplit == 1) {
        if (part == 0) *((float *) (a.dst + h * a.dnb1) + dmine) = acc;
        return;
    }
    // ---------------------------------------------------------------- 6. slices: partial state out (write-through), last arriver of the head folds
    const __amdgpu_buffer_rsrc_t prs = fa1_rsrc(a.part + (int64_t) h * a.nsplit * (D + 2), a.nsplit * (D + 2) * 4);
    const uint32_t po = (uint32_t) sp * (D + 2) * 4u;
    if (part == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc), prs, po + (uint32_t) dmine * 4u, 0, 16);
    if (threadIdx.x == 0) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(M), prs, po + (uint32_t) D * 4u, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(S), prs, po + (uint32_t) (D + 1) * 4u, 0, 16);
    }
    __shared__ unsigned last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(a.cnt + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned) (a.nsplit - 1) ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x < D) {
        const int d = threadIdx.x;
        float Mx = -INFINITY, St = 0.0f, o = 0.0f;
        float Ms[FA1_MAX_SPLIT], Ss[FA1_MAX_SPLIT], os[FA1_MAX_SPLIT];
#pragma unroll
        for (int i = 0; i < FA1_MAX_SPLIT; ++i) {
            const uint32_t so = (uint32_t) ((i < a.nsplit ? i : 0) * (D + 2)) * 4u;
            Ms[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prs, so + (uint32_t) D * 4u, 0, 16));
            Ss[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prs, so + (uint32_t) (D + 1) * 4u, 0, 16));
            os[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prs, so + (uint32_t) d * 4u, 0, 16));
        }
#pragma unroll
        for (int i = 0; i < FA1_MAX_SPLIT; ++i) { if (i >= a.nsplit) Ms[i] = -INFINITY; Mx = fmaxf(Mx, Ms[i]); }
#pragma unroll
        for (int i = 0; i < FA1_MAX_SPLIT; ++i) {
            const float f = Ms[i] == -INFINITY ? 0.0f : expf(Ms[i] - Mx);
            St += Ss[i] * f; o += os[i] * f;
        }
        ((float *) (a.dst + h * a.dnb1))[d] = St == 0.0f ? 0.0f : o * (1.0f / St);
    }
    if (threadIdx.x == 0) __hip_atomic_store(a.cnt + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
}

// one token of one sequence, q / k / v pre-stage, <= 256 cache rows, f16 mask shared by the heads of a token (or per head), D 64 / 128
static bool g_one_enabled = true;
void fattn_set_one(bool on) { g_one_enabled = on; }
bool fattn_one_ok(const fattn_args & f) {
    static const bool env_off = getenv("MI355X_FA_NO_ONE") != nullptr;
    const bool off = env_off || !g_one_enabled;
    const int64_t D = f.q.ne[0];
    if (off || !f.pre || (D != 64 && D != 128) || f.v.ne[0] != D || f.q.ne[1] != 1 || f.q.ne[3] != 1 || f.k.ne[3] != 1) return false;
    if (f.k.ne[1] < 1 || f.k.ne[1] > FA1_NKV * FA1_MAX_SPLIT || f.k.ne[2] < 1 || f.q.ne[2] % f.k.ne[2] != 0) return false;
    if (f.k.nb[1] % 16 != 0 || f.v.nb[1] % 4 != 0 || ((uintptr_t) f.k.p & 15) != 0 || ((uintptr_t) f.v.p & 3) != 0 || f.k.nb[2] % 16 != 0 || f.v.nb[2] % 4 != 0) return false;
    if (f.dst.nb[0] != 4 || f.dst.nb[1] % 8 != 0 || ((uintptr_t) f.dst.p & 7) != 0) return false;
    if (f.k.nb[1] * FA1_NKV > 0x7fffffff || f.v.nb[1] * FA1_NKV > 0x7fffffff || f.k.nb[2] > 0x7fffffff || f.v.nb[2] > 0x7fffffff) return false;
    if (f.img || f.out16) return false;
    return true;
}

int fattn_one_nsplit(const fattn_args & f) { return (int) ((f.k.ne[1] + FA1_NKV - 1) / FA1_NKV); }
void flash_attn_one(const fa_dev & f, int D, const float * rope_tab, hipStream_t st) {
    if (!rope_tab) { fprintf(stderr, "[mi355x] flash_attn_one: the (cos, sin) table of the token is missing\n"); abort(); }
    const fa_pre & P = f.pre;
    fa1_dev a;
    a.qraw = P.qraw; a.kraw = P.kraw; a.vraw = P.vraw; a.qw = P.qw; a.kw = P.kw; a.tab = rope_tab;
    a.kcache = P.kcache; a.vcache = P.vcache; a.kidx = P.kidx; a.vidx = P.vidx;
    a.k = f.k; a.v = f.v; a.mask = f.mask; a.sinks = f.sinks; a.dst = f.dst;
    a.q_hs = (int) P.q_hs; a.k_hs = (int) P.k_hs; a.v_hs = (int) P.v_hs; a.kc_rs = (int) P.kc_rs; a.vc_rs = (int) P.vc_rs;
    a.knb1 = (int) f.knb1; a.knb2 = (int) f.knb2; a.vnb1 = (int) f.vnb1; a.vnb2 = (int) f.vnb2; a.mnb2 = (int) f.mnb2; a.mne2 = (int) f.mne2; a.dnb1 = (int) f.dnb1;
    a.nkv = f.nkv; a.gq = f.gq; a.neox = (P.rd.mode & GGML_ROPE_TYPE_NEOX) ? 1 : 0; a.n_head_log2 = (int) f.n_head_log2;
    a.has_norm = P.qw != nullptr; a.vidx_st = 0; a.vidx_n = 0;
    a.nsplit = f.nsplit > 1 ? f.nsplit : 1; a.part = f.part; a.cnt = f.nsplit > 1 ? f.cnt : nullptr;
    a.eps = P.eps; a.scale = f.scale; a.max_bias = f.max_bias; a.logit_softcap = f.logit_softcap; a.m0 = f.m0; a.m1 = f.m1;
    a.n_head = f.nh; { const int nkvh = f.nh / (a.gq > 0 ? a.gq : 1); a.nkvh_log2 = (nkvh > 0 && (nkvh & (nkvh - 1)) == 0) ? __builtin_ctz((unsigned) nkvh) : -1; }
    const dim3 grid((unsigned) (f.nh * a.nsplit));
    if (D == 64) k_fattn_one<64><<<grid, dim3(256), 0, st>>>(FA1_LEAD_ARGS(a), a);
    else         k_fattn_one<128><<<grid, dim3(256), 0, st>>>(FA1_LEAD_ARGS(a), a);
}


// ---- group-slice form: applicability and launch.  parts: FGS_NSL x [n_head * D] partial outputs, then FGS_NSL x [n_head] x (M, S)
size_t fattn_gs_parts_bytes(int n_head, int D) { return (size_t) FGS_NSL * n_head * D * 4 + (size_t) FGS_NSL * n_head * 2 * 4; }
int    fattn_gs_nslice() { return FGS_NSL; }
static int  g_gs_mode = -1;                          // option "fattn_gs": -1 = MI355X_FA_NO_GS decides (default on), 0 off, 1 on
static long g_gs_launches = 0, g_gs_far_launches = 0;
void fattn_set_gs(int m) { g_gs_mode = m; }
long fattn_gs_launches() { return g_gs_launches; }
long fattn_gs_far_launches() { return g_gs_far_launches; }
bool fattn_gs_ok(const fattn_args & f) {
    static const bool env_off = getenv("MI355X_FA_NO_GS") != nullptr;
    if (!(g_gs_mode >= 0 ? g_gs_mode != 0 : !env_off) || !fattn_one_ok(f)) return false;
    const fattn_pre & P = *f.pre;
    const int64_t D = f.q.ne[0], nh = f.q.ne[2], nkvh = f.k.ne[2];
    if (D != 128 || nkvh < 1 || nkvh > 255 || nh != 4 * nkvh || f.k.ne[1] > FA1_NKV || f.sinks || f.max_bias != 0.0f || f.logit_softcap != 0.0f) return false;
    if (P.q_hs != D * 4 || P.k_hs != D * 4 || P.v_hs != D * 4 || f.k.nb[2] != (size_t) D * 2 || f.v.nb[2] != (size_t) D * 2 || f.k.nb[1] != f.v.nb[1] || f.k.nb[1] != (size_t) nkvh * D * 2) return false;
    if (((uintptr_t) P.qraw & 15) != 0 || ((uintptr_t) P.kraw & 15) != 0 || ((uintptr_t) P.vraw & 15) != 0 || ((uintptr_t) f.k.p & 15) != 0 || ((uintptr_t) f.v.p & 15) != 0) return false;
    if (f.mask && ((((uintptr_t) f.mask->p) & 3) != 0 || f.mask->nb[2] % 4 != 0)) return false;
    const int64_t ko = (const char *) P.kraw - (const char *) P.qraw, vo = (const char *) P.vraw - (const char *) P.qraw, vco = ((const char *) f.v.p - (const char *) f.k.p) / 16;
    (void) ko; (void) vo; (void) vco;                                 // (out of an offset's reach: the launch takes the pointers from the argument block, fattn_gs_far)
    if (f.mask) {                                                    // one mask row for every head, 16-byte aligned, within +-32 GB of the q rows
        const int64_t mo = (const char *) f.mask->p - (const char *) P.qraw;
        if (f.mask->ne[2] != 1 || (((uintptr_t) f.mask->p) & 15) != 0 || mo == 0) return false;
    }
    if (P.qw && !P.kw) return false;
    {   // the (cos, sin) table and the new token's row indices are reached through pre-loaded offsets from the q rows: the table 16-byte units, k's index 8-byte units, v's index
        // within +-32 KB of k's (the two index tensors are graph inputs allocated side by side)
        if (!f.rope_tab || !P.kidx || !P.vidx) return false;
        const int64_t to = (const char *) f.rope_tab - (const char *) P.qraw, io = (const char *) P.kidx - (const char *) P.qraw, vi = (const char *) P.vidx - (const char *) P.kidx;
        if ((to & 15) != 0 || (io & 7) != 0 || (vi & 7) != 0) return false;
    }
    return true;
}
void flash_attn_gs(const fa_dev & f, int D, const float * rope_tab, float * parts, hipStream_t st) {
    if (!rope_tab || !parts || D != 128) { fprintf(stderr, "[mi355x] flash_attn_gs: bad arguments\n"); abort(); }
    const fa_pre & P = f.pre;
    fa1_dev a{};
    a.qraw = P.qraw; a.kraw = P.kraw; a.vraw = P.vraw; a.qw = P.qw; a.kw = P.kw; a.tab = rope_tab;
    a.kcache = P.kcache; a.vcache = P.vcache; a.kidx = P.kidx; a.vidx = P.vidx;
    a.k = f.k; a.v = f.v; a.mask = f.mask; a.sinks = nullptr; a.dst = f.dst;
    a.q_hs = (int) P.q_hs; a.k_hs = (int) P.k_hs; a.v_hs = (int) P.v_hs; a.kc_rs = (int) P.kc_rs; a.vc_rs = (int) P.vc_rs;
    a.knb1 = (int) f.knb1; a.knb2 = (int) f.knb2; a.vnb1 = (int) f.vnb1; a.vnb2 = (int) f.vnb2; a.mnb2 = (int) f.mnb2; a.mne2 = (int) f.mne2; a.dnb1 = (int) f.dnb1;
    a.nkv = f.nkv; a.gq = f.gq; a.neox = (P.rd.mode & GGML_ROPE_TYPE_NEOX) ? 1 : 0; a.n_head_log2 = (int) f.n_head_log2;
    a.has_norm = P.qw != nullptr; a.nsplit = FGS_NSL; a.part = parts; a.cnt = nullptr;
    a.eps = P.eps; a.scale = f.scale; a.max_bias = 0.0f; a.logit_softcap = 0.0f; a.m0 = 1.0f; a.m1 = 1.0f;
    a.n_head = f.nh; a.nkvh_log2 = -1;
    const int nkvh = f.nh / 4;
    // does every pre-loaded offset reach its operand?  (k / v rows +-512 KB from q's, table / mask +-32 GB, k's index +-16 GB, v's index +-16 KB from k's, k's norm weights +-2 GB
    // from q's, the V view +-32 GB from the K view)  If not: FAR, the kernel reads the pointers from the argument block
    auto fits = [](int64_t v, int bits) { return v >= -((int64_t) 1 << (bits - 1)) && v < ((int64_t) 1 << (bits - 1)); };
    const int64_t ko = (a.kraw - a.qraw) / 16, vo = (a.vraw - a.qraw) / 16, to = ((const char *) a.tab - a.qraw) / 16, io = (a.kidx - a.qraw) / 8, vi = (a.vidx - a.kidx) / 8,
                  mo = a.mask ? (a.mask - a.qraw) / 16 : 0, wo = a.qw ? (const char *) a.kw - (const char *) a.qw : 0, vco = (a.v - a.k) / 16;
    static const bool force_far = getenv("MI355X_FA_GS_FAR") != nullptr;          // (test / A-B switch)
    const bool far = force_far || !(fits(ko, 16) && fits(vo, 16) && fits(to, 32) && fits(io, 32) && fits(vi, 12) && fits(mo, 32) && fits(wo, 32) && fits(vco, 32));
    if (far) ++g_gs_far_launches;
    const uint32_t pk = (uint32_t) nkvh | ((uint32_t) a.neox << 8) | ((uint32_t) a.has_norm << 9) | ((uint32_t) f.nkv << 10) | (far ? 0x80000000u : (((uint32_t) (int32_t) vi & 0xfffu) << 19));
    ++g_gs_launches;
    const uint32_t kv16 = far ? 0u : ((uint32_t) (uint16_t) (int16_t) ko | ((uint32_t) (uint16_t) (int16_t) vo << 16));
    k_fattn_gs<128><<<dim3((unsigned) (nkvh * FGS_NSL)), dim3(64 * FGS_W), 0, st>>>(a.qraw, a.qw, a.k, far ? 0 : (int) to, far ? 0 : (int) io, kv16, far ? 0 : (int) mo, far ? 0 : (int) wo,
                                                                                      far ? 0 : (int) vco, a.eps, pk, a);
}
// the slices' partial states folded into the f32 rows [n_head * D] (what the wo launch does in its prologue; used when that launch cannot take the parts)
__global__ void k_fattn_gs_merge(const float * parts, float * dst, int n_head, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_head * D) return;
    const int h = i / D;
    const float * ms = parts + (size_t) FGS_NSL * n_head * D;
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < FGS_NSL; ++s) M = fmaxf(M, ms[((size_t) s * n_head + h) * 2]);
    float o = 0.0f, den = 0.0f;
#pragma unroll
    for (int s = 0; s < FGS_NSL; ++s) {
        const float Ms = ms[((size_t) s * n_head + h) * 2], f = Ms == -INFINITY ? 0.0f : expf(Ms - M);
        o = fmaf(f, parts[(size_t) s * n_head * D + i], o); den = fmaf(f, ms[((size_t) s * n_head + h) * 2 + 1], den);
    }
    dst[i] = den == 0.0f ? 0.0f : o * (1.0f / den);
}
void fattn_gs_merge(const float * parts, float * dst, int n_head, int D, hipStream_t st) {
    k_fattn_gs_merge<<<dim3((unsigned) ((n_head * D + 255) / 256)), dim3(256), 0, st>>>(parts, dst, n_head, D);
}

// ---- host side of the soft-max path
bool attn_one_sm_ok(const attn_sm_args & f) {
    static const bool env_off = getenv("MI355X_NO_ATTN_SM") != nullptr;
    if (env_off || !g_one_enabled) return false;
    const int nsplit = (f.nkv + FA1_NKV - 1) / FA1_NKV;
    if ((f.D != 64 && f.D != 128) || f.nkv < 1 || nsplit > FA1_MAX_SPLIT || f.nkv % 8 != 0 || f.n_head < 1 || f.n_head_kv < 1 || f.n_head % f.n_head_kv != 0) return false;
    if (nsplit > 1 && (!f.part || !f.counters || f.n_head > 1024 || f.part_bytes < (size_t) f.n_head * nsplit * (f.D + 2) * 4)) return false;      // slices need the partial-state scratch and the arrival counters
    if (f.knb1 % 16 != 0 || f.knb2 % 16 != 0 || ((uintptr_t) f.k & 15) != 0 || f.vnb1 % 16 != 0 || f.vnb2 % 16 != 0 || ((uintptr_t) f.v & 15) != 0) return false;
    if (f.knb1 * FA1_NKV > 0x7fffffff || f.vnb1 * f.D > 0x7fffffff || f.knb2 > 0x7fffffff || f.vnb2 > 0x7fffffff) return false;
    if (f.dnb1 % 4 != 0 || ((uintptr_t) f.dst & 3) != 0 || !f.pre || !f.rope_tab) return false;
    if (f.vidx_n < (int64_t) f.n_head_kv * f.D) return false;
    return true;
}
void attn_one_sm(const attn_sm_args & f, hipStream_t st) {
    if (!attn_one_sm_ok(f)) { fprintf(stderr, "[mi355x] attn_one_sm: unsupported arguments\n"); abort(); }
    const fattn_pre & P = *f.pre;
    fa1_dev a;
    a.qraw = (const char *) P.qraw; a.kraw = (const char *) P.kraw; a.vraw = (const char *) P.vraw; a.qw = P.qw; a.kw = P.kw; a.tab = f.rope_tab;
    a.kcache = (char *) P.kcache; a.vcache = (char *) P.vcache; a.kidx = (const char *) P.kidx; a.vidx = (const char *) P.vidx;
    a.k = (const char *) f.k; a.v = (const char *) f.v; a.mask = (const char *) f.mask; a.sinks = nullptr; a.dst = (char *) f.dst;
    a.q_hs = (int) P.q_hs; a.k_hs = (int) P.k_hs; a.v_hs = (int) P.v_hs; a.kc_rs = (int) P.kc_rs; a.vc_rs = 2;
    a.knb1 = (int) f.knb1; a.knb2 = (int) f.knb2; a.vnb1 = (int) f.vnb1; a.vnb2 = (int) f.vnb2; a.mnb2 = (int) f.mnb2; a.mne2 = (int) (f.mne2 > 0 ? f.mne2 : 1); a.dnb1 = (int) f.dnb1;
    a.nkv = f.nkv; a.gq = f.n_head / f.n_head_kv; a.neox = (P.rp.mode & GGML_ROPE_TYPE_NEOX) ? 1 : 0; a.n_head_log2 = 0;
    const int nsplit = (f.nkv + FA1_NKV - 1) / FA1_NKV;
    a.vidx_st = P.idx_is64 ? 8 : 4; a.vidx_n = (int) f.vidx_n; a.has_norm = P.qw != nullptr; a.nsplit = nsplit; a.part = nsplit > 1 ? (float *) f.part : nullptr; a.cnt = nsplit > 1 ? f.counters : nullptr;
    a.eps = P.eps; a.scale = f.scale; a.max_bias = 0.0f; a.logit_softcap = 0.0f; a.m0 = 1.0f; a.m1 = 1.0f;
    a.n_head = f.n_head; a.nkvh_log2 = (f.n_head_kv & (f.n_head_kv - 1)) == 0 ? __builtin_ctz((unsigned) f.n_head_kv) : -1;
    const dim3 grid((unsigned) (f.n_head * nsplit));
    if (f.D == 64) k_attn_one_sm<64><<<grid, dim3(256), 0, st>>>(FA1_LEAD_ARGS(a), a);
    else           k_attn_one_sm<128><<<grid, dim3(256), 0, st>>>(FA1_LEAD_ARGS(a), a);
}

} // namespace mi

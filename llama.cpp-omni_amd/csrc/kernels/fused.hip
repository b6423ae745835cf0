// fused.hip -- multi-node kernels of the decode graph (gfx950, wave64).
//
// k_norm_rope : up to three "jobs" on [D, H, T] activations in ONE launch, one wave per (job, head, token):
//                 RMS_NORM -> MUL(w) -> ROPE  [-> SET_ROWS into the f16 KV cache]     (the q and k chains of llm_build_qwen3,
//                                                                                      reference src/llama-model.cpp:9331-9349)
//                 plain f32 -> f16 SET_ROWS                                           (the v store; llama_kv_cache::cpy_v,
//                                                                                      src/llama-kv-cache.cpp:1088)
//               The arithmetic of each stage is exactly the stand-alone kernels' (elementwise.hip): sum of squares in double,
//               (x*scale)*w, theta by sequential products, f32 -> f16 round-to-nearest-even (norm_rope_dev.hpp).
#include "norm_rope_dev.hpp"

namespace mi {

struct nr_job {
    const char * x; int64_t xnb1, xnb2;               // [D, H, T] f32 input (head / token byte strides)
    const float * w;                                  // [D] norm weight; null = no norm
    int plain;                                        // plain f32 -> f16 store job (no norm, no rope)
    char * y; int64_t ynb1, ynb2;                     // f32 rope output (may be null)
    char * y16; int64_t y16_rs;                       // f16 rope output as image rows h * T + t (may be null)
    char * kv; int64_t kv_rs;                         // f16 table base + row stride (may be null)
    const char * idx; int idx_is64; int64_t idx_nb0;  // row index per token
    int H; int wave_end;                              // waves [prev.wave_end, wave_end) belong to this job
    int nsplit; int64_t split_bytes;                  // > 1 (k_norm_rope_v4 only): x is slab 0 of `nsplit` split-K slabs of the mat-mul in front, split_bytes apart -- the rows are their sum, in slab order
};
struct nr_dev {
    nr_job j[3]; int njobs;
    const int32_t * pos; const float * ff;
    int D, T; float eps; rope_dev rd;
    const float * tab;                                // optional [T][D/2] (cos, sin) table of this ubatch (k_rope_table), else null
};

// (cos, sin) of every (token, rotation pair) of a ubatch, once per graph: 36 layers x (q heads + k heads) re-use them
__global__ void __launch_bounds__(256) k_rope_table(const int32_t * __restrict__ pos, const float * __restrict__ ff, const rope_dev rd, int T, int half, float * __restrict__ tab) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T * half) return;
    const int t = i / half, ip = i - t * half;
    float c, s;
    rope_angle((float) pos[t], ip, ff, rd, c, s);
    tab[2 * i] = c; tab[2 * i + 1] = s;
}

template <int PPL>   // rotation pairs per lane: 1 for D <= 128, 2 for D <= 256
__global__ void __launch_bounds__(256) k_norm_rope(const nr_dev a) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int) (blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (wid >= a.j[a.njobs - 1].wave_end) return;
    int ji = 0, w0 = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) if (i + 1 < a.njobs && wid >= a.j[i].wave_end) { ji = i + 1; w0 = a.j[i].wave_end; }
    const nr_job J = ji == 0 ? a.j[0] : (ji == 1 ? a.j[1] : a.j[2]);
    const int lw = wid - w0;
    const int h = lw % J.H, t = lw / J.H;
    const char * xr = J.x + h * J.xnb1 + t * J.xnb2;
    int64_t row = 0;
    if (J.kv) row = J.idx_is64 ? *(const int64_t *) (J.idx + t * J.idx_nb0) : (int64_t) *(const int32_t *) (J.idx + t * J.idx_nb0);

    if (J.plain) {                                    // plain store job: f32 -> f16 rows (k_set_rows' arithmetic)
        uint16_t * kr = (uint16_t *) (J.kv + row * J.kv_rs) + (int64_t) h * a.D;
        for (int e = lane; e < a.D; e += 64) kr[e] = f2h(*(const float *) (xr + e * 4));
        return;
    }

    float r0[PPL], r1[PPL]; int e0[PPL], e1[PPL]; bool act[PPL];
    norm_rope_wave<PPL>(xr, J.w, a.D, a.eps, (float) a.pos[t], a.ff, a.rd, lane, r0, r1, e0, e1, act, a.tab ? a.tab + (size_t) t * a.D : nullptr);
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        if (!act[q]) continue;
        if (J.y) {
            char * yr = J.y + h * J.ynb1 + t * J.ynb2;
            *(float *) (yr + e0[q] * 4) = r0[q]; *(float *) (yr + e1[q] * 4) = r1[q];
        }
        if (J.kv) {
            uint16_t * kr = (uint16_t *) (J.kv + row * J.kv_rs) + (int64_t) h * a.D;
            kr[e0[q]] = f2h(r0[q]); kr[e1[q]] = f2h(r1[q]);
        }
        if (J.y16) {
            uint16_t * hr = (uint16_t *) (J.y16 + ((int64_t) h * a.T + t) * J.y16_rs);
            hr[e0[q]] = f2h(r0[q]); hr[e1[q]] = f2h(r1[q]);
        }
    }
}

// Prefill ubatches (D = 128, NEOX pairs (i, i + 64), angles from the per-graph table): 16 lanes per (head, token) row instead of a wave --
// lane j owns elements 4j .. 4j+3 and 64+4j .. 64+4j+3 (its four rotation pairs), so every access is 16 bytes (f32) / 8 bytes (f16 cache
// row) and a wave covers four heads; everything a row needs (x, w, angles, cache row index) is requested before the first use.
// k_norm_rope's arithmetic: sum of squares in double, (x * scale) * w, v0 c - v1 s / v0 s + v1 c, f16 round-to-nearest-even.
__global__ void __launch_bounds__(256) k_norm_rope_v4(const nr_dev a) {
    constexpr int D = 128;
    const int lane = threadIdx.x & 63, sub = lane >> 4, j = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane((int) (blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (wid * 4 >= a.j[a.njobs - 1].wave_end) return;
    int ji = 0, w0 = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) if (i + 1 < a.njobs && wid * 4 >= a.j[i].wave_end) { ji = i + 1; w0 = a.j[i].wave_end; }   // (every H is a multiple of 4: a wave never straddles jobs)
    const nr_job J = ji == 0 ? a.j[0] : (ji == 1 ? a.j[1] : a.j[2]);
    const int lw = wid * 4 + sub - w0;
    const int h = lw % J.H, t = lw / J.H;
    const char * xr = J.x + h * J.xnb1 + t * J.xnb2;
    f32x4 lo = *(const f32x4 *) (xr + j * 16), hi = *(const f32x4 *) (xr + 256 + j * 16);
    if (J.nsplit > 1) {                               // the wq / wk / wv launch left its K halves as slabs: summed here (slab 0 + slab 1 + ..., k_gemm_reduce_multi's order), never written
        f32x4 l2[7], h2[7];
#pragma unroll
        for (int s = 1; s < 8; ++s) {
            l2[s - 1] = s < J.nsplit ? *(const f32x4 *) (xr + s * J.split_bytes + j * 16)       : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
            h2[s - 1] = s < J.nsplit ? *(const f32x4 *) (xr + s * J.split_bytes + 256 + j * 16) : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
        }
#pragma unroll
        for (int s = 1; s < 8; ++s) if (s < J.nsplit) { lo += l2[s - 1]; hi += h2[s - 1]; }
    }
    int64_t row = 0;
    if (J.kv) row = J.idx_is64 ? *(const int64_t *) (J.idx + t * J.idx_nb0) : (int64_t) *(const int32_t *) (J.idx + t * J.idx_nb0);
    auto pack = [](const f32x4 v) { u32x2 h; h[0] = (uint32_t) f2h(v[0]) | ((uint32_t) f2h(v[1]) << 16); h[1] = (uint32_t) f2h(v[2]) | ((uint32_t) f2h(v[3]) << 16); return h; };
    if (J.plain) {                                    // plain store job: f32 -> f16 rows
        char * kr = J.kv + row * J.kv_rs + (int64_t) h * D * 2;
        *(u32x2 *) (kr + j * 8) = pack(lo); *(u32x2 *) (kr + 128 + j * 8) = pack(hi);
        return;
    }
    f32x4 wl = { 1.0f, 1.0f, 1.0f, 1.0f }, wh = wl;
    if (J.w) { wl = *(const f32x4 *) (J.w + 4 * j); wh = *(const f32x4 *) (J.w + 64 + 4 * j); }
    const float * tb = a.tab + (size_t) t * D + 8 * j;                     // (cos, sin) of pairs 4j .. 4j+3
    const f32x4 cs0 = *(const f32x4 *) tb, cs1 = *(const f32x4 *) (tb + 4);
    double ss = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ss += (double) (lo[e] * lo[e]); ss += (double) (hi[e] * hi[e]); }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);           // the 16 lanes of this row
    const float mean  = (float) (ss / (double) D);
    const float scale = 1.0f / sqrtf(mean + a.eps);
    f32x4 r0, r1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v0 = J.w ? (lo[e] * scale) * wl[e] : lo[e], v1 = J.w ? (hi[e] * scale) * wh[e] : hi[e];
        const float c = e < 2 ? cs0[2 * e] : cs1[2 * e - 4], sn = e < 2 ? cs0[2 * e + 1] : cs1[2 * e - 3];
        r0[e] = v0 * c - v1 * sn; r1[e] = v0 * sn + v1 * c;
    }
    if (J.y) {
        char * yr = J.y + h * J.ynb1 + t * J.ynb2;
        *(f32x4 *) (yr + j * 16) = r0; *(f32x4 *) (yr + 256 + j * 16) = r1;
    }
    if (J.kv) {
        char * kr = J.kv + row * J.kv_rs + (int64_t) h * D * 2;
        *(u32x2 *) (kr + j * 8) = pack(r0); *(u32x2 *) (kr + 128 + j * 8) = pack(r1);
    }
    if (J.y16) {
        char * hr = J.y16 + ((int64_t) h * a.T + t) * J.y16_rs;
        *(u32x2 *) (hr + j * 8) = pack(r0); *(u32x2 *) (hr + 128 + j * 8) = pack(r1);
    }
}

void rope_table(const int32_t * pos, const float * ff, const rope_params & rp, int T, int D, float * tab, hipStream_t st) {
    const int n = T * (D / 2);
    if (n <= 0) return;
    k_rope_table<<<dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st>>>(pos, ff, make_rope_dev(rp), T, D / 2, tab);
}

static long g_norm_rope_split_launches = 0;
long norm_rope_split_launches() { return g_norm_rope_split_launches; }
// would norm_rope_store run the kernel that can sum split-K slabs (k_norm_rope_v4), for these jobs?  (the same test as below, on the host arguments)
bool norm_rope_takes_split(const norm_rope_args & f) {
    static const bool off = getenv("MI355X_NO_NORM_ROPE_V4") != nullptr;
    if (off || f.D != 128 || f.T < 16 || !(f.rp.mode & GGML_ROPE_TYPE_NEOX) || !f.rope_tab || f.njobs <= 0) return false;
    for (int i = 0; i < f.njobs; ++i) {
        const norm_rope_job & d = f.j[i];
        const bool plain = !d.w && !d.rope_only;
        if (!(d.H % 4 == 0 && ((uintptr_t) d.x & 15) == 0 && d.xnb1 % 16 == 0 && d.xnb2 % 16 == 0 && (!d.w || ((uintptr_t) d.w & 15) == 0) &&
              (!d.y || (((uintptr_t) d.y & 15) == 0 && d.ynb1 % 16 == 0 && d.ynb2 % 16 == 0)) && (!d.kv || (((uintptr_t) d.kv & 7) == 0 && d.kv_rs % 8 == 0)) && (!plain || d.kv) &&
              (!d.y16 || (((uintptr_t) d.y16 & 7) == 0 && d.y16_rs % 8 == 0)) && d.nsplit >= 1 && d.nsplit <= 8 && d.split_bytes % 16 == 0)) return false;
    }
    return true;
}
void norm_rope_store(const norm_rope_args & f, hipStream_t st) {
    if (f.D == 0 || f.T == 0 || f.njobs == 0) return;
    nr_dev a;
    a.njobs = f.njobs; a.pos = f.pos; a.ff = f.ff; a.D = f.D; a.T = f.T; a.eps = f.eps; a.rd = make_rope_dev(f.rp);
    a.tab = f.rope_tab;
    if (f.rope_tab && !f.rope_tab_valid) {
        const int n = f.T * (f.D / 2);
        k_rope_table<<<dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st>>>(f.pos, f.ff, a.rd, f.T, f.D / 2, f.rope_tab);
    }
    int acc = 0;
    for (int i = 0; i < 3; ++i) {
        const norm_rope_job & s = f.j[i < f.njobs ? i : 0];
        nr_job & d = a.j[i];
        d.x = (const char *) s.x; d.xnb1 = s.xnb1; d.xnb2 = s.xnb2; d.w = s.w; d.plain = (!s.w && !s.rope_only) ? 1 : 0;
        d.y = (char *) s.y; d.ynb1 = s.ynb1; d.ynb2 = s.ynb2; d.y16 = (char *) s.y16; d.y16_rs = s.y16_rs;
        d.kv = (char *) s.kv; d.kv_rs = s.kv_rs; d.idx = (const char *) s.idx; d.idx_is64 = s.idx_is64; d.idx_nb0 = s.idx_nb0;
        d.H = s.H; d.nsplit = s.nsplit; d.split_bytes = s.split_bytes;
        if (i < f.njobs) acc += s.H * f.T;
        d.wave_end = acc;
    }
    {   // many tokens, D = 128, NEOX, table angles, 16-byte aligned rows, heads in fours: the 16-lanes-per-row kernel
        static const bool off = getenv("MI355X_NO_NORM_ROPE_V4") != nullptr;
        bool v4 = !off && f.D == 128 && f.T >= 16 && (a.rd.mode & GGML_ROPE_TYPE_NEOX) && a.tab != nullptr;
        for (int i = 0; i < f.njobs && v4; ++i) {
            const nr_job & d = a.j[i];
            v4 = d.H % 4 == 0 && ((uintptr_t) d.x & 15) == 0 && d.xnb1 % 16 == 0 && d.xnb2 % 16 == 0 && (!d.w || ((uintptr_t) d.w & 15) == 0) &&
                 (!d.y || (((uintptr_t) d.y & 15) == 0 && d.ynb1 % 16 == 0 && d.ynb2 % 16 == 0)) && (!d.kv || (((uintptr_t) d.kv & 7) == 0 && d.kv_rs % 8 == 0)) && (!d.plain || d.kv) && (!d.y16 || (((uintptr_t) d.y16 & 7) == 0 && d.y16_rs % 8 == 0));
        }
        if (v4) { for (int i = 0; i < f.njobs; ++i) if (f.j[i].nsplit > 1) { ++g_norm_rope_split_launches; break; }
                  k_norm_rope_v4<<<dim3((unsigned) ((acc / 4 + 3) / 4)), dim3(256), 0, st>>>(a); return; }
        for (int i = 0; i < f.njobs; ++i) if (f.j[i].nsplit > 1) { fprintf(stderr, "[mi355x] norm_rope_store: split-K sources need the 16-lanes-per-row kernel (ask norm_rope_takes_split first)\n"); abort(); }
    }
    dim3 grid((unsigned) ((acc + 3) / 4));
    if (f.D <= 128) k_norm_rope<1><<<grid, dim3(256), 0, st>>>(a);
    else            k_norm_rope<2><<<grid, dim3(256), 0, st>>>(a);
}

} // namespace mi

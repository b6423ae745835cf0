// fused.hip -- multi-node kernels of the decode graph (gfx950, wave64).
//
// k_norm_rope : RMS_NORM -> MUL(w) -> ROPE [-> SET_ROWS into the f16 KV cache] for one [D, H, T] activation in ONE launch
//               (the q and k chains of llm_build_qwen3, reference src/llama-model.cpp:9331-9349; the store is
//               llama_kv_cache::cpy_k, src/llama-kv-cache.cpp:1053).  The arithmetic of each stage is exactly the stand-alone
//               kernels' (elementwise.hip): sum of squares in double, (x*scale)*w, theta by sequential products, f32 -> f16 RNE.
#include "../kernels.hpp"

namespace mi {

struct nr_dev {
    const char * x; int64_t xnb1, xnb2, xnb3;         // [D, H, T, S] f32
    const float * w;                                  // [D]
    const int32_t * pos;                              // [T]
    const float * ff;                                 // freq factors or null
    char * y; int64_t ynb1, ynb2, ynb3;               // f32 out (may be null)
    char * kv; int64_t kv_rs;                         // f16 cache base + row stride (may be null)
    const char * idx; int idx_is64; int64_t idx_nb0;  // row index per token
    int D, H, T;
    int mode, n_dims;
    float eps, theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1;
};

template <int PPL>   // rotation pairs per lane: D/128 rounded up (1 for D <= 128, 2 for D <= 256)
__global__ void __launch_bounds__(256) k_norm_rope(const nr_dev a) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwork = (int64_t) a.H * a.T;
    if (wid >= nwork) return;
    const int h = (int) (wid % a.H); const int64_t t = wid / a.H;
    const char * xr = a.x + h * a.xnb1 + t * a.xnb2;
    const int  half = a.D / 2;
    const bool neox = a.mode & GGML_ROPE_TYPE_NEOX;

    float x0[PPL], x1[PPL], w0[PPL], w1[PPL]; int e0[PPL], e1[PPL]; bool act[PPL];
    double ss = 0.0;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const int ip = lane + 64 * p;
        act[p] = ip < half;
        e0[p] = neox ? ip : 2 * ip;
        e1[p] = neox ? ip + half : 2 * ip + 1;
        if (act[p]) {
            x0[p] = *(const float *) (xr + e0[p] * 4); x1[p] = *(const float *) (xr + e1[p] * 4);
            w0[p] = a.w[e0[p]]; w1[p] = a.w[e1[p]];
            ss += (double) (x0[p] * x0[p]); ss += (double) (x1[p] * x1[p]);
        } else { x0[p] = x1[p] = w0[p] = w1[p] = 0.0f; }
    }
    ss = wave_sum<double>(ss);
    const float mean  = (float) (ss / (double) a.D);
    const float scale = 1.0f / sqrtf(mean + a.eps);
    const float p = (float) a.pos[t];
    int64_t row = 0;
    if (a.kv) row = a.idx_is64 ? *(const int64_t *) (a.idx + t * a.idx_nb0) : (int64_t) *(const int32_t *) (a.idx + t * a.idx_nb0);
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        if (!act[q]) continue;
        const int ip = lane + 64 * q;
        const float v0 = (x0[q] * scale) * w0[q], v1 = (x1[q] * scale) * w1[q];
        float theta = p;
        for (int k = 0; k < ip; ++k) theta *= a.theta_scale;                    // sequential, as ggml_rope_cache_init
        const float f = a.ff ? a.ff[ip] : 1.0f;
        const float theta_extrap = theta / f;
        const float theta_interp = a.freq_scale * theta_extrap;
        float th = theta_interp, mscale = a.attn_factor;
        if (a.ext_factor != 0.0f) {
            const float yv = ((float) ip - a.corr0) / fmaxf(0.001f, a.corr1 - a.corr0);
            const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, yv))) * a.ext_factor;
            th = theta_interp * (1.0f - ramp_mix) + theta_extrap * ramp_mix;
            mscale *= 1.0f + 0.1f * logf(1.0f / a.freq_scale);
        }
        const float c = cosf(th) * mscale, s = sinf(th) * mscale;
        const float r0 = v0 * c - v1 * s, r1 = v0 * s + v1 * c;
        if (a.y) {
            char * yr = a.y + h * a.ynb1 + t * a.ynb2;
            *(float *) (yr + e0[q] * 4) = r0; *(float *) (yr + e1[q] * 4) = r1;
        }
        if (a.kv) {
            uint16_t * kr = (uint16_t *) (a.kv + row * a.kv_rs) + (int64_t) h * a.D;
            kr[e0[q]] = f2h(r0); kr[e1[q]] = f2h(r1);
        }
    }
}

static float rope_corr_dim2(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(base));
}

void norm_rope_store(const norm_rope_args & f, hipStream_t st) {
    if (f.D == 0 || f.H == 0 || f.T == 0) return;
    nr_dev a;
    a.x = (const char *) f.x; a.xnb1 = f.xnb1; a.xnb2 = f.xnb2; a.xnb3 = 0;
    a.w = f.w; a.pos = f.pos; a.ff = f.ff;
    a.y = (char *) f.y; a.ynb1 = f.ynb1; a.ynb2 = f.ynb2; a.ynb3 = 0;
    a.kv = (char *) f.kv; a.kv_rs = f.kv_rs; a.idx = (const char *) f.idx; a.idx_is64 = f.idx_is64; a.idx_nb0 = f.idx_nb0;
    a.D = f.D; a.H = f.H; a.T = f.T; a.mode = f.rp.mode; a.n_dims = f.rp.n_dims; a.eps = f.eps;
    a.theta_scale = powf(f.rp.freq_base, -2.0f / f.rp.n_dims);
    a.freq_scale = f.rp.freq_scale; a.ext_factor = f.rp.ext_factor; a.attn_factor = f.rp.attn_factor;
    const float start = floorf(rope_corr_dim2(f.rp.n_dims, f.rp.n_ctx_orig, f.rp.beta_fast, f.rp.freq_base));
    const float end   = ceilf (rope_corr_dim2(f.rp.n_dims, f.rp.n_ctx_orig, f.rp.beta_slow, f.rp.freq_base));
    a.corr0 = fmaxf(0.0f, start); a.corr1 = fminf((float) f.rp.n_dims - 1, end);
    const int64_t nwork = (int64_t) f.H * f.T;
    dim3 grid((unsigned) ((nwork + 3) / 4));
    if (f.D <= 128) k_norm_rope<1><<<grid, dim3(256), 0, st>>>(a);
    else            k_norm_rope<2><<<grid, dim3(256), 0, st>>>(a);
}

} // namespace mi

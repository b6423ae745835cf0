// fused.hip -- multi-node kernels of the decode graph (gfx950, wave64).
//
// k_norm_rope : up to three "jobs" on [D, H, T] activations in ONE launch, one wave per (job, head, token):
//                 RMS_NORM -> MUL(w) -> ROPE  [-> SET_ROWS into the f16 KV cache]     (the q and k chains of llm_build_qwen3,
//                                                                                      reference src/llama-model.cpp:9331-9349)
//                 plain f32 -> f16 SET_ROWS                                           (the v store; llama_kv_cache::cpy_v,
//                                                                                      src/llama-kv-cache.cpp:1088)
//               The arithmetic of each stage is exactly the stand-alone kernels' (elementwise.hip): sum of squares in double,
//               (x*scale)*w, theta by sequential products, f32 -> f16 round-to-nearest-even.
#include "../kernels.hpp"

namespace mi {

struct nr_job {
    const char * x; int64_t xnb1, xnb2;               // [D, H, T] f32 input (head / token byte strides)
    const float * w;                                  // [D] norm weight; null = no norm / rope (plain store job)
    char * y; int64_t ynb1, ynb2;                     // f32 rope output (may be null)
    char * kv; int64_t kv_rs;                         // f16 table base + row stride (may be null)
    const char * idx; int idx_is64; int64_t idx_nb0;  // row index per token
    int H; int wave_end;                              // waves [prev.wave_end, wave_end) belong to this job
};
struct nr_dev {
    nr_job j[3]; int njobs;
    const int32_t * pos; const float * ff;
    int D, T, mode;
    float eps, theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1;
};

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
    const int  half = a.D / 2;
    const bool neox = a.mode & GGML_ROPE_TYPE_NEOX;
    int64_t row = 0;
    if (J.kv) row = J.idx_is64 ? *(const int64_t *) (J.idx + t * J.idx_nb0) : (int64_t) *(const int32_t *) (J.idx + t * J.idx_nb0);

    if (!J.w) {                                       // plain store job: f32 -> f16 rows (k_set_rows' arithmetic)
        uint16_t * kr = (uint16_t *) (J.kv + row * J.kv_rs) + (int64_t) h * a.D;
        for (int e = lane; e < a.D; e += 64) kr[e] = f2h(*(const float *) (xr + e * 4));
        return;
    }

    float x0[PPL], x1[PPL], w0v[PPL], w1v[PPL]; int e0[PPL], e1[PPL]; bool act[PPL];
    double ss = 0.0;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const int ip = lane + 64 * p;
        act[p] = ip < half;
        e0[p] = neox ? ip : 2 * ip;
        e1[p] = neox ? ip + half : 2 * ip + 1;
        if (act[p]) {
            x0[p] = *(const float *) (xr + e0[p] * 4); x1[p] = *(const float *) (xr + e1[p] * 4);
            w0v[p] = J.w[e0[p]]; w1v[p] = J.w[e1[p]];
            ss += (double) (x0[p] * x0[p]); ss += (double) (x1[p] * x1[p]);
        } else { x0[p] = x1[p] = w0v[p] = w1v[p] = 0.0f; }
    }
    ss = wave_sum<double>(ss);
    const float mean  = (float) (ss / (double) a.D);
    const float scale = 1.0f / sqrtf(mean + a.eps);
    const float p = (float) a.pos[t];
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        if (!act[q]) continue;
        const int ip = lane + 64 * q;
        const float v0 = (x0[q] * scale) * w0v[q], v1 = (x1[q] * scale) * w1v[q];
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
        if (J.y) {
            char * yr = J.y + h * J.ynb1 + t * J.ynb2;
            *(float *) (yr + e0[q] * 4) = r0; *(float *) (yr + e1[q] * 4) = r1;
        }
        if (J.kv) {
            uint16_t * kr = (uint16_t *) (J.kv + row * J.kv_rs) + (int64_t) h * a.D;
            kr[e0[q]] = f2h(r0); kr[e1[q]] = f2h(r1);
        }
    }
}

static float rope_corr_dim2(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(base));
}

void norm_rope_store(const norm_rope_args & f, hipStream_t st) {
    if (f.D == 0 || f.T == 0 || f.njobs == 0) return;
    nr_dev a;
    a.njobs = f.njobs; a.pos = f.pos; a.ff = f.ff; a.D = f.D; a.T = f.T; a.mode = f.rp.mode; a.eps = f.eps;
    int acc = 0;
    for (int i = 0; i < 3; ++i) {
        const norm_rope_job & s = f.j[i < f.njobs ? i : 0];
        nr_job & d = a.j[i];
        d.x = (const char *) s.x; d.xnb1 = s.xnb1; d.xnb2 = s.xnb2; d.w = s.w;
        d.y = (char *) s.y; d.ynb1 = s.ynb1; d.ynb2 = s.ynb2;
        d.kv = (char *) s.kv; d.kv_rs = s.kv_rs; d.idx = (const char *) s.idx; d.idx_is64 = s.idx_is64; d.idx_nb0 = s.idx_nb0;
        d.H = s.H;
        if (i < f.njobs) acc += s.H * f.T;
        d.wave_end = acc;
    }
    a.theta_scale = powf(f.rp.freq_base, -2.0f / f.rp.n_dims);
    a.freq_scale = f.rp.freq_scale; a.ext_factor = f.rp.ext_factor; a.attn_factor = f.rp.attn_factor;
    const float start = floorf(rope_corr_dim2(f.rp.n_dims, f.rp.n_ctx_orig, f.rp.beta_fast, f.rp.freq_base));
    const float end   = ceilf (rope_corr_dim2(f.rp.n_dims, f.rp.n_ctx_orig, f.rp.beta_slow, f.rp.freq_base));
    a.corr0 = fmaxf(0.0f, start); a.corr1 = fminf((float) f.rp.n_dims - 1, end);
    dim3 grid((unsigned) ((acc + 3) / 4));
    if (f.D <= 128) k_norm_rope<1><<<grid, dim3(256), 0, st>>>(a);
    else            k_norm_rope<2><<<grid, dim3(256), 0, st>>>(a);
}

} // namespace mi

// norm_rope_dev.hpp -- RMS_NORM -> MUL(w) -> ROPE of one head by one wave, shared by k_norm_rope (fused.hip) and the decode
// attention kernel's q/k/v pre-stage (fattn.hip) so both produce the same bits.
//   reference: ggml_compute_forward_rms_norm_f32 (ops.cpp:3517-3566, sum in double), the following MUL, and
//   ggml_compute_forward_rope_f32 (ops.cpp:5534-5720): theta by sequential products as ggml_rope_cache_init (:5460), YaRN mix rope_yarn (:5443)
#pragma once
#include "../kernels.hpp"

namespace mi {

struct rope_dev { int mode; float theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1; };

static inline float rope_corr_dim_host(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(base));
}
static inline rope_dev make_rope_dev(const rope_params & rp) {
    rope_dev r;
    r.mode = rp.mode;
    r.theta_scale = powf(rp.freq_base, -2.0f / rp.n_dims);
    r.freq_scale = rp.freq_scale; r.ext_factor = rp.ext_factor; r.attn_factor = rp.attn_factor;
    const float start = floorf(rope_corr_dim_host(rp.n_dims, rp.n_ctx_orig, rp.beta_fast, rp.freq_base));
    const float end   = ceilf (rope_corr_dim_host(rp.n_dims, rp.n_ctx_orig, rp.beta_slow, rp.freq_base));
    r.corr0 = fmaxf(0.0f, start); r.corr1 = fminf((float) rp.n_dims - 1, end);
    return r;
}

// (cos, sin) * mscale of rotation pair ip at position pos -- the reference's rope_yarn on the sequentially multiplied theta
static __device__ __forceinline__ void rope_angle(float pos, int ip, const float * ff, const rope_dev rd, float & c, float & s) {
    float theta = pos;
    for (int k = 0; k < ip; ++k) theta *= rd.theta_scale;                        // sequential, as ggml_rope_cache_init
    const float f = ff ? ff[ip] : 1.0f;
    const float theta_extrap = theta / f;
    const float theta_interp = rd.freq_scale * theta_extrap;
    float th = theta_interp, mscale = rd.attn_factor;
    if (rd.ext_factor != 0.0f) {
        const float yv = ((float) ip - rd.corr0) / fmaxf(0.001f, rd.corr1 - rd.corr0);
        const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, yv))) * rd.ext_factor;
        th = theta_interp * (1.0f - ramp_mix) + theta_extrap * ramp_mix;
        mscale *= 1.0f + 0.1f * logf(1.0f / rd.freq_scale);
    }
    c = cosf(th) * mscale; s = sinf(th) * mscale;
}

// one wave, one head of D elements at xr (f32, contiguous): lane l owns rotation pairs l + 64*p, p < PPL.
// out: rotated values r0/r1 at element indices e0/e1 (valid where act[p]).  tab != null: (cos, sin) pairs of this token, [D/2] float2,
// produced by rope_angle (the per-graph table of a prefill ubatch: every layer and head re-uses the same angles).
// w == null: no norm (the llama-architecture chains -- the TTS decoder -- are ROPE only): the raw values are rotated
template <int PPL>
static __device__ __forceinline__ void norm_rope_wave(const char * xr, const float * w, int D, float eps, float pos, const float * ff, const rope_dev rd,
                                                      int lane, float (&r0)[PPL], float (&r1)[PPL], int (&e0)[PPL], int (&e1)[PPL], bool (&act)[PPL],
                                                      const float * tab = nullptr) {
    const int  half = D / 2;
    const bool neox = rd.mode & GGML_ROPE_TYPE_NEOX;
    float x0[PPL], x1[PPL], w0v[PPL], w1v[PPL];
    double ss = 0.0;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const int ip = lane + 64 * p;
        act[p] = ip < half;
        e0[p] = neox ? ip : 2 * ip;
        e1[p] = neox ? ip + half : 2 * ip + 1;
        if (act[p]) {
            x0[p] = *(const float *) (xr + e0[p] * 4); x1[p] = *(const float *) (xr + e1[p] * 4);
            w0v[p] = w ? w[e0[p]] : 1.0f; w1v[p] = w ? w[e1[p]] : 1.0f;
            ss += (double) (x0[p] * x0[p]); ss += (double) (x1[p] * x1[p]);
        } else { x0[p] = x1[p] = w0v[p] = w1v[p] = 0.0f; }
    }
    ss = wave_sum<double>(ss);
    const float mean  = (float) (ss / (double) D);
    const float scale = 1.0f / sqrtf(mean + eps);
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        r0[q] = r1[q] = 0.0f;
        if (!act[q]) continue;
        const int ip = lane + 64 * q;
        const float v0 = w ? (x0[q] * scale) * w0v[q] : x0[q], v1 = w ? (x1[q] * scale) * w1v[q] : x1[q];
        float c, s;
        if (tab) { c = tab[2 * ip]; s = tab[2 * ip + 1]; }
        else     rope_angle(pos, ip, ff, rd, c, s);
        r0[q] = v0 * c - v1 * s; r1[q] = v0 * s + v1 * c;
    }
}

} // namespace mi

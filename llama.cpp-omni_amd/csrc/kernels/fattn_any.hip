// fattn_any.hip -- FLASH_ATTN_EXT for the head sizes the specialised kernels do not take (D not in {64, 128}, or K and V heads of different
// size: 40 / 48 / 72 / 80 / 96 / 192 / 256, 576 x 512): one WAVE per (query row, head, sequence), lanes over the head dimension.
// Not a fast kernel -- it exists so that such a node stays on the device instead of being handed to the CPU backend by supports_op (a
// scheduler split + two transfers per layer); the Qwen3 / TTS / Whisper shapes never come here.
// reference: ggml_compute_forward_flash_attn_ext_f16, ops.cpp:7912-8148 -- q rounded to f16 (q_to_vec_dot of an F16 K), s = K.q * scale
// [-> softcap * tanh] + slope * mask, masked cells skipped, online soft-max, sinks, 1 / S at the end (V accumulated in f32 like the other kernels).
#include "fattn_dev.hpp"

namespace mi {

constexpr int FA_ANY_MAXI = 9;                    // 64 x 9 = 576 head elements

template <int NI>
__global__ void __launch_bounds__(256) k_fattn_any(const fa_dev a, const int Dk, const int Dv) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (int64_t) a.nq * a.nh * a.ns) return;
    const int q = (int) (w % a.nq), h = (int) ((w / a.nq) % a.nh), is3 = (int) (w / ((int64_t) a.nq * a.nh));
    const int ikv = h / a.gq;
    const uint32_t hu = (uint32_t) h;
    const float slope = a.max_bias > 0.0f ? (hu < a.n_head_log2 ? powf(a.m0, (float) (hu + 1)) : powf(a.m1, (float) (2 * (hu - a.n_head_log2) + 1))) : 1.0f;
    const char * qr = a.q + q * a.qnb1 + h * a.qnb2 + is3 * a.qnb3;
    const char * kb = a.k + ikv * a.knb2 + is3 * a.knb3;
    const char * vb = a.v + ikv * a.vnb2 + is3 * a.vnb3;
    const uint16_t * mrow = a.mask ? (const uint16_t *) (a.mask + q * a.mnb1 + (h % (int) a.mne2) * a.mnb2 + (is3 % (int) a.mne3) * a.mnb3) : nullptr;
    float qv[NI], o[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int d = lane + 64 * i;
        qv[i] = d < Dk ? h2f(f2h(*(const float *) (qr + d * 4))) : 0.0f;
        o[i] = 0.0f;
    }
    float M = -INFINITY, S = 0.0f;
    for (int kv = 0; kv < a.nkv; ++kv) {
        const float mv = mrow ? slope * h2f(mrow[kv]) : 0.0f;
        if (mv == -INFINITY) continue;                                  // (wave-uniform) ops.cpp:8047-8050
        const uint16_t * kr = (const uint16_t *) (kb + (int64_t) kv * a.knb1);
        float part = 0.0f;
#pragma unroll
        for (int i = 0; i < NI; ++i) { const int d = lane + 64 * i; if (d < Dk) part = fmaf(qv[i], h2f(kr[d]), part); }
        float s = wave_sum_f32(part) * a.scale;
        if (a.logit_softcap != 0.0f) s = a.logit_softcap * tanhf(s);
        s += mv;
        float ms = 1.0f, vs = 1.0f;
        if (s > M) { ms = expf(M - s); M = s; } else vs = expf(s - M);
        const uint16_t * vr = (const uint16_t *) (vb + (int64_t) kv * a.vnb1);
#pragma unroll
        for (int i = 0; i < NI; ++i) { const int d = lane + 64 * i; if (d < Dv) o[i] = fmaf(vs, h2f(vr[d]), o[i] * ms); }
        S = S * ms + vs;
    }
    if (a.sinks) {                                                       // ops.cpp:8116-8130
        const float sk = a.sinks[h];
        float ms = 1.0f, vs = 1.0f;
        if (sk > M) { ms = expf(M - sk); M = sk; } else vs = expf(sk - M);
#pragma unroll
        for (int i = 0; i < NI; ++i) o[i] *= ms;
        S = S * ms + vs;
    }
    const float inv = S == 0.0f ? 0.0f : 1.0f / S;
    float * out = (float *) (a.dst + h * a.dnb1 + q * a.dnb2 + is3 * a.dnb3);
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int d = lane + 64 * i; if (d < Dv) out[d] = o[i] * inv; }
}

bool fattn_any_ok(int64_t Dk, int64_t Dv) { return Dk >= 1 && Dv >= 1 && Dk <= 64 * FA_ANY_MAXI && Dv <= 64 * FA_ANY_MAXI; }

void flash_attn_ext_any(const fa_dev & a, int Dk, int Dv, hipStream_t st) {
    const int64_t waves = (int64_t) a.nq * a.nh * a.ns;
    if (waves == 0) return;
    const dim3 grid((unsigned) ((waves + 3) / 4));
    const int ni = (int) (((Dk > Dv ? Dk : Dv) + 63) / 64);
    switch (ni) {
        case 1: k_fattn_any<1><<<grid, dim3(256), 0, st>>>(a, Dk, Dv); break;
        case 2: k_fattn_any<2><<<grid, dim3(256), 0, st>>>(a, Dk, Dv); break;
        case 3: k_fattn_any<3><<<grid, dim3(256), 0, st>>>(a, Dk, Dv); break;
        case 4: k_fattn_any<4><<<grid, dim3(256), 0, st>>>(a, Dk, Dv); break;
        default: k_fattn_any<FA_ANY_MAXI><<<grid, dim3(256), 0, st>>>(a, Dk, Dv); break;
    }
}

} // namespace mi

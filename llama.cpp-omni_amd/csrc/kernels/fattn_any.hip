// fattn_any.hip -- FLASH_ATTN_EXT for what the specialised kernels do not take: other head sizes (D not in {64, 128}, or K and V heads of
// different size: 40 / 48 / 72 / 80 / 96 / 192 / 256, 576 x 512) and other cache types (F32, BF16, Q8_0, Q4_0 K / V -- a quantised KV cache):
// one WAVE per (query row, head, sequence), lanes over the head dimension.
// Not a fast kernel -- it exists so that such a node stays on the device instead of being handed to the CPU backend by supports_op (a
// scheduler split + two transfers per layer); the Qwen3 / TTS / Whisper shapes never come here.
// reference: ggml_compute_forward_flash_attn_ext_f16, ops.cpp:7912-8148 -- q converted to K's vec_dot_type (f16 for F16, bf16 for BF16, f32 for F32,
// Q8_0 blocks for Q8_0 / Q4_0: d = amax / 127 stored f16, round-half-even like the compiled x86 quantiser), s = K.q * scale [-> softcap * tanh]
// + slope * mask, masked cells skipped, online soft-max, sinks, 1 / S at the end.  The Q8_0 x Q8_0 / Q4_0 x Q8_0 block dots are integer sums times
// d_k * d_q in the reference; here the same de-quantised factors are multiplied in f32 (differs by f32 rounding only).  V is accumulated in f32
// (the reference's path for every V type but F16, where it accumulates in f16).
#include "fattn_dev.hpp"

namespace mi {

constexpr int FA_ANY_MAXI = 9;                    // 64 x 9 = 576 head elements

// element d of a K / V row of type T (de-quantised to f32)
template <int T> static __device__ __forceinline__ float fa_any_elem(const char * row, int d) {
    if (T == GGML_TYPE_F16)  return h2f(((const uint16_t *) row)[d]);
    if (T == GGML_TYPE_F32)  return ((const float *) row)[d];
    if (T == GGML_TYPE_BF16) return __uint_as_float((uint32_t) ((const uint16_t *) row)[d] << 16);
    if (T == GGML_TYPE_Q8_0) { const char * b = row + (d >> 5) * 34; return h2f(*(const uint16_t *) b) * (float) (int) (int8_t) b[2 + (d & 31)]; }
    /* Q4_0 */               { const char * b = row + (d >> 5) * 18; const int j = d & 31; const uint32_t by = (uint8_t) b[2 + (j & 15)];
                               return h2f(*(const uint16_t *) b) * (float) ((int) (j < 16 ? by & 0xFu : by >> 4) - 8); }
}

template <int NI, int T>
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
        float x = d < Dk ? *(const float *) (qr + d * 4) : 0.0f;
        if (T == GGML_TYPE_F16) x = h2f(f2h(x));
        else if (T == GGML_TYPE_BF16) {                                  // ggml_compute_fp32_to_bf16: round to nearest even (NaN kept quiet)
            uint32_t u = __float_as_uint(x);
            u = (u & 0x7fffffffu) > 0x7f800000u ? (u | 0x00400000u) & 0xffff0000u : (u + (0x7fffu + ((u >> 16) & 1u))) & 0xffff0000u;
            x = __uint_as_float(u);
        } else if (T == GGML_TYPE_Q8_0 || T == GGML_TYPE_Q4_0) {          // quantize_row_q8_0 on the 32-block the lane's element belongs to (a half wave)
            float amax = fabsf(x);
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
            const float dq = amax / 127.0f, id = amax != 0.0f ? 127.0f / amax : 0.0f;
            x = h2f(f2h(dq)) * rintf(x * id);
        }
        qv[i] = x;
        o[i] = 0.0f;
    }
    float M = -INFINITY, S = 0.0f;
    for (int kv = 0; kv < a.nkv; ++kv) {
        const float mv = mrow ? slope * h2f(mrow[kv]) : 0.0f;
        if (mv == -INFINITY) continue;                                  // (wave-uniform) ops.cpp:8047-8050
        const char * kr = kb + (int64_t) kv * a.knb1;
        float part = 0.0f;
#pragma unroll
        for (int i = 0; i < NI; ++i) { const int d = lane + 64 * i; if (d < Dk) part = fmaf(qv[i], fa_any_elem<T>(kr, d), part); }
        float s = wave_sum_f32(part) * a.scale;
        if (a.logit_softcap != 0.0f) s = a.logit_softcap * tanhf(s);
        s += mv;
        float ms = 1.0f, vs = 1.0f;
        if (s > M) { ms = expf(M - s); M = s; } else vs = expf(s - M);
        const char * vr = vb + (int64_t) kv * a.vnb1;
#pragma unroll
        for (int i = 0; i < NI; ++i) { const int d = lane + 64 * i; if (d < Dv) o[i] = fmaf(vs, fa_any_elem<T>(vr, d), o[i] * ms); }
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

template <int T>
static void fa_any_go(const fa_dev & a, int Dk, int Dv, hipStream_t st) {
    const int64_t waves = (int64_t) a.nq * a.nh * a.ns;
    const dim3 grid((unsigned) ((waves + 3) / 4));
    const int ni = (int) (((Dk > Dv ? Dk : Dv) + 63) / 64);
    switch (ni) {
        case 1: k_fattn_any<1, T><<<grid, dim3(256), 0, st>>>(a, Dk, Dv); break;
        case 2: k_fattn_any<2, T><<<grid, dim3(256), 0, st>>>(a, Dk, Dv); break;
        case 3: k_fattn_any<3, T><<<grid, dim3(256), 0, st>>>(a, Dk, Dv); break;
        case 4: k_fattn_any<4, T><<<grid, dim3(256), 0, st>>>(a, Dk, Dv); break;
        default: k_fattn_any<FA_ANY_MAXI, T><<<grid, dim3(256), 0, st>>>(a, Dk, Dv); break;
    }
}
void flash_attn_ext_any(const fa_dev & a, int Dk, int Dv, int kv_type, hipStream_t st) {
    if ((int64_t) a.nq * a.nh * a.ns == 0) return;
    switch (kv_type) {
        case GGML_TYPE_F16:  fa_any_go<GGML_TYPE_F16>(a, Dk, Dv, st); break;
        case GGML_TYPE_F32:  fa_any_go<GGML_TYPE_F32>(a, Dk, Dv, st); break;
        case GGML_TYPE_BF16: fa_any_go<GGML_TYPE_BF16>(a, Dk, Dv, st); break;
        case GGML_TYPE_Q8_0: fa_any_go<GGML_TYPE_Q8_0>(a, Dk, Dv, st); break;
        case GGML_TYPE_Q4_0: fa_any_go<GGML_TYPE_Q4_0>(a, Dk, Dv, st); break;
        default: fprintf(stderr, "[mi355x] flash_attn: unsupported K / V type %d\n", kv_type); abort();
    }
}

} // namespace mi

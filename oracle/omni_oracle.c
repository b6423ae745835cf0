/* oracle/omni_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's arithmetic on the hot path (quantised mat-mul / attention behind
 * ggml_backend_i).  It is the *checker* for the HIP kernels: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product (libggml-mi355x.so) never links or calls it.
 *
 * PARITY PINNED: every function below is checked bit-for-bit (integer stages) / to 1e-6 (float stages)
 * against the real reference code compiled from /root/reference (oracle/_ref/libggml-ref.so) by
 * oracle/make_golden.py, and against the committed vectors tests/golden/*.npz by tests/test_oracle.py.
 *
 * Each function cites the reference file:line it restates.  Build: make -C oracle  (gcc -O2 -ffp-contract=off).
 * Floating-point contraction is OFF so that every multiply/add rounds separately, as the C source reads.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define QK_K 256
#define QK8_0 32

/* ---- block formats (ggml/src/ggml-common.h:219-224, :295-305, :330-343) ---- */
#pragma pack(push, 1)
typedef struct { uint16_t d; int8_t qs[QK8_0]; } blk_q8_0;                               /* 34 B  */
typedef struct { uint16_t d, dmin; uint8_t scales[12]; uint8_t qs[QK_K / 2]; } blk_q4_K;  /* 144 B */
typedef struct { uint8_t ql[QK_K / 2], qh[QK_K / 4]; int8_t scales[QK_K / 16]; uint16_t d; } blk_q6_K; /* 210 B */
#pragma pack(pop)
typedef struct { float d; int8_t qs[QK_K]; int16_t bsums[QK_K / 16]; } blk_q8_K;          /* 292 B */

/* ---- IEEE half <-> float, written from the format definition (the reference uses F16C / a bit trick,
 *      ggml-impl.h:378-430; both are exact / round-to-nearest-even, so results are identical) ---- */
float orc_h2f(uint16_t h) {
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0)       v = ldexpf((float) m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else              v = ldexpf((float) (m + 1024), e - 25);
    return s ? -v : v;
}
uint16_t orc_f2h(float f) {
    uint32_t w; memcpy(&w, &f, 4);
    const uint16_t s = (uint16_t) ((w >> 16) & 0x8000);
    const uint32_t a = w & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t) (s | 0x7e00);                 /* NaN */
    if (a >= 0x47800000u) return (uint16_t) (s | 0x7c00);                /* >= 65536 (or inf) -> inf */
    if (a >= 0x477ff000u) return (uint16_t) (s | 0x7c00);                /* rounds up to 65536 -> inf */
    const int e = (int) (a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;                            /* 24-bit significand */
    int shift;                                                           /* bits dropped */
    uint32_t he;
    if (e >= -14) { shift = 13; he = (uint32_t) (e + 15); }
    else { shift = 13 + (-14 - e); he = 0; if (shift > 25) return s; }
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) q++;
    uint32_t out = he ? ((he - 1) << 10) + q : q;                        /* q carries the implicit bit when normal */
    return (uint16_t) (s | out);
}
void orc_f32_to_f16_row(const float * x, uint16_t * y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = orc_f2h(x[i]); }
void orc_f16_to_f32_row(const uint16_t * x, float * y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = orc_h2f(x[i]); }

/* ---- Q4_K scale/min unpack: get_scale_min_k4, ggml/src/ggml-quants.c:703-710 ---- */
static void scale_min_k4(int j, const uint8_t * q, uint8_t * sc, uint8_t * m) {
    if (j < 4) { *sc = q[j] & 63; *m = q[j + 4] & 63; }
    else { *sc = (uint8_t) ((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4)); *m = (uint8_t) ((q[j + 4] >> 4) | ((q[j] >> 6) << 4)); }
}

/* ---- dequantize_row_q4_K: ggml/src/ggml-quants.c:1352-1374 ---- */
void orc_dequantize_row_q4_K(const void * vx, float * y, int64_t k) {
    const blk_q4_K * x = (const blk_q4_K *) vx;
    for (int64_t i = 0; i < k / QK_K; ++i) {
        const float d = orc_h2f(x[i].d), mn = orc_h2f(x[i].dmin);
        const uint8_t * q = x[i].qs;
        for (int j = 0, is = 0; j < QK_K; j += 64, is += 2, q += 32) {
            uint8_t sc, m;
            scale_min_k4(is, x[i].scales, &sc, &m);     const float d1 = d * sc, m1 = mn * m;
            scale_min_k4(is + 1, x[i].scales, &sc, &m); const float d2 = d * sc, m2 = mn * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * (q[l] & 0xF) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * (q[l] >> 4) - m2;
        }
    }
}
/* ---- dequantize_row_q6_K: ggml/src/ggml-quants.c:1762-1791 ---- */
void orc_dequantize_row_q6_K(const void * vx, float * y, int64_t k) {
    const blk_q6_K * x = (const blk_q6_K *) vx;
    for (int64_t i = 0; i < k / QK_K; ++i) {
        const float d = orc_h2f(x[i].d);
        const uint8_t * ql = x[i].ql, * qh = x[i].qh; const int8_t * sc = x[i].scales;
        for (int n = 0; n < QK_K; n += 128, y += 128, ql += 64, qh += 32, sc += 8)
            for (int l = 0; l < 32; ++l) {
                const int is = l / 16;
                const int8_t q1 = (int8_t) ((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                const int8_t q2 = (int8_t) ((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                const int8_t q3 = (int8_t) ((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                const int8_t q4 = (int8_t) ((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                y[l] = d * sc[is] * q1; y[l + 32] = d * sc[is + 2] * q2; y[l + 64] = d * sc[is + 4] * q3; y[l + 96] = d * sc[is + 6] * q4;
            }
    }
}
/* ---- dequantize_row_q8_0: ggml/src/ggml-quants.c:401-415 ---- */
void orc_dequantize_row_q8_0(const void * vx, float * y, int64_t k) {
    const blk_q8_0 * x = (const blk_q8_0 *) vx;
    for (int64_t i = 0; i < k / QK8_0; ++i) { const float d = orc_h2f(x[i].d); for (int j = 0; j < QK8_0; ++j) y[i * QK8_0 + j] = x[i].qs[j] * d; }
}

/* ---- nearest_int: ggml/src/ggml-quants.c:444-449 (round-half-even via the 1.5*2^23 magic add) ---- */
static int nearest_int(float f) { volatile float v = f + 12582912.f; int i; float t = v; memcpy(&i, &t, 4); return (i & 0x007fffff) - 0x00400000; }

/* ---- quantize_row_q8_K_ref: ggml/src/ggml-quants.c:2555-2592 (the x86 CPU backend forwards to it, arch/x86/quants.c:493) ---- */
void orc_quantize_row_q8_K(const float * x, void * vy, int64_t k) {
    blk_q8_K * y = (blk_q8_K *) vy;
    for (int64_t i = 0; i < k / QK_K; ++i, x += QK_K) {
        float max = 0, amax = 0;
        for (int j = 0; j < QK_K; ++j) { const float ax = fabsf(x[j]); if (ax > amax) { amax = ax; max = x[j]; } }
        if (!amax) { y[i].d = 0; memset(y[i].qs, 0, QK_K); memset(y[i].bsums, 0, sizeof(y[i].bsums)); continue; }   /* bsums: see DESIGN.md (ref leaves them stale; d = 0 nulls them) */
        const float iscale = -127.f / max;
        for (int j = 0; j < QK_K; ++j) { int v = nearest_int(iscale * x[j]); y[i].qs[j] = (int8_t) (v < 127 ? v : 127); }
        for (int j = 0; j < QK_K / 16; ++j) { int s = 0; for (int ii = 0; ii < 16; ++ii) s += y[i].qs[j * 16 + ii]; y[i].bsums[j] = (int16_t) s; }
        y[i].d = 1 / iscale;
    }
}
/* ---- quantize_row_q8_0 as the x86 CPU backend computes it: ggml-cpu/arch/x86/quants.c:290-345
 *      d = amax/127 (stored f16); id = amax ? 127/amax : 0; q = round-half-even(x*id)
 *      (quantize_row_q8_0_ref, ggml-quants.c:199-222, differs: id = 1/d, roundf) ---- */
void orc_quantize_row_q8_0(const float * x, void * vy, int64_t k) {
    blk_q8_0 * y = (blk_q8_0 *) vy;
    for (int64_t i = 0; i < k / QK8_0; ++i) {
        float amax = 0; for (int j = 0; j < QK8_0; ++j) { const float a = fabsf(x[i * QK8_0 + j]); if (a > amax) amax = a; }
        const float d = amax / 127.f, id = amax != 0.0f ? 127.f / amax : 0.0f;
        y[i].d = orc_f2h(d);
        for (int j = 0; j < QK8_0; ++j) y[i].qs[j] = (int8_t) nearbyintf(x[i * QK8_0 + j] * id);
    }
}

/* ---- ggml_vec_dot_q4_K_q8_K (generic): ggml/src/ggml-cpu/quants.c:550-623 ---- */
float orc_vec_dot_q4_K_q8_K(int n, const void * vx, const void * vy) {
    const blk_q4_K * x = (const blk_q4_K *) vx; const blk_q8_K * y = (const blk_q8_K *) vy;
    float sums[8] = {0}, sumf = 0;
    for (int i = 0; i < n / QK_K; ++i) {
        int8_t a[QK_K]; int32_t aux32[8] = {0};
        const uint8_t * q4 = x[i].qs;
        for (int j = 0; j < QK_K / 64; ++j, q4 += 32) { for (int l = 0; l < 32; ++l) { a[64 * j + l] = (int8_t) (q4[l] & 0xF); a[64 * j + 32 + l] = (int8_t) (q4[l] >> 4); } }
        uint8_t sc[8], mn[8];
        for (int j = 0; j < 8; ++j) scale_min_k4(j, x[i].scales, &sc[j], &mn[j]);
        int sumi = 0;
        for (int j = 0; j < QK_K / 16; ++j) sumi += y[i].bsums[j] * mn[j / 2];
        const int8_t * q8 = y[i].qs; const int8_t * ap = a;
        for (int j = 0; j < QK_K / 32; ++j)
            for (int g = 0; g < 4; ++g, q8 += 8, ap += 8)
                for (int l = 0; l < 8; ++l) aux32[l] += (int32_t) sc[j] * (int16_t) (q8[l] * ap[l]);
        const float d = orc_h2f(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
        const float dmin = orc_h2f(x[i].dmin) * y[i].d;
        sumf -= dmin * sumi;
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}
/* ---- ggml_vec_dot_q6_K_q8_K (generic): ggml/src/ggml-cpu/quants.c:705-758 ---- */
float orc_vec_dot_q6_K_q8_K(int n, const void * vx, const void * vy) {
    const blk_q6_K * x = (const blk_q6_K *) vx; const blk_q8_K * y = (const blk_q8_K *) vy;
    float sums[8] = {0}, sumf = 0;
    for (int i = 0; i < n / QK_K; ++i) {
        int8_t a[QK_K]; int32_t aux32[8] = {0};
        const uint8_t * q4 = x[i].ql, * qh = x[i].qh; int8_t * ap = a;
        for (int j = 0; j < QK_K; j += 128, ap += 128, q4 += 64, qh += 32)
            for (int l = 0; l < 32; ++l) {
                ap[l]      = (int8_t) ((q4[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                ap[l + 32] = (int8_t) ((q4[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                ap[l + 64] = (int8_t) ((q4[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                ap[l + 96] = (int8_t) ((q4[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
            }
        const int8_t * q8 = y[i].qs; ap = a;
        for (int j = 0; j < QK_K / 16; ++j) {
            const int scale = x[i].scales[j];
            for (int g = 0; g < 2; ++g, q8 += 8, ap += 8) for (int l = 0; l < 8; ++l) aux32[l] += scale * (int16_t) (q8[l] * ap[l]);
        }
        const float d = orc_h2f(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}
/* ---- ggml_vec_dot_q8_0_q8_0 (generic): ggml/src/ggml-cpu/quants.c:305-333 ---- */
float orc_vec_dot_q8_0_q8_0(int n, const void * vx, const void * vy) {
    const blk_q8_0 * x = (const blk_q8_0 *) vx, * y = (const blk_q8_0 *) vy;
    float sumf = 0;
    for (int ib = 0; ib < n / QK8_0; ++ib) {
        int sumi = 0; for (int j = 0; j < QK8_0; ++j) sumi += x[ib].qs[j] * y[ib].qs[j];
        sumf += sumi * (orc_h2f(x[ib].d) * orc_h2f(y[ib].d));
    }
    return sumf;
}
/* ---- ggml_vec_dot_f16 (scalar tail form): ggml/src/ggml-cpu/vec.cpp ggml_vec_dot_f16: sum in ggml_float (double) ---- */
float orc_vec_dot_f16(int n, const uint16_t * x, const uint16_t * y) {
    double s = 0; for (int i = 0; i < n; ++i) s += (double) (orc_h2f(x[i]) * orc_h2f(y[i])); return (float) s;
}

/* ---- ggml_compute_forward_mul_mat for ne11 columns: ggml/src/ggml-cpu/ggml-cpu.c:1210-1402
 *      src1 (f32) is converted to the weight type's vec_dot_type (:1272-1306), then one vec_dot per output.
 *      wtype: 12 = Q4_K, 14 = Q6_K, 8 = Q8_0, 1 = F16, 0 = F32.  dst[col*M + row]. ---- */
void orc_mul_mat(int wtype, const void * W, int64_t w_rs, const float * X, int64_t K, int64_t M, int64_t N, float * dst) {
    for (int64_t c = 0; c < N; ++c) {
        const float * x = X + c * K;
        void * q = NULL;
        if (wtype == 12 || wtype == 14) { q = malloc((size_t) (K / QK_K) * sizeof(blk_q8_K)); orc_quantize_row_q8_K(x, q, K); }
        else if (wtype == 8) { q = malloc((size_t) (K / QK8_0) * sizeof(blk_q8_0)); orc_quantize_row_q8_0(x, q, K); }
        else if (wtype == 1) { q = malloc((size_t) K * 2); orc_f32_to_f16_row(x, (uint16_t *) q, K); }
        for (int64_t r = 0; r < M; ++r) {
            const char * wr = (const char *) W + r * w_rs; float v;
            switch (wtype) {
                case 12: v = orc_vec_dot_q4_K_q8_K((int) K, wr, q); break;
                case 14: v = orc_vec_dot_q6_K_q8_K((int) K, wr, q); break;
                case 8:  v = orc_vec_dot_q8_0_q8_0((int) K, wr, q); break;
                case 1:  v = orc_vec_dot_f16((int) K, (const uint16_t *) wr, (const uint16_t *) q); break;
                default: { double s = 0; for (int64_t i = 0; i < K; ++i) s += (double) (((const float *) wr)[i] * x[i]); v = (float) s; }
            }
            dst[c * M + r] = v;
        }
        free(q);
    }
}

/* ---- ggml_compute_forward_rms_norm_f32: ggml/src/ggml-cpu/ops.cpp:3517-3566 (one row) ---- */
void orc_rms_norm(const float * x, float * y, int64_t n, float eps) {
    double sum = 0; for (int64_t i = 0; i < n; ++i) sum += (double) (x[i] * x[i]);
    const float mean = (float) (sum / n), scale = 1.0f / sqrtf(mean + eps);
    for (int64_t i = 0; i < n; ++i) y[i] = x[i] * scale;
}

/* ---- rope: ggml_compute_forward_rope_f32 ops.cpp:5534-5720, rope_yarn :5443-5458, ggml_rope_cache_init :5460-5475,
 *      ggml_rope_yarn_corr_dims ggml.c:4122-4134.  One row of ne0 floats at position `pos`. mode: 0 normal, 2 neox ---- */
static float yarn_ramp(float low, float high, int i0) { const float y = (i0 / 2 - low) / fmaxf(0.001f, high - low); return 1 - fminf(1, fmaxf(0, y)); }
void orc_rope_row(const float * x, float * y, int ne0, int n_dims, int mode, int pos, int n_ctx_orig, float freq_base, float freq_scale,
                  float ext_factor, float attn_factor, float beta_fast, float beta_slow, const float * freq_factors) {
    const float theta_scale = powf(freq_base, -2.0f / n_dims);
    const float c0 = n_dims * logf(n_ctx_orig / (beta_fast * 2 * (float) M_PI)) / (2 * logf(freq_base));
    const float c1 = n_dims * logf(n_ctx_orig / (beta_slow * 2 * (float) M_PI)) / (2 * logf(freq_base));
    const float lo = fmaxf(0, floorf(c0)), hi = fminf((float) n_dims - 1, ceilf(c1));
    float theta = (float) pos;
    for (int i0 = 0; i0 < ne0; i0 += 2) {
        if (i0 < n_dims) {
            const float ff = freq_factors ? freq_factors[i0 / 2] : 1.0f;
            const float theta_extrap = theta / ff;
            const float theta_interp = freq_scale * theta_extrap;
            float th = theta_interp, mscale = attn_factor;
            if (ext_factor != 0.0f) {
                const float ramp_mix = yarn_ramp(lo, hi, i0) * ext_factor;
                th = theta_interp * (1 - ramp_mix) + theta_extrap * ramp_mix;
                mscale *= 1.0f + 0.1f * logf(1.0f / freq_scale);
            }
            const float c = cosf(th) * mscale, s = sinf(th) * mscale;
            if (mode & 2) { const int ic = i0 / 2; const float x0 = x[ic], x1 = x[ic + n_dims / 2]; y[ic] = x0 * c - x1 * s; y[ic + n_dims / 2] = x0 * s + x1 * c; }
            else { const float x0 = x[i0], x1 = x[i0 + 1]; y[i0] = x0 * c - x1 * s; y[i0 + 1] = x0 * s + x1 * c; }
            theta *= theta_scale;
        } else { y[i0] = x[i0]; y[i0 + 1] = x[i0 + 1]; }
    }
}

/* ---- ggml_compute_forward_soft_max_f32: ops.cpp:5072-5182 (one row; mask already widened to f32, slope applied by caller = 1) ---- */
void orc_soft_max_row(const float * x, const float * mask, float * y, int64_t n, float scale) {
    float mx = -INFINITY;
    for (int64_t i = 0; i < n; ++i) { y[i] = x[i] * scale + (mask ? mask[i] : 0.0f); if (y[i] > mx) mx = y[i]; }
    double sum = 0;
    for (int64_t i = 0; i < n; ++i) { const float e = expf(y[i] - mx); y[i] = e; sum += (double) e; }
    const float inv = (float) (1.0 / sum);
    for (int64_t i = 0; i < n; ++i) y[i] *= inv;
}

/* ---- ggml_vec_swiglu_f32 scalar form: vec.cpp:369 / ggml_silu_f32 vec.h:958 ---- */
void orc_swiglu(const float * x, const float * g, float * y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = (x[i] / (1.0f + expf(-x[i]))) * g[i]; }

/* ---- ggml_compute_forward_flash_attn_ext_f16: ops.cpp:7912-8148, one (query row, head).
 *      q f32[D]; K,V f16 rows with byte strides; mask f16[nkv] or NULL.  V is f16 -> the reference accumulates in f16 (:8069-8083). ---- */
void orc_flash_attn_row(const float * q, const uint16_t * K, int64_t k_rs, const uint16_t * V, int64_t v_rs, const uint16_t * mask,
                        int64_t nkv, int D, float scale, float * out) {
    uint16_t * q16 = (uint16_t *) malloc((size_t) D * 2), * acc = (uint16_t *) calloc((size_t) D, 2);
    orc_f32_to_f16_row(q, q16, D);
    float S = 0, M = -INFINITY;
    for (int64_t ic = 0; ic < nkv; ++ic) {
        const float mv = mask ? orc_h2f(mask[ic]) : 0.0f;
        if (mv == -INFINITY) continue;
        float s = orc_vec_dot_f16(D, (const uint16_t *) ((const char *) K + ic * k_rs), q16) * scale + mv;
        const float Mold = M; float ms = 1.0f, vs = 1.0f;
        if (s > M) { M = s; ms = expf(Mold - M); for (int d = 0; d < D; ++d) acc[d] = orc_f2h(orc_h2f(acc[d]) * ms); }
        else vs = expf(s - M);
        const uint16_t * v = (const uint16_t *) ((const char *) V + ic * v_rs);
        for (int d = 0; d < D; ++d) acc[d] = orc_f2h(orc_h2f(acc[d]) + orc_h2f(v[d]) * vs);
        S = S * ms + vs;
    }
    const float inv = S == 0.0f ? 0.0f : 1.0f / S;
    for (int d = 0; d < D; ++d) out[d] = orc_h2f(acc[d]) * inv;
    free(q16); free(acc);
}

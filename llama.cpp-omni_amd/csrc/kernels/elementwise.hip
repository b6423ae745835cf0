// elementwise.hip -- the non-matmul nodes of the decode / prefill graph (gfx950, wave64).
// Each kernel restates one `ggml_compute_forward_*` of the reference CPU backend
// (ggml/src/ggml-cpu/ops.cpp); the line ranges are cited per kernel.
#include "../kernels.hpp"
#include "act_dev.hpp"
#include <float.h>

namespace mi {

struct td4 { char * p; int64_t ne[4]; int64_t nb[4]; };
static td4 to_td4(const tdesc & t) {
    td4 r; r.p = (char *) t.p;
    for (int i = 0; i < 4; ++i) { r.ne[i] = t.ne[i]; r.nb[i] = (int64_t) t.nb[i]; }
    return r;
}

// ================================================================================================
// de-quantisation of whole rows (reference: dequantize_row_q4_K :1352, _q6_K :1762, _q8_0 :401 in
// ggml-quants.c).  One thread produces a run of consecutive outputs of one block; float ops are the
// same single multiply / multiply-subtract as the C source, so results are bit-exact.
// ================================================================================================
template <typename T> static __device__ __forceinline__ T cvt_out(float v);
template <> __device__ __forceinline__ float    cvt_out<float>(float v)    { return v; }
template <> __device__ __forceinline__ uint16_t cvt_out<uint16_t>(float v) { return f2h(v); }

static __device__ __forceinline__ void q4k_scale_min(int s, const uint8_t * q, int & sc, int & m) {   // get_scale_min_k4, ggml-quants.c:703-710
    if (s < 4) { sc = q[s] & 63; m = q[s + 4] & 63; }
    else       { sc = (q[s + 4] & 0xF) | ((q[s - 4] >> 6) << 4); m = (q[s + 4] >> 4) | ((q[s] >> 6) << 4); }
}

template <typename T>
__global__ void __launch_bounds__(256) k_dequant_q4k(const char * __restrict__ src, size_t src_rs, char * __restrict__ dst, size_t dst_rs, int64_t K, int64_t nrows) {
    // thread -> (row, block, sub-block s of 32, 8 outputs)  : 4 threads per sub-block
    const int64_t nb = K / 256;
    const int64_t t  = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nrows * nb * 32;
    if (t >= total) return;
    const int     part = (int) (t & 3);           // 8 outputs within the sub-block
    const int     s    = (int) ((t >> 2) & 7);    // sub-block
    const int64_t blk  = t >> 5;
    const int64_t row = blk / nb, ib = blk % nb;
    const block_q4_K * b = (const block_q4_K *) (src + row * src_rs) + ib;
    int sc, m; q4k_scale_min(s, b->scales, sc, m);
    const float d1 = h2f(b->d) * (float) sc;
    const float m1 = h2f(b->dmin) * (float) m;
    const uint8_t * q = b->qs + 32 * (s >> 1) + 8 * part;
    T * out = (T *) (dst + row * dst_rs) + ib * 256 + 32 * s + 8 * part;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const int v = (s & 1) ? (q[l] >> 4) : (q[l] & 0xF);
        out[l] = cvt_out<T>(d1 * (float) v - m1);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_dequant_q6k(const char * __restrict__ src, size_t src_rs, char * __restrict__ dst, size_t dst_rs, int64_t K, int64_t nrows) {
    // thread -> (row, block, half n, l in 0..31): 4 outputs y[128n + l + {0,32,64,96}]
    const int64_t nb = K / 256;
    const int64_t t  = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nrows * nb * 64;
    if (t >= total) return;
    const int     l   = (int) (t & 31);
    const int     n   = (int) ((t >> 5) & 1);
    const int64_t blk = t >> 6;
    const int64_t row = blk / nb, ib = blk % nb;
    const block_q6_K * b = (const block_q6_K *) (src + row * src_rs) + ib;
    const float d = h2f(b->d);
    const uint8_t * ql = b->ql + 64 * n;
    const uint8_t   qh = b->qh[32 * n + l];
    const int8_t *  sc = b->scales + 8 * n;
    const int is = l / 16;
    const int8_t q1 = (int8_t) ((ql[l +  0] & 0xF) | (((qh >> 0) & 3) << 4)) - 32;
    const int8_t q2 = (int8_t) ((ql[l + 32] & 0xF) | (((qh >> 2) & 3) << 4)) - 32;
    const int8_t q3 = (int8_t) ((ql[l +  0] >> 4)  | (((qh >> 4) & 3) << 4)) - 32;
    const int8_t q4 = (int8_t) ((ql[l + 32] >> 4)  | (((qh >> 6) & 3) << 4)) - 32;
    T * y = (T *) (dst + row * dst_rs) + ib * 256 + 128 * n;
    y[l +  0] = cvt_out<T>(d * (float) sc[is + 0] * (float) q1);
    y[l + 32] = cvt_out<T>(d * (float) sc[is + 2] * (float) q2);
    y[l + 64] = cvt_out<T>(d * (float) sc[is + 4] * (float) q3);
    y[l + 96] = cvt_out<T>(d * (float) sc[is + 6] * (float) q4);
}

template <typename T>
__global__ void __launch_bounds__(256) k_dequant_q80(const char * __restrict__ src, size_t src_rs, char * __restrict__ dst, size_t dst_rs, int64_t K, int64_t nrows) {
    const int64_t nb = K / 32;
    const int64_t t  = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nrows * K) return;
    const int64_t row = t / K, e = t % K;
    const block_q8_0 * b = (const block_q8_0 *) (src + row * src_rs) + e / 32;
    ((T *) (dst + row * dst_rs))[e] = cvt_out<T>((float) b->qs[e % 32] * h2f(b->d));
    (void) nb;
}

// ---- the other block formats (Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q2_K / Q3_K / Q5_K): one element at a time, the float operations of
// dequantize_row_q4_0 :307, _q4_1 :327, _q5_0 :348, _q5_1 :374, _q2_K :784, _q3_K :1128, _q5_K :1554 (ggml-quants.c) in the same order.
// These types have no integer-dot kernels here: MUL_MAT runs on their (resident) F16 image, GET_ROWS gathers through this function.
static __device__ __forceinline__ float h2f_at(const uint8_t * p) { return h2f((uint16_t) (p[0] | (p[1] << 8))); }
static __device__ float dq_elem_other(int type, const char * row, int64_t e) {
    switch (type) {
        case GGML_TYPE_Q4_0: case GGML_TYPE_Q4_1: {
            const int bs = type == GGML_TYPE_Q4_0 ? 18 : 20;
            const uint8_t * b = (const uint8_t *) row + (e / 32) * bs;
            const int w = (int) (e % 32), j = w & 15;
            const uint8_t q = b[(bs - 16) + j];
            const int x = w < 16 ? (q & 0x0F) : (q >> 4);
            const float d = h2f_at(b);
            if (type == GGML_TYPE_Q4_0) return (float) (x - 8) * d;
            return (float) x * d + h2f_at(b + 2);
        }
        case GGML_TYPE_Q5_0: case GGML_TYPE_Q5_1: {
            const int bs = type == GGML_TYPE_Q5_0 ? 22 : 24, hoff = bs - 20;
            const uint8_t * b = (const uint8_t *) row + (e / 32) * bs;
            const int w = (int) (e % 32), j = w & 15;
            const uint32_t qh = (uint32_t) b[hoff] | ((uint32_t) b[hoff + 1] << 8) | ((uint32_t) b[hoff + 2] << 16) | ((uint32_t) b[hoff + 3] << 24);
            const uint8_t q = b[hoff + 4 + j];
            const int x = w < 16 ? ((q & 0x0F) | (((qh >> j) << 4) & 0x10)) : ((q >> 4) | ((qh >> (j + 12)) & 0x10));
            const float d = h2f_at(b);
            if (type == GGML_TYPE_Q5_0) return (float) (x - 16) * d;
            return (float) x * d + h2f_at(b + 2);
        }
        case GGML_TYPE_Q2_K: {                                        // scales[16] qs[64] d dmin
            const uint8_t * b = (const uint8_t *) row + (e / 256) * 84;
            const int w = (int) (e % 256), n = w >> 7, r = w & 127, j = r >> 5, sub = (r >> 4) & 1, l = r & 15;
            const uint8_t sc = b[n * 8 + j * 2 + sub];
            const float dl = h2f_at(b + 80) * (float) (sc & 0xF), ml = h2f_at(b + 82) * (float) (sc >> 4);
            const uint8_t q = b[16 + n * 32 + sub * 16 + l];
            return dl * (float) (int8_t) ((q >> (2 * j)) & 3) - ml;
        }
        case GGML_TYPE_Q3_K: {                                        // hmask[32] qs[64] scales[12] d
            const uint8_t * b = (const uint8_t *) row + (e / 256) * 110;
            const int w = (int) (e % 256), n = w >> 7, r = w & 127, j = r >> 5, sub = (r >> 4) & 1, l = r & 15;
            const int k = n * 8 + j * 2 + sub, grp = k >> 2, i = k & 3;
            const uint8_t * sp = b + 96;
            const int lo = grp == 0 ? (sp[i] & 0xF) : grp == 1 ? (sp[4 + i] & 0xF) : grp == 2 ? (sp[i] >> 4) : (sp[4 + i] >> 4);
            const int hi = (sp[8 + i] >> (2 * grp)) & 3;
            const float dl = h2f_at(b + 108) * (float) ((int) (int8_t) (lo | (hi << 4)) - 32);
            const uint8_t q = b[32 + n * 32 + sub * 16 + l];
            const uint8_t hm = b[sub * 16 + l];
            const int m = 1 << (n * 4 + j);
            return dl * (float) ((int) (int8_t) ((q >> (2 * j)) & 3) - ((hm & m) ? 0 : 4));
        }
        case GGML_TYPE_Q5_K: {                                        // d dmin scales[12] qh[32] qs[128]
            const uint8_t * b = (const uint8_t *) row + (e / 256) * 176;
            const int w = (int) (e % 256), p = w >> 6, hi = (w >> 5) & 1, l = w & 31;
            int sc, m; q4k_scale_min(2 * p + hi, b + 4, sc, m);
            const float d1 = h2f_at(b) * (float) sc, m1 = h2f_at(b + 2) * (float) m;
            const uint8_t ql = b[48 + 32 * p + l];
            const int nib = hi ? (ql >> 4) : (ql & 0xF);
            const int hb = (b[16 + l] & ((hi ? 2 : 1) << (2 * p))) ? 16 : 0;
            return d1 * (float) (nib + hb) - m1;
        }
        default: return 0.0f;
    }
}
template <typename T>
__global__ void __launch_bounds__(256) k_dequant_other(int type, const char * __restrict__ src, size_t src_rs, char * __restrict__ dst, size_t dst_rs, int64_t K, int64_t nrows) {
    const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nrows * K) return;
    const int64_t row = t / K, e = t % K;
    ((T *) (dst + row * dst_rs))[e] = cvt_out<T>(dq_elem_other(type, src + row * src_rs, e));
}

template <typename T>
static void dequant_rows_t(int type, const void * src, size_t src_rs, T * dst, size_t dst_rs, int64_t K, int64_t nrows, hipStream_t st) {
    if (K == 0 || nrows == 0) return;
    int64_t nthreads;
    switch (type) {
        case GGML_TYPE_Q4_K: nthreads = nrows * (K / 256) * 32;
            k_dequant_q4k<T><<<dim3((unsigned) ((nthreads + 255) / 256)), dim3(256), 0, st>>>((const char *) src, src_rs, (char *) dst, dst_rs, K, nrows); break;
        case GGML_TYPE_Q6_K: nthreads = nrows * (K / 256) * 64;
            k_dequant_q6k<T><<<dim3((unsigned) ((nthreads + 255) / 256)), dim3(256), 0, st>>>((const char *) src, src_rs, (char *) dst, dst_rs, K, nrows); break;
        case GGML_TYPE_Q8_0: nthreads = nrows * K;
            k_dequant_q80<T><<<dim3((unsigned) ((nthreads + 255) / 256)), dim3(256), 0, st>>>((const char *) src, src_rs, (char *) dst, dst_rs, K, nrows); break;
        case GGML_TYPE_Q4_0: case GGML_TYPE_Q4_1: case GGML_TYPE_Q5_0: case GGML_TYPE_Q5_1: case GGML_TYPE_Q2_K: case GGML_TYPE_Q3_K: case GGML_TYPE_Q5_K:
            nthreads = nrows * K;
            k_dequant_other<T><<<dim3((unsigned) ((nthreads + 255) / 256)), dim3(256), 0, st>>>(type, (const char *) src, src_rs, (char *) dst, dst_rs, K, nrows); break;
        default: fprintf(stderr, "[mi355x] dequant_rows: unsupported type %d\n", type); abort();
    }
}
void dequant_rows_f32(int type, const void * src, size_t src_rs, float * dst, size_t dst_rs, int64_t K, int64_t nrows, hipStream_t st) {
    dequant_rows_t<float>(type, src, src_rs, dst, dst_rs, K, nrows, st);
}
void dequant_rows_f16(int type, const void * src, size_t src_rs, uint16_t * dst, size_t dst_rs, int64_t K, int64_t nrows, hipStream_t st) {
    dequant_rows_t<uint16_t>(type, src, src_rs, dst, dst_rs, K, nrows, st);
}

// ================================================================================================
// RMS_NORM (+ fused MUL).  reference: ggml_compute_forward_rms_norm_f32, ops.cpp:3517-3566:
//   sum = sum_i (double)(x_i*x_i);  mean = sum/ne00;  scale = 1/sqrtf(mean+eps);  y = x*scale
// The sum of squares is accumulated in double exactly like the reference's `ggml_float`, which makes
// the result independent of the summation order (bit-exact `scale` in practice).  The optional fused
// weight multiply is the graph's following MUL node (y*w, w broadcast over rows).
// ================================================================================================
// y16 != null: also (or, with y.p == null, only) emit the f16-rounded row -- the activation format of the prefill GEMM -- so the
// separate f32 -> f16 conversion launch and, when no one else reads it, the f32 write disappear (2-D inputs only)
template <bool HAS_W>
__global__ void __launch_bounds__(1024) k_rms_norm(td4 x, td4 y, td4 w, float eps, char * __restrict__ y16, int64_t y16_rs) {
    __shared__ double red[16];
    const int64_t i1 = blockIdx.x, i2 = blockIdx.y, i3 = blockIdx.z;
    const float * xr = (const float *) (x.p + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    float *       yr = (float *) (y.p + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]);
    const int64_t n = x.ne[0];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) { const float v = xr[i]; s += (double) (v * v); }
    s = block_sum<double>(s, red);
    const float mean  = (float) (s / (double) n);
    const float scale = 1.0f / sqrtf(mean + eps);
    if (HAS_W) {
        const float * wr = (const float *) (w.p + (i1 % w.ne[1]) * w.nb[1] + (i2 % w.ne[2]) * w.nb[2] + (i3 % w.ne[3]) * w.nb[3]);
        const int64_t wn = w.ne[0];
        uint16_t * hr = y16 ? (uint16_t *) (y16 + i1 * y16_rs) : nullptr;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const float v = (xr[i] * scale) * wr[wn == n ? i : i % wn];
            if (y.p) yr[i] = v;
            if (hr)  hr[i] = f2h(v);
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) yr[i] = xr[i] * scale;
    }
}

// Many rows (prefill ubatches): one WAVE per row, the row held in registers (MAXV 16-byte pieces per lane), sum of squares in double per lane and
// folded across the wave -- one read of x, no LDS, no barrier (k_rms_norm re-reads the row and issues 4-byte accesses: 1.8 TB/s on 16384 x 4096).
template <int MAXV>
__global__ void __launch_bounds__(256) k_rms_norm_rows(td4 x, td4 y, const float * __restrict__ w, float eps, char * __restrict__ y16, int64_t y16_rs, int64_t nrows, int y16_q8) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int64_t i1 = row % x.ne[1], i2 = (row / x.ne[1]) % x.ne[2], i3 = row / (x.ne[1] * x.ne[2]);
    const char * xr = x.p + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3];
    const int n = (int) x.ne[0];
    f32x4 v[MAXV];
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = (lane + 64 * k) * 4;
        v[k] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
        if (i < n) v[k] = *(const f32x4 *) (xr + (size_t) i * 4);
    }
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += (double) (v[k][e] * v[k][e]);
    ss = wave_sum<double>(ss);
    const float mean  = (float) (ss / (double) n);
    const float scale = 1.0f / sqrtf(mean + eps);
    char * yr = y.p ? y.p + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3] : nullptr;
    char * hr = y16 ? y16 + row * y16_rs : nullptr;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = (lane + 64 * k) * 4;
        if (i >= n) break;
        f32x4 o;
        if (w) { const f32x4 ww = *(const f32x4 *) (w + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[k][e] * scale) * ww[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[k][e] * scale;
        }
        if (yr) *(f32x4 *) (yr + (size_t) i * 4) = o;
        if (hr && y16_q8) o = q8k_requant4(o, lane);              // (n % 256 == 0: piece k of the wave IS the row's 256-block k, lane l its elements 4l..4l+3)
        if (hr) { u32x2 h; h[0] = (uint32_t) f2h(o[0]) | ((uint32_t) f2h(o[1]) << 16); h[1] = (uint32_t) f2h(o[2]) | ((uint32_t) f2h(o[3]) << 16); *(u32x2 *) (hr + (size_t) i * 2) = h; }
    }
}

void rms_norm(const tdesc & x, const tdesc & y, float eps, const tdesc * mul_w, hipStream_t st, uint16_t * y16, size_t y16_rs, bool write_f32, bool y16_q8) {
    if (x.ne[0] == 0 || x.ne[1] * x.ne[2] * x.ne[3] == 0) return;
    const int64_t n = x.ne[0];
    if (y16_q8 && (!y16 || n % 256 != 0)) { fprintf(stderr, "[mi355x] rms_norm: the Q8_K image needs rows of whole 256-blocks\n"); abort(); }
    {
        const int64_t nrows = x.ne[1] * x.ne[2] * x.ne[3];
        auto al16 = [](const tdesc & t) { return ((uintptr_t) t.p & 15) == 0 && t.nb[0] == 4 && t.nb[1] % 16 == 0 && t.nb[2] % 16 == 0 && t.nb[3] % 16 == 0; };
        const bool w_ok = !mul_w || (mul_w->ne[0] == n && mul_w->ne[1] * mul_w->ne[2] * mul_w->ne[3] == 1 && mul_w->nb[0] == 4 && ((uintptr_t) mul_w->p & 15) == 0);
        if (nrows >= 64 && n % 4 == 0 && n <= 8192 && al16(x) && al16(y) && w_ok && (!y16 || (y16_rs % 8 == 0 && ((uintptr_t) y16 & 7) == 0)) && (mul_w || !y16)) {
            td4 yd = to_td4(y);
            if (!write_f32) yd.p = nullptr;
            const dim3 grid((unsigned) ((nrows + 3) / 4));
            const float * wp = mul_w ? (const float *) mul_w->p : nullptr;
            if (n <= 2048)      k_rms_norm_rows<8><<<grid, dim3(256), 0, st>>>(to_td4(x), yd, wp, eps, (char *) y16, (int64_t) y16_rs, nrows, y16_q8 ? 1 : 0);
            else if (n <= 4096) k_rms_norm_rows<16><<<grid, dim3(256), 0, st>>>(to_td4(x), yd, wp, eps, (char *) y16, (int64_t) y16_rs, nrows, y16_q8 ? 1 : 0);
            else                k_rms_norm_rows<32><<<grid, dim3(256), 0, st>>>(to_td4(x), yd, wp, eps, (char *) y16, (int64_t) y16_rs, nrows, y16_q8 ? 1 : 0);
            return;
        }
    }
    int bs = n <= 128 ? 64 : n < 1024 ? 256 : n < 8192 ? 512 : 1024;
    dim3 grid((unsigned) x.ne[1], (unsigned) x.ne[2], (unsigned) x.ne[3]);
    if (y16 && (!mul_w || x.ne[2] * x.ne[3] != 1)) { fprintf(stderr, "[mi355x] rms_norm: f16 emission needs the fused weight and a 2-D input\n"); abort(); }
    td4 yd = to_td4(y);
    if (!write_f32) yd.p = nullptr;
    if (mul_w) k_rms_norm<true><<<grid, dim3(bs), 0, st>>>(to_td4(x), yd, to_td4(*mul_w), eps, (char *) y16, (int64_t) y16_rs);
    else       k_rms_norm<false><<<grid, dim3(bs), 0, st>>>(to_td4(x), yd, to_td4(x), eps, nullptr, 0);
    if (mul_w && y16 && y16_q8) requant_f16_rows_q8k(y16, y16_rs, n, x.ne[1], st);      // (this kernel left the plain f16 rows: re-quantised in place -- from f16, the few-rows path only)
}

// ================================================================================================
// IM2COL (the data movement half of ggml_conv_1d / ggml_conv_2d; the contraction is a MUL_MAT).
// reference: ggml_compute_forward_im2col_f16 / _f32, ops.cpp:6150-6301:  [N, IC, IH, IW] -> [N, OH, OW, IC*KH*KW],
// dst[.., iic*KH*KW + ikh*KW + ikw] = src[iih][iiw] with iiw = iow*s0 + ikw*d0 - p0, iih = ioh*s1 + ikh*d1 - p1, zero outside.
// ================================================================================================
struct im2col_dev { int N, IC, IH, IW, KH, KW, OH, OW, s0, s1, p0, p1, d0, d1; int64_t ofs0, ofs1; };

template <typename T>
__global__ void __launch_bounds__(256) k_im2col(const char * __restrict__ src, T * __restrict__ dst, const im2col_dev a, int64_t total) {
    const int64_t gid = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int ckk = a.IC * a.KH * a.KW;
    const int64_t pix = gid / ckk;                      // (in, ioh, iow) flattened
    const int e = (int) (gid - pix * ckk);
    const int iic = e / (a.KH * a.KW), k = e - iic * (a.KH * a.KW), ikh = k / a.KW, ikw = k - ikh * a.KW;
    const int iow = (int) (pix % a.OW); const int64_t t = pix / a.OW;
    const int ioh = (int) (t % a.OH), in = (int) (t / a.OH);
    const int iiw = iow * a.s0 + ikw * a.d0 - a.p0, iih = ioh * a.s1 + ikh * a.d1 - a.p1;
    float v = 0.0f;
    if (iih >= 0 && iih < a.IH && iiw >= 0 && iiw < a.IW) v = ((const float *) (src + in * a.ofs0 + iic * a.ofs1))[(int64_t) iih * a.IW + iiw];
    if (sizeof(T) == 2) ((uint16_t *) dst)[gid] = f2h(v); else ((float *) dst)[gid] = v;
}
void im2col_f32(const tdesc & kernel, const tdesc & x, const tdesc & y, int y_type, const int32_t * p, hipStream_t st) {
    const bool is_2D = p[6] == 1;
    im2col_dev a;
    a.s0 = p[0]; a.s1 = p[1]; a.p0 = p[2]; a.p1 = p[3]; a.d0 = p[4]; a.d1 = p[5];
    a.N  = (int) (is_2D ? x.ne[3] : x.ne[2]); a.IC = (int) (is_2D ? x.ne[2] : x.ne[1]); a.IH = (int) (is_2D ? x.ne[1] : 1); a.IW = (int) x.ne[0];
    a.KH = (int) (is_2D ? kernel.ne[1] : 1);  a.KW = (int) kernel.ne[0];
    a.OH = (int) (is_2D ? y.ne[2] : 1);       a.OW = (int) y.ne[1];
    a.ofs0 = (int64_t) (is_2D ? x.nb[3] : x.nb[2]); a.ofs1 = (int64_t) (is_2D ? x.nb[2] : x.nb[1]);
    const int64_t total = (int64_t) a.N * a.OH * a.OW * a.IC * a.KH * a.KW;
    if (total == 0) return;
    const unsigned grid = (unsigned) ((total + 255) / 256);
    if (y_type == GGML_TYPE_F16) k_im2col<uint16_t><<<dim3(grid), dim3(256), 0, st>>>((const char *) x.p, (uint16_t *) y.p, a, total);
    else                         k_im2col<float><<<dim3(grid), dim3(256), 0, st>>>((const char *) x.p, (float *) y.p, a, total);
}

// ================================================================================================
// POOL_2D / POOL_1D (avg, max).  reference: ggml_compute_forward_pool_2d, ops.cpp:7281-7355 (window clipped at the borders, the
// average still divides by k0*k1) and ggml_compute_forward_pool_1d_sk_p0, ops.cpp:7212-7260 (k == s, no padding: a 2-D pool with a
// 1-row window).  One thread per output cell, the window summed in the reference's order (ky outer, kx inner), `/ ka` a division.
struct pool_dev { int op, k0, k1, s0, s1, p0, p1; int IW, IH, OW, OH; int64_t nb1, nb2; int f16; };
__global__ void __launch_bounds__(256) k_pool2d(const char * __restrict__ x, float * __restrict__ y, const pool_dev a, int64_t total) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int) (i % a.OW); const int64_t r = i / a.OW;
    const int oy = (int) (r % a.OH); const int64_t pl = r / a.OH;
    const char * plane = x + pl * a.nb2;
    float out = a.op == GGML_OP_POOL_AVG ? 0.0f : -FLT_MAX;
    const int ix = -a.p0 + ox * a.s0, iy = -a.p1 + oy * a.s1;
    for (int ky = 0; ky < a.k1; ++ky) {
        if (iy + ky < 0 || iy + ky >= a.IH) continue;
        const char * row = plane + (int64_t) (iy + ky) * a.nb1;
        for (int kx = 0; kx < a.k0; ++kx) {
            const int j = ix + kx;
            if (j < 0 || j >= a.IW) continue;
            const float v = a.f16 ? h2f(((const uint16_t *) row)[j]) : ((const float *) row)[j];
            if (a.op == GGML_OP_POOL_AVG) out += v; else if (v > out) out = v;
        }
    }
    if (a.op == GGML_OP_POOL_AVG) out /= (float) (a.k0 * a.k1);
    y[i] = out;
}
void pool_f32(const tdesc & x, int x_type, const tdesc & y, const int32_t * p, bool is_2d, hipStream_t st) {
    pool_dev a;
    a.op = p[0]; a.f16 = x_type == GGML_TYPE_F16;
    int64_t planes;
    if (is_2d) {
        a.k0 = p[1]; a.k1 = p[2]; a.s0 = p[3]; a.s1 = p[4]; a.p0 = p[5]; a.p1 = p[6];
        a.IW = (int) x.ne[0]; a.IH = (int) x.ne[1]; a.OW = (int) y.ne[0]; a.OH = (int) y.ne[1];
        a.nb1 = (int64_t) x.nb[1]; a.nb2 = (int64_t) x.nb[2]; planes = x.ne[2] * x.ne[3];
    } else {                                                  // [k0, s0, p0] with k0 == s0, p0 == 0 (supports_op): every row of the tensor is one "plane row"
        a.k0 = p[1]; a.k1 = 1; a.s0 = p[2]; a.s1 = 1; a.p0 = 0; a.p1 = 0;
        a.IW = (int) x.ne[0]; a.IH = (int) (x.ne[1] * x.ne[2] * x.ne[3]); a.OW = (int) y.ne[0]; a.OH = a.IH;
        a.nb1 = (int64_t) x.nb[1]; a.nb2 = 0; planes = 1;
    }
    const int64_t total = (int64_t) a.OW * a.OH * planes;
    if (total == 0) return;
    k_pool2d<<<dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st>>>((const char *) x.p, (float *) y.p, a, total);
}

// ================================================================================================
// NORM (LayerNorm without affine part).  reference: ggml_compute_forward_norm_f32, ops.cpp:3450-3495: mean = sum(x)/n,
// variance = sum((x-mean)^2)/n (both sums in double, ggml_vec_sum_f32 / ggml_vec_cvar_f32), y = (x-mean) / sqrt(variance + eps).
// The omni audio / vision encoders normalise with it (tools/omni/audition.cpp, vision.cpp).
// ================================================================================================
__global__ void __launch_bounds__(1024) k_norm(td4 x, td4 y, float eps) {
    __shared__ double red[16];
    const int64_t i1 = blockIdx.x, i2 = blockIdx.y, i3 = blockIdx.z;
    const float * xr = (const float *) (x.p + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    float *       yr = (float *) (y.p + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]);
    const int64_t n = x.ne[0];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += (double) xr[i];
    s = block_sum<double>(s, red);
    const float mean = (float) s / (float) n;
    double v = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) { const float d = xr[i] - mean; v += (double) (d * d); }
    __syncthreads();
    v = block_sum<double>(v, red);
    const float variance = (float) (v / (double) n);
    const float scale = 1.0f / sqrtf(variance + eps);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) yr[i] = (xr[i] - mean) * scale;
}
// many rows (the encoders' [n_state, n_tokens] activations): one WAVE per row, the row in registers, both double sums folded across the wave
// w / b != null: the MUL by the norm weight and the ADD of the norm bias that follow a LayerNorm in the encoders (three roundings to f32, as the
// separate ops do: no contraction into a fused multiply-add); y16 != null: also (or, with y.p == null, only) the f16-rounded row for the GEMM
// sp.part != null: the row does not lie in memory yet -- it is the sum of `nsplit` split-K slabs of the GEMM in front (dense [rows][n] blocks, `split_elems` floats apart) plus up
// to two addends (a bias row with stride 0, the residual), added in k_gemm_reduce_multi's order (slab 0 + slab 1 + ... + addend 1 + addend 2), written to x (the ADD's result,
// the next residual) and normalised from the registers: the reduction launch between a split mat-mul and the LayerNorm behind it (wo / fc2 of an encoder layer) is gone
// w_bs / b_bs: the vectors of dim-2 slice i2 start w_bs / b_bs floats further on (0: one vector for all rows); mod: the result is (n * w + n) + b -- the Token2Wav DiT's
// modulation MUL(n, scale) -> ADD(n, .) -> ADD(., shift) with per-batch-element scale / shift rows, rounded as the three nodes round
struct norm_split_src { const float * part; int nsplit; size_t split_elems; const char * resid; size_t resid_cs; const char * resid2; size_t resid2_cs; size_t w_bs = 0, b_bs = 0; int mod = 0;
                        // gy: the row is first COMPUTED as x = resid + gy * gate (MUL then ADD, rounded as the two nodes round; gate one row per dim-2 slice, g_bs floats apart) and written
                        // to x -- the DiT's gated residual in front of its LayerNorm; resid / gy rows laid out like x
                        const char * gy = nullptr; const char * gres = nullptr; const float * gate = nullptr; size_t g_bs = 0; };
template <int MAXV>
__global__ void __launch_bounds__(256) k_norm_rows(td4 x, td4 y, float eps, int64_t nrows, const float * __restrict__ w, const float * __restrict__ b, char * __restrict__ y16, int64_t y16_rs,
                                                   const norm_split_src sp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int64_t i1 = row % x.ne[1], i2 = (row / x.ne[1]) % x.ne[2], i3 = row / (x.ne[1] * x.ne[2]);
    const char * xr = x.p + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3];
    char *       yr = y.p ? y.p + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3] : nullptr;
    char *       hr = y16 ? y16 + row * y16_rs : nullptr;
    const int n = (int) x.ne[0];
    f32x4 v[MAXV];
    double s = 0.0;
    if (MAXV <= 4 && sp.part) {
        // (rows of at most 1024: every slab's and addend's piece of all the lane's quads requested before the first addition -- a streaming chunk is ~50 rows = 50 waves on the
        //  whole chip, and four dependent round trips per wave made the folded launch cost what the reduction launch had cost)
        f32x4 sl[MAXV][8], r1[MAXV], r2[MAXV];
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int i = (lane + 64 * k) * 4;
            const float * p = sp.part + (size_t) row * n + i;
#pragma unroll
            for (int t = 0; t < 8; ++t) sl[k][t] = (i < n && t < sp.nsplit) ? *(const f32x4 *) (p + t * sp.split_elems) : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
            r1[k] = (i < n && sp.resid)  ? *(const f32x4 *) (sp.resid  + (size_t) row * sp.resid_cs  + (size_t) i * 4) : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
            r2[k] = (i < n && sp.resid2) ? *(const f32x4 *) (sp.resid2 + (size_t) row * sp.resid2_cs + (size_t) i * 4) : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
        }
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int i = (lane + 64 * k) * 4;
            v[k] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
            if (i < n) {
                f32x4 a = sl[k][0];
#pragma unroll
                for (int t = 1; t < 8; ++t) if (t < sp.nsplit) a += sl[k][t];
                if (sp.resid)  a += r1[k];
                if (sp.resid2) a += r2[k];
                *(f32x4 *) (const_cast<char *>(xr) + (size_t) i * 4) = a;
                v[k] = a;
                s += (double) a[0] + (double) a[1] + (double) a[2] + (double) a[3];
            }
        }
    } else
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = (lane + 64 * k) * 4;
        v[k] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
        if (i < n) {
            if (sp.part) {                                           // (launcher: x is 2-D here, row = its second index)
                const float * p = sp.part + (size_t) row * n + i;
                f32x4 sl[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) sl[t] = t < sp.nsplit ? *(const f32x4 *) (p + t * sp.split_elems) : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
                f32x4 r1 = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f }, r2 = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
                if (sp.resid)  r1 = *(const f32x4 *) (sp.resid  + (size_t) row * sp.resid_cs  + (size_t) i * 4);
                if (sp.resid2) r2 = *(const f32x4 *) (sp.resid2 + (size_t) row * sp.resid2_cs + (size_t) i * 4);
                f32x4 a = sl[0];
#pragma unroll
                for (int t = 1; t < 8; ++t) if (t < sp.nsplit) a += sl[t];
                if (sp.resid)  a += r1;
                if (sp.resid2) a += r2;
                *(f32x4 *) (const_cast<char *>(xr) + (size_t) i * 4) = a;
                v[k] = a;
            } else if (sp.gy) {
                const size_t ro = (size_t) (xr - x.p);                  // (resid / gy rows laid out like x)
                const f32x4 yv = *(const f32x4 *) (sp.gy + ro + (size_t) i * 4), rv = *(const f32x4 *) (sp.gres + ro + (size_t) i * 4), gv = *(const f32x4 *) (sp.gate + (size_t) i2 * sp.g_bs + i);
                f32x4 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = __fadd_rn(rv[e], __fmul_rn(yv[e], gv[e]));
                *(f32x4 *) (const_cast<char *>(xr) + (size_t) i * 4) = a;
                v[k] = a;
            } else v[k] = *(const f32x4 *) (xr + (size_t) i * 4);
            s += (double) v[k][0] + (double) v[k][1] + (double) v[k][2] + (double) v[k][3];
        }
    }
    s = wave_sum<double>(s);
    const float mean = (float) s / (float) n;
    double q = 0.0;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = (lane + 64 * k) * 4;
        if (i < n) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mean; v[k][e] = d; q += (double) (d * d); }
        }
    }
    q = wave_sum<double>(q);
    const float variance = (float) (q / (double) n);
    const float scale = 1.0f / sqrtf(variance + eps);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = (lane + 64 * k) * 4;
        if (i >= n) break;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[k][e] * scale;
        if (w) { const f32x4 ww = *(const f32x4 *) (w + (size_t) i2 * sp.w_bs + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = __fmul_rn(o[e], ww[e]); o[e] = sp.mod ? __fadd_rn(o[e], t) : t; }
        }
        if (b) { const f32x4 bb = *(const f32x4 *) (b + (size_t) i2 * sp.b_bs + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(o[e], bb[e]);
        }
        if (yr) *(f32x4 *) (yr + (size_t) i * 4) = o;
        if (hr) { u32x2 h; h[0] = (uint32_t) f2h(o[0]) | ((uint32_t) f2h(o[1]) << 16); h[1] = (uint32_t) f2h(o[2]) | ((uint32_t) f2h(o[3]) << 16); *(u32x2 *) (hr + (size_t) i * 2) = h; }
    }
}
// the rows kernel's shapes: many rows of at most 4096 elements, 16-byte aligned
bool norm_rows_ok(const tdesc & x, const tdesc & y) {
    auto al16 = [](const tdesc & t) { return ((uintptr_t) t.p & 15) == 0 && t.nb[0] == 4 && t.nb[1] % 16 == 0 && t.nb[2] % 16 == 0 && t.nb[3] % 16 == 0; };
    const int64_t n = x.ne[0], nrows = x.ne[1] * x.ne[2] * x.ne[3];
    return nrows >= 2 && n > 0 && n % 4 == 0 && n <= 4096 && al16(x) && al16(y);
}
// LayerNorm rows with the following MUL (w) / ADD (b) by [n] vectors folded in and, optionally, the f16 image of the result (write_f32 false: only that)
void norm_rows_f32(const tdesc & x, const tdesc & y, float eps, const float * w, const float * b, uint16_t * y16, size_t y16_rs, bool write_f32, hipStream_t st, size_t w_bs, size_t b_bs, bool mod,
                   const norm_gate * gate) {
    const int64_t n = x.ne[0], nrows = x.ne[1] * x.ne[2] * x.ne[3];
    if (!norm_rows_ok(x, y) || ((uintptr_t) w & 15) || ((uintptr_t) b & 15) || (y16 && (y16_rs % 8 != 0 || ((uintptr_t) y16 & 7) != 0)) || (!write_f32 && !y16)) { fprintf(stderr, "[mi355x] norm_rows_f32: unsupported arguments\n"); abort(); }
    td4 yd = to_td4(y); if (!write_f32) yd.p = nullptr;
    const dim3 grid((unsigned) ((nrows + 3) / 4));
    norm_split_src sp = { nullptr, 0, 0, nullptr, 0, nullptr, 0 };
    sp.w_bs = w_bs; sp.b_bs = b_bs; sp.mod = mod ? 1 : 0;
    if (gate) {
        if (((uintptr_t) gate->y | (uintptr_t) gate->resid | (uintptr_t) gate->gate) & 15) { fprintf(stderr, "[mi355x] norm_rows_f32: unaligned gated-residual operands\n"); abort(); }
        sp.gy = (const char *) gate->y; sp.gres = (const char *) gate->resid; sp.gate = gate->gate; sp.g_bs = gate->g_bs;
    }
    if (n <= 1024)      k_norm_rows<4><<<grid, dim3(256), 0, st>>>(to_td4(x), yd, eps, nrows, w, b, (char *) y16, (int64_t) y16_rs, sp);
    else if (n <= 2048) k_norm_rows<8><<<grid, dim3(256), 0, st>>>(to_td4(x), yd, eps, nrows, w, b, (char *) y16, (int64_t) y16_rs, sp);
    else                k_norm_rows<16><<<grid, dim3(256), 0, st>>>(to_td4(x), yd, eps, nrows, w, b, (char *) y16, (int64_t) y16_rs, sp);
}
// the same with the rows still lying as split-K slabs (+ up to two addends): x = slabs + addends is written, then normalised
static long g_norm_from_split_launches = 0;
long norm_from_split_launches() { return g_norm_from_split_launches; }
bool norm_rows_from_split_ok(const tdesc & x, const tdesc & y, int nsplit, size_t resid_cs, size_t resid2_cs, const void * resid, const void * resid2, const void * part) {
    return norm_rows_ok(x, y) && x.ne[2] == 1 && x.ne[3] == 1 && nsplit >= 1 && nsplit <= 8 && resid_cs % 16 == 0 && resid2_cs % 16 == 0 &&
           ((uintptr_t) resid & 15) == 0 && ((uintptr_t) resid2 & 15) == 0 && ((uintptr_t) part & 15) == 0;
}
void norm_rows_from_split(const tdesc & x, const tdesc & y, float eps, const float * w, const float * b, uint16_t * y16, size_t y16_rs, bool write_f32,
                          const float * part, int nsplit, size_t split_elems, const float * resid, size_t resid_cs, const float * resid2, size_t resid2_cs, hipStream_t st) {
    const int64_t n = x.ne[0], nrows = x.ne[1];
    if (!norm_rows_from_split_ok(x, y, nsplit, resid_cs, resid2_cs, resid, resid2, part) || ((uintptr_t) w & 15) || ((uintptr_t) b & 15) ||
        (y16 && (y16_rs % 8 != 0 || ((uintptr_t) y16 & 7) != 0)) || (!write_f32 && !y16)) { fprintf(stderr, "[mi355x] norm_rows_from_split: unsupported arguments\n"); abort(); }
    td4 yd = to_td4(y); if (!write_f32) yd.p = nullptr;
    const dim3 grid((unsigned) ((nrows + 3) / 4));
    norm_split_src sp = { part, nsplit, split_elems, (const char *) resid, resid_cs, (const char *) resid2, resid2_cs };
    ++g_norm_from_split_launches;
    if (n <= 1024)      k_norm_rows<4><<<grid, dim3(256), 0, st>>>(to_td4(x), yd, eps, nrows, w, b, (char *) y16, (int64_t) y16_rs, sp);
    else if (n <= 2048) k_norm_rows<8><<<grid, dim3(256), 0, st>>>(to_td4(x), yd, eps, nrows, w, b, (char *) y16, (int64_t) y16_rs, sp);
    else                k_norm_rows<16><<<grid, dim3(256), 0, st>>>(to_td4(x), yd, eps, nrows, w, b, (char *) y16, (int64_t) y16_rs, sp);
}
void norm_f32(const tdesc & x, const tdesc & y, float eps, hipStream_t st) {
    if (x.ne[0] == 0 || x.ne[1] * x.ne[2] * x.ne[3] == 0) return;
    const int64_t n = x.ne[0];
    if (norm_rows_ok(x, y)) { norm_rows_f32(x, y, eps, nullptr, nullptr, nullptr, 0, true, st); return; }
    const int bs = n <= 128 ? 64 : n < 1024 ? 256 : n < 8192 ? 512 : 1024;
    k_norm<<<dim3((unsigned) x.ne[1], (unsigned) x.ne[2], (unsigned) x.ne[3]), dim3(bs), 0, st>>>(to_td4(x), to_td4(y), eps);
}

// ================================================================================================
// ROPE f32.  reference: ggml_compute_forward_rope_f32 ops.cpp:5534-5720, rope_yarn :5443-5458,
// ggml_rope_cache_init :5460-5475, ggml_rope_yarn_corr_dims ggml.c.
// theta for pair i is produced by the SAME sequential product as the reference's cache init
// (theta_0 = pos; theta_{i+1} = theta_i * theta_scale) so the angle is bit-identical; only cosf/sinf
// differ (device libm vs glibc, <= 2 ulp).
// ================================================================================================
struct rope_dev_e {
    int   n_dims, mode;
    float theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1;
};

__global__ void __launch_bounds__(256) k_rope(td4 x, td4 y, const int32_t * __restrict__ pos, const float * __restrict__ ff, rope_dev_e rp) {
    // grid: (ne1 heads, ne2 tokens, ne3); threads over pairs
    const int64_t i1 = blockIdx.x, i2 = blockIdx.y, i3 = blockIdx.z;
    const char * xr = x.p + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3];
    char *       yr = y.p + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3];
    const int ne0 = (int) x.ne[0];
    const bool neox = rp.mode & GGML_ROPE_TYPE_NEOX;
    const float p = (float) pos[i2];
    for (int ip = threadIdx.x; ip < ne0 / 2; ip += blockDim.x) {
        const int i0 = 2 * ip;
        if (i0 < rp.n_dims) {
            float theta = p;
            for (int k = 0; k < ip; ++k) theta *= rp.theta_scale;          // sequential, as ggml_rope_cache_init
            const float f = ff ? ff[ip] : 1.0f;
            const float theta_extrap = theta / f;
            float theta_interp = rp.freq_scale * theta_extrap;
            float th = theta_interp, mscale = rp.attn_factor;
            if (rp.ext_factor != 0.0f) {
                const float yv = ((float) (i0 / 2) - rp.corr0) / fmaxf(0.001f, rp.corr1 - rp.corr0);
                const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, yv))) * rp.ext_factor;
                th = theta_interp * (1.0f - ramp_mix) + theta_extrap * ramp_mix;
                mscale *= 1.0f + 0.1f * logf(1.0f / rp.freq_scale);
            }
            const float c = cosf(th) * mscale, s = sinf(th) * mscale;
            if (neox) {
                const float x0 = *(const float *) (xr + (int64_t) ip * x.nb[0]);
                const float x1 = *(const float *) (xr + (int64_t) (ip + rp.n_dims / 2) * x.nb[0]);
                *(float *) (yr + (int64_t) ip * y.nb[0])                   = x0 * c - x1 * s;
                *(float *) (yr + (int64_t) (ip + rp.n_dims / 2) * y.nb[0]) = x0 * s + x1 * c;
            } else {
                const float x0 = *(const float *) (xr + (int64_t) i0 * x.nb[0]);
                const float x1 = *(const float *) (xr + (int64_t) (i0 + 1) * x.nb[0]);
                *(float *) (yr + (int64_t) i0 * y.nb[0])       = x0 * c - x1 * s;
                *(float *) (yr + (int64_t) (i0 + 1) * y.nb[0]) = x0 * s + x1 * c;
            }
        } else {                                   // pass-through channels beyond n_dims
            *(float *) (yr + (int64_t) i0 * y.nb[0])       = *(const float *) (xr + (int64_t) i0 * x.nb[0]);
            *(float *) (yr + (int64_t) (i0 + 1) * y.nb[0]) = *(const float *) (xr + (int64_t) (i0 + 1) * x.nb[0]);
        }
    }
}

// ggml_rope_yarn_corr_dim / _dims (ggml.c): restated on the host
static float rope_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(base));
}

void rope_f32(const tdesc & x, const int32_t * pos, const float * ff, const tdesc & y, const rope_params & rp, hipStream_t st) {
    if (x.ne[0] * x.ne[1] * x.ne[2] * x.ne[3] == 0) return;
    rope_dev_e d;
    d.n_dims = rp.n_dims; d.mode = rp.mode;
    d.theta_scale = powf(rp.freq_base, -2.0f / rp.n_dims);
    d.freq_scale = rp.freq_scale; d.ext_factor = rp.ext_factor; d.attn_factor = rp.attn_factor;
    const float start = floorf(rope_corr_dim(rp.n_dims, rp.n_ctx_orig, rp.beta_fast, rp.freq_base));
    const float end   = ceilf (rope_corr_dim(rp.n_dims, rp.n_ctx_orig, rp.beta_slow, rp.freq_base));
    d.corr0 = fmaxf(0.0f, start); d.corr1 = fminf((float) rp.n_dims - 1, end);
    dim3 grid((unsigned) x.ne[1], (unsigned) x.ne[2], (unsigned) x.ne[3]);
    k_rope<<<grid, dim3(64), 0, st>>>(to_td4(x), to_td4(y), pos, ff, d);
}

// ================================================================================================
// SOFT_MAX.  reference: ggml_compute_forward_soft_max_f32 ops.cpp:5072-5182:
//   wp = x*scale + slope*mask ; max ; p = expf(wp-max) ; sum in double ; y = p * (1/sum)
// mask is f16 or f32, broadcast over heads (i02 % ne12) and batches (i03 % ne13).
// ================================================================================================
template <bool MASK_F16>
__global__ void __launch_bounds__(1024) k_soft_max(td4 x, td4 y, td4 mk, bool has_mask, const float * __restrict__ sinks,
                                                  float scale, float max_bias, float m0, float m1, uint32_t n_head_log2) {
    extern __shared__ __attribute__((aligned(16))) char sm_lds[];
    __shared__ double redd[16];
    __shared__ float  redf[16];
    float * buf = (float *) sm_lds;
    const int64_t i1 = blockIdx.x, i2 = blockIdx.y, i3 = blockIdx.z;
    const int64_t n = x.ne[0];
    const float * xr = (const float *) (x.p + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    float *       yr = (float *) (y.p + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]);
    const char *  mr = has_mask ? mk.p + i1 * mk.nb[1] + (i2 % mk.ne[2]) * mk.nb[2] + (i3 % mk.ne[3]) * mk.nb[3] : nullptr;
    const uint32_t h = (uint32_t) i2;
    const float slope = max_bias > 0.0f ? (h < n_head_log2 ? powf(m0, (float) (h + 1)) : powf(m1, (float) (2 * (h - n_head_log2) + 1))) : 1.0f;

    float mx = -INFINITY;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        float v = xr[i] * scale;
        if (mr) v += slope * (MASK_F16 ? h2f(((const uint16_t *) mr)[i]) : ((const float *) mr)[i]);
        buf[i] = v;
        mx = fmaxf(mx, v);
    }
    mx = block_max(mx, redf);
    if (sinks) mx = fmaxf(mx, sinks[i2]);
    double sum = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float e = expf(buf[i] - mx);
        buf[i] = e;
        sum += (double) e;
    }
    sum = block_sum<double>(sum, redd);
    if (sinks) sum += (double) expf(sinks[i2] - mx);
    const float inv = (float) (1.0 / sum);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) yr[i] = buf[i] * inv;
}

// Many short rows (the [n_kv, n_tokens, n_head] score block of a prefill ubatch without FLASH_ATTN_EXT): one WAVE per row, the row in
// registers, 16-byte accesses, no LDS and no barrier; optionally the f16 rows the following MUL_MAT (V^T . P) wants as its activation image
// (y16; y.p == null then skips the f32 result).  Same arithmetic as k_soft_max: x * scale + slope * mask, max, expf, sum in double, * (1 / sum).
template <int MAXV, bool MASK_F16>
__global__ void __launch_bounds__(256) k_soft_max_rows(td4 x, td4 y, td4 mk, bool has_mask, float scale, float max_bias, float m0, float m1, uint32_t n_head_log2,
                                                       char * __restrict__ y16, int64_t y16_rs, int64_t nrows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int64_t i1 = row % x.ne[1], i2 = (row / x.ne[1]) % x.ne[2], i3 = row / (x.ne[1] * x.ne[2]);
    const int n = (int) x.ne[0];
    const char * xr = x.p + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3];
    const char * mr = has_mask ? mk.p + i1 * mk.nb[1] + (i2 % mk.ne[2]) * mk.nb[2] + (i3 % mk.ne[3]) * mk.nb[3] : nullptr;
    const uint32_t h = (uint32_t) i2;
    const float slope = max_bias > 0.0f ? (h < n_head_log2 ? powf(m0, (float) (h + 1)) : powf(m1, (float) (2 * (h - n_head_log2) + 1))) : 1.0f;
    f32x4 v[MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = (lane + 64 * k) * 4;
        v[k] = f32x4{ -INFINITY, -INFINITY, -INFINITY, -INFINITY };
        if (i < n) {
            const f32x4 xv = *(const f32x4 *) (xr + (size_t) i * 4);
            f32x4 m = { 0.0f, 0.0f, 0.0f, 0.0f };
            if (mr) {
                if (MASK_F16) { const u32x2 mh = *(const u32x2 *) (mr + (size_t) i * 2); m = f32x4{ h2f((uint16_t) (mh[0] & 0xffff)), h2f((uint16_t) (mh[0] >> 16)), h2f((uint16_t) (mh[1] & 0xffff)), h2f((uint16_t) (mh[1] >> 16)) }; }
                else m = *(const f32x4 *) (mr + (size_t) i * 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { float t = xv[e] * scale; if (mr) t += slope * m[e]; v[k][e] = t; mx = fmaxf(mx, t); }
        }
    }
    mx = wave_max(mx);
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = (lane + 64 * k) * 4;
        if (i < n) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float ex = expf(v[k][e] - mx); v[k][e] = ex; sum += (double) ex; }
        }
    }
    sum = wave_sum<double>(sum);
    const float inv = (float) (1.0 / sum);
    char * yr = y.p ? y.p + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3] : nullptr;
    char * hr = y16 ? y16 + row * y16_rs : nullptr;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = (lane + 64 * k) * 4;
        if (i >= n) break;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[k][e] * inv;
        if (yr) *(f32x4 *) (yr + (size_t) i * 4) = o;
        if (hr) { u32x2 hh; hh[0] = (uint32_t) f2h(o[0]) | ((uint32_t) f2h(o[1]) << 16); hh[1] = (uint32_t) f2h(o[2]) | ((uint32_t) f2h(o[3]) << 16); *(u32x2 *) (hr + (size_t) i * 2) = hh; }
    }
}

bool soft_max_rows_ok(const tdesc & x, const tdesc * mask, int mask_type, const float * sinks, const tdesc & y) {
    const int64_t n = x.ne[0], nrows = x.ne[1] * x.ne[2] * x.ne[3];
    auto al = [](const tdesc & t, int es) { return ((uintptr_t) t.p & 15) == 0 && t.nb[0] == (size_t) es && t.nb[1] % (es * 4) == 0 && t.nb[2] % (es * 4) == 0 && t.nb[3] % (es * 4) == 0; };
    if (sinks || nrows < 64 || n % 4 != 0 || n > 8192 || !al(x, 4) || (y.p && !al(y, 4))) return false;
    if (mask && !al(*mask, mask_type == GGML_TYPE_F16 ? 2 : 4)) return false;
    if (mask && mask_type == GGML_TYPE_F16 && ((uintptr_t) mask->p & 7) != 0) return false;
    return true;
}

void soft_max_f32(const tdesc & x, const tdesc * mask, int mask_type, const float * sinks, const tdesc & y, float scale, float max_bias, hipStream_t st, uint16_t * y16, size_t y16_rs, bool write_f32) {
    if (x.ne[0] * x.ne[1] * x.ne[2] * x.ne[3] == 0) return;
    const int64_t n = x.ne[0];
    if (soft_max_rows_ok(x, mask, mask_type, sinks, y)) {
        const int64_t nrows = x.ne[1] * x.ne[2] * x.ne[3];
        const uint32_t n_head = (uint32_t) x.ne[2];
        const uint32_t nhl2 = 1u << (uint32_t) floorf(log2f((float) n_head));
        const float m0 = powf(2.0f, -(max_bias) / nhl2), m1 = powf(2.0f, -(max_bias / 2.0f) / nhl2);
        td4 yd = to_td4(y); if (!write_f32) yd.p = nullptr;
        td4 mk = mask ? to_td4(*mask) : to_td4(x);
        const dim3 grid((unsigned) ((nrows + 3) / 4));
        const bool mh = mask && mask_type == GGML_TYPE_F16;
#define SM_GO(V) do { if (mh) k_soft_max_rows<V, true><<<grid, dim3(256), 0, st>>>(to_td4(x), yd, mk, mask != nullptr, scale, max_bias, m0, m1, nhl2, (char *) y16, (int64_t) y16_rs, nrows); \
                      else    k_soft_max_rows<V, false><<<grid, dim3(256), 0, st>>>(to_td4(x), yd, mk, mask != nullptr, scale, max_bias, m0, m1, nhl2, (char *) y16, (int64_t) y16_rs, nrows); } while (0)
        if (n <= 512) SM_GO(2); else if (n <= 1024) SM_GO(4); else if (n <= 2048) SM_GO(8); else if (n <= 4096) SM_GO(16); else SM_GO(32);
#undef SM_GO
        return;
    }
    if (y16) { fprintf(stderr, "[mi355x] soft_max: f16 emission needs the row kernel (soft_max_rows_ok)\n"); abort(); }
    int bs = n <= 64 ? 64 : n <= 1024 ? 256 : 1024;
    const uint32_t n_head = (uint32_t) x.ne[2];
    const uint32_t n_head_log2 = 1u << (uint32_t) floorf(log2f((float) n_head));
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2);
    const float m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    dim3 grid((unsigned) x.ne[1], (unsigned) x.ne[2], (unsigned) x.ne[3]);
    const size_t lds = (size_t) n * 4;
    td4 mk = mask ? to_td4(*mask) : to_td4(x);
    if (mask && mask_type == GGML_TYPE_F16) {
        if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *) k_soft_max<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        k_soft_max<true><<<grid, dim3(bs), lds, st>>>(to_td4(x), to_td4(y), mk, true, sinks, scale, max_bias, m0, m1, n_head_log2);
    } else {
        if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *) k_soft_max<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        k_soft_max<false><<<grid, dim3(bs), lds, st>>>(to_td4(x), to_td4(y), mk, mask != nullptr, sinks, scale, max_bias, m0, m1, n_head_log2);
    }
}

// ================================================================================================
// GLU family (ops.cpp:2934-2990 swiglu; reglu/geglu variants alongside) and unary ops (unary-ops.cpp)
// ================================================================================================
static __device__ __forceinline__ float op_silu(float x) { return x / (1.0f + expf(-x)); }                 // vec.h:958
// GELU / GELU_QUICK of an f32 value go through the reference's f16 tables (GGML_GELU_FP16 / GGML_GELU_QUICK_FP16, vec.h:17-18, :892-906, :933-941;
// tables filled in ggml-cpu.c:3555-3556 with f16(ggml_gelu_f32(f)) for every f16 value f): the argument is rounded to f16, the formula evaluated in
// f32 and the result rounded to f16 -- evaluated here instead of looked up (same value unless the device tanhf / expf differs from glibc's by more
// than the f16 rounding absorbs); GELU short-cuts x <= -10 to 0 and x >= 10 to x before the table.
// (op_gelu / op_gelu_quick: kernels/act_dev.hpp -- the split-K reduction applies them too)
static __device__ __forceinline__ float op_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__global__ void __launch_bounds__(256) k_glu(int op, const char * __restrict__ a, int64_t a_rs, const char * __restrict__ b, int64_t b_rs,
                                            char * __restrict__ y, int64_t y_rs, int64_t nc, int64_t nr, char * __restrict__ y16, int64_t y16_rs) {
    const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nc * nr) return;
    const int64_t r = t / nc, i = t % nc;
    const float x = ((const float *) (a + r * a_rs))[i];
    const float g = ((const float *) (b + r * b_rs))[i];
    float v;
    switch (op) {
        case GGML_GLU_OP_REGLU:       v = (x > 0.0f ? x : 0.0f) * g; break;
        case GGML_GLU_OP_GEGLU:       v = op_gelu(x) * g; break;
        case GGML_GLU_OP_SWIGLU:      v = op_silu(x) * g; break;
        case GGML_GLU_OP_GEGLU_ERF:   v = op_gelu_erf(x) * g; break;
        case GGML_GLU_OP_GEGLU_QUICK: v = op_gelu_quick(x) * g; break;
        default: v = 0.0f;
    }
    if (y)   ((float *) (y + r * y_rs))[i] = v;
    if (y16) ((uint16_t *) (y16 + r * y16_rs))[i] = f2h(v);           // activation image of the following GEMM
}

// SWIGLU on 16-byte aligned rows: four elements per lane, one row per blockIdx.y (no 64-bit index arithmetic) -- the prefill shape
__global__ void __launch_bounds__(256) k_swiglu_v4(const char * __restrict__ a, int64_t a_rs, const char * __restrict__ b, int64_t b_rs,
                                                  char * __restrict__ y, int64_t y_rs, int nc4, char * __restrict__ y16, int64_t y16_rs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nc4) return;
    const int64_t r = blockIdx.y;
    const f32x4 x = ((const f32x4 *) (a + r * a_rs))[i];
    const f32x4 g = ((const f32x4 *) (b + r * b_rs))[i];
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = op_silu(x[e]) * g[e];
    if (y) ((f32x4 *) (y + r * y_rs))[i] = v;
    if (y16) {
        u32x2 h;
        h[0] = (uint32_t) f2h(v[0]) | ((uint32_t) f2h(v[1]) << 16); h[1] = (uint32_t) f2h(v[2]) | ((uint32_t) f2h(v[3]) << 16);
        ((u32x2 *) (y16 + r * y16_rs))[i] = h;
    }
}

// SWIGLU whose only consumers are K-quant mat-muls (ffn_down at several columns): one wave per (row, 256-element block) computes
// silu(a) * b and writes the Q8_K block of the row's activation image directly (quantize_row_q8_K arithmetic, common.hpp) --
// the separate quantiser launch disappears; y (the f32 result) is optional
__global__ void __launch_bounds__(256) k_swiglu_q8k(const char * __restrict__ a, int64_t a_rs, const char * __restrict__ b, int64_t b_rs,
                                                   char * __restrict__ y, int64_t y_rs, char * __restrict__ img, size_t img_bytes, int nblk, int64_t total) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= total) return;
    const int64_t r = wid / nblk; const int ib = (int) (wid - r * nblk);
    const f32x4 x = *(const f32x4 *) (a + r * a_rs + (int64_t) ib * 1024 + lane * 16);
    const f32x4 g = *(const f32x4 *) (b + r * b_rs + (int64_t) ib * 1024 + lane * 16);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = op_silu(x[e]) * g[e];
    if (y) *(f32x4 *) (y + r * y_rs + (int64_t) ib * 1024 + lane * 16) = v;
    char * im = img + r * img_bytes;
    const int64_t K = (int64_t) nblk * 256;
    q8k_block_from_regs(v, lane, (int8_t *) im + ib * 256, (int16_t *) (im + K) + ib * 16, (float *) (im + K + K / 8) + ib);
}
bool swiglu_q8k_ok(const tdesc & a, const tdesc & b, const tdesc & y) {
    return y.ne[0] % 256 == 0 && y.ne[2] == 1 && y.ne[3] == 1 && ((uintptr_t) a.p & 15) == 0 && ((uintptr_t) b.p & 15) == 0 && a.nb[1] % 16 == 0 && b.nb[1] % 16 == 0 &&
           ((uintptr_t) y.p & 15) == 0 && y.nb[1] % 16 == 0;
}
void swiglu_q8k(const tdesc & a, const tdesc & b, const tdesc & y, bool write_f32, void * img, hipStream_t st) {
    const int nblk = (int) (y.ne[0] / 256);
    const int64_t total = (int64_t) nblk * y.ne[1];
    if (total == 0) return;
    k_swiglu_q8k<<<dim3((unsigned) ((total + 3) / 4)), dim3(256), 0, st>>>((const char *) a.p, (int64_t) a.nb[1], (const char *) b.p, (int64_t) b.nb[1],
                                                                         write_f32 ? (char *) y.p : nullptr, (int64_t) y.nb[1], (char *) img, q8k_image_bytes(y.ne[0]), nblk, total);
}

void glu_f32(int glu_op, const tdesc & a, const tdesc * b, bool swapped, const tdesc & y, hipStream_t st, uint16_t * y16, size_t y16_rs, bool write_f32) {
    // rows are contiguous_1 (checked by supports_op): treat as [nc, nr] with a row stride
    const int64_t nc = y.ne[0];
    const int64_t nr = y.ne[1] * y.ne[2] * y.ne[3];
    if (nc * nr == 0) return;
    const char * ap = (const char *) a.p; const char * bp;
    int64_t a_rs = (int64_t) a.nb[1], b_rs;
    if (b) { bp = (const char *) b->p; b_rs = (int64_t) b->nb[1]; }
    else   { bp = ap + (swapped ? 0 : nc * 4); ap = ap + (swapped ? nc * 4 : 0); b_rs = a_rs; }
    if (glu_op == GGML_GLU_OP_SWIGLU && nc % 4 == 0 && nr <= 65535 && ((uintptr_t) ap & 15) == 0 && ((uintptr_t) bp & 15) == 0 && a_rs % 16 == 0 && b_rs % 16 == 0 &&
        ((uintptr_t) y.p & 15) == 0 && y.nb[1] % 16 == 0 && ((uintptr_t) y16 & 7) == 0 && y16_rs % 8 == 0) {
        k_swiglu_v4<<<dim3((unsigned) ((nc / 4 + 255) / 256), (unsigned) nr), dim3(256), 0, st>>>(ap, a_rs, bp, b_rs, write_f32 ? (char *) y.p : nullptr, (int64_t) y.nb[1],
                                                                                              (int) (nc / 4), (char *) y16, (int64_t) y16_rs);
        return;
    }
    k_glu<<<dim3((unsigned) ((nc * nr + 255) / 256)), dim3(256), 0, st>>>(glu_op, ap, a_rs, bp, b_rs, write_f32 ? (char *) y.p : nullptr, (int64_t) y.nb[1], nc, nr, (char *) y16, (int64_t) y16_rs);
}

static __device__ __forceinline__ float unary_apply(int op, float v) {
    switch (op) {
        case GGML_UNARY_OP_ABS:        return fabsf(v);
        case GGML_UNARY_OP_SGN:        return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
        case GGML_UNARY_OP_NEG:        return -v;
        case GGML_UNARY_OP_STEP:       return v > 0.0f ? 1.0f : 0.0f;
        case GGML_UNARY_OP_TANH:       return tanhf(v);
        case GGML_UNARY_OP_ELU:        return v > 0.0f ? v : expm1f(v);
        case GGML_UNARY_OP_RELU:       return v > 0.0f ? v : 0.0f;
        case GGML_UNARY_OP_SIGMOID:    return 1.0f / (1.0f + expf(-v));
        case GGML_UNARY_OP_GELU:       return op_gelu(v);
        case GGML_UNARY_OP_GELU_QUICK: return op_gelu_quick(v);
        case GGML_UNARY_OP_SILU:       return op_silu(v);
        case GGML_UNARY_OP_HARDSWISH:  return v * fminf(1.0f, fmaxf(0.0f, (v + 3.0f) / 6.0f));
        case GGML_UNARY_OP_HARDSIGMOID:return fminf(1.0f, fmaxf(0.0f, (v + 3.0f) / 6.0f));
        case GGML_UNARY_OP_EXP:        return expf(v);
        case GGML_UNARY_OP_GELU_ERF:   return op_gelu_erf(v);
        default: return v;
    }
}
__global__ void __launch_bounds__(256) k_unary(int op, const float * __restrict__ x, float * __restrict__ y, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] = unary_apply(op, x[i]);
}
// four elements per thread (16-byte aligned, n % 4 == 0); y16 != null: also (or, with y == null, only) the f16-rounded values -- the dense f16
// activation image of the MFMA GEMM that reads the result (the encoders' GELU between fc1 and fc2)
__global__ void __launch_bounds__(256) k_unary_v4(int op, const f32x4 * __restrict__ x, f32x4 * __restrict__ y, u32x2 * __restrict__ y16, int64_t n4) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = x[i]; f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = unary_apply(op, v[e]);
    if (y) y[i] = r;
    if (y16) { u32x2 h; h[0] = (uint32_t) f2h(r[0]) | ((uint32_t) f2h(r[1]) << 16); h[1] = (uint32_t) f2h(r[2]) | ((uint32_t) f2h(r[3]) << 16); y16[i] = h; }
}
void unary_f32(int uop, const float * x, float * y, int64_t n, hipStream_t st, uint16_t * y16, bool write_f32) {
    if (n == 0) return;
    if (n % 4 == 0 && (((uintptr_t) x | (uintptr_t) y) & 15) == 0 && ((uintptr_t) y16 & 7) == 0) {
        k_unary_v4<<<dim3((unsigned) ((n / 4 + 255) / 256)), dim3(256), 0, st>>>(uop, (const f32x4 *) x, write_f32 ? (f32x4 *) y : nullptr, (u32x2 *) y16, n / 4);
        return;
    }
    if (y16 || !write_f32) { fprintf(stderr, "[mi355x] unary_f32: f16 emission needs 16-byte aligned operands and n %% 4 == 0\n"); abort(); }
    k_unary<<<dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st>>>(uop, x, y, n);
}

__global__ void __launch_bounds__(256) k_scale(const float * __restrict__ x, float * __restrict__ y, int64_t n, float s, float b) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * s + b;
}
void scale_f32(const float * x, float * y, int64_t n, float s, float b, hipStream_t st) {
    if (n == 0) return;
    k_scale<<<dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st>>>(x, y, n, s, b);
}

// ================================================================================================
// ADD / SUB / MUL / DIV with broadcast (binary-ops.cpp): dst and src0 same shape, src1 repeats.
// ================================================================================================
template <int OP>
__global__ void __launch_bounds__(256) k_bin(td4 a, td4 b, td4 y) {
    const int64_t n0 = y.ne[0];
    const int64_t total = y.ne[0] * y.ne[1] * y.ne[2] * y.ne[3];
    for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t) gridDim.x * blockDim.x) {
        const int64_t i0 = t % n0; int64_t r = t / n0;
        const int64_t i1 = r % y.ne[1]; r /= y.ne[1];
        const int64_t i2 = r % y.ne[2]; const int64_t i3 = r / y.ne[2];
        const float va = *(const float *) (a.p + i0 * a.nb[0] + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
        const float vb = *(const float *) (b.p + (i0 % b.ne[0]) * b.nb[0] + (i1 % b.ne[1]) * b.nb[1] + (i2 % b.ne[2]) * b.nb[2] + (i3 % b.ne[3]) * b.nb[3]);
        float v;
        if (OP == GGML_OP_ADD) v = va + vb; else if (OP == GGML_OP_SUB) v = va - vb; else if (OP == GGML_OP_MUL) v = va * vb; else v = va / vb;
        *(float *) (y.p + i0 * y.nb[0] + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]) = v;
    }
}
// dense operands, four elements per thread, 32-bit indices: b either has a's shape or is ONE row broadcast over every row (bias adds, norm
// gains, the DiT's modulation vectors) -- the generic kernel's eight 64-bit div / mod per element made a bias add over [1024, 1500] 11.6 us
template <int OP, bool BROW>
__global__ void __launch_bounds__(256) k_bin_v4(const f32x4 * __restrict__ a, const f32x4 * __restrict__ b, f32x4 * __restrict__ y, uint32_t total4, uint32_t n04) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total4) return;
    const f32x4 va = a[i], vb = b[BROW ? i % n04 : i];
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = OP == GGML_OP_ADD ? va[e] + vb[e] : OP == GGML_OP_SUB ? va[e] - vb[e] : OP == GGML_OP_MUL ? va[e] * vb[e] : va[e] / vb[e];
    y[i] = v;
}
template <int OP>
static bool bin_v4_go(const tdesc & a, const tdesc & b, const tdesc & y, hipStream_t st) {
    const int64_t total = y.ne[0] * y.ne[1] * y.ne[2] * y.ne[3];
    auto dense = [](const tdesc & t) { return t.nb[0] == 4 && t.nb[1] == (size_t) t.ne[0] * 4 && t.nb[2] == t.nb[1] * (size_t) t.ne[1] && t.nb[3] == t.nb[2] * (size_t) t.ne[2] && ((uintptr_t) t.p & 15) == 0; };
    if (total >= (1ll << 31) || y.ne[0] % 4 != 0 || !dense(a) || !dense(y) || !dense(b)) return false;
    for (int i = 0; i < 4; ++i) if (a.ne[i] != y.ne[i]) return false;
    const bool same = b.ne[0] == y.ne[0] && b.ne[1] == y.ne[1] && b.ne[2] == y.ne[2] && b.ne[3] == y.ne[3];
    const bool row  = b.ne[0] == y.ne[0] && b.ne[1] * b.ne[2] * b.ne[3] == 1;
    if (!same && !row) return false;
    const uint32_t total4 = (uint32_t) (total / 4), n04 = (uint32_t) (y.ne[0] / 4);
    const dim3 grid((total4 + 255) / 256);
    if (same) k_bin_v4<OP, false><<<grid, dim3(256), 0, st>>>((const f32x4 *) a.p, (const f32x4 *) b.p, (f32x4 *) y.p, total4, n04);
    else      k_bin_v4<OP, true><<<grid, dim3(256), 0, st>>>((const f32x4 *) a.p, (const f32x4 *) b.p, (f32x4 *) y.p, total4, n04);
    return true;
}
void bin_bcast_f32(int op, const tdesc & a, const tdesc & b, const tdesc & y, hipStream_t st) {
    const int64_t total = y.ne[0] * y.ne[1] * y.ne[2] * y.ne[3];
    if (total == 0) return;
    if (total >= 4096) {
        bool done = false;
        switch (op) {
            case GGML_OP_ADD: done = bin_v4_go<GGML_OP_ADD>(a, b, y, st); break;
            case GGML_OP_SUB: done = bin_v4_go<GGML_OP_SUB>(a, b, y, st); break;
            case GGML_OP_MUL: done = bin_v4_go<GGML_OP_MUL>(a, b, y, st); break;
            case GGML_OP_DIV: done = bin_v4_go<GGML_OP_DIV>(a, b, y, st); break;
            default: break;
        }
        if (done) return;
    }
    int64_t g = (total + 255) / 256; if (g > 8192) g = 8192;
    dim3 grid((unsigned) g), blk(256);
    switch (op) {
        case GGML_OP_ADD: k_bin<GGML_OP_ADD><<<grid, blk, 0, st>>>(to_td4(a), to_td4(b), to_td4(y)); break;
        case GGML_OP_SUB: k_bin<GGML_OP_SUB><<<grid, blk, 0, st>>>(to_td4(a), to_td4(b), to_td4(y)); break;
        case GGML_OP_MUL: k_bin<GGML_OP_MUL><<<grid, blk, 0, st>>>(to_td4(a), to_td4(b), to_td4(y)); break;
        case GGML_OP_DIV: k_bin<GGML_OP_DIV><<<grid, blk, 0, st>>>(to_td4(a), to_td4(b), to_td4(y)); break;
        default: abort();
    }
}

// ================================================================================================
// chains of element-wise nodes (kernels.hpp ew_chain_args): the reference's Token2Wav graphs spell Mish as sub / exp / exp / add / log / tanh / mul and the DiT's
// modulation as mul / add / add -- 3500 of a window's launches are links of such chains, each a ~4 us dependent launch over a few hundred KB
// ================================================================================================
struct ew_chain_dev { int n_ops, n_in; uint32_t total4; ew_op_desc op[8]; const float * in[6]; int in_mode[6]; uint32_t in_n04[6], in_per4[6], in_bs4[6]; float * out; };
static __device__ __forceinline__ float ew_apply(const ew_op_desc & o, float a, float b) {
    switch (o.kind) {
        case GGML_OP_ADD:        return a + b;
        case GGML_OP_SUB:        return a - b;
        case GGML_OP_MUL:        return a * b;
        case GGML_OP_DIV:        return a / b;
        case GGML_OP_SCALE:      return a * o.p0 + o.p1;                                   // k_scale
        case GGML_OP_UNARY:      return unary_apply(o.sub, a);
        case GGML_OP_SQR:        return a * a;                                             // k_math (t2w_ops.hip)
        case GGML_OP_SQRT:       return sqrtf(a);
        case GGML_OP_LOG:        return logf(a);
        case GGML_OP_SIN:        return sinf(a);
        case GGML_OP_COS:        return cosf(a);
        case GGML_OP_CLAMP:      return fmaxf(fminf(a, o.p1), o.p0);
        case GGML_OP_LEAKY_RELU: return (a > 0.0f ? a : 0.0f) + o.p0 * (a < 0.0f ? a : 0.0f);
        default:                 return a;
    }
}
// V elements per thread: 4 (16-byte accesses), or 1 when the chain is short on threads and long on arithmetic -- Mish over a DiT activation of 57 344 floats is exp, exp,
// log, tanh per element: 56 workgroups of four-element threads left three quarters of the chip idle behind a ~10 us dependent instruction chain per thread
template <int V>
__global__ void __launch_bounds__(256) k_ew_chain(const ew_chain_dev c) {
    typedef float vec __attribute__((ext_vector_type(V)));
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= c.total4 * (4 / V)) return;
    vec v[6 + 8];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (k >= c.n_in) break;
        const uint32_t n0 = c.in_n04[k] * (4 / V);
        uint32_t idx = i;
        if (c.in_mode[k] == 2) idx = 0;
        else if (c.in_mode[k] == 3) idx = (i / (c.in_per4[k] * (4 / V))) * (c.in_bs4[k] * (4 / V)) + i % n0;
        else if (c.in_mode[k] == 1) idx = i % n0;
        if (c.in_mode[k] == 2) { const float s = c.in[k][0]; vec t; for (int e = 0; e < V; ++e) t[e] = s; v[k] = t; }
        else v[k] = ((const vec *) c.in[k])[idx];
    }
    vec r = v[0];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j >= c.n_ops) break;
        const ew_op_desc o = c.op[j];
        vec a = r, b = r;
        // (selectors are wave-uniform: a short uniform switch instead of dynamically indexed registers)
#pragma unroll
        for (int k = 0; k < 14; ++k) { if ((o.a < 8 ? o.a : 6 + o.a - 8) == k) a = v[k]; if ((o.b < 8 ? o.b : 6 + o.b - 8) == k) b = v[k]; }
#pragma unroll
        for (int e = 0; e < V; ++e) r[e] = ew_apply(o, a[e], b[e]);
        v[6 + j] = r;
    }
    ((vec *) c.out)[i] = r;
}
void ew_chain(const ew_chain_args & a, hipStream_t st) {
    if (a.total == 0) return;
    if (a.n_ops < 1 || a.n_ops > 8 || a.n_in < 1 || a.n_in > 6 || a.total % 4 != 0 || a.total / 4 >= (1ll << 30) || ((uintptr_t) a.out & 15) != 0) { fprintf(stderr, "[mi355x] ew_chain: unsupported arguments\n"); abort(); }
    ew_chain_dev c;
    c.n_ops = a.n_ops; c.n_in = a.n_in; c.total4 = (uint32_t) (a.total / 4); c.out = a.out;
    for (int j = 0; j < 8; ++j) c.op[j] = a.op[j < a.n_ops ? j : 0];
    for (int k = 0; k < 6; ++k) { const int q = k < a.n_in ? k : 0; c.in[k] = a.in[q]; c.in_mode[k] = a.in_mode[q]; c.in_n04[k] = a.in_n04[q] ? a.in_n04[q] : 1; c.in_per4[k] = a.in_per4[q] ? a.in_per4[q] : 1; c.in_bs4[k] = a.in_bs4[q]; }
    int heavy = 0;                                                  // libm-grade ops in the chain
    for (int j = 0; j < a.n_ops; ++j) heavy += a.op[j].kind == GGML_OP_UNARY || a.op[j].kind == GGML_OP_LOG || a.op[j].kind == GGML_OP_SIN || a.op[j].kind == GGML_OP_COS;
    static const bool no_v1 = getenv("MI355X_EW_CHAIN_NO_V1") != nullptr;
    if (!no_v1 && heavy >= 2 && c.total4 <= 256u * 1024u) k_ew_chain<1><<<dim3((c.total4 * 4 + 255) / 256), dim3(256), 0, st>>>(c);
    else k_ew_chain<4><<<dim3((c.total4 + 255) / 256), dim3(256), 0, st>>>(c);
}

// ================================================================================================
// CPY / CONT / DUP: element i (row-major over src->ne) of src -> element i of dst (row-major over dst->ne)
// (ggml_compute_forward_dup, ops.cpp): types f32 <-> f16, arbitrary strides on both sides.
// ================================================================================================
template <typename TS, typename TD> static __device__ __forceinline__ TD cvt_elem(TS v);
template <> __device__ __forceinline__ float    cvt_elem<float, float>(float v)          { return v; }
template <> __device__ __forceinline__ uint16_t cvt_elem<float, uint16_t>(float v)       { return f2h(v); }
template <> __device__ __forceinline__ float    cvt_elem<uint16_t, float>(uint16_t v)    { return h2f(v); }
template <> __device__ __forceinline__ uint16_t cvt_elem<uint16_t, uint16_t>(uint16_t v) { return v; }
template <> __device__ __forceinline__ int32_t  cvt_elem<int32_t, int32_t>(int32_t v)    { return v; }
struct bf16_t { uint16_t v; };
template <> __device__ __forceinline__ float    cvt_elem<bf16_t, float>(bf16_t b)        { return __uint_as_float((uint32_t) b.v << 16); }

// IX: index type of the element counter -- 32-bit whenever the tensor has fewer than 2^31 elements (a 64-bit div/mod is a ~100-instruction
// sequence and there are eight per element: the KQ-mask cast of every libllama decode graph took 63 us with them, 2.4 % of the step)
template <typename TS, typename TD, typename IX>
__global__ void __launch_bounds__(256) k_cpy(td4 s, td4 d) {
    const IX total = (IX) (s.ne[0] * s.ne[1] * s.ne[2] * s.ne[3]);
    const IX sn0 = (IX) s.ne[0], sn1 = (IX) s.ne[1], sn2 = (IX) s.ne[2], dn0 = (IX) d.ne[0], dn1 = (IX) d.ne[1], dn2 = (IX) d.ne[2];
    for (IX t = (IX) blockIdx.x * (IX) blockDim.x + (IX) threadIdx.x; t < total; t += (IX) gridDim.x * (IX) blockDim.x) {
        IX r = t;
        const int64_t s0 = r % sn0; r /= sn0;
        const int64_t s1 = r % sn1; r /= sn1;
        const int64_t s2 = r % sn2; const int64_t s3 = r / sn2;
        r = t;
        const int64_t d0 = r % dn0; r /= dn0;
        const int64_t d1 = r % dn1; r /= dn1;
        const int64_t d2 = r % dn2; const int64_t d3 = r / dn2;
        const TS v = *(const TS *) (s.p + s0 * s.nb[0] + s1 * s.nb[1] + s2 * s.nb[2] + s3 * s.nb[3]);
        *(TD *) (d.p + d0 * d.nb[0] + d1 * d.nb[1] + d2 * d.nb[2] + d3 * d.nb[3]) = cvt_elem<TS, TD>(v);
    }
}
// both sides dense in the same element order: a linear convert (the f32 -> f16 KQ-mask cast, CONT of dense tensors)
template <typename TS, typename TD>
__global__ void __launch_bounds__(256) k_cpy_lin(const TS * __restrict__ s, TD * __restrict__ d, int64_t total) {
    for (int64_t t = (int64_t) blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t) gridDim.x * 256) d[t] = cvt_elem<TS, TD>(s[t]);
}
static bool dense_rowmajor(const tdesc & t, size_t es) {
    size_t nb = es;
    for (int i = 0; i < 4; ++i) { if (t.ne[i] != 1 && t.nb[i] != nb) return false; nb *= (size_t) t.ne[i]; }
    return true;
}
template <typename TS, typename TD>
static void cpy_go(const tdesc & src, const tdesc & dst, int64_t total, hipStream_t st) {
    int64_t g = (total + 255) / 256; if (g > 16384) g = 16384;
    const dim3 grid((unsigned) g), blk(256);
    if (dense_rowmajor(src, sizeof(TS)) && dense_rowmajor(dst, sizeof(TD))) { k_cpy_lin<TS, TD><<<grid, blk, 0, st>>>((const TS *) src.p, (TD *) dst.p, total); return; }
    const td4 s = to_td4(src), d = to_td4(dst);
    if (total < ((int64_t) 1 << 31)) k_cpy<TS, TD, uint32_t><<<grid, blk, 0, st>>>(s, d);
    else                             k_cpy<TS, TD, int64_t><<<grid, blk, 0, st>>>(s, d);
}
void cpy_strided(const tdesc & src, int src_type, const tdesc & dst, int dst_type, hipStream_t st) {
    const int64_t total = src.ne[0] * src.ne[1] * src.ne[2] * src.ne[3];
    if (total == 0) return;
    if      (src_type == GGML_TYPE_F32 && dst_type == GGML_TYPE_F32) cpy_go<float, float>(src, dst, total, st);
    else if (src_type == GGML_TYPE_F32 && dst_type == GGML_TYPE_F16) cpy_go<float, uint16_t>(src, dst, total, st);
    else if (src_type == GGML_TYPE_F16 && dst_type == GGML_TYPE_F32) cpy_go<uint16_t, float>(src, dst, total, st);
    else if (src_type == GGML_TYPE_F16 && dst_type == GGML_TYPE_F16) cpy_go<uint16_t, uint16_t>(src, dst, total, st);
    else if (src_type == GGML_TYPE_I32 && dst_type == GGML_TYPE_I32) cpy_go<int32_t, int32_t>(src, dst, total, st);
    else { fprintf(stderr, "[mi355x] cpy: unsupported %d -> %d\n", src_type, dst_type); abort(); }
}

// ---- a batch of same-type strided copies in ONE launch (round 6): runs of layout-only nodes (CONT / CONCAT / CPY of Token2Wav's cache packing, the head-flattening copies in
// front of an attention chain) are mutually independent more often than not, and each is a ~2.7 us launch that moves a few KB.  The executor queues them (graph_exec.cpp
// copy_queue: byte-range hazards against everything pending) and hands up to COPY_BATCH_MAX of them over by value: blockIdx.y = the job, blockIdx.x strides over its units.
// A job is the linear-index copy of k_cpy (source and destination decompose the same element index over their OWN shapes), its element 2, 4 or -- when both innermost
// dimensions are contiguous, a multiple of 16 bytes and everything is 16-byte aligned -- 16 bytes wide.
struct copy_job_dev { const char * sp; char * dp; int sne[4]; int dne[4]; long long snb[4]; long long dnb[4]; unsigned total; int es; };
struct copy_batch_dev { copy_job_dev j[COPY_BATCH_MAX]; };
template <typename T>
static __device__ __forceinline__ void copy_job_run(const copy_job_dev & J) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < J.total; i += gridDim.x * blockDim.x) {
        unsigned r = i;
        const unsigned s0 = r % (unsigned) J.sne[0]; r /= (unsigned) J.sne[0];
        const unsigned s1 = r % (unsigned) J.sne[1]; r /= (unsigned) J.sne[1];
        const unsigned s2 = r % (unsigned) J.sne[2]; const unsigned s3 = r / (unsigned) J.sne[2];
        r = i;
        const unsigned d0 = r % (unsigned) J.dne[0]; r /= (unsigned) J.dne[0];
        const unsigned d1 = r % (unsigned) J.dne[1]; r /= (unsigned) J.dne[1];
        const unsigned d2 = r % (unsigned) J.dne[2]; const unsigned d3 = r / (unsigned) J.dne[2];
        *(T *) (J.dp + d0 * J.dnb[0] + d1 * J.dnb[1] + d2 * J.dnb[2] + d3 * J.dnb[3]) = *(const T *) (J.sp + s0 * J.snb[0] + s1 * J.snb[1] + s2 * J.snb[2] + s3 * J.snb[3]);
    }
}
__global__ void __launch_bounds__(256) k_copy_batch(const copy_batch_dev b) {
    const copy_job_dev & J = b.j[blockIdx.y];
    if (J.es == 16) copy_job_run<uint4>(J); else if (J.es == 4) copy_job_run<uint32_t>(J); else copy_job_run<uint16_t>(J);
}
bool copy_batch_ok(const tdesc & src, const tdesc & dst, int es) {
    if (es != 2 && es != 4) return false;
    int64_t ts = 1, tdn = 1;
    for (int i = 0; i < 4; ++i) { if (src.ne[i] < 1 || dst.ne[i] < 1 || src.ne[i] > 0x7fffffff || dst.ne[i] > 0x7fffffff) return false; ts *= src.ne[i]; tdn *= dst.ne[i]; }
    return ts == tdn && ts < ((int64_t) 1 << 31);
}
void copy_batch(const copy_pair * jobs, int n, hipStream_t st) {
    if (n < 1 || n > COPY_BATCH_MAX) { fprintf(stderr, "[mi355x] copy_batch: %d jobs\n", n); abort(); }
    copy_batch_dev b;
    unsigned max_units = 1;
    for (int k = 0; k < n; ++k) {
        const tdesc & s = jobs[k].src; const tdesc & d = jobs[k].dst; const int es = jobs[k].es;
        if (!copy_batch_ok(s, d, es)) { fprintf(stderr, "[mi355x] copy_batch: job %d not a same-type copy of < 2^31 elements\n", k); abort(); }
        copy_job_dev & J = b.j[k];
        J.sp = (const char *) s.p; J.dp = (char *) d.p; J.es = es;
        int64_t total = 1;
        for (int i = 0; i < 4; ++i) { J.sne[i] = (int) s.ne[i]; J.dne[i] = (int) d.ne[i]; J.snb[i] = (long long) s.nb[i]; J.dnb[i] = (long long) d.nb[i]; total *= s.ne[i]; }
        const int V = 16 / es;
        bool wide = s.nb[0] == (size_t) es && d.nb[0] == (size_t) es && s.ne[0] % V == 0 && d.ne[0] % V == 0 && (((uintptr_t) s.p | (uintptr_t) d.p) & 15) == 0;
        for (int i = 1; i < 4 && wide; ++i) if (((s.ne[i] > 1 ? s.nb[i] : 0) | (d.ne[i] > 1 ? d.nb[i] : 0)) & 15) wide = false;
        if (wide) { J.sne[0] /= V; J.dne[0] /= V; J.snb[0] = 16; J.dnb[0] = 16; J.es = 16; total /= V; }
        J.total = (unsigned) total;
        if (J.total > max_units) max_units = J.total;
    }
    unsigned gx = (max_units + 255) / 256; if (gx > 2048) gx = 2048;
    k_copy_batch<<<dim3(gx, (unsigned) n), dim3(256), 0, st>>>(b);
}

// ================================================================================================
// GET_ROWS (ops.cpp:4500-4665): dst[:, i10, i11, i12] = to_float(src0[:, idx[i10,i11,i12], i11, i12])
// ================================================================================================
template <typename TS>
__global__ void __launch_bounds__(256) k_get_rows(td4 s, td4 idx, td4 d) {
    const int64_t i10 = blockIdx.x, i11 = blockIdx.y, i12 = blockIdx.z;
    const int64_t row = *(const int32_t *) (idx.p + i10 * idx.nb[0] + i11 * idx.nb[1] + i12 * idx.nb[2]);
    const char * sr = s.p + row * s.nb[1] + i11 * s.nb[2] + i12 * s.nb[3];
    float *      dr = (float *) (d.p + i10 * d.nb[1] + i11 * d.nb[2] + i12 * d.nb[3]);
    for (int64_t i = threadIdx.x; i < s.ne[0]; i += blockDim.x) dr[i] = cvt_elem<TS, float>(((const TS *) sr)[i]);
}
__global__ void k_get_rows_ptrs(td4 s, td4 idx, td4 d, const char ** src_rows, char ** dst_rows) {
    const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = idx.ne[0] * idx.ne[1] * idx.ne[2];
    if (t >= n) return;
    const int64_t i10 = t % idx.ne[0], i11 = (t / idx.ne[0]) % idx.ne[1], i12 = t / (idx.ne[0] * idx.ne[1]);
    const int64_t row = *(const int32_t *) (idx.p + i10 * idx.nb[0] + i11 * idx.nb[1] + i12 * idx.nb[2]);
    src_rows[t] = s.p + row * s.nb[1] + i11 * s.nb[2] + i12 * s.nb[3];
    dst_rows[t] = d.p + i10 * d.nb[1] + i11 * d.nb[2] + i12 * d.nb[3];
}
// quantised tables: one workgroup per gathered row, dequantised with the same arithmetic as dequant_rows
__global__ void __launch_bounds__(256) k_get_rows_q(int type, td4 s, td4 idx, td4 d) {
    const int64_t i10 = blockIdx.x, i11 = blockIdx.y, i12 = blockIdx.z;
    const int64_t row = *(const int32_t *) (idx.p + i10 * idx.nb[0] + i11 * idx.nb[1] + i12 * idx.nb[2]);
    const char * sr = s.p + row * s.nb[1] + i11 * s.nb[2] + i12 * s.nb[3];
    float *      y  = (float *) (d.p + i10 * d.nb[1] + i11 * d.nb[2] + i12 * d.nb[3]);
    const int64_t K = s.ne[0];
    for (int64_t e = threadIdx.x; e < K; e += blockDim.x) {
        float v;
        if (type == GGML_TYPE_Q8_0) {
            const block_q8_0 * b = (const block_q8_0 *) sr + e / 32;
            v = (float) b->qs[e % 32] * h2f(b->d);
        } else if (type == GGML_TYPE_Q4_K) {
            const block_q4_K * b = (const block_q4_K *) sr + e / 256;
            const int w = (int) (e % 256), sb = w / 32, l = w % 32;
            int sc, m; q4k_scale_min(sb, b->scales, sc, m);
            const uint8_t q = b->qs[32 * (sb >> 1) + l];
            const int qv = (sb & 1) ? (q >> 4) : (q & 0xF);
            v = (h2f(b->d) * (float) sc) * (float) qv - h2f(b->dmin) * (float) m;
        } else if (type != GGML_TYPE_Q6_K) {
            v = dq_elem_other(type, sr, e);
        } else { // Q6_K
            const block_q6_K * b = (const block_q6_K *) sr + e / 256;
            const int w = (int) (e % 256), n = w / 128, r = w % 128, k = r / 32, l = r % 32;
            const uint8_t qlb = b->ql[64 * n + l + ((k & 1) ? 32 : 0)];
            const int lo = (k >= 2) ? (qlb >> 4) : (qlb & 0xF);
            const int hi = (b->qh[32 * n + l] >> (2 * k)) & 3;
            const int8_t q = (int8_t) (lo | (hi << 4)) - 32;
            v = h2f(b->d) * (float) b->scales[8 * n + l / 16 + 2 * k] * (float) q;
        }
        y[e] = v;
    }
}
void get_rows(const tdesc & src, int src_type, const tdesc & idx, const tdesc & dst, hipStream_t st) {
    if (idx.ne[0] * idx.ne[1] * idx.ne[2] == 0 || src.ne[0] == 0) return;
    dim3 grid((unsigned) idx.ne[0], (unsigned) idx.ne[1], (unsigned) idx.ne[2]);
    const td4 s = to_td4(src), i = to_td4(idx), d = to_td4(dst);
    switch (src_type) {
        case GGML_TYPE_F32: k_get_rows<float><<<grid, dim3(256), 0, st>>>(s, i, d); break;
        case GGML_TYPE_I32: k_get_rows<float><<<grid, dim3(256), 0, st>>>(s, i, d); break;   // bit copy (4-byte elements)
        case GGML_TYPE_F16: k_get_rows<uint16_t><<<grid, dim3(256), 0, st>>>(s, i, d); break;
        case GGML_TYPE_BF16: k_get_rows<bf16_t><<<grid, dim3(256), 0, st>>>(s, i, d); break;
        case GGML_TYPE_Q8_0: case GGML_TYPE_Q4_K: case GGML_TYPE_Q6_K:
        case GGML_TYPE_Q4_0: case GGML_TYPE_Q4_1: case GGML_TYPE_Q5_0: case GGML_TYPE_Q5_1: case GGML_TYPE_Q2_K: case GGML_TYPE_Q3_K: case GGML_TYPE_Q5_K:
            k_get_rows_q<<<grid, dim3(256), 0, st>>>(src_type, s, i, d); break;
        default: fprintf(stderr, "[mi355x] get_rows: unsupported type %d\n", src_type); abort();
    }
}

// ================================================================================================
// SET_ROWS (ops.cpp:4739-4787): dst[:, idx[i, i02%ne11, i03%ne12], i02, i03] = from_float(src0[:, i, i02, i03])
// (KV-cache store: f32 -> f16 RNE, 64-bit indices)
// ================================================================================================
template <typename TD, typename TI>
__global__ void __launch_bounds__(256) k_set_rows(td4 s, td4 idx, td4 d) {
    const int64_t i = blockIdx.x, i02 = blockIdx.y, i03 = blockIdx.z;
    const int64_t i1 = (int64_t) *(const TI *) (idx.p + i * idx.nb[0] + (i02 % idx.ne[1]) * idx.nb[1] + (i03 % idx.ne[2]) * idx.nb[2]);
    const float * sr = (const float *) (s.p + i * s.nb[1] + i02 * s.nb[2] + i03 * s.nb[3]);
    TD *          dr = (TD *) (d.p + i1 * d.nb[1] + i02 * d.nb[2] + i03 * d.nb[3]);
    for (int64_t c = threadIdx.x; c < s.ne[0]; c += blockDim.x) dr[c] = cvt_elem<float, TD>(sr[c]);
}
// Single-element rows (the transposed V cache of the flash-attention-off graphs: llama-kv-cache.cpp:1091-1109 scatters every element of a
// [n_embd_v_gqa, n_tokens] block to index d * kv_size + cell): one workgroup per row would be half a million 64-thread workgroups per layer
// at ubatch 512 (115 us).  Rows r = t * P + j are read coalesced along j, and -- after a 64 x 64 exchange through LDS -- written along t,
// where the reference's indices are consecutive cells: coalesced on both sides whenever the indices have that shape, correct for any indices
// (P only orders the work).
template <typename TD, typename TI>
__global__ void __launch_bounds__(256) k_set_rows_1elem(const char * __restrict__ src, int64_t s_rs, const char * __restrict__ idx, int64_t i_rs, char * __restrict__ dst, int64_t d_rs, int P, int T) {
    __shared__ TD v[64][65];
    __shared__ long long ix[64][65];
    const int j0 = blockIdx.x * 64, t0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int tl = ty + 4 * k, t = t0 + tl, j = j0 + tx;
        if (t < T && j < P) {
            const int64_t r = (int64_t) t * P + j;
            v[tl][tx] = cvt_elem<float, TD>(*(const float *) (src + r * s_rs));
            ix[tl][tx] = (long long) *(const TI *) (idx + r * i_rs);
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int jl = ty + 4 * k, t = t0 + tx, j = j0 + jl;
        if (t < T && j < P) *(TD *) (dst + ix[tx][jl] * d_rs) = v[tx][jl];
    }
}

// SET_ROWS into a quantised / BF16 table (a KV cache kept as Q8_0 / Q4_0 / BF16: llama_kv_cache::cpy_k / cpy_v with -ctk / -ctv): the row goes
// through the type's from_float like ggml_compute_forward_set_rows_f32 does -- quantize_row_q8_0 as the x86 build compiles it (d = amax / 127 stored
// as f16, id = 127 / amax, round-half-even), quantize_row_q4_0_ref (ggml-quants.c: the FIRST value of largest magnitude decides d = max / -8,
// q = min(15, (int8) (x / d + 8.5))), ggml_compute_fp32_to_bf16.  One thread per 32-element block (per element for BF16).
template <int T, typename TI>
__global__ void __launch_bounds__(256) k_set_rows_q(td4 s, td4 idx, td4 d, int64_t total) {
    const int64_t t = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int64_t per_row = T == GGML_TYPE_BF16 ? s.ne[0] : s.ne[0] / 32;
    const int64_t ib = t % per_row; int64_t r = t / per_row;
    const int64_t i = r % s.ne[1]; r /= s.ne[1];
    const int64_t i02 = r % s.ne[2], i03 = r / s.ne[2];
    const int64_t i1 = (int64_t) *(const TI *) (idx.p + i * idx.nb[0] + (i02 % idx.ne[1]) * idx.nb[1] + (i03 % idx.ne[2]) * idx.nb[2]);
    const float * sr = (const float *) (s.p + i * s.nb[1] + i02 * s.nb[2] + i03 * s.nb[3]);
    char * dr = d.p + i1 * d.nb[1] + i02 * d.nb[2] + i03 * d.nb[3];
    if (T == GGML_TYPE_BF16) {
        uint32_t u = __float_as_uint(sr[ib]);
        u = (u & 0x7fffffffu) > 0x7f800000u ? (u >> 16) | 64u : (u + (0x7fffu + ((u >> 16) & 1u))) >> 16;
        ((uint16_t *) dr)[ib] = (uint16_t) u;
        return;
    }
    const float * x = sr + ib * 32;
    if (T == GGML_TYPE_Q8_0) {
        float amax = 0.0f;
        for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(x[j]));
        const float dd = amax / 127.0f, id = amax != 0.0f ? 127.0f / amax : 0.0f;
        char * b = dr + ib * 34;
        *(uint16_t *) b = f2h(dd);
        for (int j = 0; j < 32; ++j) b[2 + j] = (char) (int8_t) (int) rintf(x[j] * id);
    } else {                                                             // Q4_0
        float amax = 0.0f, mx = 0.0f;
        for (int j = 0; j < 32; ++j) { const float v = x[j]; if (amax < fabsf(v)) { amax = fabsf(v); mx = v; } }
        const float dd = mx / -8.0f, id = dd != 0.0f ? 1.0f / dd : 0.0f;
        char * b = dr + ib * 18;
        *(uint16_t *) b = amax == 0.0f ? (uint16_t) 0x8000u : f2h(dd);     // an all-zero block: d = +0 / -8 = -0 in the reference (kept explicit: the compiler's division drops the sign)
        for (int j = 0; j < 16; ++j) {
            const int q0 = (int) (int8_t) (x[j] * id + 8.5f), q1 = (int) (int8_t) (x[16 + j] * id + 8.5f);
            b[2 + j] = (char) ((q0 < 15 ? q0 : 15) | ((q1 < 15 ? q1 : 15) << 4));
        }
    }
}

void set_rows(const tdesc & src, const tdesc & idx, int idx_type, const tdesc & dst, int dst_type, hipStream_t st, int64_t period) {
    if (src.ne[0] * src.ne[1] * src.ne[2] * src.ne[3] == 0) return;
    if (dst_type == GGML_TYPE_Q8_0 || dst_type == GGML_TYPE_Q4_0 || dst_type == GGML_TYPE_BF16) {
        const int64_t per_row = dst_type == GGML_TYPE_BF16 ? src.ne[0] : src.ne[0] / 32;
        const int64_t total = per_row * src.ne[1] * src.ne[2] * src.ne[3];
        const dim3 grid((unsigned) ((total + 255) / 256));
        const td4 s = to_td4(src), i = to_td4(idx), d = to_td4(dst);
        const bool i64 = idx_type == GGML_TYPE_I64;
#define SRQ(T) do { if (i64) k_set_rows_q<T, int64_t><<<grid, dim3(256), 0, st>>>(s, i, d, total); else k_set_rows_q<T, int32_t><<<grid, dim3(256), 0, st>>>(s, i, d, total); } while (0)
        if (dst_type == GGML_TYPE_Q8_0) SRQ(GGML_TYPE_Q8_0); else if (dst_type == GGML_TYPE_Q4_0) SRQ(GGML_TYPE_Q4_0); else SRQ(GGML_TYPE_BF16);
#undef SRQ
        return;
    }
    if (src.ne[0] == 1 && src.ne[2] == 1 && src.ne[3] == 1 && src.ne[1] >= 4096 && (dst_type == GGML_TYPE_F16 || dst_type == GGML_TYPE_F32) && idx.ne[1] * idx.ne[2] * idx.ne[3] == 1) {
        const int64_t R = src.ne[1];
        int64_t P = period > 0 && R % period == 0 ? period : (R % 1024 == 0 ? 1024 : (R % 64 == 0 ? 64 : 1));
        if (P > 1 && R / P <= 0x7fffffff && P <= 0x7fffffff) {
            const int T = (int) (R / P);
            const dim3 grid((unsigned) ((P + 63) / 64), (unsigned) ((T + 63) / 64));
            const bool i64 = idx_type == GGML_TYPE_I64;
            const char * sp = (const char *) src.p; const char * ip = (const char *) idx.p; char * dp = (char *) dst.p;
            if (dst_type == GGML_TYPE_F16) {
                if (i64) k_set_rows_1elem<uint16_t, int64_t><<<grid, dim3(256), 0, st>>>(sp, (int64_t) src.nb[1], ip, (int64_t) idx.nb[0], dp, (int64_t) dst.nb[1], (int) P, T);
                else     k_set_rows_1elem<uint16_t, int32_t><<<grid, dim3(256), 0, st>>>(sp, (int64_t) src.nb[1], ip, (int64_t) idx.nb[0], dp, (int64_t) dst.nb[1], (int) P, T);
            } else {
                if (i64) k_set_rows_1elem<float, int64_t><<<grid, dim3(256), 0, st>>>(sp, (int64_t) src.nb[1], ip, (int64_t) idx.nb[0], dp, (int64_t) dst.nb[1], (int) P, T);
                else     k_set_rows_1elem<float, int32_t><<<grid, dim3(256), 0, st>>>(sp, (int64_t) src.nb[1], ip, (int64_t) idx.nb[0], dp, (int64_t) dst.nb[1], (int) P, T);
            }
            return;
        }
    }
    dim3 grid((unsigned) src.ne[1], (unsigned) src.ne[2], (unsigned) src.ne[3]);
    const int bs = src.ne[0] <= 64 ? 64 : 256;
    const td4 s = to_td4(src), i = to_td4(idx), d = to_td4(dst);
    const bool i64 = idx_type == GGML_TYPE_I64;
    if (dst_type == GGML_TYPE_F16) {
        if (i64) k_set_rows<uint16_t, int64_t><<<grid, dim3(bs), 0, st>>>(s, i, d); else k_set_rows<uint16_t, int32_t><<<grid, dim3(bs), 0, st>>>(s, i, d);
    } else if (dst_type == GGML_TYPE_F32) {
        if (i64) k_set_rows<float, int64_t><<<grid, dim3(bs), 0, st>>>(s, i, d); else k_set_rows<float, int32_t><<<grid, dim3(bs), 0, st>>>(s, i, d);
    } else { fprintf(stderr, "[mi355x] set_rows: unsupported dst type %d\n", dst_type); abort(); }
}

// ------------------------------------------------------------------------------------------------ batched small uploads
// one workgroup per entry; 16-byte pieces where source and destination allow, bytes at the edges
__global__ void __launch_bounds__(256) k_upload_small(const upload_ent * __restrict__ ents, const char * __restrict__ base) {
    const upload_ent e = ents[blockIdx.x];
    const char * src = base + e.off; char * dst = (char *) e.dst;
    const uint32_t n = e.size;
    if ((((uintptr_t) src | (uintptr_t) dst) & 15) == 0) {
        const uint32_t nv = n >> 4;
        for (uint32_t i = threadIdx.x; i < nv; i += 256) ((u32x4 *) dst)[i] = ((const u32x4 *) src)[i];
        for (uint32_t i = (nv << 4) + threadIdx.x; i < n; i += 256) dst[i] = src[i];
    } else {
        for (uint32_t i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
    }
}
void upload_small(const upload_ent * ents, const char * base, int n, hipStream_t st) {
    if (n > 0) k_upload_small<<<dim3(n), dim3(256), 0, st>>>(ents, base);
}

} // namespace mi

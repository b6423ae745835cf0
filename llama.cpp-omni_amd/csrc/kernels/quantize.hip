// quantize.hip -- on-device activation quantisers for the quantised mat-vec / mat-mul path.
//
// The reference CPU backend converts src1 (f32) to the weight type's `vec_dot_type` before every
// mul_mat (ggml-cpu.c:1272-1306): Q8_K for K-quants, Q8_0 for Q8_0, F16 for F16 weights.  To make the
// integer stages bit-identical with that oracle the device does the SAME quantisation -- not the
// per-32 Q8_1 scheme of the CUDA backend.
#include "../kernels.hpp"

namespace mi {

// Q8_K image: one wave per 256-element block
__global__ void __launch_bounds__(256) k_quantize_q8k(const char * __restrict__ x, size_t xs, char * __restrict__ img,
                                                     int64_t K, int64_t nrows, size_t img_bytes) {
    const int     lane = threadIdx.x & 63;
    const int64_t nb   = K / 256;
    const int64_t blk  = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);     // global block id
    if (blk >= nb * nrows) return;
    const int64_t row = blk / nb, ib = blk % nb;
    const float * xr = (const float *) (x + row * xs) + ib * 256;
    char *        im = img + row * img_bytes;
    const f32x4 v = *(const f32x4 *) (xr + 4 * lane);
    q8k_block_from_regs(v, lane, (int8_t *) im + ib * 256, (int16_t *) (im + K) + ib * 16, (float *) (im + K + K / 8) + ib);
}

void quantize_q8k_image(const float * x, size_t xs, void * img, int64_t K, int64_t nrows, hipStream_t st) {
    const int64_t nblk = K / 256 * nrows;
    if (nblk == 0) return;
    k_quantize_q8k<<<dim3((unsigned) ((nblk + 3) / 4)), dim3(256), 0, st>>>((const char *) x, xs, (char *) img, K, nrows, q8k_image_bytes(K));
}

// ------------------------------------------------------------------------------------------------
// RMS_NORM + MUL(w) + Q8_K image in one launch: one workgroup per row.
//   y = (x * (1/sqrtf(mean(x^2)+eps))) * w      (ops.cpp:3517-3566 then the graph's MUL node; sum of squares in double)
//   img = Q8_K(y)                                (what the following MUL_MATs would compute from y)
// ------------------------------------------------------------------------------------------------
// The row and the weight vector are fetched ONCE, up front, into registers (wave w owns blocks w, w+nw, ...; lane l the
// elements 4l..4l+3 of each): one memory round trip, then arithmetic only.  MAXB blocks per wave bound the row length.
template <int MAXB>
__global__ void __launch_bounds__(512) k_rms_norm_mul_quant(const char * __restrict__ x, size_t xs, const float * __restrict__ w, char * __restrict__ y, size_t ys,
                                                           char * __restrict__ img, int n, float eps, size_t img_bytes) {
    __shared__ double red[16];
    const int64_t row = blockIdx.x;
    const float * xr = (const float *) (x + row * xs);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, nb = n >> 8;
    f32x4 xv[MAXB], wv[MAXB];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int ib = wave + b * nw;
        if (ib < nb) { xv[b] = *(const f32x4 *) (xr + ib * 256 + 4 * lane); wv[b] = *(const f32x4 *) (w + ib * 256 + 4 * lane); }
        else { xv[b] = f32x4{0, 0, 0, 0}; wv[b] = f32x4{0, 0, 0, 0}; }
    }
    double s = 0.0;
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) s += (double) (xv[b][i] * xv[b][i]);
    s = block_sum<double>(s, red);
    const float mean  = (float) (s / (double) n);
    const float scale = 1.0f / sqrtf(mean + eps);
    char * im = img + row * img_bytes;
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int ib = wave + b * nw;
        if (ib >= nb) break;
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (xv[b][i] * scale) * wv[b][i];
        if (y) *(f32x4 *) ((float *) (y + row * ys) + ib * 256 + 4 * lane) = v;
        q8k_block_from_regs(v, lane, (int8_t *) im + ib * 256, (int16_t *) (im + n) + ib * 16, (float *) (im + n + n / 8) + ib);
    }
}

bool rms_norm_mul_quant_ok(int64_t n) { return n % 256 == 0 && n / 256 <= 8 * 8; }

void rms_norm_mul_quant(const float * x, size_t xs, const float * w, float * y, size_t ys, void * img, int64_t n, int64_t nrows, float eps, hipStream_t st) {
    if (n == 0 || nrows == 0) return;
    const int nb = (int) (n / 256);
    const int bs = nb >= 8 ? 512 : (nb >= 4 ? 256 : (nb >= 2 ? 128 : 64));
    const int per = (nb + bs / 64 - 1) / (bs / 64);
    const size_t ib = q8k_image_bytes(n);
    if (per <= 1)      k_rms_norm_mul_quant<1><<<dim3((unsigned) nrows), dim3(bs), 0, st>>>((const char *) x, xs, w, (char *) y, ys, (char *) img, (int) n, eps, ib);
    else if (per <= 2) k_rms_norm_mul_quant<2><<<dim3((unsigned) nrows), dim3(bs), 0, st>>>((const char *) x, xs, w, (char *) y, ys, (char *) img, (int) n, eps, ib);
    else if (per <= 4) k_rms_norm_mul_quant<4><<<dim3((unsigned) nrows), dim3(bs), 0, st>>>((const char *) x, xs, w, (char *) y, ys, (char *) img, (int) n, eps, ib);
    else               k_rms_norm_mul_quant<8><<<dim3((unsigned) nrows), dim3(bs), 0, st>>>((const char *) x, xs, w, (char *) y, ys, (char *) img, (int) n, eps, ib);
}

// ------------------------------------------------------------------------------------------------
// Q8_0 image.  32 lanes per 32-element block (two blocks per wave).
//   reference (what the x86 CPU backend actually runs): quantize_row_q8_0, ggml-cpu/arch/x86/quants.c:290-345
//     d  = amax / 127 -> stored as f16        id = amax != 0 ? 127 / amax : 0
//     q  = round-half-even(x * id)
//   (the portable quantize_row_q8_0_ref, ggml-quants.c:199-222, uses id = 1/d and roundf; the oracle that
//   parity is defined against is the compiled x86 backend, so its arithmetic is the one restated here.)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_quantize_q80(const char * __restrict__ x, size_t xs, char * __restrict__ img,
                                                     int64_t K, int64_t nrows, size_t img_bytes) {
    const int64_t nb  = K / 32;
    const int64_t blk = (int64_t) blockIdx.x * 8 + (threadIdx.x >> 5);
    if (blk >= nb * nrows) return;
    const int     l   = threadIdx.x & 31;
    const int64_t row = blk / nb, ib = blk % nb;
    const float   v   = ((const float *) (x + row * xs))[ib * 32 + l];
    float amax = fabsf(v);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float d  = amax / 127.0f;
    const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
    const int   q  = (int) __builtin_rintf(v * id);
    char * im = img + row * img_bytes;
    ((int8_t *) im)[ib * 32 + l] = (int8_t) q;
    if (l == 0) ((float *) (im + K))[ib] = h2f(f2h(d));
}

void quantize_q80_image(const float * x, size_t xs, void * img, int64_t K, int64_t nrows, hipStream_t st) {
    const int64_t nblk = K / 32 * nrows;
    if (nblk == 0) return;
    k_quantize_q80<<<dim3((unsigned) ((nblk + 7) / 8)), dim3(256), 0, st>>>((const char *) x, xs, (char *) img, K, nrows, q80_image_bytes(K));
}

// ------------------------------------------------------------------------------------------------
// f32 -> f16 rows (RNE), the F16 weights' vec_dot_type conversion (ggml_cpu_fp32_to_fp16)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_f32_to_f16_rows(const char * __restrict__ x, size_t xs, char * __restrict__ y, size_t ys, int64_t K, int64_t nrows) {
    const int64_t row = blockIdx.y;
    const float * xr = (const float *) (x + row * xs);
    uint16_t *    yr = (uint16_t *) (y + row * ys);
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < K; i += (int64_t) gridDim.x * blockDim.x) yr[i] = f2h(xr[i]);
}

// same, rows enumerated over three strided dimensions (a permuted activation, e.g. q [D, n_tokens, n_head] seen per head):
// row r = (i1, i2, i3) with i1 fastest, source offset i1*nb1 + i2*nb2 + i3*nb3, destination rows dense in r
__global__ void __launch_bounds__(256) k_f32_to_f16_rows3(const char * __restrict__ x, size_t nb1, size_t nb2, size_t nb3, int n1, int n2,
                                                         char * __restrict__ y, size_t ys, int64_t K) {
    const int64_t row = blockIdx.y;
    const int i1 = (int) (row % n1), i2 = (int) ((row / n1) % n2), i3 = (int) (row / ((int64_t) n1 * n2));
    const float * xr = (const float *) (x + i1 * nb1 + i2 * nb2 + i3 * nb3);
    uint16_t *    yr = (uint16_t *) (y + row * ys);
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < K; i += (int64_t) gridDim.x * blockDim.x) yr[i] = f2h(xr[i]);
}
// short rows (attention heads: K = 64 / 128 per (token, head) row -- a workgroup per row would be 24 000 workgroups of one or two live waves): four elements
// per thread, K / 4 threads per row, rows packed into the workgroups
__global__ void __launch_bounds__(256) k_f32_to_f16_rows3_v4(const char * __restrict__ x, size_t nb1, size_t nb2, size_t nb3, int n1, int n2,
                                                            char * __restrict__ y, size_t ys, int k4, int64_t nrows) {
    const int64_t gt = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int64_t row = gt / k4;
    if (row >= nrows) return;
    const int i = (int) (gt - row * k4) * 4;
    const int i1 = (int) (row % n1), i2 = (int) ((row / n1) % n2), i3 = (int) (row / ((int64_t) n1 * n2));
    const f32x4 v = *(const f32x4 *) (x + i1 * nb1 + i2 * nb2 + i3 * nb3 + (size_t) i * 4);
    u32x2 h; h[0] = (uint32_t) f2h(v[0]) | ((uint32_t) f2h(v[1]) << 16); h[1] = (uint32_t) f2h(v[2]) | ((uint32_t) f2h(v[3]) << 16);
    *(u32x2 *) (y + row * ys + (size_t) i * 2) = h;
}
void convert_f32_f16_rows3(const float * x, size_t nb1, size_t nb2, size_t nb3, int64_t n1, int64_t n2, int64_t n3, uint16_t * y, size_t ys, int64_t K, hipStream_t st) {
    const int64_t nrows = n1 * n2 * n3;
    if (K == 0 || nrows == 0) return;
    if (K % 4 == 0 && K <= 1024 && (((uintptr_t) x | nb1 | nb2 | nb3) & 15) == 0 && (((uintptr_t) y | ys) & 7) == 0 && nrows * (K / 4) < ((int64_t) 1 << 38)) {
        const int64_t threads = nrows * (K / 4);
        k_f32_to_f16_rows3_v4<<<dim3((unsigned) ((threads + 255) / 256)), dim3(256), 0, st>>>((const char *) x, nb1, nb2, nb3, (int) n1, (int) n2, (char *) y, ys, (int) (K / 4), nrows);
        return;
    }
    unsigned gx = (unsigned) ((K + 255) / 256); if (gx > 64) gx = 64;
    k_f32_to_f16_rows3<<<dim3(gx, (unsigned) nrows), dim3(256), 0, st>>>((const char *) x, nb1, nb2, nb3, (int) n1, (int) n2, (char *) y, ys, K);
}

// f32 rows -> f16 rows of the Q8_K-quantised values (q8k_requant4): one wave per (row, 256-block); IN16: the source rows are f16 already (an attention / SwiGLU
// launch wrote them into the image: re-quantised in place)
template <bool IN16>
__global__ void __launch_bounds__(256) k_rows_to_f16q(const char * __restrict__ x, size_t xs, char * __restrict__ y, size_t ys, int nblk, int64_t nrows) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= nrows * nblk) return;
    const int64_t row = wid / nblk; const int b = (int) (wid % nblk);
    f32x4 v;
    if (IN16) { const u32x2 h = *(const u32x2 *) (x + row * xs + (size_t) b * 512 + lane * 8);
                v[0] = h2f((uint16_t) (h[0] & 0xffff)); v[1] = h2f((uint16_t) (h[0] >> 16)); v[2] = h2f((uint16_t) (h[1] & 0xffff)); v[3] = h2f((uint16_t) (h[1] >> 16)); }
    else v = *(const f32x4 *) (x + row * xs + (size_t) b * 1024 + lane * 16);
    const f32x4 o = q8k_requant4(v, lane);
    u32x2 h; h[0] = (uint32_t) f2h(o[0]) | ((uint32_t) f2h(o[1]) << 16); h[1] = (uint32_t) f2h(o[2]) | ((uint32_t) f2h(o[3]) << 16);
    *(u32x2 *) (y + row * ys + (size_t) b * 512 + lane * 8) = h;
}
void convert_f32_f16q_rows(const float * x, size_t xs, uint16_t * y, size_t ys, int64_t K, int64_t nrows, hipStream_t st) {
    if (K == 0 || nrows == 0) return;
    if (K % 256 != 0 || xs % 16 != 0 || ys % 8 != 0 || ((uintptr_t) x & 15) != 0 || ((uintptr_t) y & 7) != 0) { fprintf(stderr, "[mi355x] convert_f32_f16q_rows: K %% 256 == 0 and aligned rows\n"); abort(); }
    const int64_t waves = nrows * (K / 256);
    k_rows_to_f16q<false><<<dim3((unsigned) ((waves + 3) / 4)), dim3(256), 0, st>>>((const char *) x, xs, (char *) y, ys, (int) (K / 256), nrows);
}
void requant_f16_rows_q8k(uint16_t * y, size_t ys, int64_t K, int64_t nrows, hipStream_t st) {
    if (K == 0 || nrows == 0) return;
    if (K % 256 != 0 || ys % 8 != 0 || ((uintptr_t) y & 7) != 0) { fprintf(stderr, "[mi355x] requant_f16_rows_q8k: K %% 256 == 0 and aligned rows\n"); abort(); }
    const int64_t waves = nrows * (K / 256);
    k_rows_to_f16q<true><<<dim3((unsigned) ((waves + 3) / 4)), dim3(256), 0, st>>>((const char *) y, ys, (char *) y, ys, (int) (K / 256), nrows);
}

void convert_f32_f16_rows(const float * x, size_t xs, uint16_t * y, size_t ys, int64_t K, int64_t nrows, hipStream_t st) {
    if (K == 0 || nrows == 0) return;
    unsigned gx = (unsigned) ((K + 255) / 256); if (gx > 64) gx = 64;
    k_f32_to_f16_rows<<<dim3(gx, (unsigned) nrows), dim3(256), 0, st>>>((const char *) x, xs, (char *) y, ys, K, nrows);
}

} // namespace mi

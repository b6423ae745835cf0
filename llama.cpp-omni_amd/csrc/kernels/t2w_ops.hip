// t2w_ops.hip -- the ops the Token2Wav graphs (flow-matching DiT + HiFT vocoder, reference tools/omni/token2wav/token2wav-impl.cpp) need
// beyond the text decoder's set.  Token2Wav drives ONE backend with ggml_backend_graph_compute directly
// (token2wav-impl.cpp:6280-6345: no scheduler, no CPU fallback per op), so a graph runs on this backend only if every op in it does.
// All of these are data movement or element-wise f32 arithmetic: HBM-bound, one pass, coalesced along ne[0]; each restates one
// ggml_compute_forward_* of the reference CPU backend (file:line at the kernel) and is validated by the reference's own
// test-backend-ops against this plug-in (tools/run_tbo.sh) and by tests/test_t2w_gpu.py against the reference CPU backend.
#include "../kernels.hpp"

namespace mi {

struct t4 { char * p; int64_t ne[4]; int64_t nb[4]; };
static t4 to_t4(const tdesc & t) {
    t4 d; d.p = (char *) t.p;
    for (int i = 0; i < 4; ++i) { d.ne[i] = t.ne[i]; d.nb[i] = (int64_t) t.nb[i]; }
    return d;
}
static inline dim3 grid_for(int64_t total) { int64_t g = (total + 255) / 256; if (g > 16384) g = 16384; if (g < 1) g = 1; return dim3((unsigned) g); }
#define T2W_LOOP(total) for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < (total); t += (int64_t) gridDim.x * blockDim.x)
static __device__ __forceinline__ void unravel(int64_t t, const int64_t * ne, int64_t & i0, int64_t & i1, int64_t & i2, int64_t & i3) {
    i0 = t % ne[0]; int64_t r = t / ne[0];
    i1 = r % ne[1]; r /= ne[1];
    i2 = r % ne[2]; i3 = r / ne[2];
}

// ---------------------------------------------------------------------------------------------- element-wise math on dense f32
// SQR / SQRT / LOG / SIN / COS (ops.cpp unary family via vec.h: x*x, sqrtf, logf, sinf, cosf), CLAMP (ops.cpp:5305-5345),
// LEAKY_RELU (vec.h:834: max(x, 0) + slope * min(x, 0))
__global__ void __launch_bounds__(256) k_math(int op, const float * __restrict__ x, float * __restrict__ y, int64_t n, float p0, float p1) {
    T2W_LOOP(n) {
        const float v = x[t]; float r;
        switch (op) {
            case GGML_OP_SQR:        r = v * v; break;
            case GGML_OP_SQRT:       r = sqrtf(v); break;
            case GGML_OP_LOG:        r = logf(v); break;
            case GGML_OP_SIN:        r = sinf(v); break;
            case GGML_OP_COS:        r = cosf(v); break;
            case GGML_OP_CLAMP:      r = fmaxf(fminf(v, p1), p0); break;                       // MAX(MIN(x, max), min)
            case GGML_OP_LEAKY_RELU: r = (v > 0.0f ? v : 0.0f) + p0 * (v < 0.0f ? v : 0.0f); break;
            default: r = v;
        }
        y[t] = r;
    }
}
void math_f32(int op, const float * x, float * y, int64_t n, float p0, float p1, hipStream_t st) {
    if (n == 0) return;
    k_math<<<grid_for(n), dim3(256), 0, st>>>(op, x, y, n, p0, p1);
}

// ---------------------------------------------------------------------------------------------- CONCAT (ops.cpp:1968-2009), 4- or 2-byte elements
template <typename T>
__global__ void __launch_bounds__(256) k_concat(t4 a, t4 b, t4 y, int dim) {
    const int64_t total = y.ne[0] * y.ne[1] * y.ne[2] * y.ne[3];
    T2W_LOOP(total) {
        int64_t i[4]; unravel(t, y.ne, i[0], i[1], i[2], i[3]);
        const char * s;
        if (i[dim] < a.ne[dim]) s = a.p + i[0] * a.nb[0] + i[1] * a.nb[1] + i[2] * a.nb[2] + i[3] * a.nb[3];
        else { int64_t j[4] = { i[0], i[1], i[2], i[3] }; j[dim] -= a.ne[dim]; s = b.p + j[0] * b.nb[0] + j[1] * b.nb[1] + j[2] * b.nb[2] + j[3] * b.nb[3]; }
        *(T *) (y.p + i[0] * y.nb[0] + i[1] * y.nb[1] + i[2] * y.nb[2] + i[3] * y.nb[3]) = *(const T *) s;
    }
}
// dense operands: the result is `outer` repetitions of [A elements of a][B elements of b] (A, B = the operands' extents below and including `dim`), 16 bytes per thread
__global__ void __launch_bounds__(256) k_concat_dense(const uint4 * __restrict__ a, const uint4 * __restrict__ b, uint4 * __restrict__ y, uint32_t A4, uint32_t B4, uint32_t total4) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total4) return;
    const uint32_t o = i / (A4 + B4), r = i - o * (A4 + B4);
    y[i] = r < A4 ? a[o * A4 + r] : b[o * B4 + (r - A4)];
}
void concat(const tdesc & a, const tdesc & b, const tdesc & y, int dim, int elem_size, hipStream_t st) {
    const int64_t total = y.ne[0] * y.ne[1] * y.ne[2] * y.ne[3];
    if (total == 0) return;
    {
        auto dense = [&](const tdesc & t) { size_t s = (size_t) elem_size; for (int d = 0; d < 4; ++d) { if (t.ne[d] > 1 && t.nb[d] != s) return false; s *= (size_t) t.ne[d]; } return ((uintptr_t) t.p & 15) == 0; };
        int64_t A = elem_size, B = elem_size;
        for (int d = 0; d <= dim; ++d) { A *= a.ne[d]; B *= b.ne[d]; }
        if (dim >= 0 && dim < 4 && dense(a) && dense(b) && dense(y) && A % 16 == 0 && B % 16 == 0 && A > 0 && B > 0 && total * elem_size / 16 < (1ll << 32)) {
            const uint32_t total4 = (uint32_t) (total * elem_size / 16);
            k_concat_dense<<<dim3((total4 + 255) / 256), dim3(256), 0, st>>>((const uint4 *) a.p, (const uint4 *) b.p, (uint4 *) y.p, (uint32_t) (A / 16), (uint32_t) (B / 16), total4);
            return;
        }
    }
    if (elem_size == 4) k_concat<uint32_t><<<grid_for(total), dim3(256), 0, st>>>(to_t4(a), to_t4(b), to_t4(y), dim);
    else                k_concat<uint16_t><<<grid_for(total), dim3(256), 0, st>>>(to_t4(a), to_t4(b), to_t4(y), dim);
}

// ---------------------------------------------------------------------------------------------- 1-D convolution kernel [KW, C, Cout] -> rows [Cout][KW][C]
// (the image the fused streaming causal convolution multiplies against KW consecutive C-fastest frames: graph_exec.cpp exec_causal_conv)
__global__ void __launch_bounds__(256) k_conv1d_weight_rows(const float * __restrict__ w, float * __restrict__ y, int KW, int C, int64_t total) {
    T2W_LOOP(total) {
        const int64_t row = t / ((int64_t) KW * C); const int r = (int) (t - row * (int64_t) KW * C);
        const int k = r / C, c = r - k * C;
        y[t] = w[row * (int64_t) KW * C + (int64_t) c * KW + k];
    }
}
void conv1d_weight_rows(const float * w, float * y, int KW, int C, int Cout, hipStream_t st) {
    const int64_t total = (int64_t) KW * C * Cout;
    if (total == 0) return;
    k_conv1d_weight_rows<<<grid_for(total), dim3(256), 0, st>>>(w, y, KW, C, total);
}

// ---------------------------------------------------------------------------------------------- 1-D convolution over a T-fastest signal without the im2col matrix
// The HiFT vocoder's convolutions (token2wav-impl.cpp:5136-5235: x [T, Cin], kernel [KW, Cin, Cout], stride 1, dilation d, zero padding p) are spelled IM2COL (a
// [KW*Cin, T] f32 matrix: 21 MB for 64 channels x 11 taps x 7681 samples) -> CONT -> MUL_MAT -> REPEAT(bias) -> ADD: 40..60 us of copies around 0.7 GFLOP.  Here the
// product is formed straight from x on v_mfma_f32_16x16x4f32 with i = output channel, j = sample: the B operand of contraction index kk = c * KW + k is x[t + k d - p + T c]
// -- sixteen consecutive samples per lane group, coalesced -- and the A operand comes from the kernel transposed once to [KW*Cin][Cout] (resident image, coalesced over
// the output channels).  One workgroup = 16 channels x 16 samples, its four waves a quarter of the contraction each.  f32 products and sums like the separate nodes, summed in
// kk order; the bias added last, one rounding, as the ADD node does.
struct conv1d_tc_dev { const float * x; const float * wt; const float * bias; float * y; int T, OW, Cin, Cout, KW, dil, pad; };
__global__ void __launch_bounds__(256) k_conv1d_tc(const conv1d_tc_dev a) {
    typedef float acc4 __attribute__((ext_vector_type(4)));
    __shared__ float red[3][64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const int t0 = (int) blockIdx.x * 16, co0 = (int) blockIdx.y * 16;
    const int KK = a.KW * a.Cin;
    const int t = t0 + r16, co = co0 + r16;
    const bool co_ok = co < a.Cout, t_ok = t < a.OW;
    // the four waves split the contraction (chunks that are multiples of 4), folded through LDS in wave order: one 16 x 16 tile per workgroup keeps small signals
    // (512 samples x 256 channels) on every CU and the chain per wave short
    const int chunk = (((KK + 3) / 4) + 3) & ~3;
    const int k_lo = wave * chunk, k_hi = k_lo + chunk < KK ? k_lo + chunk : KK;
    const float inv_kw = 1.0f / (float) a.KW;              // kk -> (c, k) without a division or a loop: exact for kk < 2^20 (the + 0.5 keeps the product off the integers' edges)
    acc4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
    constexpr int U = 16;                                  // contraction steps requested before the first MFMA of a round (the loads are the latency here)
    for (int kk0 = k_lo; kk0 < k_hi; kk0 += 4 * U) {
        float xv[U], wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = kk0 + 4 * u + g;
            const int c = (int) (((float) kk + 0.5f) * inv_kw), k = kk - c * a.KW;
            const int ti = t + k * a.dil - a.pad;
            const bool ok = kk < k_hi;
            const bool xin = ok && t_ok && ti >= 0 && ti < a.T;
            xv[u] = a.x[xin ? ti + a.T * c : 0];
            wv[u] = a.wt[(ok && co_ok) ? (size_t) kk * a.Cout + co : 0];
            if (!xin) xv[u] = 0.0f;
            if (!(ok && co_ok)) wv[u] = 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u], xv[u], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave - 1][e * 64 + lane] = acc[e];
    }
    __syncthreads();
    if (wave > 0) return;
    // acc[e] = y[sample t0 + r16][channel co0 + 4 g + e]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int oc = co0 + 4 * g + e;
        float v = acc[e];
#pragma unroll
        for (int w = 0; w < 3; ++w) v += red[w][e * 64 + lane];
        if (t_ok && oc < a.Cout) { if (a.bias) v = __fadd_rn(v, a.bias[oc]); a.y[t + (size_t) a.OW * oc] = v; }
    }
}
__global__ void __launch_bounds__(256) k_conv1d_weight_t(const float * __restrict__ w, float * __restrict__ y, int KK, int Cout, int64_t total) {      // [Cout][KK] -> [KK][Cout]
    T2W_LOOP(total) { const int64_t kk = t / Cout; const int co = (int) (t - kk * Cout); y[t] = w[kk + (int64_t) KK * co]; }
}
void conv1d_weight_t(const float * w, float * y, int KK, int Cout, hipStream_t st) {
    const int64_t total = (int64_t) KK * Cout;
    if (total == 0) return;
    k_conv1d_weight_t<<<grid_for(total), dim3(256), 0, st>>>(w, y, KK, Cout, total);
}
void conv1d_tc(const float * x, const float * wt, const float * bias, float * y, int T, int OW, int Cin, int Cout, int KW, int dil, int pad, hipStream_t st) {
    if (OW <= 0 || Cout <= 0) return;
    const conv1d_tc_dev a = { x, wt, bias, y, T, OW, Cin, Cout, KW, dil, pad };
    k_conv1d_tc<<<dim3((unsigned) ((OW + 15) / 16), (unsigned) ((Cout + 15) / 16)), dim3(256), 0, st>>>(a);
}

// ---------------------------------------------------------------------------------------------- REPEAT (ops.cpp:1637-1679): dst[i] = src[i mod ne_src]
template <typename T>
__global__ void __launch_bounds__(256) k_repeat(t4 x, t4 y) {
    const int64_t total = y.ne[0] * y.ne[1] * y.ne[2] * y.ne[3];
    T2W_LOOP(total) {
        int64_t i0, i1, i2, i3; unravel(t, y.ne, i0, i1, i2, i3);
        *(T *) (y.p + i0 * y.nb[0] + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]) =
            *(const T *) (x.p + (i0 % x.ne[0]) * x.nb[0] + (i1 % x.ne[1]) * x.nb[1] + (i2 % x.ne[2]) * x.nb[2] + (i3 % x.ne[3]) * x.nb[3]);
    }
}
void repeat(const tdesc & x, const tdesc & y, int elem_size, hipStream_t st) {
    const int64_t total = y.ne[0] * y.ne[1] * y.ne[2] * y.ne[3];
    if (total == 0) return;
    if (elem_size == 4) k_repeat<uint32_t><<<grid_for(total), dim3(256), 0, st>>>(to_t4(x), to_t4(y));
    else                k_repeat<uint16_t><<<grid_for(total), dim3(256), 0, st>>>(to_t4(x), to_t4(y));
}

// ---------------------------------------------------------------------------------------------- PAD (ops.cpp:7592-7638; dense dst), PAD_REFLECT_1D (ops.cpp:7660-7691)
struct pad_dev { int lp[4], rp[4]; };
__global__ void __launch_bounds__(256) k_pad(t4 x, t4 y, pad_dev pd) {
    const int64_t total = y.ne[0] * y.ne[1] * y.ne[2] * y.ne[3];
    T2W_LOOP(total) {
        int64_t i0, i1, i2, i3; unravel(t, y.ne, i0, i1, i2, i3);
        float v = 0.0f;
        if (i0 >= pd.lp[0] && i0 < y.ne[0] - pd.rp[0] && i1 >= pd.lp[1] && i1 < y.ne[1] - pd.rp[1] && i2 >= pd.lp[2] && i2 < y.ne[2] - pd.rp[2] && i3 >= pd.lp[3] && i3 < y.ne[3] - pd.rp[3])
            v = *(const float *) (x.p + (i0 - pd.lp[0]) * x.nb[0] + (i1 - pd.lp[1]) * x.nb[1] + (i2 - pd.lp[2]) * x.nb[2] + (i3 - pd.lp[3]) * x.nb[3]);
        ((float *) y.p)[t] = v;                                          // the reference writes dst as a dense array (dst_idx)
    }
}
void pad_f32(const tdesc & x, const tdesc & y, const int32_t * p, hipStream_t st) {
    const int64_t total = y.ne[0] * y.ne[1] * y.ne[2] * y.ne[3];
    if (total == 0) return;
    pad_dev pd; for (int d = 0; d < 4; ++d) { pd.lp[d] = p[2 * d]; pd.rp[d] = p[2 * d + 1]; }
    k_pad<<<grid_for(total), dim3(256), 0, st>>>(to_t4(x), to_t4(y), pd);
}
__global__ void __launch_bounds__(256) k_pad_reflect_1d(t4 x, t4 y, int p0, int p1) {
    const int64_t total = y.ne[0] * y.ne[1] * y.ne[2] * y.ne[3];
    T2W_LOOP(total) {
        int64_t i0, i1, i2, i3; unravel(t, y.ne, i0, i1, i2, i3);
        int64_t j = i0 - p0;                                             // left[-k] = left[k]; right[k] = right[-k] with right = element ne00 - 1
        if (j < 0) j = -j;
        if (j >= x.ne[0]) j = 2 * (x.ne[0] - 1) - j;
        *(float *) (y.p + i0 * y.nb[0] + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]) = *(const float *) (x.p + j * x.nb[0] + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    }
    (void) p1;
}
void pad_reflect_1d_f32(const tdesc & x, const tdesc & y, int p0, int p1, hipStream_t st) {
    const int64_t total = y.ne[0] * y.ne[1] * y.ne[2] * y.ne[3];
    if (total == 0) return;
    k_pad_reflect_1d<<<grid_for(total), dim3(256), 0, st>>>(to_t4(x), to_t4(y), p0, p1);
}

// ---------------------------------------------------------------------------------------------- ARANGE (ops.cpp:7762-7783), TIMESTEP_EMBEDDING (ops.cpp:7800-7831)
__global__ void __launch_bounds__(256) k_arange(float * __restrict__ y, int64_t n, float start, float step) {
    T2W_LOOP(n) y[t] = start + step * (float) t;
}
void arange_f32(float * y, int64_t n, float start, float step, hipStream_t st) {
    if (n == 0) return;
    k_arange<<<grid_for(n), dim3(256), 0, st>>>(y, n, start, step);
}
__global__ void __launch_bounds__(256) k_timestep_embedding(const float * __restrict__ ts, char * __restrict__ y, int64_t y_nb1, int64_t n, int dim, int max_period) {
    const int half = dim / 2;
    const int64_t per = half + ((dim & 1) ? 1 : 0);
    T2W_LOOP(n * per) {
        const int64_t i = t / per; const int j = (int) (t - i * per);
        float * e = (float *) (y + i * y_nb1);
        if (j == half) { e[2 * half] = 0.0f; continue; }                 // odd dim: the last element is zero
        const float freq = expf(-logf((float) max_period) * (float) j / (float) half);
        const float arg = ts[i] * freq;
        e[j] = cosf(arg); e[j + half] = sinf(arg);
    }
}
void timestep_embedding_f32(const float * ts, const tdesc & y, int64_t n, int dim, int max_period, hipStream_t st) {
    if (n == 0 || dim < 1) return;
    const int64_t per = dim / 2 + (dim & 1);
    k_timestep_embedding<<<grid_for(n * per), dim3(256), 0, st>>>(ts, (char *) y.p, (int64_t) y.nb[1], n, dim, max_period);
}

// ---------------------------------------------------------------------------------------------- SUM_ROWS (ops.cpp:1399-1430): one wave per row, double accumulation
__global__ void __launch_bounds__(256) k_sum_rows(t4 x, t4 y) {
    const int64_t nrows = x.ne[1] * x.ne[2] * x.ne[3];
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6); r < nrows; r += (int64_t) gridDim.x * 4) {
        const int64_t i1 = r % x.ne[1], i2 = (r / x.ne[1]) % x.ne[2], i3 = r / (x.ne[1] * x.ne[2]);
        const char * row = x.p + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3];
        double s = 0.0;
        for (int64_t i = lane; i < x.ne[0]; i += 64) s += (double) *(const float *) (row + i * 4);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) *(float *) (y.p + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]) = (float) s;
    }
}
void sum_rows_f32(const tdesc & x, const tdesc & y, hipStream_t st) {
    const int64_t nrows = x.ne[1] * x.ne[2] * x.ne[3];
    if (nrows == 0) return;
    int64_t g = (nrows + 3) / 4; if (g > 16384) g = 16384;
    k_sum_rows<<<dim3((unsigned) g), dim3(256), 0, st>>>(to_t4(x), to_t4(y));
}

// ---------------------------------------------------------------------------------------------- CONV_TRANSPOSE_1D (ops.cpp:5952-6038 f16 kernel, :6040-6122 f32 kernel)
// kernel [K, Cout, Cin], x [L, Cin] f32 -> y [(L - 1) * s0 + K, Cout]:  y[o][l * s0 + k] += sum_c x[c][l] * w[c][o][k].
// One thread per output element: the taps (l, k) with l * s0 + k == pos are visited in ascending l like the reference's accumulation into
// dst; with an f16 kernel x is rounded to f16 first (the reference's wdata copy), products accumulate in f32 (ggml_vec_dot_f16).
template <bool W16>
__global__ void __launch_bounds__(256) k_conv_transpose_1d(const char * __restrict__ w, int64_t w_nb1, int64_t w_nb2, const char * __restrict__ x, int64_t x_nb1,
                                                           char * __restrict__ y, int64_t y_nb1, int K, int Cout, int Cin, int L, int OL, int s0) {
    T2W_LOOP((int64_t) OL * Cout) {
        const int o = (int) (t / OL), pos = (int) (t - (int64_t) o * OL);
        int l0 = pos >= K ? (pos - K + s0) / s0 : 0;                     // smallest l with pos - l * s0 <= K - 1
        float acc = 0.0f;
        for (int l = l0; l < L && l * s0 <= pos; ++l) {
            const int k = pos - l * s0;
            float v = 0.0f;
#pragma unroll 8
            for (int c = 0; c < Cin; ++c) {                               // (unrolled: eight channel pairs of loads in flight, the additions in the same order)
                const float xv = *(const float *) (x + (int64_t) c * x_nb1 + (int64_t) l * 4);
                if (W16) v += h2f(f2h(xv)) * h2f(*(const uint16_t *) (w + (int64_t) c * w_nb2 + (int64_t) o * w_nb1 + (int64_t) k * 2));
                else     v += xv * *(const float *) (w + (int64_t) c * w_nb2 + (int64_t) o * w_nb1 + (int64_t) k * 4);
            }
            acc += v;
        }
        *(float *) (y + (int64_t) o * y_nb1 + (int64_t) pos * 4) = acc;
    }
}
// The HiFT up-sampling stages (512 -> 256 channels, K 16, stride 8, ...): one workgroup per (output channel, 256 output positions).  The channel's taps
// w[c][o][0..K) of ALL input channels are staged in LDS once (coalesced K-float runs; Cin * K floats: 32 KB at most there) instead of being re-read from global
// memory with a channel stride by every thread of every position; x[c][l] is read coalesced (neighbouring positions share or neighbour l).  Same visiting
// order as the kernel above: taps in ascending l, channels in ascending c inside a tap.
template <bool W16>
__global__ void __launch_bounds__(256) k_conv_transpose_1d_lds(const char * __restrict__ w, int64_t w_nb1, int64_t w_nb2, const char * __restrict__ x, int64_t x_nb1,
                                                               char * __restrict__ y, int64_t y_nb1, int K, int Cin, int L, int OL, int s0) {
    extern __shared__ float ct_w[];                                        // [Cin][K]
    const int o = (int) blockIdx.y, pos = (int) blockIdx.x * 256 + (int) threadIdx.x;
    for (int i = (int) threadIdx.x; i < Cin * K; i += 256) {
        const int c = i / K, k = i - c * K;
        const char * p = w + (int64_t) c * w_nb2 + (int64_t) o * w_nb1;
        ct_w[i] = W16 ? h2f(*(const uint16_t *) (p + (int64_t) k * 2)) : *(const float *) (p + (int64_t) k * 4);
    }
    __syncthreads();
    if (pos >= OL) return;
    const int l0 = pos >= K ? (pos - K + s0) / s0 : 0;
    float acc = 0.0f;
    for (int l = l0; l < L && l * s0 <= pos; ++l) {
        const int k = pos - l * s0;
        float v = 0.0f;
#pragma unroll 8
        for (int c = 0; c < Cin; ++c) {
            const float xv = *(const float *) (x + (int64_t) c * x_nb1 + (int64_t) l * 4);
            v += (W16 ? h2f(f2h(xv)) : xv) * ct_w[c * K + k];
        }
        acc += v;
    }
    *(float *) (y + (int64_t) o * y_nb1 + (int64_t) pos * 4) = acc;
}
void conv_transpose_1d_f32(const tdesc & w, int w_type, const tdesc & x, const tdesc & y, int s0, hipStream_t st) {
    const int K = (int) w.ne[0], Cout = (int) w.ne[1], Cin = (int) w.ne[2], L = (int) x.ne[0], OL = (int) y.ne[0];
    if ((int64_t) OL * Cout == 0) return;
    static const bool no_lds = getenv("MI355X_NO_CONVT_LDS") != nullptr;
    const size_t lds = (size_t) Cin * (size_t) K * 4;
    if (!no_lds && lds <= 48 * 1024 && Cout <= 65535 && Cin >= 16) {
        const dim3 grid((unsigned) ((OL + 255) / 256), (unsigned) Cout);
        if (w_type == GGML_TYPE_F16) k_conv_transpose_1d_lds<true><<<grid, dim3(256), lds, st>>>((const char *) w.p, (int64_t) w.nb[1], (int64_t) w.nb[2], (const char *) x.p, (int64_t) x.nb[1], (char *) y.p, (int64_t) y.nb[1], K, Cin, L, OL, s0);
        else                         k_conv_transpose_1d_lds<false><<<grid, dim3(256), lds, st>>>((const char *) w.p, (int64_t) w.nb[1], (int64_t) w.nb[2], (const char *) x.p, (int64_t) x.nb[1], (char *) y.p, (int64_t) y.nb[1], K, Cin, L, OL, s0);
        return;
    }
    if (w_type == GGML_TYPE_F16) k_conv_transpose_1d<true><<<grid_for((int64_t) OL * Cout), dim3(256), 0, st>>>((const char *) w.p, (int64_t) w.nb[1], (int64_t) w.nb[2], (const char *) x.p, (int64_t) x.nb[1], (char *) y.p, (int64_t) y.nb[1], K, Cout, Cin, L, OL, s0);
    else                         k_conv_transpose_1d<false><<<grid_for((int64_t) OL * Cout), dim3(256), 0, st>>>((const char *) w.p, (int64_t) w.nb[1], (int64_t) w.nb[2], (const char *) x.p, (int64_t) x.nb[1], (char *) y.p, (int64_t) y.nb[1], K, Cout, Cin, L, OL, s0);
}

// ---------------------------------------------------------------------------------------------- CPY f32 <-> i32 (ggml_compute_forward_dup_flt<float, int32_t>: C casts)
template <typename TS, typename TD>
__global__ void __launch_bounds__(256) k_cast_fi(t4 s, t4 d) {
    const int64_t total = s.ne[0] * s.ne[1] * s.ne[2] * s.ne[3];
    T2W_LOOP(total) {
        int64_t a0, a1, a2, a3, b0, b1, b2, b3; unravel(t, s.ne, a0, a1, a2, a3); unravel(t, d.ne, b0, b1, b2, b3);
        *(TD *) (d.p + b0 * d.nb[0] + b1 * d.nb[1] + b2 * d.nb[2] + b3 * d.nb[3]) = (TD) *(const TS *) (s.p + a0 * s.nb[0] + a1 * s.nb[1] + a2 * s.nb[2] + a3 * s.nb[3]);
    }
}
void cast_f32_i32(const tdesc & src, bool src_is_f32, const tdesc & dst, hipStream_t st) {
    const int64_t total = src.ne[0] * src.ne[1] * src.ne[2] * src.ne[3];
    if (total == 0) return;
    if (src_is_f32) k_cast_fi<float, int32_t><<<grid_for(total), dim3(256), 0, st>>>(to_t4(src), to_t4(dst));
    else            k_cast_fi<int32_t, float><<<grid_for(total), dim3(256), 0, st>>>(to_t4(src), to_t4(dst));
}

} // namespace mi

// gemm_any.hip -- MUL_MAT with more than 8 columns for the operand shapes the F16 MFMA GEMM (gemm.hip) does not take: F32 weights (the
// Token2Wav graphs are all-F32; SigLip2's K . Q^T, tools/omni/vision.cpp:648-703) and F16 weights whose contraction length is not a
// multiple of 32 (Whisper's V^T . P over 1500 frames, audition.cpp:600-640).  Before this kernel those mat-muls ran as mat-vecs in chunks
// of 8 columns: 3136 launches / 86 ms for ONE Whisper encoder layer, 4608 / 56 ms for one SigLip2 layer.
//
//     dst[n][m] = sum_k W[m][k] * X[n][k]      W: F32 or F16 rows, X: f32 rows (rounded to f16 first when W is F16 -- the reference's
//                                              vec_dot_type conversion, ggml-cpu.c:1245-1268), f32 accumulate
// on v_mfma_f32_32x32x2f32: f32 x f32 fused multiply-add into f32, i.e. the CPU's arithmetic (ggml_vec_dot_f32 / _f16 accumulate in f32
// FMA lanes), only the summation order differs.  Any M, N, K >= 1 (tiles are zero-filled past the edges), rows K-contiguous with any
// row stride, broadcast batch over dims 2, 3 like ggml_compute_forward_mul_mat.  Workgroup = 4 waves on a 64 x 64 tile, 32-wide K-steps
// through two padded LDS buffers (one barrier per step, the next step's global loads in flight under the MFMAs): these matrices are small --
// the point is one launch instead of thousands, and a short per-workgroup latency chain.
#include "../kernels.hpp"
#include "act_dev.hpp"

namespace mi {

typedef float ga_acc __attribute__((ext_vector_type(16)));

struct gemm_any_dev {
    const char * W; size_t w_rs, w_nb2, w_nb3;
    const char * X; size_t x_rs, x_nb2, x_nb3;
    char * dst; size_t dst_cs, dst_nb2, dst_nb3;
    const float * bias;
    int act;                                             // 1: GELU (the reference's f16-table arithmetic, act_dev.hpp op_gelu) on the finished value, behind the bias
    int M, N, K, tiles_m, ne12, r2, r3, round_x, accumulate;      // round_x: 0 none, 1 activations rounded to f16, 2 to bf16 (and the 16-bit weights are bf16)
    float * partial; unsigned * counters; int ksplit;             // k_gemm_f32_sk128: K split over gridDim.z workgroups (1: none)
    const char * W_more[2]; char * dst_more[2]; const float * bias_more[2];     // k_gemm_f32_t16: matrices 1, 2 of a grouped launch (blockIdx.z)
};

template <typename WT, typename XT>
__global__ void __launch_bounds__(256) k_gemm_any(const gemm_any_dev g) {
    constexpr int KS = 32, LD = KS + 1, PT = 64 * KS / 256;        // 8 elements of each operand per thread and K-step
    __shared__ float Ws[2][64 * LD], Xs[2][64 * LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave & 1, wn = wave >> 1;
    const int tm = (int) blockIdx.x % g.tiles_m, tn = (int) blockIdx.x / g.tiles_m;
    const int i12 = (int) blockIdx.y % g.ne12, i13 = (int) blockIdx.y / g.ne12;
    const char * W = g.W + (size_t) (i12 / g.r2) * g.w_nb2 + (size_t) (i13 / g.r3) * g.w_nb3;
    const char * X = g.X + (size_t) i12 * g.x_nb2 + (size_t) i13 * g.x_nb3;
    const int m0 = tm * 64, n0 = tn * 64;
    ga_acc acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int fr = lane & 31, kh = lane >> 5;
    // element i of this thread: tile row (t >> 5) + 8 i, column t & 31 -- consecutive lanes read consecutive k of one row
    const int c = t & 31, r0 = t >> 5;
    float wv[PT], xv[PT];
    auto fetch = [&](int k0) {                                     // branch-free: clamped addresses, zero selected afterwards (a conditional load is a branch + a full wait)
        const int k = k0 + c, kc = k < g.K ? k : g.K - 1;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int r = r0 + 8 * i;
            const int mr = m0 + r < g.M ? m0 + r : g.M - 1, nr = n0 + r < g.N ? n0 + r : g.N - 1;
            const char * pw = W + (size_t) mr * g.w_rs + (size_t) kc * sizeof(WT);
            const char * px = X + (size_t) nr * g.x_rs + (size_t) kc * sizeof(XT);
            wv[i] = sizeof(WT) == 2 ? (g.round_x == 2 ? __uint_as_float((uint32_t) *(const uint16_t *) pw << 16) : h2f(*(const uint16_t *) pw)) : *(const float *) pw;
            xv[i] = sizeof(XT) == 2 ? h2f(*(const uint16_t *) px) : *(const float *) px;       // (F16 x F16: the im2col columns of the encoders' convolutions)
        }
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int r = r0 + 8 * i;
            if (sizeof(XT) == 4 && g.round_x == 1) xv[i] = h2f(f2h(xv[i]));
            if (sizeof(XT) == 4 && g.round_x == 2) { uint32_t u = __float_as_uint(xv[i]); u = (u & 0x7fffffffu) > 0x7f800000u ? (u | 0x00400000u) & 0xffff0000u : (u + (0x7fffu + ((u >> 16) & 1u))) & 0xffff0000u; xv[i] = __uint_as_float(u); }
            if (k >= g.K || m0 + r >= g.M) wv[i] = 0.0f;
            if (k >= g.K || n0 + r >= g.N) xv[i] = 0.0f;
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PT; ++i) { Ws[buf][(r0 + 8 * i) * LD + c] = wv[i]; Xs[buf][(r0 + 8 * i) * LD + c] = xv[i]; }
    };
    fetch(0);
    int buf = 0;
    for (int k0 = 0; k0 < g.K; k0 += KS, buf ^= 1) {
        park(buf);
        __syncthreads();                                           // K-step k0 visible; everybody is past the MFMAs that read the other buffer
        if (k0 + KS < g.K) fetch(k0 + KS);                         // in flight under this step's MFMAs
#pragma unroll
        for (int kk = 0; kk < KS / 2; ++kk) {
            const float a = Xs[buf][(wn * 32 + fr) * LD + 2 * kk + kh], b = Ws[buf][(wm * 32 + fr) * LD + 2 * kk + kh];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    char * dst = g.dst + (size_t) i12 * g.dst_nb2 + (size_t) i13 * g.dst_nb3;
    const int m = m0 + wm * 32 + fr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = n0 + wn * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (m < g.M && n < g.N) { float * p = (float *) (dst + (size_t) n * g.dst_cs + (size_t) m * 4); float v = g.accumulate ? *p + acc[e] : acc[e]; if (g.bias) v = __fadd_rn(v, g.bias[m]); if (g.act) v = op_gelu(v); *p = v; }
    }
}

// F16 weights on the f16 matrix cores: the same tiles and edge rules as k_gemm_any, both operands parked in LDS as f16 (the activations rounded
// first -- what the reference's vec_dot_type conversion does; f16 x f16 products are exact in f32, f32 accumulate) and multiplied with
// v_mfma_f32_32x32x16_f16: 64-wide K-steps, 4 MFMAs per wave and step instead of 16 f32 ones per 32.  Whisper's V^T . P (K = 1500 frames, rows
// of 3000 bytes: no 16-byte alignment for the DMA GEMMs of gemm.hip) went from 118 us to the figure in DESIGN.md.  VEC = 2: element pairs
// (4-byte f16 / 8-byte f32 loads; the launcher checks alignment and K even), VEC = 1: single elements, any alignment.
template <typename XT, int VEC>
__global__ void __launch_bounds__(256) k_gemm_any_h(const gemm_any_dev g) {
    constexpr int KS = 64, LD = KS + 8, CPR = KS / VEC, RPP = 256 / CPR, PT = 64 / RPP;     // LD in halves: rows of 144 bytes (16-byte aligned 8-half reads)
    __shared__ __attribute__((aligned(16))) uint16_t Ws[2][64 * LD], Xs[2][64 * LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave & 1, wn = wave >> 1;
    const int tm = (int) blockIdx.x % g.tiles_m, tn = (int) blockIdx.x / g.tiles_m;
    const int i12 = (int) blockIdx.y % g.ne12, i13 = (int) blockIdx.y / g.ne12;
    const char * W = g.W + (size_t) (i12 / g.r2) * g.w_nb2 + (size_t) (i13 / g.r3) * g.w_nb3;
    const char * X = g.X + (size_t) i12 * g.x_nb2 + (size_t) i13 * g.x_nb3;
    const int m0 = tm * 64, n0 = tn * 64;
    ga_acc acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int fr = lane & 31, kh = lane >> 5;
    const int c = t % CPR, r0 = t / CPR;                           // element (pair) i of this thread: tile row r0 + RPP i, columns VEC c ..
    uint32_t wv[PT], xv[PT];                                       // VEC f16 values each
    auto fetch = [&](int k0) {                                     // branch-free: clamped addresses, zero selected afterwards
        const int k = k0 + VEC * c, kc = k < g.K ? k : g.K - VEC;  // (VEC = 2: K is even, a pair is inside or outside as a whole)
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int r = r0 + RPP * i;
            const int mr = m0 + r < g.M ? m0 + r : g.M - 1, nr = n0 + r < g.N ? n0 + r : g.N - 1;
            const char * pw = W + (size_t) mr * g.w_rs + (size_t) kc * 2;
            const char * px = X + (size_t) nr * g.x_rs + (size_t) kc * sizeof(XT);
            wv[i] = VEC == 2 ? *(const uint32_t *) pw : (uint32_t) *(const uint16_t *) pw;
            if (sizeof(XT) == 2) xv[i] = VEC == 2 ? *(const uint32_t *) px : (uint32_t) *(const uint16_t *) px;
            else if (VEC == 2)   { const float2 f = *(const float2 *) px; xv[i] = (uint32_t) f2h(f.x) | ((uint32_t) f2h(f.y) << 16); }
            else                 xv[i] = (uint32_t) f2h(*(const float *) px);
        }
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int r = r0 + RPP * i;
            if (k >= g.K || m0 + r >= g.M) wv[i] = 0u;
            if (k >= g.K || n0 + r >= g.N) xv[i] = 0u;
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int o = (r0 + RPP * i) * LD + VEC * c;
            if (VEC == 2) { *(uint32_t *) &Ws[buf][o] = wv[i]; *(uint32_t *) &Xs[buf][o] = xv[i]; }
            else          { Ws[buf][o] = (uint16_t) wv[i]; Xs[buf][o] = (uint16_t) xv[i]; }
        }
    };
    fetch(0);
    int buf = 0;
    for (int k0 = 0; k0 < g.K; k0 += KS, buf ^= 1) {
        park(buf);
        __syncthreads();                                           // K-step k0 visible; everybody is past the MFMAs that read the other buffer
        if (k0 + KS < g.K) fetch(k0 + KS);                         // in flight under this step's MFMAs
#pragma unroll
        for (int kk = 0; kk < KS / 16; ++kk) {
            const f16x8 a = *(const f16x8 *) &Xs[buf][(wn * 32 + fr) * LD + 16 * kk + 8 * kh];
            const f16x8 b = *(const f16x8 *) &Ws[buf][(wm * 32 + fr) * LD + 16 * kk + 8 * kh];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
    }
    char * dst = g.dst + (size_t) i12 * g.dst_nb2 + (size_t) i13 * g.dst_nb3;
    const int m = m0 + wm * 32 + fr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = n0 + wn * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (m < g.M && n < g.N) { float * p = (float *) (dst + (size_t) n * g.dst_cs + (size_t) m * 4); float v = g.accumulate ? *p + acc[e] : acc[e]; if (g.bias) v = __fadd_rn(v, g.bias[m]); if (g.act) v = op_gelu(v); *p = v; }
    }
}

// Small products (a Token2Wav DiT projection: 512 x 200 x 512 is 32 tiles of 64 x 64): a wave's 32 x 32 tile needs K / 2 dependent f32 MFMAs of 64
// cycles each whatever the workgroup tile, so few large tiles mean a long chain on a few CUs.  Here a workgroup owns ONE 32 x 32 tile and its
// four waves split K (each stages its own 32-wide K-steps through a wave-private LDS region: no workgroup barrier inside the loop); the four
// partial tiles are folded through LDS in wave order.
template <typename WT, typename XT>
__global__ void __launch_bounds__(256) k_gemm_any_sk(const gemm_any_dev g) {
    constexpr int KS = 32, LD = KS + 1;
    __shared__ float Ws[4][32 * LD], Xs[4][32 * LD];               // wave-private: a wave's LDS operations execute in order, one buffer is enough
    __shared__ float red[3][64 * 16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tm = (int) blockIdx.x % g.tiles_m, tn = (int) blockIdx.x / g.tiles_m;        // (tiles_m counts 32-row tiles here)
    const int i12 = (int) blockIdx.y % g.ne12, i13 = (int) blockIdx.y / g.ne12;
    const char * W = g.W + (size_t) (i12 / g.r2) * g.w_nb2 + (size_t) (i13 / g.r3) * g.w_nb3;
    const char * X = g.X + (size_t) i12 * g.x_nb2 + (size_t) i13 * g.x_nb3;
    const int m0 = tm * 32, n0 = tn * 32;
    // this wave's K range: whole K-steps, the ranges of the four waves cover [0, K)
    const int nsteps = (g.K + KS - 1) / KS, per = (nsteps + 3) / 4;
    const int s_lo = wave * per, s_hi = s_lo + per < nsteps ? s_lo + per : nsteps;
    ga_acc acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int fr = lane & 31, kh = lane >> 5, c = lane & 31, r0 = lane >> 5;
    float wv[16], xv[16];
    auto fetch = [&](int k0) {                                     // branch-free: clamped addresses, zero selected afterwards
        const int k = k0 + c, kc = k < g.K ? k : g.K - 1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + 2 * i;
            const int mr = m0 + r < g.M ? m0 + r : g.M - 1, nr = n0 + r < g.N ? n0 + r : g.N - 1;
            const char * pw = W + (size_t) mr * g.w_rs + (size_t) kc * sizeof(WT);
            const char * px = X + (size_t) nr * g.x_rs + (size_t) kc * sizeof(XT);
            wv[i] = sizeof(WT) == 2 ? (g.round_x == 2 ? __uint_as_float((uint32_t) *(const uint16_t *) pw << 16) : h2f(*(const uint16_t *) pw)) : *(const float *) pw;
            xv[i] = sizeof(XT) == 2 ? h2f(*(const uint16_t *) px) : *(const float *) px;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + 2 * i;
            if (sizeof(XT) == 4 && g.round_x == 1) xv[i] = h2f(f2h(xv[i]));
            if (sizeof(XT) == 4 && g.round_x == 2) { uint32_t u = __float_as_uint(xv[i]); u = (u & 0x7fffffffu) > 0x7f800000u ? (u | 0x00400000u) & 0xffff0000u : (u + (0x7fffu + ((u >> 16) & 1u))) & 0xffff0000u; xv[i] = __uint_as_float(u); }
            if (k >= g.K || m0 + r >= g.M) wv[i] = 0.0f;
            if (k >= g.K || n0 + r >= g.N) xv[i] = 0.0f;
        }
    };
    if (s_lo < s_hi) fetch(s_lo * KS);
    for (int s = s_lo; s < s_hi; ++s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { Ws[wave][(r0 + 2 * i) * LD + c] = wv[i]; Xs[wave][(r0 + 2 * i) * LD + c] = xv[i]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (s + 1 < s_hi) fetch((s + 1) * KS);
#pragma unroll
        for (int kk = 0; kk < KS / 2; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[wave][fr * LD + 2 * kk + kh], Ws[wave][fr * LD + 2 * kk + kh], acc, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave - 1][e * 64 + lane] = acc[e];
    }
    __syncthreads();
    if (wave > 0) return;
    char * dst = g.dst + (size_t) i12 * g.dst_nb2 + (size_t) i13 * g.dst_nb3;
    const int m = m0 + fr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        float v = acc[e];
#pragma unroll
        for (int w = 0; w < 3; ++w) v += red[w][e * 64 + lane];
        if (m < g.M && n < g.N) { float * p = (float *) (dst + (size_t) n * g.dst_cs + (size_t) m * 4); if (g.accumulate) v = *p + v; if (g.bias) v = __fadd_rn(v, g.bias[m]); if (g.act) v = op_gelu(v); *p = v; }
    }
}

// The same small f32 x f32 products without the LDS staging, for 16-byte aligned rows (the reference's Token2Wav: every DiT / conformer projection of a streaming
// window is 512..2048 rows x 56 columns x K 512..2048, 8600 of them per second of audio).  k_gemm_any_sk spends its time in round trips: 32-wide K-steps, each a
// dependent global load -> LDS -> 16 MFMAs (17-26 us per product).  In the v_mfma_f32_32x32x2f32 operand layout lane l supplies row l % 32 and ONE k per MFMA, and
// any assignment of a step's k values to (MFMA index, lane half) is valid as long as both operands use the same one -- so each lane reads 64 CONTIGUOUS floats of
// its own row (half 0: k0 .. k0+63, half 1: k0+64 .. k0+127) with sixteen 16-byte loads per operand, all in flight at once, and feeds element kk to MFMA kk:
// one round trip per 128 k.  NW waves split K (4 up to K 512, 8 beyond); the partial tiles are folded through LDS in wave order (deterministic).
template <int NW>
__global__ void __launch_bounds__(NW * 64) k_gemm_f32_rows(const gemm_any_dev g) {
    constexpr int KS = 128;
    __shared__ float red[NW - 1][64 * 16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tm = (int) blockIdx.x % g.tiles_m, tn = (int) blockIdx.x / g.tiles_m;        // (tiles_m counts 32-row tiles)
    const int i12 = (int) blockIdx.y % g.ne12, i13 = (int) blockIdx.y / g.ne12;
    const char * W = g.W + (size_t) (i12 / g.r2) * g.w_nb2 + (size_t) (i13 / g.r3) * g.w_nb3;
    const char * X = g.X + (size_t) i12 * g.x_nb2 + (size_t) i13 * g.x_nb3;
    const int m0 = tm * 32, n0 = tn * 32, fr = lane & 31, kh = lane >> 5;
    const int mr = m0 + fr < g.M ? m0 + fr : g.M - 1, nr = n0 + fr < g.N ? n0 + fr : g.N - 1;      // rows past the edge re-read the last row; their results are not stored
    const char * pw = W + (size_t) mr * g.w_rs, * px = X + (size_t) nr * g.x_rs;
    const int nsteps = (g.K + KS - 1) / KS, per = (nsteps + NW - 1) / NW;
    const int s_lo = wave * per, s_hi = s_lo + per < nsteps ? s_lo + per : nsteps;
    ga_acc acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    for (int s = s_lo; s < s_hi; ++s) {
        const int kb = s * KS + kh * 64;
        float4 wv[16], xv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {                             // branch-free: clamped addresses (K % 4 == 0: a quad is inside or outside as a whole), zero selected afterwards
            const int k = kb + 4 * j, kc = k < g.K ? k : g.K - 4;
            wv[j] = *(const float4 *) (pw + (size_t) kc * 4);
            xv[j] = *(const float4 *) (px + (size_t) kc * 4);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (kb + 4 * j >= g.K) { wv[j] = make_float4(0.f, 0.f, 0.f, 0.f); xv[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[j].x, wv[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[j].y, wv[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[j].z, wv[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[j].w, wv[j].w, acc, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave - 1][e * 64 + lane] = acc[e];
    }
    __syncthreads();
    if (wave > 0) return;
    char * dst = g.dst + (size_t) i12 * g.dst_nb2 + (size_t) i13 * g.dst_nb3;
    const int m = m0 + fr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        float v = acc[e];
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) v += red[w][e * 64 + lane];
        if (m < g.M && n < g.N) { float * p = (float *) (dst + (size_t) n * g.dst_cs + (size_t) m * 4); if (g.accumulate) v = *p + v; if (g.bias) v = __fadd_rn(v, g.bias[m]); if (g.act) v = op_gelu(v); *p = v; }
    }
}

// The small f32 x f32 products once more, with COALESCED operand reads.  k_gemm_f32_rows' per-lane row reads fetch each 128-byte line with eight separate 16-byte
// requests from eight different instructions; with a few waves per CU in flight the lines are gone from the 16 KB L1 in between and every request goes back to L2
// (4 MB of fc2 weights: 23 us on 32 workgroups).  Here a wave reads TWO rows x 512 contiguous bytes per instruction (whole lines), stages its own 32 x 128 slice of
// each operand in a wave-private LDS region (rows padded to 132 floats) and feeds v_mfma_f32_32x32x2f32 from 16-byte LDS reads -- lane (row, half) takes the quad at
// k = 8 q + 4 half of its row for MFMAs 4 q .. 4 q + 3: both operands use the same assignment, which is all the instruction needs.  K is split over the four waves in
// 128-deep steps, the next step's global reads are in flight under the current step's 64 MFMAs, no workgroup barrier inside the loop; partial tiles folded in wave order.
extern __shared__ float ga_dyn_lds[];
__global__ void __launch_bounds__(256) k_gemm_f32_sk128(const gemm_any_dev g) {
    constexpr int KS = 128, LD = KS + 4;
    float * Ws = ga_dyn_lds + (threadIdx.x >> 6) * (2 * 32 * LD), * Xs = Ws + 32 * LD;
    float * red = ga_dyn_lds + 4 * (2 * 32 * LD);                  // [3][64 * 16]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tm = (int) blockIdx.x % g.tiles_m, tn = (int) blockIdx.x / g.tiles_m;        // (tiles_m counts 32-row tiles)
    const int i12 = (int) blockIdx.y % g.ne12, i13 = (int) blockIdx.y / g.ne12;
    const char * W = g.W + (size_t) (i12 / g.r2) * g.w_nb2 + (size_t) (i13 / g.r3) * g.w_nb3;
    const char * X = g.X + (size_t) i12 * g.x_nb2 + (size_t) i13 * g.x_nb3;
    const int m0 = tm * 32, n0 = tn * 32, fr = lane & 31, kh = lane >> 5;
    const int nsteps = (g.K + KS - 1) / KS;
    const int kz = (int) blockIdx.z, spz = (nsteps + g.ksplit - 1) / g.ksplit;                     // this workgroup's steps [z_lo, z_hi)
    const int z_lo = kz * spz, z_hi = z_lo + spz < nsteps ? z_lo + spz : nsteps;
    const int per = (z_hi - z_lo + 3) / 4;
    const int s_lo = z_lo + wave * per, s_hi = s_lo + per < z_hi ? s_lo + per : z_hi;
    ga_acc acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    // staging map: instruction i of 16 covers tile rows 2 i, 2 i + 1; lane (kh, fr) the quad fr of row 2 i + kh
    float4 wv[16], xv[16];
    auto fetch = [&](int k0) {                                     // branch-free: clamped addresses (K % 4 == 0), zero selected afterwards
        const int k = k0 + 4 * fr, kc = k < g.K ? k : g.K - 4;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = 2 * i + kh;
            const int mr = m0 + r < g.M ? m0 + r : g.M - 1, nr = n0 + r < g.N ? n0 + r : g.N - 1;
            wv[i] = *(const float4 *) (W + (size_t) mr * g.w_rs + (size_t) kc * 4);
            xv[i] = *(const float4 *) (X + (size_t) nr * g.x_rs + (size_t) kc * 4);
        }
        if (k >= g.K) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { wv[i] = make_float4(0.f, 0.f, 0.f, 0.f); xv[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
    };
    if (s_lo < s_hi) fetch(s_lo * KS);
    for (int s = s_lo; s < s_hi; ++s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { *(float4 *) &Ws[(2 * i + kh) * LD + 4 * fr] = wv[i]; *(float4 *) &Xs[(2 * i + kh) * LD + 4 * fr] = xv[i]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (s + 1 < s_hi) fetch((s + 1) * KS);
#pragma unroll
        for (int q = 0; q < KS / 8; ++q) {
            const float4 a = *(const float4 *) &Xs[fr * LD + 8 * q + 4 * kh], b = *(const float4 *) &Ws[fr * LD + 8 * q + 4 * kh];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[(wave - 1) * 1024 + e * 64 + lane] = acc[e];
    }
    __syncthreads();
    if (wave > 0) return;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        v[e] = acc[e];
#pragma unroll
        for (int w = 0; w < 3; ++w) v[e] += red[w * 1024 + e * 64 + lane];
    }
    if (g.ksplit > 1) {
        const int tile = (int) (blockIdx.y * gridDim.x + blockIdx.x), ntiles = (int) (gridDim.x * gridDim.y);
        float * mine = g.partial + ((size_t) kz * ntiles + tile) * 1024;
#pragma unroll
        for (int e = 0; e < 16; ++e) mine[e * 64 + lane] = v[e];
        __threadfence();                                           // the partial tile is visible device-wide before the ticket is taken
        unsigned ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(g.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != (unsigned) (g.ksplit - 1)) return;
        __threadfence();
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0.0f;
        for (int z = 0; z < g.ksplit; ++z) {                       // split order, whoever arrives last: deterministic
            const float * p = g.partial + ((size_t) z * ntiles + tile) * 1024;
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += __builtin_nontemporal_load(p + e * 64 + lane);
        }
        if (lane == 0) __hip_atomic_store(g.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next launch
    }
    char * dst = g.dst + (size_t) i12 * g.dst_nb2 + (size_t) i13 * g.dst_nb3;
    const int m = m0 + fr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        float r = v[e];
        if (m < g.M && n < g.N) { float * p = (float *) (dst + (size_t) n * g.dst_cs + (size_t) m * 4); if (g.accumulate) r = *p + r; if (g.bias) r = __fadd_rn(r, g.bias[m]); if (g.act) r = op_gelu(r); *p = r; }
    }
}

// The small f32 x f32 products of a Token2Wav DiT block once more (512 .. 2048 rows x 50 .. 56 frames x batch 2, K 512 .. 2048; the attention's V^T . P with K = 200):
// with 32 x 32 tiles such a product is 64 workgroups -- a quarter of the chip -- and each wave's chain is K / 8 dependent 64-cycle MFMAs (K = 2048: 16 us per product;
// splitting K over workgroups buys nothing, the device-scope fences of the ticket fold cost what the split saves).  Here the tile is 16 x 16 on
// v_mfma_f32_16x16x4f32 (32 cycles, the same 32 MAC per cycle): four times the workgroups, a quarter of the chain, the fold stays inside the workgroup.  Staging as in
// k_gemm_f32_sk128: whole 512-byte row pieces per instruction into a wave-private LDS region (rows padded to 132 floats), 16-byte LDS reads -- lane (row, g) takes the
// quad at k = 16 q + 4 g for MFMAs 4 q .. 4 q + 3, both operands alike.  The four waves split K in chunks that are multiples of 16, so K = 200 still uses all four.
__global__ void __launch_bounds__(256) k_gemm_f32_t16(const gemm_any_dev g) {
    constexpr int KS = 128, LD = KS + 4;
    float * Ws = ga_dyn_lds + (threadIdx.x >> 6) * (2 * 16 * LD), * Xs = Ws + 16 * LD;
    float * red = ga_dyn_lds + 4 * (2 * 16 * LD);                  // [3][64 * 4]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tm = (int) blockIdx.x % g.tiles_m, tn = (int) blockIdx.x / g.tiles_m;        // (tiles_m counts 16-row tiles)
    const int i12 = (int) blockIdx.y % g.ne12, i13 = (int) blockIdx.y / g.ne12;
    const int z = (int) blockIdx.z;                               // (grouped launch: which matrix)
    const char * W = (z == 0 ? g.W : g.W_more[z - 1]) + (size_t) (i12 / g.r2) * g.w_nb2 + (size_t) (i13 / g.r3) * g.w_nb3;
    const char * X = g.X + (size_t) i12 * g.x_nb2 + (size_t) i13 * g.x_nb3;
    const int m0 = tm * 16, n0 = tn * 16, fr = lane & 31, kh = lane >> 5, r16 = lane & 15, gq = lane >> 4;
    const int chunk = ((g.K + 3) / 4 + 15) & ~15;                  // this wave's k range [k_lo, k_hi)
    const int k_lo = wave * chunk, k_hi = k_lo + chunk < g.K ? k_lo + chunk : g.K;
    typedef float acc4 __attribute__((ext_vector_type(4)));
    acc4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
    float4 wv[8], xv[8];
    auto fetch = [&](int k0) {                                     // branch-free: clamped addresses (K % 4 == 0), zero selected afterwards
        const int k = k0 + 4 * fr, kc = k < g.K ? k : g.K - 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 2 * i + kh;
            const int mr = m0 + r < g.M ? m0 + r : g.M - 1, nr = n0 + r < g.N ? n0 + r : g.N - 1;
            wv[i] = *(const float4 *) (W + (size_t) mr * g.w_rs + (size_t) kc * 4);
            xv[i] = *(const float4 *) (X + (size_t) nr * g.x_rs + (size_t) kc * 4);
        }
        if (k >= k_hi) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { wv[i] = make_float4(0.f, 0.f, 0.f, 0.f); xv[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
    };
    if (k_lo < k_hi) fetch(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += KS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { *(float4 *) &Ws[(2 * i + kh) * LD + 4 * fr] = wv[i]; *(float4 *) &Xs[(2 * i + kh) * LD + 4 * fr] = xv[i]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (k0 + KS < k_hi) fetch(k0 + KS);
        const int nq = (k_hi - k0 < KS ? k_hi - k0 + 15 : KS) / 16;      // (wave-uniform)
        for (int q = 0; q < nq; ++q) {
            const float4 a = *(const float4 *) &Xs[r16 * LD + 16 * q + 4 * gq], b = *(const float4 *) &Ws[r16 * LD + 16 * q + 4 * gq];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(wave - 1) * 256 + e * 64 + lane] = acc[e];
    }
    __syncthreads();
    if (wave > 0) return;
    char * dst = (z == 0 ? g.dst : g.dst_more[z - 1]) + (size_t) i12 * g.dst_nb2 + (size_t) i13 * g.dst_nb3;
    const float * bias = z == 0 ? g.bias : g.bias_more[z - 1];
    const int m = m0 + r16;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = n0 + 4 * gq + e;
        float v = acc[e];
#pragma unroll
        for (int w = 0; w < 3; ++w) v += red[w * 256 + e * 64 + lane];
        if (m < g.M && n < g.N) { float * p = (float *) (dst + (size_t) n * g.dst_cs + (size_t) m * 4); if (g.accumulate) v = *p + v; if (bias) v = __fadd_rn(v, bias[m]); if (g.act) v = op_gelu(v); *p = v; }
    }
}

// K split over workgroups for k_gemm_f32_sk128: only when the tiles leave most CUs idle and every workgroup still gets whole 128-deep steps for its four waves
static int gemm_any_ksplit(int64_t M, int64_t N, int64_t K, int nbatch) {
    static const bool off = getenv("MI355X_NO_GEMM_F32_KSPLIT") != nullptr;
    if (off || nbatch < 1 || K % 4 != 0) return 1;
    const int64_t tiles = ((M + 31) / 32) * ((N + 31) / 32) * nbatch, nsteps = (K + 127) / 128;
    if (tiles > 96 || nsteps < 8) return 1;
    int s = (int) (nsteps / 4); if (s > 4) s = 4;
    while (s > 1 && tiles * s > 256) --s;
    return s < 1 ? 1 : s;
}
size_t gemm_any_split_scratch_bytes(int64_t M, int64_t N, int64_t K, int nbatch, bool f32_operands) {
    if (!f32_operands) return 0;
    const int s = gemm_any_ksplit(M, N, K, nbatch);
    return s > 1 ? (size_t) s * (size_t) (((M + 31) / 32) * ((N + 31) / 32) * nbatch) * 1024 * 4 : 0;
}

static bool gemm_any_t16_shape(const gemm_any_args & a) {
    static const bool no_t16 = getenv("MI355X_NO_GEMM_F32_T16") != nullptr, no_sk = getenv("MI355X_GEMM_ANY_NO_SPLIT") != nullptr;
    static const int64_t t16_max_tiles = getenv("MI355X_GEMM_T16_MAX_TILES") ? atoll(getenv("MI355X_GEMM_T16_MAX_TILES")) : 128;
    return !no_t16 && !no_sk && !a.x_f16 && !a.w_f16 && !a.w_bf16 && a.K % 4 == 0 && a.K >= 64 && ((a.M + 31) / 32) * ((a.N + 31) / 32) * a.nbatch < t16_max_tiles && ((a.M + 15) / 16) * ((a.N + 15) / 16) <= 65535 &&
           (((uintptr_t) a.W | a.w_rs | a.w_nb2 | a.w_nb3 | (uintptr_t) a.X | a.x_rs | a.x_nb2 | a.x_nb3) & 15) == 0;
}
bool gemm_any_group_ok(const gemm_any_args & a) {
    if (a.nmat < 1 || a.nmat > 3 || !gemm_any_t16_shape(a) || a.accumulate) return false;
    for (int q = 1; q < a.nmat; ++q) if (!a.W_more[q - 1] || !a.dst_more[q - 1] || ((uintptr_t) a.W_more[q - 1] & 15) != 0) return false;
    return true;
}
void gemm_any(const gemm_any_args & a, hipStream_t st) {
    if (a.M == 0 || a.N == 0 || a.nbatch == 0) return;
    if (a.nmat > 1 && !gemm_any_group_ok(a)) { fprintf(stderr, "[mi355x] gemm_any: grouped launch outside the 16 x 16-tile kernel's shapes\n"); abort(); }
    gemm_any_dev g;
    g.W = (const char *) a.W; g.w_rs = a.w_rs; g.w_nb2 = a.w_nb2; g.w_nb3 = a.w_nb3;
    g.X = (const char *) a.X; g.x_rs = a.x_rs; g.x_nb2 = a.x_nb2; g.x_nb3 = a.x_nb3;
    g.dst = (char *) a.dst; g.dst_cs = a.dst_cs; g.dst_nb2 = a.dst_nb2; g.dst_nb3 = a.dst_nb3;
    g.partial = nullptr; g.counters = nullptr; g.ksplit = 1;
    g.M = (int) a.M; g.N = (int) a.N; g.K = (int) a.K; g.tiles_m = (int) ((a.M + 63) / 64); g.ne12 = a.ne12; g.r2 = a.r2; g.r3 = a.r3; g.round_x = a.w_bf16 ? 2 : (a.w_f16 ? 1 : 0); g.accumulate = a.accumulate ? 1 : 0; g.bias = a.bias; g.act = a.act;
    static const bool no_sk = getenv("MI355X_GEMM_ANY_NO_SPLIT") != nullptr;
    // (F16 weights have the f16 matrix cores below -- 8x the K per MFMA, paired loads: their chains are short without a split; measured on Whisper's
    //  V^T . P of a streaming chunk, 64 x 50 x 400 x 16 heads: 24 us here, 6 us there)
    const bool h_path = a.w_f16 && !a.w_bf16 && !getenv("MI355X_NO_GEMM_ANY_H") && a.K < 2048;
    if (gemm_any_t16_shape(a)) {       // f32 x f32, 16-byte aligned rows, fewer than 128 tiles of 32 x 32: 16 x 16 tiles
        constexpr int lds = (4 * 2 * 16 * 132 + 3 * 256) * 4;       // 70 656 B: two workgroups per CU
        static bool attr[64] = {};
        int dev = 0; HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr[dev]) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemm_f32_t16, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); if (dev >= 0 && dev < 64) attr[dev] = true; }
        g.tiles_m = (int) ((a.M + 15) / 16);
        for (int q = 0; q < 2; ++q) { g.W_more[q] = (const char *) a.W_more[q]; g.dst_more[q] = (char *) a.dst_more[q]; g.bias_more[q] = a.bias_more[q]; }
        k_gemm_f32_t16<<<dim3((unsigned) (g.tiles_m * ((a.N + 15) / 16)), (unsigned) a.nbatch, (unsigned) a.nmat), dim3(256), lds, st>>>(g);
        return;
    }
    if (!no_sk && !h_path && (int64_t) g.tiles_m * ((a.N + 63) / 64) * a.nbatch < 128 && a.K >= 256) {          // few tiles, long chains: one 32 x 32 tile per workgroup, K split over its waves
        g.tiles_m = (int) ((a.M + 31) / 32);
        const dim3 grid((unsigned) (g.tiles_m * ((a.N + 31) / 32)), (unsigned) a.nbatch);
        static const bool no_rows = getenv("MI355X_NO_GEMM_F32_ROWS") != nullptr;
        if (!no_rows && !a.x_f16 && !a.w_f16 && !a.w_bf16 && a.K % 4 == 0 &&
            (((uintptr_t) a.W | a.w_rs | a.w_nb2 | a.w_nb3 | (uintptr_t) a.X | a.x_rs | a.x_nb2 | a.x_nb3) & 15) == 0) {       // f32 x f32, 16-byte aligned rows: no LDS staging, one round trip per 128 k
            static const bool no_sk128 = getenv("MI355X_NO_GEMM_F32_SK128") != nullptr;
            if (!no_sk128) {
                constexpr int lds = (4 * 2 * 32 * 132 + 3 * 1024) * 4;     // 147 456 B: one workgroup per CU
                static bool attr[64] = {};
                int dev = 0; HIP_CHECK(hipGetDevice(&dev));
                if (dev < 0 || dev >= 64 || !attr[dev]) { HIP_CHECK(hipFuncSetAttribute((const void *) k_gemm_f32_sk128, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); if (dev >= 0 && dev < 64) attr[dev] = true; }
                const int ks = gemm_any_ksplit(a.M, a.N, a.K, a.nbatch);
                const size_t need = gemm_any_split_scratch_bytes(a.M, a.N, a.K, a.nbatch, true);
                static const bool dbg = getenv("MI355X_GEMM_ANY_DEBUG") != nullptr;
                if (dbg) fprintf(stderr, "[mi355x] gemm_any sk128: M %lld N %lld K %lld batch %d: ksplit %d, scratch need %zu have %zu, partial %p counters %p (%d)\n", (long long) a.M, (long long) a.N, (long long) a.K, a.nbatch, ks, need, a.partial_bytes, (void *) a.partial, (void *) a.counters, a.n_counters);
                if (ks > 1 && a.partial && a.counters && need <= a.partial_bytes && (int64_t) grid.x * grid.y <= a.n_counters) {
                    g.partial = a.partial; g.counters = a.counters; g.ksplit = ks;
                    k_gemm_f32_sk128<<<dim3(grid.x, grid.y, (unsigned) ks), dim3(256), lds, st>>>(g);
                } else
                    k_gemm_f32_sk128<<<grid, dim3(256), lds, st>>>(g);
                return;
            }
            static const int force_nw = getenv("MI355X_GEMM_F32_ROWS_NW") ? atoi(getenv("MI355X_GEMM_F32_ROWS_NW")) : 0;
            if (force_nw == 4 || (force_nw == 0 && a.K <= 512)) k_gemm_f32_rows<4><<<grid, dim3(256), 0, st>>>(g); else k_gemm_f32_rows<8><<<grid, dim3(512), 0, st>>>(g);
            return;
        }
        if (a.x_f16)      k_gemm_any_sk<uint16_t, uint16_t><<<grid, dim3(256), 0, st>>>(g);
        else if (a.w_f16 || a.w_bf16) k_gemm_any_sk<uint16_t, float><<<grid, dim3(256), 0, st>>>(g);
        else              k_gemm_any_sk<float, float><<<grid, dim3(256), 0, st>>>(g);
        return;
    }
    const dim3 grid((unsigned) (g.tiles_m * ((a.N + 63) / 64)), (unsigned) a.nbatch);
    static const bool no_h = getenv("MI355X_NO_GEMM_ANY_H") != nullptr;
    if (a.w_f16 && !a.w_bf16 && !no_h) {                            // F16 weights: the f16 matrix cores (pairs when every row start and K allow it)
        const size_t xe = a.x_f16 ? 2 : 4;
        const bool vec2 = a.K % 2 == 0 && ((uintptr_t) a.W | a.w_rs | a.w_nb2 | a.w_nb3) % 4 == 0 && ((uintptr_t) a.X | a.x_rs | a.x_nb2 | a.x_nb3) % (2 * xe) == 0;
        if (a.x_f16) { if (vec2) k_gemm_any_h<uint16_t, 2><<<grid, dim3(256), 0, st>>>(g); else k_gemm_any_h<uint16_t, 1><<<grid, dim3(256), 0, st>>>(g); }
        else         { if (vec2) k_gemm_any_h<float, 2><<<grid, dim3(256), 0, st>>>(g);    else k_gemm_any_h<float, 1><<<grid, dim3(256), 0, st>>>(g); }
        return;
    }
    if (a.x_f16)      k_gemm_any<uint16_t, uint16_t><<<grid, dim3(256), 0, st>>>(g);       // (F16 activations only come with F16 weights: supports_op)
    else if (a.w_f16 || a.w_bf16) k_gemm_any<uint16_t, float><<<grid, dim3(256), 0, st>>>(g);
    else              k_gemm_any<float, float><<<grid, dim3(256), 0, st>>>(g);
}

} // namespace mi

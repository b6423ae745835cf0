// mmv1q.hip -- the batch-1 decode mat-vec for Q8_0 (and, k_mv1h below, F16) weights (the omni TTS / Token2Wav modules ship Q8_0: BASELINE configs[4]), the twin of
// mmv1.hip: one f32 activation row in, everything in front of and behind the dot products inside the launch.
//
//     dst[row] = vec_dot_q8_0_q8_0(W[row, :], Q8_0(act))          ggml-cpu/quants.c:305-333: sum over 32-blocks of sumi * d_w * d_a
//     act      = x  or  rms_norm(x) * w                           ops.cpp:3517-3566, sum of squares in double
//     Q8_0(.)  : the compiled x86 form (arch/x86/quants.c:290-345): d = amax / 127 (stored f16), id = 127 / amax, q = round-half-even(x * id)
//     epilogues: + resid, or silu(gate) * up for the ffn_gate / ffn_up pair
//
// Why: the TTS decoder is 20 layers of 0.6 .. 2.5 MB matrices -- pure launch latency.  Through the round-1 kernels one decoded token is 426
// launches (21 per layer: 7 mat-vecs, 4 activation quantisers, 2 norms, 2 ropes, 2 stores, 2 adds, GLU, attention); with this kernel and the
// one-token attention kernel it is 5 per layer.
//
// Layout: 16 waves per workgroup; every workgroup builds the Q8_0 image of the row in LDS ([qs : K int8][d : K / 32 f32]); a wave owns
// whole rows (four lanes per 34-byte block, 16 blocks per step, hardware-unaligned 8-byte loads like k_mmv_q80); ALL weight loads of a wave
// are requested before the image is built (a row is at most 8 steps: K <= 4096).
#include "../kernels.hpp"

namespace mi {

extern __shared__ __attribute__((aligned(16))) char mv1q_lds[];

struct mv1q_mat { const char * W; size_t w_rs; char * dst; const char * resid; int nrows; int task_end; };   // tasks [prev.task_end, task_end) belong to this matrix
struct mv1q_dev { mv1q_mat m[3]; int nmat; const char * W1; const float * x; const float * nw; const char * img; float eps; int K; int ntasks; };

static __device__ __forceinline__ float mv1q_silu(float x) { return x / (1.0f + expf(-x)); }    // ggml_silu_f32, vec.h:958
template <int CTRL>
static __device__ __forceinline__ float dpp_row_f32(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)); }

// NIT: steps per row the instance is unrolled for (nit = ceil(K / 32 / 16) <= NIT); XE: image elements per thread (K <= XE * 1024)
template <int NIT, int XE, bool PAIR>
__global__ void __launch_bounds__(1024) k_mv1q(const mv1q_dev a) {
    typedef u32x2 __attribute__((aligned(2))) u32x2a2;
    constexpr int NW = 16, R = PAIR ? 2 : 1;
    __shared__ double red[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 2, lp = lane & 3;
    const int K = a.K, nb = K >> 5, nit = (nb + 15) >> 4;
    const int task = __builtin_amdgcn_readfirstlane((int) (blockIdx.x * NW + wave));
    const bool have = task < a.ntasks;

    // ---- 1. the activation row (and norm weights) first, then every weight load of this wave's row(s)
    float xv[XE], wv[XE];
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *) a.x, (short) 0, a.img ? 0 : K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *) a.nw, (short) 0, a.nw ? K * 4 : 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < XE; ++c) xv[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, (uint32_t) (threadIdx.x + 1024 * c) * 4u, 0, 0));
#pragma unroll
    for (int c = 0; c < XE; ++c) wv[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wr, (uint32_t) (threadIdx.x + 1024 * c) * 4u, 0, 0));

    int mi_ = 0, t0 = 0;
    if (!PAIR) {
#pragma unroll
        for (int i = 0; i < 2; ++i) if (i + 1 < a.nmat && task >= a.m[i].task_end) { mi_ = i + 1; t0 = a.m[i].task_end; }
    }
    const mv1q_mat M = mi_ == 0 ? a.m[0] : (mi_ == 1 ? a.m[1] : a.m[2]);
    const int row = task - t0;
    u32x2 q[NIT][R]; uint32_t dw[NIT][R];
    if (have) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int ib = it * 16 + g; ib = ib < nb ? ib : nb - 1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const char * bp = ((PAIR && r == 1) ? a.W1 : M.W) + (size_t) row * M.w_rs + (size_t) ib * 34;
                dw[it][r] = *(const uint16_t *) bp;
                q[it][r]  = *(const u32x2a2 *) (bp + 2 + 8 * lp);
            }
        }
    }

    // ---- 2. the Q8_0 image of the row in LDS
    char * im = mv1q_lds;
    if (a.img) {                                          // ready-made image (common.hpp layout: [qs : K][d : K / 32 f32])
        for (int i = threadIdx.x; i < (K + nb * 4) / 4; i += 1024) ((uint32_t *) im)[i] = ((const uint32_t *) a.img)[i];
    } else {
        float scale = 1.0f;
        if (a.nw) {
            double ss = 0.0;
#pragma unroll
            for (int c = 0; c < XE; ++c) ss += (double) (xv[c] * xv[c]);                  // (elements past K read as zero)
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            if (lane == 0) red[wave] = ss;
            __syncthreads();
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += red[w];
            scale = 1.0f / sqrtf((float) (tot / (double) K) + a.eps);
        }
#pragma unroll
        for (int c = 0; c < XE; ++c) {
            const int e = threadIdx.x + 1024 * c;
            const float v = a.nw ? (xv[c] * scale) * wv[c] : xv[c];
            float amax = fabsf(v);
            amax = fmaxf(amax, dpp_row_f32<0xB1>(amax)); amax = fmaxf(amax, dpp_row_f32<0x4E>(amax));
            amax = fmaxf(amax, dpp_row_f32<0x141>(amax)); amax = fmaxf(amax, dpp_row_f32<0x140>(amax));      // the 16-lane row
            amax = fmaxf(amax, __shfl_xor(amax, 16, 64));                                                  // the 32-element block
            const float d  = amax / 127.0f;
            const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
            if (e < K) {
                ((int8_t *) im)[e] = (int8_t) (int) __builtin_rintf(v * id);
                if ((lane & 31) == 0) ((float *) (im + K))[e >> 5] = h2f(f2h(d));
            }
        }
    }
    __syncthreads();
    if (!have) return;

    // ---- 3. dot products from registers
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (it >= nit) break;
        const int  ib = it * 16 + g;
        const bool valid = ib < nb;
        const int  ibc = valid ? ib : nb - 1;
        const u32x2 av = *(const u32x2 *) (im + ibc * 32 + 8 * lp);
        const float yd = *(const float *) (im + K + ibc * 4);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float t = (float) dot4(q[it][r][0], av[0], dot4(q[it][r][1], av[1], 0)) * (h2f((uint16_t) dw[it][r]) * yd);
            acc[r] += valid ? t : 0.0f;
        }
    }
    if (PAIR) {
        const float gs = wave_sum_f32(acc[0]), us = wave_sum_f32(acc[R - 1]);
        if (lane == 0) *(float *) (M.dst + (size_t) row * 4) = mv1q_silu(gs) * us;
    } else {
        float s = wave_sum_f32(acc[0]);
        if (lane == 0) {
            if (M.resid) s += *(const float *) (M.resid + (size_t) row * 4);
            *(float *) (M.dst + (size_t) row * 4) = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------ F16 weights
// The same launch shape for F16 rows (the TTS module also ships as F16): the activation is rounded to f16 (vec_dot_type of F16 weights,
// ggml_cpu_fp32_to_fp16) and kept as f16 in LDS; lane l of step s multiplies the eight halves (64 s + l) * 8 .. + 7 of the row in f32 FMAs,
// in the order of k_mmv_f (mmvq.hip), so the two kernels give the same bits.  K % 8 == 0, 16-byte aligned rows.
template <int NIT, int XE, bool PAIR>
__global__ void __launch_bounds__(1024) k_mv1h(const mv1q_dev a) {
    constexpr int NW = 16, R = PAIR ? 2 : 1;
    __shared__ double red[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = a.K, np = K >> 3, nit = (np + 63) >> 6;                  // 16-byte pieces per row, steps per row
    const int task = __builtin_amdgcn_readfirstlane((int) (blockIdx.x * NW + wave));
    const bool have = task < a.ntasks;

    float xv[XE], wv[XE];
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *) a.x, (short) 0, a.img ? 0 : K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *) a.nw, (short) 0, a.nw ? K * 4 : 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < XE; ++c) xv[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, (uint32_t) (threadIdx.x + 1024 * c) * 4u, 0, 0));
#pragma unroll
    for (int c = 0; c < XE; ++c) wv[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wr, (uint32_t) (threadIdx.x + 1024 * c) * 4u, 0, 0));

    int mi_ = 0, t0 = 0;
    if (!PAIR) {
#pragma unroll
        for (int i = 0; i < 2; ++i) if (i + 1 < a.nmat && task >= a.m[i].task_end) { mi_ = i + 1; t0 = a.m[i].task_end; }
    }
    const mv1q_mat M = mi_ == 0 ? a.m[0] : (mi_ == 1 ? a.m[1] : a.m[2]);
    const int row = task - t0;
    u32x4 q[NIT][R];
    if (have) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int p = it * 64 + lane; p = p < np ? p : np - 1;
#pragma unroll
            for (int r = 0; r < R; ++r) q[it][r] = *(const u32x4 *) (((PAIR && r == 1) ? a.W1 : M.W) + (size_t) row * M.w_rs + (size_t) p * 16);
        }
    }

    char * im = mv1q_lds;                                   // f16 row [K]
    if (a.img) {
        for (int i = threadIdx.x; i < K / 2; i += 1024) ((uint32_t *) im)[i] = ((const uint32_t *) a.img)[i];
    } else {
        float scale = 1.0f;
        if (a.nw) {
            double ss = 0.0;
#pragma unroll
            for (int c = 0; c < XE; ++c) ss += (double) (xv[c] * xv[c]);
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            if (lane == 0) red[wave] = ss;
            __syncthreads();
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += red[w];
            scale = 1.0f / sqrtf((float) (tot / (double) K) + a.eps);
        }
#pragma unroll
        for (int c = 0; c < XE; ++c) {
            const int e = threadIdx.x + 1024 * c;
            if (e < K) ((uint16_t *) im)[e] = f2h(a.nw ? (xv[c] * scale) * wv[c] : xv[c]);
        }
    }
    __syncthreads();
    if (!have) return;

    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (it >= nit) break;
        const int  p = it * 64 + lane;
        const bool valid = p < np;
        const u32x4 av = *(const u32x4 *) (im + (size_t) (valid ? p : np - 1) * 16);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t = acc[r];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t = fmaf(h2f((uint16_t) (q[it][r][k] & 0xffff)), h2f((uint16_t) (av[k] & 0xffff)), t);
                t = fmaf(h2f((uint16_t) (q[it][r][k] >> 16)), h2f((uint16_t) (av[k] >> 16)), t);
            }
            acc[r] = valid ? t : acc[r];
        }
    }
    if (PAIR) {
        const float gs = wave_sum_f32(acc[0]), us = wave_sum_f32(acc[R - 1]);
        if (lane == 0) *(float *) (M.dst + (size_t) row * 4) = mv1q_silu(gs) * us;
    } else {
        float s = wave_sum_f32(acc[0]);
        if (lane == 0) {
            if (M.resid) s += *(const float *) (M.resid + (size_t) row * 4);
            *(float *) (M.dst + (size_t) row * 4) = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
bool mmv1q_ok(const mv1_args & a) {
    if (a.nmat < 1 || a.nmat > 3 || a.K <= 0 || a.K % 32 != 0 || a.K > 4096) return false;
    if (a.W_up && a.nmat != 1) return false;
    const bool f16 = a.m[0].type == GGML_TYPE_F16;
    int64_t tasks = 0;
    for (int i = 0; i < a.nmat; ++i) {
        const mmv_mat & m = a.m[i];
        if (m.type != (f16 ? GGML_TYPE_F16 : GGML_TYPE_Q8_0) || m.nrows <= 0) return false;
        if (f16 ? (m.w_rs % 16 != 0 || ((uintptr_t) m.W & 15) != 0) : (m.w_rs % 2 != 0 || ((uintptr_t) m.W & 1) != 0)) return false;
        if (((uintptr_t) m.dst & 3) != 0 || ((uintptr_t) m.resid & 3) != 0) return false;
        tasks += m.nrows;
    }
    if (tasks > 65536) return false;                     // (a 150 k-row lm-head streams better through the row-loop kernels: one image build per workgroup there)
    if (a.W_up && (((uintptr_t) a.W_up & (f16 ? 15 : 1)) != 0 || a.m[0].resid)) return false;
    if (a.img) return ((uintptr_t) a.img & 3) == 0;
    return a.x && ((uintptr_t) a.x & 3) == 0 && ((uintptr_t) a.norm_w & 3) == 0;
}

template <int NIT, int XE>
static void mv1q_go(const mv1q_dev & d, bool f16, bool pair, int grid, hipStream_t st) {
    const size_t lds = f16 ? (size_t) d.K * 2 + 16 : (size_t) d.K + (size_t) (d.K / 32) * 4 + 16;
    if (f16) {
        if (pair) k_mv1h<NIT, XE, true><<<dim3(grid), dim3(1024), lds, st>>>(d);
        else      k_mv1h<NIT, XE, false><<<dim3(grid), dim3(1024), lds, st>>>(d);
        return;
    }
    if (pair) k_mv1q<NIT, XE, true><<<dim3(grid), dim3(1024), lds, st>>>(d);
    else      k_mv1q<NIT, XE, false><<<dim3(grid), dim3(1024), lds, st>>>(d);
}

void mmv1q(const mv1_args & a, hipStream_t st) {
    if (!mmv1q_ok(a)) { fprintf(stderr, "[mi355x] mmv1q: unsupported arguments (K=%lld)\n", (long long) a.K); abort(); }
    mv1q_dev d;
    d.nmat = a.nmat; d.K = (int) a.K; d.W1 = (const char *) a.W_up; d.x = a.img ? nullptr : a.x; d.nw = a.img ? nullptr : a.norm_w; d.img = (const char *) a.img; d.eps = a.eps;
    int acc = 0;
    for (int i = 0; i < 3; ++i) {
        const mmv_mat & m = a.m[i < a.nmat ? i : 0];
        if (i < a.nmat) acc += (int) m.nrows;
        d.m[i] = { (const char *) m.W, m.w_rs, (char *) m.dst, (const char *) m.resid, (int) m.nrows, acc };
    }
    d.ntasks = acc;
    const int grid = (acc + 15) / 16;
    const bool f16 = a.m[0].type == GGML_TYPE_F16;
    const int nit = f16 ? (int) ((a.K / 8 + 63) / 64) : (int) ((a.K / 32 + 15) / 16), xe = (int) ((a.K + 1023) / 1024);      // (both: one step per 512 elements)
    const bool pair = a.W_up != nullptr;
    if      (nit <= 2 && xe <= 1) mv1q_go<2, 1>(d, f16, pair, grid, st);
    else if (nit <= 4 && xe <= 2) mv1q_go<4, 2>(d, f16, pair, grid, st);
    else if (nit <= 6 && xe <= 3) mv1q_go<6, 3>(d, f16, pair, grid, st);
    else                          mv1q_go<8, 4>(d, f16, pair, grid, st);
}

} // namespace mi

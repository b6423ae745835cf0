// mmq.hip -- MUL_MAT of K-quant weights against a FEW activation columns (2 .. 64 tokens: several sequences decoded together,
// speculative drafts, small ubatches) on the int8 matrix cores of gfx950.
//
// reference arithmetic: ggml_vec_dot_q4_K_q8_K / ggml_vec_dot_q6_K_q8_K (ggml-cpu/quants.c:550-623, 705-758) on activations quantised by
// quantize_row_q8_K (ggml-quants.c:2555-2592): per 256-block
//     Q4_K:  d*yd * sum_j sc_j * (q4 . q8)_j  -  dmin*yd * sum_j m_j * bsum_j          (8 sub-blocks of 32)
//     Q6_K:  d*yd * sum_s sc_s * ((q6 - 32) . q8)_s                                    (16 sub-blocks of 16)
// The integer sums are exact (the same integers as the reference); only the f32 accumulation across blocks is re-associated, exactly as
// in the mat-vec kernels (mmvk.hip).  The dot4 mat-vec spends one VALU op per 4 weights PER COLUMN, so at 8+ columns it is
// compute-bound long before HBM; here the sub-block dot products of 32 weight rows x 32 tokens are ONE MFMA:
//     tokens are the M side (A operand, int8 activations from the Q8_K image), weight rows the N side (B operand, unpacked nibbles),
//     so a lane owns ONE weight row (lane % 32) and 16 tokens (its accumulator registers): the per-(row, sub-block) scale is a
//     lane-uniform multiplier (v_mad_i32_i24 on the int32 tile) and the per-(token, block) scale is applied once per block.
//     Q4_K mins:  sum_j m_j * bsum_j == sum_k m_{j(k)} * q8_k  -- a second MFMA per sub-block whose B operand is m_j replicated,
//                 accumulated over the block by the matrix core itself (no bsums needed).
//     Q6_K -32 :  sum_s sc_s * 32 * sum_{k in s} q8_k          -- likewise an MFMA with B = sc_s replicated; the weights go in unsigned.
// One wave = 32 weight rows x the token tile; the KS waves of a workgroup split the K blocks of the SAME rows and fold through LDS, so
// that small matrices (wk: 1024 rows) still put thousands of waves on the chip.  Weights stream straight from HBM into registers, one
// block ahead; activations (a few hundred KB, L2-resident) are read per sub-block.
#include "../kernels.hpp"

namespace mi {

typedef int i32x4  __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

struct mmq_mat_dev { const char * W; size_t w_rs; float * dst; size_t dst_cs; const char * resid; size_t resid_cs; int nrows; int type; int tile_end; };
struct mmq_dev {
    mmq_mat_dev m[3]; int nmat;
    const char * act; size_t act_cs;          // Q8_K images (q8k_image_bytes(K) each): [K int8][K/16 int16][K/256 f32]
    int K, ncols;
};

// Q6_K blocks are 210 bytes: only 2-byte aligned.  Global memory takes unaligned vector loads (as the mat-vec kernels' ld_w relies on)
static __device__ __forceinline__ u32x4 ld16u(const char * p) { return *(const u32x4 *) p; }
static __device__ __forceinline__ u32x2 ld8u (const char * p) { return *(const u32x2 *) p; }

template <int NT, int KS>      // NT: token groups of 8 in use (1..4), KS: waves per workgroup splitting K
__global__ void __launch_bounds__(64 * KS) __attribute__((amdgpu_waves_per_eu(2))) k_mmq_kquant(const mmq_dev a) {
    extern __shared__ __attribute__((aligned(16))) char mmq_lds[];
    const int nblk = a.K >> 8;
    float * yd  = (float *) mmq_lds;                       // [nblk][32] token scales of every block (transposed)
    float * red = yd + nblk * 32;                          // [KS - 1][64][NT * 4] fold area

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lq = lane & 31, hb = lane >> 5;
    // ---- which matrix / which 32-row tile
    int tile = (int) blockIdx.x, mi = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) if (i + 1 < a.nmat && (int) blockIdx.x >= a.m[i].tile_end) { mi = i + 1; tile = (int) blockIdx.x - a.m[i].tile_end; }
    const mmq_mat_dev M = mi == 0 ? a.m[0] : (mi == 1 ? a.m[1] : a.m[2]);
    const int row = tile * 32 + lq;
    const bool row_ok = row < M.nrows;
    const char * wrow = M.W + (size_t) (row_ok ? row : M.nrows - 1) * M.w_rs;

    // ---- token scales of all blocks -> LDS, transposed: yd[b][t]  (called after the first block's loads are in flight)
    auto stage_scales = [&]() {
        for (int i = threadIdx.x; i < nblk * 32; i += 64 * KS) {
            const int b = i >> 5, t = i & 31;
            yd[i] = t < a.ncols ? *(const float *) (a.act + (size_t) t * a.act_cs + a.K + (a.K >> 3) + 4 * b) : 0.0f;
        }
        __syncthreads();
    };

    const bool tok_ok = lq < a.ncols;                                    // A-operand role: lane = token lq
    const char * arow = a.act + (size_t) (tok_ok ? lq : 0) * a.act_cs + 16 * hb;

    float out[NT * 4];
#pragma unroll
    for (int i = 0; i < NT * 4; ++i) out[i] = 0.0f;

    if (M.type == GGML_TYPE_Q4_K || M.type == GGML_TYPE_Q5_K) {
        // Q5_K (176-byte blocks: d, dmin, scales[12], qh[32], qs[128]): the same sub-block structure, bit j of qh[l] is the fifth bit of
        // sub-block j's weight l -- OR-ed into the unpacked nibbles, everything else as Q4_K
        const bool q5 = M.type == GGML_TYPE_Q5_K;
        const int bs = q5 ? 176 : 144, qoff = q5 ? 48 : 16;
        struct wblk { u32x4 hdr, qs[4], qh; };                         // one block of this lane's row (its half of the nibbles)
        struct ablk { u32x4 av[8]; };                                  // the token's int8 of one block (this lane's 16 of every 32)
        auto fetch = [&](int b, wblk & G) {
            const char * p = wrow + (size_t) b * bs;
            G.hdr = *(const u32x4 *) p;
#pragma unroll
            for (int q = 0; q < 4; ++q) G.qs[q] = *(const u32x4 *) (p + qoff + q * 32 + 16 * hb);
            if (q5) G.qh = *(const u32x4 *) (p + 16 + 16 * hb);
        };
        auto fetch_a = [&](int b, ablk & G) {
            const char * ab = arow + (size_t) b * 256;
#pragma unroll
            for (int j = 0; j < 8; ++j) G.av[j] = *(const u32x4 *) (ab + j * 32);
        };
        auto reduce = [&](int b, const wblk & G, const ablk & GA) {
            // scales / mins of the 8 sub-blocks (get_scale_min_k4, ggml-quants.c:703-710)
            const uint32_t s0 = G.hdr[1], s1 = G.hdr[2], s2 = G.hdr[3];          // scales[0..3], [4..7], [8..11]
            int sc[8], mn[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = (s0 >> (8 * j)) & 0xff, mid = (s1 >> (8 * j)) & 0xff, hi = (s2 >> (8 * j)) & 0xff;
                sc[j]     = (int) (lo & 63);                 mn[j]     = (int) (mid & 63);
                sc[j + 4] = (int) ((hi & 0xf) | ((lo >> 6) << 4));   mn[j + 4] = (int) ((hi >> 4) | ((mid >> 6) << 4));
            }
            i32x16 acc, mins;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[e] = 0; mins[e] = 0; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32x4 av = GA.av[j];                          // (lanes past ncols carry token 0's bytes: their outputs are never stored)
                const u32x4 wq = G.qs[j >> 1];
                i32x4 wv, mv, aa;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t w5 = (wq[e] >> (4 * (j & 1))) & 0x0f0f0f0fu;
                    if (q5) w5 |= ((G.qh[e] >> j) & 0x01010101u) << 4;
                    wv[e] = (int) w5;
                    mv[e] = mn[j] * 0x01010101;
                    aa[e] = (int) av[e];
                }
                i32x16 z;
#pragma unroll
                for (int e = 0; e < 16; ++e) z[e] = 0;
                const i32x16 sj = __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, wv, z, 0, 0, 0);
                mins = __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, mv, mins, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < NT * 4; ++e) acc[e] += __mul24(sj[e], sc[j]);
            }
            const float d = h2f((uint16_t) (G.hdr[0] & 0xffff)), dmin = h2f((uint16_t) (G.hdr[0] >> 16));
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                const f32x4 y4 = *(const f32x4 *) (yd + b * 32 + 8 * g + 4 * hb);
#pragma unroll
                for (int i = 0; i < 4; ++i) out[4 * g + i] += y4[i] * (d * (float) acc[4 * g + i] - dmin * (float) mins[4 * g + i]);
            }
        };
        // weights ring: RD - 1 blocks of this lane's row in flight ahead of the one being reduced (HBM latency ~2 us, a block's
        // arithmetic well under 1 us); the tokens' int8 come from L2 at use
        constexpr int RD = 4;
        wblk R[RD]; ablk GA;
        const int nb_w = wave < nblk ? (nblk - wave + KS - 1) / KS : 0;         // this wave's blocks: wave, wave + KS, ...
#pragma unroll
        for (int u = 0; u < RD - 1; ++u) if (u < nb_w) fetch(wave + u * KS, R[u]);
        stage_scales();
        for (int i0 = 0; i0 < nb_w; i0 += RD) {
#pragma unroll
            for (int u = 0; u < RD; ++u) {
                const int i = i0 + u;
                if (i < nb_w) {
                    if (i + RD - 1 < nb_w) fetch(wave + (i + RD - 1) * KS, R[(u + RD - 1) % RD]);
                    fetch_a(wave + i * KS, GA);
                    reduce(wave + i * KS, R[u], GA);
                }
            }
        }
    } else {                                                             // GGML_TYPE_Q6_K
        struct wblk { u32x2 ql[8], qh[4]; u32x4 sc; uint32_t d; };
        struct ablk { u32x2 av[16]; };
        auto fetch = [&](int b, wblk & G) {
            const char * p = wrow + (size_t) b * 210;
            // ql chunk c = n*4 + par*2 + is -> bytes n*64 + par*32 + is*16 + 8*hb ; qh chunk c = n*2 + is -> bytes 128 + n*32 + is*16 + 8*hb
#pragma unroll
            for (int c = 0; c < 8; ++c) G.ql[c] = ld8u(p + (c >> 2) * 64 + ((c >> 1) & 1) * 32 + (c & 1) * 16 + 8 * hb);
#pragma unroll
            for (int c = 0; c < 4; ++c) G.qh[c] = ld8u(p + 128 + (c >> 1) * 32 + (c & 1) * 16 + 8 * hb);
            G.sc = ld16u(p + 192);
            G.d  = (uint32_t) *(const uint16_t *) (p + 208);
        };
        auto fetch_a = [&](int b, ablk & G) {
            const char * ab = a.act + (size_t) (tok_ok ? lq : 0) * a.act_cs + (size_t) b * 256 + 8 * hb;
#pragma unroll
            for (int t = 0; t < 16; ++t) G.av[t] = *(const u32x2 *) (ab + t * 16);
        };
        auto reduce = [&](int b, const wblk & G, const ablk & GA) {
            i32x16 acc, corr;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[e] = 0; corr[e] = 0; }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                // sub-block s: half n = s/8, value group q = (s%8)/2 (ql nibble / qh bit pair), is = s%2  (dequantize_row_q6_K, ggml-quants.c:1762-1791)
                const int n = s >> 3, q = (s & 7) >> 1, is = s & 1;
                const u32x2 av = GA.av[s];
                const u32x2 l = G.ql[n * 4 + (q & 1) * 2 + is], h = G.qh[n * 2 + is];
                const int scs = (int) (int8_t) ((G.sc[s >> 2] >> (8 * (s & 3))) & 0xff);
                union { u32x2 u; long l; } wv, sv, aa;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    wv.u[e] = ((l[e] >> (4 * (q >> 1))) & 0x0f0f0f0fu) | (((h[e] >> (2 * q)) & 0x03030303u) << 4);
                    sv.u[e] = (uint32_t) (scs & 0xff) * 0x01010101u;
                }
                aa.u = av;
                i32x16 z;
#pragma unroll
                for (int e = 0; e < 16; ++e) z[e] = 0;
                const i32x16 sj = __builtin_amdgcn_mfma_i32_32x32x16_i8(aa.l, wv.l, z, 0, 0, 0);
                corr = __builtin_amdgcn_mfma_i32_32x32x16_i8(aa.l, sv.l, corr, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < NT * 4; ++e) acc[e] += __mul24(sj[e], scs);
            }
            const float d = h2f((uint16_t) G.d);
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                const f32x4 y4 = *(const f32x4 *) (yd + b * 32 + 8 * g + 4 * hb);
#pragma unroll
                for (int i = 0; i < 4; ++i) out[4 * g + i] += y4[i] * (d * (float) (acc[4 * g + i] - 32 * corr[4 * g + i]));
            }
        };
        // weights ring: RD - 1 blocks of this lane's row in flight ahead of the one being reduced (HBM latency ~2 us, a block's
        // arithmetic well under 1 us); the tokens' int8 come from L2 at use
        constexpr int RD = 3;
        wblk R[RD]; ablk GA;
        const int nb_w = wave < nblk ? (nblk - wave + KS - 1) / KS : 0;         // this wave's blocks: wave, wave + KS, ...
#pragma unroll
        for (int u = 0; u < RD - 1; ++u) if (u < nb_w) fetch(wave + u * KS, R[u]);
        stage_scales();
        for (int i0 = 0; i0 < nb_w; i0 += RD) {
#pragma unroll
            for (int u = 0; u < RD; ++u) {
                const int i = i0 + u;
                if (i < nb_w) {
                    if (i + RD - 1 < nb_w) fetch(wave + (i + RD - 1) * KS, R[(u + RD - 1) % RD]);
                    fetch_a(wave + i * KS, GA);
                    reduce(wave + i * KS, R[u], GA);
                }
            }
        }
    }

    // ---- fold the KS waves' partial sums (same lane layout), store: lane = row, registers = tokens (reg&3) + 8*(reg>>2) + 4*hb
    if (KS > 1) {
        if (wave > 0) {
            float * mine = red + ((size_t) (wave - 1) * 64 + lane) * (NT * 4);
#pragma unroll
            for (int g = 0; g < NT; ++g) *(f32x4 *) (mine + 4 * g) = f32x4{ out[4 * g], out[4 * g + 1], out[4 * g + 2], out[4 * g + 3] };
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 1; w < KS; ++w) {
            const float * oth = red + ((size_t) (w - 1) * 64 + lane) * (NT * 4);
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                const f32x4 o4 = *(const f32x4 *) (oth + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) out[4 * g + i] += o4[i];
            }
        }
    }
    if (!row_ok) return;
#pragma unroll
    for (int g = 0; g < NT; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = 8 * g + 4 * hb + i;
            if (t >= a.ncols) continue;
            float v = out[4 * g + i];
            if (M.resid) v += *(const float *) (M.resid + (size_t) t * M.resid_cs + (size_t) row * 4);        // residual ADD epilogue
            *(float *) ((char *) M.dst + (size_t) t * M.dst_cs + (size_t) row * 4) = v;
        }
}

bool mmq_ok(int type, int64_t K, const void * W, size_t w_rs) {
    if (type == GGML_TYPE_Q4_K || type == GGML_TYPE_Q5_K) return K % 256 == 0 && w_rs % 16 == 0 && ((uintptr_t) W & 15) == 0;
    if (type == GGML_TYPE_Q6_K) return K % 256 == 0 && w_rs % 2 == 0 && ((uintptr_t) W & 1) == 0;
    return false;
}

template <int NT, int KS>
static void mmq_launch(const mmq_dev & d, int ntiles, hipStream_t st) {
    const size_t lds = (size_t) (d.K >> 8) * 32 * 4 + (size_t) (KS > 1 ? (KS - 1) * 64 * NT * 4 * 4 : 0);
    k_mmq_kquant<NT, KS><<<dim3((unsigned) ntiles), dim3(64 * KS), lds, st>>>(d);
}

// up to 3 matrices sharing the activation images, at most 32 columns per call (the caller walks wider batches in chunks)
void mmq_kquant(const mmq_args & a, hipStream_t st) {
    if (a.ncols < 1 || a.ncols > 32 || a.nmat < 1 || a.nmat > 3) { fprintf(stderr, "[mi355x] mmq_kquant: bad shape\n"); abort(); }
    mmq_dev d;
    d.nmat = a.nmat; d.act = (const char *) a.act; d.act_cs = a.act_cs; d.K = (int) a.K; d.ncols = a.ncols;
    int acc = 0;
    for (int i = 0; i < 3; ++i) {
        const mmq_mat & s = a.m[i < a.nmat ? i : 0];
        d.m[i] = { (const char *) s.W, s.w_rs, s.dst, s.dst_cs, (const char *) s.resid, s.resid_cs, (int) s.nrows, s.type, 0 };
        if (i < a.nmat) acc += (int) ((s.nrows + 31) / 32);
        d.m[i].tile_end = acc;
    }
    const int nblk = (int) (a.K >> 8);
    // waves per 32-row tile: towards the chip's 2048 wave slots (two per SIMD) in one round, at least 4 K blocks per wave
    int ks = 1;
    while (ks < 8 && acc * ks * 2 <= 3072 && ks * 4 <= nblk) ks *= 2;
    const int nt = (a.ncols + 7) / 8;
#define MMQ_GO(NTT) do { if (ks == 8) mmq_launch<NTT, 8>(d, acc, st); else if (ks == 4) mmq_launch<NTT, 4>(d, acc, st); else if (ks == 2) mmq_launch<NTT, 2>(d, acc, st); else mmq_launch<NTT, 1>(d, acc, st); } while (0)
    switch (nt) { case 1: MMQ_GO(1); break; case 2: MMQ_GO(2); break; case 3: MMQ_GO(3); break; default: MMQ_GO(4); break; }
#undef MMQ_GO
}

} // namespace mi

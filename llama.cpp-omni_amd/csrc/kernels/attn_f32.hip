// attn_f32.hip -- attention spelled as separate f32 nodes (MUL_MAT K.Q -> SCALE -> SOFT_MAX -> MUL_MAT V^T.P -> PERMUTE + CONT) over a few hundred keys: the
// reference's Token2Wav flow-matching DiT (token2wav-impl.cpp:406-439: head size 64, 8 heads x batch 2, a streaming window's 50..56 frames against 200..206
// cached + new keys), 160 times per window, five launches of 2.5 .. 6 us each.  One launch here: a workgroup owns 16 queries of one (head, batch element);
//   1. S = (Q . K^T) on v_mfma_f32_16x16x4f32, the four waves taking 16-key tiles in turn; operands straight from global memory as 16-byte quads of the
//      K-contiguous rows (lane (row, g) takes the quad at d = 16 j + 4 g for MFMAs 4 j .. 4 j + 3 -- both operands alike), the SCALE node's x * s (+ b) and the
//      soft-max's own scale applied as the separate nodes round them; S to LDS
//   2. soft-max over each row in LDS (16 lanes per row): max, expf(x - max), sum, * 1 / sum -- the arithmetic of k_soft_max_rows (ggml_vec_soft_max_f32)
//   3. O = P . V on the same MFMA, one 16-wide slice of the head per wave, P from LDS, the rows of V^T from global; written through the CONT's strides
// f32 products and accumulation throughout, i.e. the separate nodes' arithmetic with another summation order.
#include "../kernels.hpp"

namespace mi {

struct attn_f32_dev {
    const char * q, * k, * vt; char * dst;
    size_t q_rs, q_bs, k_rs, k_bs, v_rs, v_bs;          // row / head-batch strides (bytes)
    size_t q_bs2, k_bs2; int q_H, k_H;                  // Q / K read where a permuted view leaves them: head-batch index hb = h + H s at h * bs + s * bs2 (H = HB: one level)
    size_t v_bs2, v_ks; int v_H;                        // v_ks != 0: V itself instead of V^T -- element (key k, dim d) at k * v_ks + d * 4, head-batch offsets like Q / K
    size_t d_nb_q, d_nb_h, d_nb_s;                      // dst: element (d, q, h, s) at d * 4 + q * d_nb_q + h * d_nb_h + s * d_nb_s
    int nq, nkv, H, ldp;                                // H: heads per batch element of the destination's split of the head-batch index
    float s1, b1, s2;
    int has_scale;
    int xcd_map;                                        // head-batch count % 8 == 0: heads are dealt to XCDs (see the kernel)
};

extern __shared__ float af_lds[];

template <int D, bool V4>
__global__ void __launch_bounds__(256) k_attn_f32(const attn_f32_dev a) {
    typedef float acc4 __attribute__((ext_vector_type(4)));
    float * S = af_lds;                                  // [16][ldp]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r16 = lane & 15, gq = lane >> 4;
    // workgroup -> (query tile, head-batch): dispatch order (x fastest) deals consecutive workgroups round-robin over the 8 XCDs, each with an L2 of its own.  With the plain
    // (blockIdx.x, blockIdx.y) reading every XCD sees EVERY head's K and V^T (SigLip2: 16 heads x 590 KB = 9.4 MB through each 4 MB L2: FETCH_SIZE 80 MB per launch); when the
    // head-batch count is a multiple of 8 the workgroups an XCD receives (linear id mod 8) are given heads xcd, xcd + 8, ... -- two heads, 1.2 MB per L2
    int qt = (int) blockIdx.x, hb = (int) blockIdx.y;
    if (a.xcd_map) {
        const int L = (int) blockIdx.x + (int) gridDim.x * (int) blockIdx.y, xcd = L & 7, idx = L >> 3;
        hb = xcd + 8 * (idx / (int) gridDim.x); qt = idx % (int) gridDim.x;
    }
    const int q0 = qt * 16;
    const char * Q = a.q + (size_t) (hb % a.q_H) * a.q_bs + (size_t) (hb / a.q_H) * a.q_bs2, * K = a.k + (size_t) (hb % a.k_H) * a.k_bs + (size_t) (hb / a.k_H) * a.k_bs2,
               * VT = a.vt + (size_t) (hb % a.v_H) * a.v_bs + (size_t) (hb / a.v_H) * a.v_bs2;
    // ---- 1. scores
    {
        const int qr = q0 + r16 < a.nq ? q0 + r16 : a.nq - 1;
        constexpr int NJ = (D + 15) / 16;                 // (D % 4 == 0: a quad of the last, partial 16-wide step is inside or outside the head as a whole; outside -> zeros)
        float4 qv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int d = 16 * j + 4 * gq;
            qv[j] = *(const float4 *) (Q + (size_t) qr * a.q_rs + (size_t) (d < D ? d : 0) * 4);
            if (d >= D) qv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // four 16-key tiles of this wave at a time, every operand quad requested before the first MFMA (the loads are the latency here, not the arithmetic); head sizes up to
        // 80 keep TWO such batches in registers: the next batch's quads are requested before the current batch's MFMAs (round 6: the SigLip2 shape, 1024 keys, waited ~1 us per
        // batch: 118 -> 110.7 us per launch, same sums in the same order).  Measured and NOT kept: a key-quarter form of step 3 for head sizes of five slices (every wave all
        // slices over a quarter of the keys, partial outputs added through LDS): 130 us -- the launch is bound by its 8 waves per CU waiting on operand quads straight from
        // global memory, not by wave 0's second slice; what it needs is K / V tiles staged through LDS for all four waves (tools/attn_f32_bench.py)
        auto load_batch = [&](int kb, float4 (&kv)[4][NJ]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k0 = kb + 64 * u;
                const int kr = k0 + r16 < a.nkv ? k0 + r16 : a.nkv - 1;
#pragma unroll
                for (int j = 0; j < NJ; ++j) { const int d = 16 * j + 4 * gq; kv[u][j] = *(const float4 *) (K + (size_t) kr * a.k_rs + (size_t) (d < D ? d : 0) * 4); }      // (Q's zeros cover the quads past D)
            }
        };
        auto mma_batch = [&](int kb, const float4 (&kv)[4][NJ]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k0 = kb + 64 * u;
                if (k0 >= a.nkv) break;
                acc4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[j].x, kv[u][j].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[j].y, kv[u][j].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[j].z, kv[u][j].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[j].w, kv[u][j].w, acc, 0, 0, 0);
                }
                // acc[e] = S[query 4 gq + e][key k0 + r16]
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[e];
                    if (a.has_scale) v = __fadd_rn(__fmul_rn(v, a.s1), a.b1);
                    v = __fmul_rn(v, a.s2);
                    if (k0 + r16 < a.nkv) S[(4 * gq + e) * a.ldp + k0 + r16] = v;
                }
            }
        };
        if constexpr (NJ <= 5) {
            float4 kva[4][NJ], kvb[4][NJ];
            int kb = wave * 16;
            if (kb < a.nkv) load_batch(kb, kva);
            for (; kb < a.nkv; kb += 512) {
                if (kb + 256 < a.nkv) load_batch(kb + 256, kvb);
                mma_batch(kb, kva);
                if (kb + 256 >= a.nkv) break;
                if (kb + 512 < a.nkv) load_batch(kb + 512, kva);
                mma_batch(kb + 256, kvb);
            }
        } else {
            for (int kb = wave * 16; kb < a.nkv; kb += 256) {
                float4 kv[4][NJ];
                load_batch(kb, kv);
                mma_batch(kb, kv);
            }
        }
    }
    __syncthreads();
    // ---- 2. soft-max: row t / 16, sixteen lanes per row
    {
        float * row = S + (t >> 4) * a.ldp;
        const int l = t & 15;
        float m = -INFINITY;
        for (int k = l; k < a.nkv; k += 16) m = fmaxf(m, row[k]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float sum = 0.0f;
        for (int k = l; k < a.nkv; k += 16) { const float e = expf(row[k] - m); row[k] = e; sum += e; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = 1.0f / sum;
        for (int k = l; k < a.nkv; k += 16) row[k] *= inv;
        const int kpad = (a.nkv + 15) & ~15;             // the tail of the last 16-key step reads zeros
        for (int k = a.nkv + l; k < kpad; k += 16) row[k] = 0.0f;
    }
    __syncthreads();
    // ---- 3. O = P . V: wave w the head slice d0 = 16 w .. (D = 64: one slice per wave)
    for (int d0 = wave * 16; d0 < D; d0 += 64) {
        acc4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
        const bool d_ok = d0 + r16 < D;                  // (head sizes that are not multiples of 16: the last slice's spare rows re-read row D - 1 and are not stored)
        const char * vrow = VT + (size_t) (d_ok ? d0 + r16 : D - 1) * a.v_rs;
        for (int kb = 0; kb < a.nkv; kb += 128) {        // eight 16-key steps at a time, the quads of V^T requested up front
            float4 vq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = kb + 16 * u + 4 * gq;
                float4 v;
                if (V4) {                                // nkv % 4 == 0 and 16-byte aligned rows: a quad is inside or outside as a whole
                    v = *(const float4 *) (vrow + (size_t) (k < a.nkv ? k : a.nkv - 4) * 4);
                    if (k >= a.nkv) v = make_float4(0.f, 0.f, 0.f, 0.f);
                } else if (a.v_ks) {                     // V as the graph has it ([D, H, n, B] permuted): four keys of this lane's dim, one row apart (16 lanes = 64 consecutive bytes of a row)
                    const char * vb = VT + (size_t) (d_ok ? d0 + r16 : D - 1) * 4;
                    v.x = k + 0 < a.nkv ? *(const float *) (vb + (size_t) (k + 0) * a.v_ks) : 0.f; v.y = k + 1 < a.nkv ? *(const float *) (vb + (size_t) (k + 1) * a.v_ks) : 0.f;
                    v.z = k + 2 < a.nkv ? *(const float *) (vb + (size_t) (k + 2) * a.v_ks) : 0.f; v.w = k + 3 < a.nkv ? *(const float *) (vb + (size_t) (k + 3) * a.v_ks) : 0.f;
                } else {                                 // (206 keys: rows of 824 bytes)
                    const float * vr = (const float *) vrow;
                    v.x = vr[k + 0 < a.nkv ? k + 0 : 0]; v.y = vr[k + 1 < a.nkv ? k + 1 : 0]; v.z = vr[k + 2 < a.nkv ? k + 2 : 0]; v.w = vr[k + 3 < a.nkv ? k + 3 : 0];
                    if (k + 0 >= a.nkv) v.x = 0.f; if (k + 1 >= a.nkv) v.y = 0.f; if (k + 2 >= a.nkv) v.z = 0.f; if (k + 3 >= a.nkv) v.w = 0.f;
                }
                vq[u] = v;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k0 = kb + 16 * u;
                if (k0 >= a.nkv) break;
                const float4 p = *(const float4 *) &S[r16 * a.ldp + k0 + 4 * gq];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p.x, vq[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p.y, vq[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p.z, vq[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p.w, vq[u].w, acc, 0, 0, 0);
            }
        }
        // acc[e] = O[query 4 gq + e][d0 + r16]
        const int h = hb % a.H, sidx = hb / a.H;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = q0 + 4 * gq + e;
            if (q < a.nq && d_ok) *(float *) (a.dst + (size_t) (d0 + r16) * 4 + (size_t) q * a.d_nb_q + (size_t) h * a.d_nb_h + (size_t) sidx * a.d_nb_s) = acc[e];
        }
    }
}

// dynamic LDS of a launch: the 16 score rows of a workgroup, (nkv rounded up to 64) + 4 floats each.  A gfx950 workgroup owns at most 160 KiB, so the chain is taken
// up to 2496 keys (refused beyond: the separate nodes run -- exec_attn_f32 falls back); 1 KiB is left for the kernel's static arrays
static size_t attn_f32_lds_bytes(int64_t nkv) { return (size_t) 16 * (size_t) (((nkv + 63) / 64) * 64 + 4) * 4; }
bool attn_f32_ok(const attn_f32_args & a) {
    static const bool off = getenv("MI355X_NO_ATTN_F32") != nullptr;
    if (off || (a.D != 64 && a.D != 72 && a.D != 80 && a.D != 96 && a.D != 128) || a.nq < 1 || a.nkv < 1 || attn_f32_lds_bytes(a.nkv) > 160 * 1024 - 1024 || a.HB < 1 || a.HB > 65535 || a.H < 1) return false;
    if ((((uintptr_t) a.q | a.q_rs | a.q_bs | (uintptr_t) a.k | a.k_rs | a.k_bs) & 15) != 0) return false;
    if ((a.q_H > 0 && ((a.q_bs2 & 15) != 0 || a.HB % a.q_H != 0)) || (a.k_H > 0 && ((a.k_bs2 & 15) != 0 || a.HB % a.k_H != 0))) return false;
    if (a.v_ks && ((a.v_ks & 3) != 0 || a.v_H < 1 || a.HB % a.v_H != 0 || (a.v_bs2 & 3) != 0)) return false;
    return (((uintptr_t) a.dst | (uintptr_t) a.vt | a.v_rs | a.v_bs) & 3) == 0;
}
void attn_f32(const attn_f32_args & a, hipStream_t st) {
    if (!attn_f32_ok(a)) { fprintf(stderr, "[mi355x] attn_f32: unsupported arguments\n"); abort(); }
    attn_f32_dev d;
    d.q = (const char *) a.q; d.k = (const char *) a.k; d.vt = (const char *) a.vt; d.dst = (char *) a.dst;
    d.q_rs = a.q_rs; d.q_bs = a.q_bs; d.k_rs = a.k_rs; d.k_bs = a.k_bs; d.v_rs = a.v_rs; d.v_bs = a.v_bs;
    d.d_nb_q = a.d_nb_q; d.d_nb_h = a.d_nb_h; d.d_nb_s = a.d_nb_s;
    d.q_H = a.q_H > 0 ? (int) a.q_H : (int) a.HB; d.q_bs2 = a.q_H > 0 ? a.q_bs2 : 0; d.k_H = a.k_H > 0 ? (int) a.k_H : (int) a.HB; d.k_bs2 = a.k_H > 0 ? a.k_bs2 : 0;
    d.v_ks = a.v_ks; d.v_H = a.v_ks ? (int) a.v_H : (int) a.HB; d.v_bs2 = a.v_ks ? a.v_bs2 : 0;
    d.nq = (int) a.nq; d.nkv = (int) a.nkv; d.H = (int) a.H; d.ldp = (int) (((a.nkv + 63) / 64) * 64 + 4);
    d.s1 = a.s1; d.b1 = a.b1; d.s2 = a.s2; d.has_scale = a.has_scale ? 1 : 0;
    static const bool no_xcd = getenv("MI355X_ATTN_F32_NO_XCD") != nullptr;
    d.xcd_map = !no_xcd && a.HB % 8 == 0 ? 1 : 0;
    const int lds = 16 * d.ldp * 4;
    const bool v4 = !a.v_ks && a.nkv % 4 == 0 && (((uintptr_t) a.vt | a.v_rs | a.v_bs) & 15) == 0;
    const dim3 grid((unsigned) ((a.nq + 15) / 16), (unsigned) a.HB);
    int dev = 0; HIP_CHECK(hipGetDevice(&dev));
    auto go = [&](auto k4, auto k1, int slot) {
        static int attr_lds[8][64] = {};
        if (lds > 65536 && (dev < 0 || dev >= 64 || attr_lds[slot][dev] < lds)) {
            HIP_CHECK(hipFuncSetAttribute((const void *) k4, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            HIP_CHECK(hipFuncSetAttribute((const void *) k1, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            if (dev >= 0 && dev < 64) attr_lds[slot][dev] = lds;
        }
        if (v4) k4<<<grid, dim3(256), lds, st>>>(d); else k1<<<grid, dim3(256), lds, st>>>(d);
    };
    switch ((int) a.D) {
        case 64:  go(k_attn_f32<64, true>,  k_attn_f32<64, false>,  0); break;
        case 72:  go(k_attn_f32<72, true>,  k_attn_f32<72, false>,  1); break;      // SigLip2: 1152 / 16 heads
        case 80:  go(k_attn_f32<80, true>,  k_attn_f32<80, false>,  2); break;
        case 96:  go(k_attn_f32<96, true>,  k_attn_f32<96, false>,  3); break;
        default:  go(k_attn_f32<128, true>, k_attn_f32<128, false>, 4); break;
    }
}

} // namespace mi

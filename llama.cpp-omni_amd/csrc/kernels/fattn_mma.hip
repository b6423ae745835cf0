// fattn_mma.hip -- FLASH_ATTN_EXT for batches of query rows (prefill) on the CDNA4 matrix cores (gfx950, wave64).
//
// reference semantics: ggml_compute_forward_flash_attn_ext_f16, ggml-cpu/ops.cpp:7912-8148 (Q rounded to f16, s = K.Q * scale
// [softcap] + slope * mask, online softmax, V weighted sum, / S, sinks, output permuted to [DV, n_head, n_q, n_seq]); what the
// reference's GPU backend does for the same node is fattn-mma-f16.cuh -- this kernel shares the maths, not the tiling.
// Accuracy bar: the reference's own NMSE 5e-4 for this op (tests/test-backend-ops.cpp:5085): P is rounded to f16 for the second
// matrix product and exp is the hardware v_exp_f32 -- both far inside that bar (measured ~1e-7).
//
// Tiling: a wave owns 32 query rows of one head; a workgroup is NW waves on consecutive 32-row blocks of the same head, so one
// staged K / V tile (32 KV rows) feeds NW*32 queries.  Everything is computed TRANSPOSED so that a lane owns one query column:
//     S^T[kv, q] = K[kv, :] . Q[q, :]      A = K fragment  (LDS row kv = lane%32, 8 consecutive d),   B = Q fragment (registers)
//     O^T[d, q]  = V^T[d, kv] . P^T[kv, q] A = V^T fragment (LDS, staged transposed),                 B = P fragment
// v_mfma_f32_32x32x16_f16 leaves C[i][j] with j = lane%32 and i = (reg&3) + 8*(reg>>2) + 4*(lane/32): a lane holds 16 scores
// of ITS query, so the softmax row maximum is 15 in-register max + one exchange with lane^32, and the per-query rescale factor is
// lane-uniform.  The S^T accumulator registers, rounded to f16, ARE the B operand of the second product: the contraction index is
// only a label, so K-slot (lane/32, e) of MFMA step s is declared to be kv = (e&3) + 4*(lane/32) + 8*(e>>2) + 16*s -- exactly
// the kv a lane already holds in registers 8s..8s+7 -- and the V^T fragment is gathered to match (two 8-byte LDS reads per step).
// V is transposed while it is staged (pairs of KV rows packed into one 32-bit LDS word).
// KV tiles whose mask is -inf for every query of the workgroup are skipped before any K/V traffic (causal prefill touches half
// of the tiles; a cache view padded past the used cells costs nothing).
#include "fattn_dev.hpp"

namespace mi {

typedef _Float16 h8v  __attribute__((ext_vector_type(8)));
typedef float    f16a __attribute__((ext_vector_type(16)));

constexpr int FM_KT = 32;                       // KV rows per tile

template <int D> struct fm_cfg {
    static constexpr int KLD = D + 8;           // K tile row pitch in halfs (16-byte aligned rows, conflict-free b128 reads)
    static constexpr int VLD = FM_KT + 2;       // V^T row pitch in halfs (17 words: odd, spreads the transposing writes)
};

template <int D, int NW>
__global__ void __launch_bounds__(64 * NW) k_fattn_mma(const fa_dev a, const int nqt) {
    constexpr int KLD = fm_cfg<D>::KLD, VLD = fm_cfg<D>::VLD;
    constexpr int NKS = D / 16;                 // MFMA k-steps of the score product
    constexpr int NDB = D / 32;                 // 32-row blocks of O^T
    __shared__ __attribute__((aligned(16))) _Float16 Ks[FM_KT * KLD];
    __shared__ __attribute__((aligned(16))) uint32_t Vt[D * VLD / 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hb = lane >> 5;
    int b = (int) blockIdx.x;
    const int qt  = nqt - 1 - b % nqt; b /= nqt;          // longest (latest, for causal masks) query tiles first
    const int h   = b % a.nh;  const int is3 = b / a.nh;
    const int ikv = h / a.gq;
    const int q0  = (qt * NW + wave) * 32;                // this wave's first query row
    const int q   = q0 + lq;
    const int qc  = q < a.nq ? q : a.nq - 1;
    const bool wave_has_rows = q0 < a.nq;

    // ---- Q fragments: 8 consecutive d of row q per k-step, f32 -> f16 (q_to_vec_dot, ops.cpp:8040)
    h8v qf[NKS];
    {
        const char * qr = a.q + qc * a.qnb1 + h * a.qnb2 + is3 * a.qnb3;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const f32x4 v0 = *(const f32x4 *) (qr + (ks * 16 + hb * 8) * 4);
            const f32x4 v1 = *(const f32x4 *) (qr + (ks * 16 + hb * 8 + 4) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qf[ks][e] = (_Float16) v0[e]; qf[ks][4 + e] = (_Float16) v1[e]; }
        }
    }
    const uint32_t hu = (uint32_t) h;
    const float slope = a.max_bias > 0.0f ? (hu < a.n_head_log2 ? powf(a.m0, (float) (hu + 1)) : powf(a.m1, (float) (2 * (hu - a.n_head_log2) + 1))) : 1.0f;
    const uint16_t * mrow = a.mask ? (const uint16_t *) (a.mask + qc * a.mnb1 + (h % (int) a.mne2) * a.mnb2 + (is3 % (int) a.mne3) * a.mnb3) : nullptr;
    const bool mask_vec = a.mask && (a.mnb1 % 8 == 0) && (((uintptr_t) mrow) % 8 == 0);

    f16a acc_o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[db][e] = 0.0f;
    float M = -INFINITY, S = 0.0f;

    const char * kbase = a.k + ikv * a.knb2 + is3 * a.knb3;
    const char * vbase = a.v + ikv * a.vnb2 + is3 * a.vnb3;
    const int ntile = (a.nkv + FM_KT - 1) / FM_KT;

    for (int t = 0; t < ntile; ++t) {
        const int kv0 = t * FM_KT;
        // ---- mask values of this lane's query: kv = kv0 + 4*hb + 8*g + {0..3}, g = 0..3  (register e = 4*g + i)
        float mv[16];
        bool live = false;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kvb = kv0 + 4 * hb + 8 * g;
            if (mrow && mask_vec && kvb + 3 < a.nkv) {
                const u32x2 w = *(const u32x2 *) (mrow + kvb);
                mv[4 * g + 0] = h2f((uint16_t) (w[0] & 0xffff)); mv[4 * g + 1] = h2f((uint16_t) (w[0] >> 16));
                mv[4 * g + 2] = h2f((uint16_t) (w[1] & 0xffff)); mv[4 * g + 3] = h2f((uint16_t) (w[1] >> 16));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int kv = kvb + i;
                    mv[4 * g + i] = kv < a.nkv ? (mrow ? h2f(mrow[kv]) : 0.0f) : -INFINITY;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            mv[e] = mv[e] == -INFINITY ? -INFINITY : slope * mv[e];
            live |= mv[e] != -INFINITY;
        }
        live = live && q < a.nq;
        const bool wave_live = __any(live) && wave_has_rows;
        if (!__syncthreads_or(wave_live ? 1 : 0)) continue;      // also: every wave is done with the previous tile's LDS

        // ---- stage K (row-major) and V (transposed, kv pairs packed) ; rows past nkv are zero
        for (int c = tid; c < FM_KT * (D / 8); c += 64 * NW) {
            const int row = c / (D / 8), col = c % (D / 8);
            u32x4 w = { 0u, 0u, 0u, 0u };
            if (kv0 + row < a.nkv) w = *(const u32x4 *) (kbase + (int64_t) (kv0 + row) * a.knb1 + col * 16);
            *(u32x4 *) &Ks[row * KLD + col * 8] = w;
        }
        for (int c = tid; c < (FM_KT / 2) * (D / 8); c += 64 * NW) {
            const int o = c % (D / 8), p = c / (D / 8);
            u32x4 w0 = { 0u, 0u, 0u, 0u }, w1 = { 0u, 0u, 0u, 0u };
            if (kv0 + 2 * p     < a.nkv) w0 = *(const u32x4 *) (vbase + (int64_t) (kv0 + 2 * p)     * a.vnb1 + o * 16);
            if (kv0 + 2 * p + 1 < a.nkv) w1 = *(const u32x4 *) (vbase + (int64_t) (kv0 + 2 * p + 1) * a.vnb1 + o * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Vt[(8 * o + 2 * i)     * (VLD / 2) + p] = (w0[i] & 0xffffu) | (w1[i] << 16);
                Vt[(8 * o + 2 * i + 1) * (VLD / 2) + p] = (w0[i] >> 16)     | (w1[i] & 0xffff0000u);
            }
        }
        __syncthreads();
        if (!wave_live) continue;

        // ---- S^T = K . Q^T
        f16a sc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const h8v kf = *(const h8v *) &Ks[lq * KLD + ks * 16 + hb * 8];
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sc, 0, 0, 0);
        }
        // ---- scale / softcap / mask, online softmax down this lane's query column
        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = sc[e] * a.scale;
            if (a.logit_softcap != 0.0f) v = a.logit_softcap * tanhf(v);
            v = mv[e] == -INFINITY ? -INFINITY : v + mv[e];
            sc[e] = v;
            tmax = fmaxf(tmax, v);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float Mn = fmaxf(M, tmax);
        const float Mu = Mn == -INFINITY ? 0.0f : Mn;
        const float alpha = __expf(M - Mu);                           // M == -inf -> 0
        float psum = 0.0f;
        h8v pf[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float p = __expf(sc[e] - Mu);                       // masked -> 0
            psum += p;
            pf[e >> 3][e & 7] = (_Float16) p;
        }
        S = S * alpha + psum;
        M = Mn;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc_o[db][e] *= alpha;
        }
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const uint32_t * vr = &Vt[(db * 32 + lq) * (VLD / 2) + 8 * s2 + 2 * hb];
                union { uint32_t u[4]; h8v v; } vf;
                vf.u[0] = vr[0]; vf.u[1] = vr[1]; vf.u[2] = vr[4]; vf.u[3] = vr[5];
                acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf.v, pf[s2], acc_o[db], 0, 0, 0);
            }
        }
    }

    // ---- finish: fold the two lane halves' partial sums, sinks (ops.cpp:8116-8130), normalise, store permuted
    S += __shfl_xor(S, 32, 64);
    float osc = 1.0f;
    if (a.sinks) {
        const float sk = a.sinks[h];
        if (sk > M) { const float f = __expf(M - sk); S = S * f + 1.0f; osc = f; }
        else S += __expf(sk - M);
    }
    const float inv = S == 0.0f ? 0.0f : osc / S;
    if (q < a.nq) {
        char * out = a.dst + h * a.dnb1 + q * a.dnb2 + is3 * a.dnb3;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o4;
#pragma unroll
                for (int i = 0; i < 4; ++i) o4[i] = acc_o[db][4 * g + i] * inv;
                *(f32x4 *) (out + (db * 32 + 8 * g + 4 * hb) * 4) = o4;
            }
    }
}

template <int D>
static void launch_fm(const fa_dev & a, hipStream_t st) {
    const int nqt4 = (a.nq + 127) / 128;
    if ((int64_t) nqt4 * a.nh * a.ns >= 256 || a.nq <= 32) {
        if (a.nq <= 32) { const int nqt = 1;  k_fattn_mma<D, 1><<<dim3((unsigned) (nqt * a.nh * a.ns)), dim3(64), 0, st>>>(a, nqt); }
        else            { k_fattn_mma<D, 4><<<dim3((unsigned) (nqt4 * a.nh * a.ns)), dim3(256), 0, st>>>(a, nqt4); }
    } else {
        const int nqt = (a.nq + 63) / 64;
        k_fattn_mma<D, 2><<<dim3((unsigned) (nqt * a.nh * a.ns)), dim3(128), 0, st>>>(a, nqt);
    }
}

void flash_attn_ext_mma(const fa_dev & a, int D, hipStream_t st) {
    if (D == 64) launch_fm<64>(a, st); else launch_fm<128>(a, st);
}

} // namespace mi

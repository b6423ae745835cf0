// fattn_mma.hip -- FLASH_ATTN_EXT for batches of query rows (prefill) on the CDNA4 matrix cores (gfx950, wave64).
//
// reference semantics: ggml_compute_forward_flash_attn_ext_f16, ggml-cpu/ops.cpp:7912-8148 (Q rounded to f16, s = K.Q * scale
// [softcap] + slope * mask, online softmax, V weighted sum, / S, sinks, output permuted to [DV, n_head, n_q, n_seq]); what the
// reference's GPU backend does for the same node is fattn-mma-f16.cuh -- this kernel shares the maths, not the tiling.
// Accuracy bar: the reference's own NMSE 5e-4 for this op (tests/test-backend-ops.cpp:5085): P is rounded to f16 for the second
// matrix product and the exponentials are v_exp_f32 in the base-2 domain -- both far inside that bar (measured ~1e-7).
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
//
// Mask tile map: the mask is shared by every head (and every layer of a graph), so a tiny pre-pass (k_fattn_mask_map) classifies
// each 32 x 32 mask tile once -- 0: -inf everywhere, 1: zero everywhere, 2: mixed -- and the attention kernel never touches the
// mask itself except on mixed (diagonal) tiles.  Dead tiles are skipped without K/V traffic or barriers (causal prefill touches
// half of the tiles; a cache view padded past the used cells costs nothing), all-zero tiles skip the mask arithmetic.
// Pipeline (one barrier per live KV tile): the next live tile's K/V rows are fetched into registers BEFORE the current tile's
// matrix work and written to the other LDS buffer after it.
#include "fattn_dev.hpp"

namespace mi {

typedef _Float16 h8v  __attribute__((ext_vector_type(8)));
typedef _Float16 h2v  __attribute__((ext_vector_type(2)));
typedef float    f16a __attribute__((ext_vector_type(16)));
typedef float    f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void * lds_ptr_t;
typedef const __attribute__((address_space(1))) void * gbl_ptr_t;

constexpr int FM_KT = 32;                       // KV rows per tile
constexpr int FM_MAXT = 4096;                   // live-tile bitmap capacity: nkv <= 131072
constexpr float FM_LOG2E = 1.4426950408889634f;

template <int D> struct fm_cfg {
    static constexpr int KLD = D + 8;           // K tile row pitch in halfs (16-byte aligned rows, conflict-free b128 reads)
    static constexpr int VLD = FM_KT + 2;       // V^T row pitch in halfs (17 words: odd, spreads the transposing writes)
    static constexpr int KB  = FM_KT * KLD * 2; // bytes of one K buffer
    static constexpr int VB  = D * VLD * 2;     // bytes of one V^T buffer
};

// ---- the finished O^T of a wave (lane = query lq, 16 x NDB f32 values = d scattered as db*32 + 8g + 4hb + i) out to memory as whole rows: through a wave-private LDS block
// [32 queries][D + 4] (the +4 keeps the 16-byte writes of eight consecutive queries on distinct banks), read back row-wise, so that one store instruction writes 64 / (D / 4)
// whole rows of D f32 (or D f16 for the image) instead of 64 separate 16-byte pieces a row stride apart -- the per-lane form cost 6.4 us per workgroup at 512 x 512 x 128
// (a CU's memory path takes a request per line touched: 64 per instruction against 8)
template <int D>
__device__ __forceinline__ void fa_store_rows(const f16a (&acc_o)[D / 32], const float inv, float * scr, const int lane, const fa_dev & a, const int h, const int q0, const int is3) {
    constexpr int NDB = D / 32, P = D + 4, LPR = D / 4, RPI = 64 / LPR;
    const int lq = lane & 31, hb = lane >> 5;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o4;
#pragma unroll
            for (int i = 0; i < 4; ++i) o4[i] = acc_o[db][4 * g + i] * inv;
            *(f32x4 *) (scr + lq * P + db * 32 + 8 * g + 4 * hb) = o4;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int rl = lane / LPR, c4 = lane % LPR;
#pragma unroll
    for (int r = 0; r < 32; r += RPI) {
        const int row = r + rl, qrow = q0 + row;
        const f32x4 o4 = *(const f32x4 *) (scr + row * P + c4 * 4);
        if (qrow < a.nq) {
            if (a.write_f32) *(f32x4 *) (a.dst + h * a.dnb1 + qrow * a.dnb2 + is3 * a.dnb3 + c4 * 16) = o4;
            if (a.out16) {
                u32x2 hw;
                hw[0] = (uint32_t) f2h(o4[0]) | ((uint32_t) f2h(o4[1]) << 16); hw[1] = (uint32_t) f2h(o4[2]) | ((uint32_t) f2h(o4[3]) << 16);
                *(u32x2 *) (a.out16 + ((int64_t) is3 * a.nq + qrow) * a.out16_rs + ((int64_t) h * D + c4 * 4) * 2) = hw;
            }
        }
    }
}

// KS = KV split inside the workgroup: NW*KS waves; wave (qb, ks) owns query block qb and the ks-th run of SQ 32-row tiles of every staged group of
// KS * SQ tiles, with its own running (M, S, O); the KS partial states of a query block are merged through LDS at the end.  Used when a
// single wave per 32 queries would leave half of the chip's SIMDs without a wave (one 512-token ubatch of one sequence: 512 blocks).
// SQ = tiles a wave works through, one after the other, between two barriers: the K / V rows of the NEXT group are requested right after the barrier that
// publishes this one and have the whole group's matrix work to arrive -- with one tile per barrier (16 KB per workgroup in flight against a 1-2 us
// round trip) every tile waited for its rows: the kernel ran at the request latency, not at the matrix or vector rate.
template <int D, int NW, int KS, int SQ, int ABL = 0>      // ABL (measurement only, MI355X_FA_ABL): 1 no K / V requests or LDS stores inside the loop (stale tiles), 2 no soft-max arithmetic, 4 no P.V product
__global__ void __launch_bounds__(64 * NW * KS) __attribute__((amdgpu_waves_per_eu(2))) k_fattn_mma(const fa_dev a, const int nqt) {
    constexpr int KLD = fm_cfg<D>::KLD, VLD = fm_cfg<D>::VLD;
    constexpr int NKS = D / 16;                 // MFMA k-steps of the score product
    constexpr int NDB = D / 32;                 // 32-row blocks of O^T
    constexpr int NT  = 64 * NW * KS;
    constexpr int NTL = KS * SQ;                // 32-row tiles per staged group
    constexpr int GT  = FM_KT * NTL;            // KV rows staged per iteration
    constexpr int KCH = (GT * (D / 8) + NT - 1) / NT;               // 16-byte K chunks per thread per group
    constexpr int VIT = ((GT / 2) * (D / 8) + NT - 1) / NT;         // V row-pair items per thread per group
    // one pool: the K and V^T tiles of both buffers, and -- after the loop, when the tiles are dead -- the parked states of the KV split (KS = 4 at head size 128 would
    // not fit next to them: 139 + 101 KB)
    constexpr int KSB = 2 * NTL * FM_KT * KLD * 2, VTB = 2 * NTL * (D * VLD / 2) * 4;
    constexpr int MRGB = KS > 1 ? (KS - 1) * NW * 64 * (NDB * 16 + 2) * 4 : 0;
    constexpr int POOLB = KSB + VTB > MRGB ? KSB + VTB : MRGB;
    constexpr int OWB = 32 * (D + 4) * 4, ORW = POOLB / OWB >= NW ? NW : POOLB / OWB;      // output rows' staging: bytes per wave, waves per round
    static_assert(ORW >= 1 && (KS == 1 || ORW == NW), "the output rows' staging fits the pool");
    __shared__ __attribute__((aligned(16))) char pool[POOLB];
    _Float16 (* const Ks)[NTL][FM_KT * KLD] = (_Float16 (*)[NTL][FM_KT * KLD]) pool;
    uint32_t (* const Vt)[NTL][D * VLD / 2] = (uint32_t (*)[NTL][D * VLD / 2]) (pool + KSB);

    const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6;
    int n_st = 0;
    auto stamp = [&]() { if (a.stamps && blockIdx.x == 0 && tid == 0 && n_st < 30) a.stamps[n_st++] = __builtin_amdgcn_s_memrealtime(); };
    stamp();
    const int wave = wave_all % NW, kvs = wave_all / NW;    // query block within the workgroup, KV split
    const int lq = lane & 31, hb = lane >> 5;
    int b = (int) blockIdx.x;
    const int qt  = nqt - 1 - b % nqt; b /= nqt;          // longest (latest, for causal masks) query tiles first
    const int h   = b % a.nh;  const int is3 = b / a.nh;
    const int ikv = h / a.gq;
    const int q0  = (qt * NW + wave) * 32;                // this wave's first query row
    const int q   = q0 + lq;
    const int qc  = q < a.nq ? q : a.nq - 1;
    const bool row_ok = q < a.nq;

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
    const float slope2 = slope * FM_LOG2E;
    const float c2 = a.logit_softcap != 0.0f ? a.scale : a.scale * FM_LOG2E;    // scores live in the base-2 domain (softcap: after tanh)
    const uint16_t * mrow = a.mask ? (const uint16_t *) (a.mask + qc * a.mnb1 + (h % (int) a.mne2) * a.mnb2 + (is3 % (int) a.mne3) * a.mnb3) : nullptr;
    const bool mask_vec = a.mask && (a.mnb1 % 8 == 0) && (((uintptr_t) mrow) % 8 == 0);

    stamp();                                              // 1: q fragments
    f16a acc_o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[db][e] = 0.0f;
    float M = -INFINITY, S = 0.0f;                        // running max (base-2 domain) and this lane's partial sum

    const char * kbase = a.k + ikv * a.knb2 + is3 * a.knb3;
    const char * vbase = a.v + ikv * a.vnb2 + is3 * a.vnb3;
    const int ntile32 = (a.nkv + FM_KT - 1) / FM_KT;              // 32-row mask / compute tiles
    const int ntile = (ntile32 + NTL - 1) / NTL;                  // staged groups of NTL tiles
    // K / V rows of tile t into registers (rows past nkv are zero)
    // two register sets: the rows of the next TWO live groups are in flight (one group's matrix work is shorter than a round trip to L2 / HBM on the small grids this kernel
    // serves: with one set every group waited ~2 us for its rows)
    u32x4 kregA[KCH], vregA[VIT][2], kregB[KCH], vregB[VIT][2];
    auto load_kv = [&](int t, u32x4 (&kreg)[KCH], u32x4 (&vreg)[VIT][2]) {
        const int kv0 = t * GT;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = tid + i * NT;
            const int row = c / (D / 8), col = c % (D / 8);
            kreg[i] = u32x4{ 0u, 0u, 0u, 0u };
            if (c < GT * (D / 8) && kv0 + row < a.nkv) kreg[i] = *(const u32x4 *) (kbase + (int64_t) (kv0 + row) * a.knb1 + col * 16);
        }
        if (a.vt) {
            // V given TRANSPOSED (the flash-attention-off graphs: a [n_kv, D] view of the transposed cache, or an encoder's permuted V): row d holds its kv cells
            // contiguously -- already the layout of the V^T tile; item = 8 cells (16 bytes) of one row
#pragma unroll
            for (int i = 0; i < 2 * VIT; ++i) {
                const int c = tid + i * NT;
                const int ch = c % (GT / 8), d = c / (GT / 8);
                u32x4 r = u32x4{ 0u, 0u, 0u, 0u };
                if (c < D * (GT / 8)) {
                    const int kvc = kv0 + 8 * ch;
                    const char * src = vbase + (int64_t) d * a.vnb1 + (int64_t) kvc * 2;
                    if (kvc + 7 < a.nkv && a.vt == 16) r = *(const u32x4 *) src;
                    else if (kvc + 7 < a.nkv && a.vt == 4) { const uint32_t * s4 = (const uint32_t *) src; r = u32x4{ s4[0], s4[1], s4[2], s4[3] }; }
                    else {
                        const uint16_t * s2 = (const uint16_t *) src;
                        uint32_t hv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) hv[e] = kvc + e < a.nkv ? (uint32_t) s2[e] : 0u;
                        r = u32x4{ hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16), hv[4] | (hv[5] << 16), hv[6] | (hv[7] << 16) };
                    }
                }
                vreg[i >> 1][i & 1] = r;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < VIT; ++i) {
            const int c = tid + i * NT;
            const int o = c % (D / 8), p = c / (D / 8);
            vreg[i][0] = vreg[i][1] = u32x4{ 0u, 0u, 0u, 0u };
            if (c < (GT / 2) * (D / 8)) {
                if (kv0 + 2 * p     < a.nkv) vreg[i][0] = *(const u32x4 *) (vbase + (int64_t) (kv0 + 2 * p)     * a.vnb1 + o * 16);
                if (kv0 + 2 * p + 1 < a.nkv) vreg[i][1] = *(const u32x4 *) (vbase + (int64_t) (kv0 + 2 * p + 1) * a.vnb1 + o * 16);
            }
        }
    };

    // ---- which tiles does any query of this workgroup see?  The classes of the workgroup's NW x ntile32 tiles are copied out of the mask tile map once, two bits each,
    // into LDS (a class looked up in global memory inside the loop was a dependent L2 round trip per group and wave, in front of the barrier: the loop ran at that
    // latency); one bit per staged group says whether anybody needs it
    __shared__ uint64_t live_bits[FM_MAXT / 64];
    __shared__ uint32_t cls2[NW][FM_MAXT / 16];
    const int qb0 = qt * NW;                                        // first 32-row query block of the workgroup
    const uint8_t * maprow = a.tile_map ? a.tile_map + (((int64_t) (is3 % (int) a.mne3) * a.mne2 + (h % (int) a.mne2)) * a.map_nqb) * ntile32 : nullptr;
    stamp();                                              // 1b: before the classes
    if (maprow) {
        const int nwords = (ntile32 + 15) / 16;
        for (int j = tid; j < NW * nwords; j += NT) {
            const int w = j / nwords, wd = j % nwords;
            uint32_t word = 0;
            if (qb0 + w < a.map_nqb) {
                const uint8_t * r = maprow + (int64_t) (qb0 + w) * ntile32;
                if (wd * 16 + 16 <= ntile32 && (((uintptr_t) (r + wd * 16)) & 15) == 0) {       // sixteen classes in one request (sixteen byte loads in a row were ~2 us of this workgroup's life)
                    const u32x4 v = *(const u32x4 *) (r + wd * 16);
#pragma unroll
                    for (int i = 0; i < 16; ++i) word |= ((v[i >> 2] >> (8 * (i & 3))) & 3u) << (2 * i);
                } else
#pragma unroll
                for (int i = 0; i < 16; ++i) { const int t32 = wd * 16 + i; word |= (t32 < ntile32 ? (uint32_t) r[t32] & 3u : 0u) << (2 * i); }
            }
            cls2[w][wd] = word;
        }
        __syncthreads();
    }
    stamp();                                              // 1c: classes in LDS
    for (int c = wave_all; c * 64 < ntile; c += NW * KS) {
        const int tt = c * 64 + lane;
        bool lv = false;
        if (tt < ntile) {
            if (!maprow) lv = true;
            else
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int k = 0; k < NTL; ++k) {
                        const int t32 = tt * NTL + k;
                        if (t32 < ntile32) lv |= ((cls2[w][t32 >> 4] >> (2 * (t32 & 15))) & 3u) != 0;
                    }
        }
        const uint64_t bits = __ballot(lv);
        if (lane == 0) live_bits[c] = bits;
    }
    __syncthreads();
    auto next_live = [&](int u) {                                    // smallest live tile >= u, or ntile (wave-uniform)
        while (u < ntile) {
            const uint64_t w = live_bits[u >> 6] >> (u & 63);
            if (w) { u += __builtin_ctzll(w); break; }
            u = (u | 63) + 1;
        }
        return __builtin_amdgcn_readfirstlane(u < ntile ? u : ntile);
    };
    auto tile_class = [&](int g, int sq) -> int {                    // this wave's 32 queries x its sq-th tile of group g
        const int t = g * NTL + kvs * SQ + sq;
        if (q0 >= a.nq || t >= ntile32) return 0;
        if (!maprow) return (t + 1) * FM_KT <= a.nkv ? 1 : 2;
        return (int) ((cls2[wave][t >> 4] >> (2 * (t & 15))) & 3u);
    };

    // mask words of this lane's query for tile t: word pair g (0..3) = halfs kv0 + 4*hb + 8*g + {0..3}; cells past nkv read as -inf
    auto load_mask = [&](int t, u32x2 (&w)[4]) {
        const int kv0 = t * FM_KT;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kvb = kv0 + 4 * hb + 8 * g;
            if (kvb + 3 < a.nkv && (!mrow || mask_vec)) {
                if (mrow) w[g] = *(const u32x2 *) (mrow + kvb); else { w[g][0] = 0u; w[g][1] = 0u; }
            } else {
                uint32_t hv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[i] = kvb + i < a.nkv ? (mrow ? (uint32_t) mrow[kvb + i] : 0u) : 0xfc00u;
                w[g][0] = hv[0] | (hv[1] << 16); w[g][1] = hv[2] | (hv[3] << 16);
            }
        }
    };
    auto store_kv = [&](int buf, const u32x4 (&kreg)[KCH], const u32x4 (&vreg)[VIT][2]) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = tid + i * NT;
            const int row = c / (D / 8), col = c % (D / 8);
            if (c < GT * (D / 8)) *(u32x4 *) &Ks[buf][row / FM_KT][(row % FM_KT) * KLD + col * 8] = kreg[i];
        }
        if (a.vt) {
#pragma unroll
            for (int i = 0; i < 2 * VIT; ++i) {
                const int c = tid + i * NT;
                const int ch = c % (GT / 8), d = c / (GT / 8);
                if (c < D * (GT / 8)) {
                    uint32_t * w = &Vt[buf][ch / 4][d * (VLD / 2) + (ch % 4) * 4];      // (cell pairs 4 (ch % 4) .. + 3 of tile ch / 4)
                    const u32x4 r = vreg[i >> 1][i & 1];
                    w[0] = r[0]; w[1] = r[1]; w[2] = r[2]; w[3] = r[3];
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < VIT; ++i) {
            const int c = tid + i * NT;
            const int o = c % (D / 8), p = c / (D / 8);
            if (c < (GT / 2) * (D / 8)) {
                const int sub = p / (FM_KT / 2), pp = p % (FM_KT / 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    Vt[buf][sub][(8 * o + 2 * e)     * (VLD / 2) + pp] = (vreg[i][0][e] & 0xffffu) | (vreg[i][1][e] << 16);
                    Vt[buf][sub][(8 * o + 2 * e + 1) * (VLD / 2) + pp] = (vreg[i][0][e] >> 16)     | (vreg[i][1][e] & 0xffff0000u);
                }
            }
        }
    };

    stamp();                                              // 2: classes, live bits
    int t = next_live(0), nlive = 0;
    int t_nx = t < ntile ? next_live(t + 1) : ntile;
    if (t < ntile) load_kv(t, kregA, vregA);
    if (t_nx < ntile) load_kv(t_nx, kregB, vregB);
    const bool c2pos = c2 > 0.0f;                                     // (then the row maximum may be taken before the scale is applied)
    auto step = [&](u32x4 (&kreg)[KCH], u32x4 (&vreg)[VIT][2]) {
        const int buf = nlive & 1;
        int cls[SQ]; u32x2 mw[SQ][4] = {};
#pragma unroll
        for (int sq = 0; sq < SQ; ++sq) {
            cls[sq] = tile_class(t, sq);
            if (cls[sq] == 2) load_mask(t * NTL + kvs * SQ + sq, mw[sq]);       // mixed tile: this lane's mask words (latency under the stores)
        }
        stamp();                                          // group: top
        if (!(ABL & 1) || nlive == 0) store_kv(buf, kreg, vreg);
        stamp();                                          // group: stored (rows had landed)
        // barrier: group t is visible; everybody is past the previous live group's matrix work, so the other buffer may be refilled
        __syncthreads();
        stamp();                                          // group: past the barrier
        const int t2 = t_nx < ntile ? next_live(t_nx + 1) : ntile;
        if (t2 < ntile && !(ABL & 1)) load_kv(t2, kreg, vreg);       // (this set has just been stored) in flight under this group's and the next one's MFMAs

#pragma unroll
        for (int sq = 0; sq < SQ; ++sq) {
            if (cls[sq] == 0) continue;
            const int li = kvs * SQ + sq;
            // ---- S^T = K . Q^T
            f16a sc;
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[e] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const h8v kf = *(const h8v *) &Ks[buf][li][lq * KLD + ks * 16 + hb * 8];
                sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sc, 0, 0, 0);
            }
            // ---- scale / softcap / mask (all-zero mask tiles skip the mask arithmetic), base-2 online softmax
            float tmax = -INFINITY;
            const bool plain = cls[sq] == 1 && a.logit_softcap == 0.0f && slope == 1.0f && c2pos;
            if (ABL & 2) {
            } else if (plain) {                                              // p = exp2(s * c2 - M): the scale rides in the exponent's FMA
#pragma unroll
                for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, sc[e]);
                tmax *= c2;
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t w = mw[sq][e >> 2][(e >> 1) & 1];
                    const uint16_t hbits = (uint16_t) ((e & 1) ? (w >> 16) : (w & 0xffffu));
                    float v = sc[e] * c2;
                    if (a.logit_softcap != 0.0f) v = a.logit_softcap * FM_LOG2E * tanhf(v);
                    v = (hbits == 0xfc00u || !row_ok) ? -INFINITY : (cls[sq] == 1 ? v : v + slope2 * h2f(hbits));
                    sc[e] = v;
                    tmax = fmaxf(tmax, v);
                }
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float Mn = fmaxf(M, tmax);
            const float Mu = Mn == -INFINITY ? 0.0f : Mn;
            const float alpha = __builtin_amdgcn_exp2f(M - Mu);           // M == -inf -> 0
            float psum = 0.0f;
            union { h2v h2[4]; h8v v; } pf[2];
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const float p0 = (ABL & 2) ? sc[e] : __builtin_amdgcn_exp2f(plain ? fmaf(sc[e], c2, -Mu) : sc[e] - Mu), p1 = (ABL & 2) ? sc[e + 1] : __builtin_amdgcn_exp2f(plain ? fmaf(sc[e + 1], c2, -Mu) : sc[e + 1] - Mu);   // masked -> 0
                psum += p0 + p1;
                const f32x2 pp = { p0, p1 };
                pf[e >> 3].h2[(e & 7) >> 1] = __builtin_convertvector(pp, h2v);          // v_cvt_pk_f16_f32 (round-to-nearest-even)
            }
            S = S * alpha + psum;
            M = Mn;
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc_o[db][e] *= alpha;
            }
            // ---- O^T += V^T . P^T  (the NDB accumulators in turn: consecutive MFMAs are independent)
#pragma unroll
            for (int s2 = 0; s2 < ((ABL & 4) ? 0 : 2); ++s2) {
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const uint32_t * vr = &Vt[buf][li][(db * 32 + lq) * (VLD / 2) + 8 * s2 + 2 * hb];
                    union { uint32_t u[4]; h8v v; } vf;
                    vf.u[0] = vr[0]; vf.u[1] = vr[1]; vf.u[2] = vr[4]; vf.u[3] = vr[5];
                    acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf.v, pf[s2].v, acc_o[db], 0, 0, 0);
                }
            }
        }
        t = t_nx; t_nx = t2; ++nlive;
    };
    while (t < ntile) { step(kregA, vregA); if (t >= ntile) break; step(kregB, vregB); }
    stamp();                                              // loop done

    // ---- merge the KS partial states of each query block (same lane layout in every wave of a block): the ks > 0 waves park
    // (M, S, O^T) in LDS, the ks == 0 wave folds them in and finishes
    if (KS > 1) {
        float * const mrg = (float *) pool;
        __syncthreads();                                             // (every wave is past its last tile: the pool changes hands)
        float * mine = mrg + ((size_t) ((kvs > 0 ? kvs - 1 : 0) * NW + wave) * 64 + lane) * (NDB * 16 + 2);
        if (kvs > 0) {
            mine[0] = M; mine[1] = S;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) mine[2 + db * 16 + e] = acc_o[db][e];
        }
        __syncthreads();
        if (kvs > 0) return;
#pragma unroll
        for (int k = 1; k < KS; ++k) {
            const float * oth = mrg + ((size_t) ((k - 1) * NW + wave) * 64 + lane) * (NDB * 16 + 2);
            const float Mo = oth[0], So = oth[1];
            const float Mn = fmaxf(M, Mo);
            const float Mu = Mn == -INFINITY ? 0.0f : Mn;
            const float f0 = __builtin_amdgcn_exp2f(M - Mu), f1 = __builtin_amdgcn_exp2f(Mo - Mu);
            S = S * f0 + So * f1; M = Mn;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc_o[db][e] = acc_o[db][e] * f0 + oth[2 + db * 16 + e] * f1;
        }
    }
    stamp();                                              // merged
    // ---- finish: fold the two lane halves' partial sums, sinks (ops.cpp:8116-8130), normalise, store permuted
    S += __shfl_xor(S, 32, 64);
    float osc = 1.0f;
    if (a.sinks) {
        const float sk = a.sinks[h] * FM_LOG2E;
        if (sk > M) { const float f = __builtin_amdgcn_exp2f(M - sk); S = S * f + 1.0f; osc = f; }
        else S += __builtin_amdgcn_exp2f(sk - M);
    }
    const float inv = S == 0.0f ? 0.0f : osc / S;
    // (KS == 1: every wave is past its last tile before the pool becomes the rows' staging; KS > 1: the slot this wave has just folded in)
    if (KS == 1) {
#pragma unroll
        for (int rnd = 0; rnd * ORW < NW; ++rnd) {
            __syncthreads();
            if (wave / ORW == rnd) fa_store_rows<D>(acc_o, inv, (float *) (pool + (wave % ORW) * OWB), lane, a, h, q0, is3);
        }
    } else fa_store_rows<D>(acc_o, inv, (float *) (pool + wave * OWB), lane, a, h, q0, is3);
    stamp();                                              // stored
    if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[31] = (unsigned long long) n_st;
}

// ---- head size 128, many query blocks: the same maths with K / V tiles that never pass through registers.
// The kernel above stages a tile's rows global -> VGPR -> LDS (V transposed on the way: 8 ds_write_b32 + 16 bit operations per thread and tile) and has ONE group of
// rows in flight per workgroup; PMC and knock-outs (profiles/r04_fattn.txt) show its waves parked 40 % of their life behind that chain and its vector pipe busier
// than its matrix pipe.  Here:
//   * K and V tiles (32 rows x 256 B each) arrive by LDS-DMA (global_load_lds_dwordx4) in a 4-slot ring, three tiles ahead, behind a counted s_waitcnt vmcnt(8) and one
//     barrier per tile; rows past the end of the cache re-read its last row (finite data: their probabilities are zero)
//   * both tiles stay ROW-MAJOR in LDS.  K fragments (lane = kv row, 8 consecutive d) are ds_read_b128 at the 16-byte chunk (ks*2 + hb) ^ (row & 15) -- the DMA
//     source side applies the same XOR, so the 16 lanes the hardware serves together hit 16 different chunk positions
//   * V^T fragments come from ds_read_b64_tr_b16: in every group of 16 lanes, lane 4i + r passes the address of 4 consecutive halves (d0 + 4r ..) of kv row k0 + i, and
//     lane j gets back column d0 + j of rows k0 .. k0 + 3 (measured: tools/tr_probe.hip) -- exactly the four consecutive kv of one d that half an MFMA operand is under the
//     kv labelling above; V chunks are XORed with 4 * (row & 3) so that the four rows of a read lie in different bank quarters
// Same arithmetic as k_fattn_mma (P rounded to f16, exponentials in the base-2 domain, the scale folded into the exponent's FMA on unmasked tiles).
constexpr float FD_THR = 8.0f;
constexpr int FD_NST = 4, FD_ROWB = 256, FD_TILEB = FM_KT * FD_ROWB, FD_STAGEB = 2 * FD_TILEB;
constexpr int FD_LIVE_OFF = FD_NST * FD_STAGEB, FD_CLS_OFF = FD_LIVE_OFF + FM_MAXT / 8, FD_LDS = FD_CLS_OFF + 4 * (FM_MAXT / 16) * 4;
extern __shared__ __attribute__((aligned(16))) char fd_lds[];

// the 16 V^T reads of a tile (4 d-blocks x {s2 = 0: rows +0 / +8, s2 = 1: rows +16 / +24}) are ISSUED in one statement and WAITED for in another: the soft-max
// arithmetic between the two runs under their latency.  The registers are written by the hardware after the first statement returns: nothing may read, move or
// spill them before fd_tr_wait (the kernel has no scratch: checked with -Rpass-analysis=kernel-resource-usage).
struct fd_vfrag { u32x2 r[4][4]; };                               // [db][2 * s2 + half]
static __device__ __forceinline__ void fd_tr_issue(const uint32_t (&ad)[4], fd_vfrag & f) {
    asm volatile("ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:2048\n\tds_read_b64_tr_b16 %2, %16 offset:4096\n\tds_read_b64_tr_b16 %3, %16 offset:6144\n\t"
                 "ds_read_b64_tr_b16 %4, %17\n\tds_read_b64_tr_b16 %5, %17 offset:2048\n\tds_read_b64_tr_b16 %6, %17 offset:4096\n\tds_read_b64_tr_b16 %7, %17 offset:6144\n\t"
                 "ds_read_b64_tr_b16 %8, %18\n\tds_read_b64_tr_b16 %9, %18 offset:2048\n\tds_read_b64_tr_b16 %10, %18 offset:4096\n\tds_read_b64_tr_b16 %11, %18 offset:6144\n\t"
                 "ds_read_b64_tr_b16 %12, %19\n\tds_read_b64_tr_b16 %13, %19 offset:2048\n\tds_read_b64_tr_b16 %14, %19 offset:4096\n\tds_read_b64_tr_b16 %15, %19 offset:6144"
                 : "=&v"(f.r[0][0]), "=&v"(f.r[0][1]), "=&v"(f.r[0][2]), "=&v"(f.r[0][3]), "=&v"(f.r[1][0]), "=&v"(f.r[1][1]), "=&v"(f.r[1][2]), "=&v"(f.r[1][3]),
                   "=&v"(f.r[2][0]), "=&v"(f.r[2][1]), "=&v"(f.r[2][2]), "=&v"(f.r[2][3]), "=&v"(f.r[3][0]), "=&v"(f.r[3][1]), "=&v"(f.r[3][2]), "=&v"(f.r[3][3])
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]) : "memory");
}
static __device__ __forceinline__ void fd_tr_wait(fd_vfrag & f) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.r[0][0]), "+v"(f.r[0][1]), "+v"(f.r[0][2]), "+v"(f.r[0][3]), "+v"(f.r[1][0]), "+v"(f.r[1][1]), "+v"(f.r[1][2]), "+v"(f.r[1][3]),
                   "+v"(f.r[2][0]), "+v"(f.r[2][1]), "+v"(f.r[2][2]), "+v"(f.r[2][3]), "+v"(f.r[3][0]), "+v"(f.r[3][1]), "+v"(f.r[3][2]), "+v"(f.r[3][3]) :: "memory");
}
// head size 64: two d-blocks, the rows of a tile 128 bytes apart (+8 rows = 1024 bytes)
static __device__ __forceinline__ void fd_tr_issue64(const uint32_t (&ad)[2], fd_vfrag & f) {
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:1024\n\tds_read_b64_tr_b16 %2, %8 offset:2048\n\tds_read_b64_tr_b16 %3, %8 offset:3072\n\t"
                 "ds_read_b64_tr_b16 %4, %9\n\tds_read_b64_tr_b16 %5, %9 offset:1024\n\tds_read_b64_tr_b16 %6, %9 offset:2048\n\tds_read_b64_tr_b16 %7, %9 offset:3072"
                 : "=&v"(f.r[0][0]), "=&v"(f.r[0][1]), "=&v"(f.r[0][2]), "=&v"(f.r[0][3]), "=&v"(f.r[1][0]), "=&v"(f.r[1][1]), "=&v"(f.r[1][2]), "=&v"(f.r[1][3])
                 : "v"(ad[0]), "v"(ad[1]) : "memory");
}
static __device__ __forceinline__ void fd_tr_wait64(fd_vfrag & f) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.r[0][0]), "+v"(f.r[0][1]), "+v"(f.r[0][2]), "+v"(f.r[0][3]), "+v"(f.r[1][0]), "+v"(f.r[1][1]), "+v"(f.r[1][2]), "+v"(f.r[1][3]) :: "memory");
}
static __device__ __forceinline__ float fd_max_halves(float x) {      // max of lanes l and l ^ 32 without the LDS crossbar (v_permlane32_swap)
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#else
    return fmaxf(x, __shfl_xor(x, 32, 64));
#endif
}

// HW = heads per workgroup: 2 = eight waves, the four query blocks of TWO heads of one KV head (same mask tiles, same K / V rows): half the requests per unit of matrix work
// D = 64 (the omni encoders: Whisper 1500 x 1500 x 16 heads, round 5): rows of 128 bytes = 8 chunks, one DMA instruction = 8 rows (one per wave and tile, HW = 1 only);
// K chunk c of row r sits at c ^ ((r >> 1) & 7) (rows two apart share a bank half), V's 64-byte d-block at db ^ ((r >> 1) & 1) (rows r, r + 2 of a transpose read
// 64 bytes apart); the V^T reads' row offsets are 8 x 128 bytes.
// KS = 2 (D = 64, grids that leave the chip under-filled -- Whisper's 1500 x 1500 x 16 heads is 192 workgroups): eight waves, waves 4 .. 7 work on the same four query
// blocks but on every other PAIR of live tiles, with their own half of the ring (each group of four waves requests its own tiles); the two (O, M, S) states are
// folded through LDS at the end.  Both groups pass the same number of barriers (the step count comes from the number of live tiles).
template <int D, int ABL, int HW, int KS = 1>
__global__ void __launch_bounds__(256 * HW * KS) __attribute__((amdgpu_waves_per_eu(2))) k_fattn_dma(const fa_dev a, const int nqt) {
    static_assert((D == 128 && KS == 1) || (D == 64 && HW == 1), "head sizes 128 and 64 (one head per workgroup); the key split only at 64 (at 128 the register-staged four-way split is faster on small grids: 24.7 vs 26.0 us at pp512)");
    constexpr int NW = 4, NKS = D / 16, NDB = D / 32, NT = 64 * NW * HW * KS;
    constexpr int ROWB = 2 * D, CPR = ROWB / 16, RPI = 64 / CPR, TILEB = FM_KT * ROWB, STAGEB = 2 * TILEB;      // chunks per row, rows per DMA instruction
    constexpr int GRP_RING = FD_NST * STAGEB, LIVE_OFF = KS * GRP_RING, CLS_OFF = LIVE_OFF + FM_MAXT / 8;
    char * const lds = fd_lds;
    uint64_t * const live_bits = (uint64_t *) (lds + LIVE_OFF);
    uint32_t (* const cls2)[FM_MAXT / 16] = (uint32_t (*)[FM_MAXT / 16]) (lds + CLS_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hb = lane >> 5;
    int b = (int) blockIdx.x;
    const int qt  = nqt - 1 - b % nqt; b /= nqt;          // longest (latest, for causal masks) query tiles first
    const int qbw = wave & 3;                                       // query block of this wave within the workgroup's 128 queries
    const int h   = (b % (a.nh / HW)) * HW + (HW == 2 ? wave >> 2 : 0);  const int is3 = b / (a.nh / HW);
    const int grp = KS == 2 ? wave >> 2 : 0, wq = KS == 2 ? (wave & 3) : wave;        // key-split group; wave within the group's DMA team
    const int ikv = h / a.gq;
    const int q0  = (qt * NW + qbw) * 32;
    const int q   = q0 + lq;
    const int qc  = q < a.nq ? q : a.nq - 1;
    const bool row_ok = q < a.nq;

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
    const float slope2 = slope * FM_LOG2E;
    const float c2 = a.logit_softcap != 0.0f ? a.scale : a.scale * FM_LOG2E;
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

    // ---- tile classes of the workgroup's four query blocks -> LDS, live bits (as above)
    const int qb0 = qt * NW;
    const uint8_t * maprow = a.tile_map ? a.tile_map + (((int64_t) (is3 % (int) a.mne3) * a.mne2 + (h % (int) a.mne2)) * a.map_nqb) * ntile : nullptr;
    if (maprow) {
        const int nwords = (ntile + 15) / 16;
        for (int j = tid; j < NW * nwords; j += NT) {
            const int w = j / nwords, wd = j % nwords;
            uint32_t word = 0;
            if (qb0 + w < a.map_nqb) {
                const uint8_t * r = maprow + (int64_t) (qb0 + w) * ntile;
#pragma unroll
                for (int i = 0; i < 16; ++i) { const int t32 = wd * 16 + i; word |= (t32 < ntile ? (uint32_t) r[t32] & 3u : 0u) << (2 * i); }
            }
            cls2[w][wd] = word;
        }
        __syncthreads();
    }
    for (int c = wave; c * 64 < ntile; c += NW * HW * KS) {
        const int tt = c * 64 + lane;
        bool lv = false;
        if (tt < ntile) {
            if (!maprow) lv = true;
            else
#pragma unroll
                for (int w = 0; w < NW; ++w) lv |= ((cls2[w][tt >> 4] >> (2 * (tt & 15))) & 3u) != 0;
        }
        const uint64_t bits = __ballot(lv);
        if (lane == 0) live_bits[c] = bits;
    }
    __syncthreads();
    auto next_live = [&](int u) {
        while (u < ntile) {
            const uint64_t w = live_bits[u >> 6] >> (u & 63);
            if (w) { u += __builtin_ctzll(w); break; }
            u = (u | 63) + 1;
        }
        return __builtin_amdgcn_readfirstlane(u < ntile ? u : ntile);
    };
    auto tile_class = [&](int t) -> int {
        if (q0 >= a.nq || t >= ntile) return 0;
        if (!maprow) return (t + 1) * FM_KT <= a.nkv ? 1 : 2;
        return (int) ((cls2[qbw][t >> 4] >> (2 * (t & 15))) & 3u);
    };
    auto load_mask = [&](int t, u32x2 (&w)[4]) {
        const int kv0 = t * FM_KT;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kvb = kv0 + 4 * hb + 8 * g;
            if (kvb + 3 < a.nkv && (!mrow || mask_vec)) {
                if (mrow) w[g] = *(const u32x2 *) (mrow + kvb); else { w[g][0] = 0u; w[g][1] = 0u; }
            } else {
                uint32_t hv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[i] = kvb + i < a.nkv ? (mrow ? (uint32_t) mrow[kvb + i] : 0u) : 0xfc00u;
                w[g][0] = hv[0] | (hv[1] << 16); w[g][1] = hv[2] | (hv[3] << 16);
            }
        }
    };

    // ---- DMA side: instruction j of wave w fills rows (w*NJ + j)*4 + [0,4) of a tile, lane l the chunk l & 15 of row l >> 4 from the source chunk the swizzle asks for
    constexpr int NJ = FM_KT / RPI / (NW * HW);                     // K (and V) instructions per wave and tile (D = 128: 2 / HW, D = 64: 1)
    const int c16 = lane & (CPR - 1);
    int rit[NJ]; uint32_t kso[NJ], vso[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        rit[j] = (wq * NJ + j) * RPI + lane / CPR;
        kso[j] = (uint32_t) ((c16 ^ (D == 128 ? (rit[j] & 15) : ((rit[j] >> 1) & 7))) * 16);
        vso[j] = (uint32_t) ((c16 ^ (4 * (D == 128 ? (rit[j] & 3) : ((rit[j] >> 1) & 1)))) * 16);
    }
    auto skip_live = [&](int t_, int n_) { for (int i_ = 0; i_ < n_ && t_ < ntile; ++i_) t_ = next_live(t_ + 1); return t_; };      // the n-th live tile after t
    int nlive = 0;
    for (int c = 0; c * 64 < ntile; ++c) nlive += __builtin_popcountll(live_bits[c]);
    nlive = __builtin_amdgcn_readfirstlane(nlive);
    const int npairs = (nlive + 1) / 2, nsteps = KS == 2 ? (npairs + 1) / 2 : npairs;
    const int t_first = grp == 1 ? skip_live(next_live(0), 2) : next_live(0);
    int slot_d = 0, t_dma = t_first, t_last = 0, n_dma = 0;
    auto dma_tile = [&]() {                                          // the next live tile (past the last one: that one again, into a slot nobody reads)
        const int t = t_dma < ntile ? t_dma : t_last;
        char * const sb = lds + grp * GRP_RING + slot_d * STAGEB + (wq * NJ * RPI) * ROWB;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int r = t * FM_KT + rit[j]; r = r < a.nkv ? r : a.nkv - 1;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (kbase + (int64_t) r * a.knb1 + kso[j]), (lds_ptr_t) (sb + j * RPI * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (vbase + (int64_t) r * a.vnb1 + vso[j]), (lds_ptr_t) (sb + TILEB + j * RPI * ROWB), 16, 0, 0);
        }
        if (t_dma < ntile) { t_last = t_dma; t_dma = (KS == 2 && (n_dma & 1)) ? skip_live(t_dma, 3) : next_live(t_dma + 1); }      // (key split: the other group's pair lies in between)
        ++n_dma;
        slot_d = slot_d == FD_NST - 1 ? 0 : slot_d + 1;
    };

    // ---- reader side, per lane: K row lq, chunk (ks*2 + hb) ^ (lq & 15); V^T: group g = lane / 16, s = lane % 16 -> row 4*hb + s/4 (+ 8 r + 16 s2), halves d0 + 4 (s & 3) ..
    const uint32_t lds0 = (uint32_t) (uintptr_t) lds + (uint32_t) (grp * GRP_RING);
    const char * const krow = lds + grp * GRP_RING + lq * ROWB;
    const int ksw = D == 128 ? (lq & 15) : ((lq >> 1) & 7);
    const int gi = lane >> 4, si = lane & 15;
    const uint32_t vlane = lds0 + TILEB + (uint32_t) ((4 * (gi >> 1) + (si >> 2)) * ROWB + (2 * (gi & 1) + ((si & 3) >> 1)) * 16 + (si & 1) * 8);
    uint32_t vdb[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) vdb[db] = vlane + (uint32_t) (64 * (db ^ (D == 128 ? (si >> 2) : ((si >> 3) & 1))));

    const bool c2pos = c2 > 0.0f;
    // ---- the phases of a tile; the two tiles of a pair are interleaved below so that one's soft-max arithmetic issues while the other's matrix products execute
    // the running maximum is only a reference point of the exponents: it follows the row maximum when that has grown by more than 2^FD_THR since the last update
    // (wave-uniform decision: then every lane rescales), so p <= 2^FD_THR instead of <= 1 -- f16 holds it, O and S are f32 -- and the 64-multiply rescale of O^T,
    // which random scores trigger in almost every tile of the first hundred, runs a handful of times per row
    typedef union { h2v h2[4]; h8v v; } pfrag;
    auto soft_max = [&](const int cls, const u32x2 (&mw)[4], f16a & sc, pfrag (&pf)[2]) {
        float tmax = -INFINITY, Mn, Mu, alpha, psum = 0.0f;
        const bool plain = cls == 1 && a.logit_softcap == 0.0f && slope == 1.0f && c2pos;
        if (plain) {                                                  // p = exp2(s * c2 - M): the scale rides in the exponent's FMA; three-input maxima
#pragma unroll
            for (int e = 0; e < 16; e += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(sc[e]), "v"(sc[e + 1]));
            tmax *= c2;
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t w = mw[e >> 2][(e >> 1) & 1];
                const uint16_t hbits = (uint16_t) ((e & 1) ? (w >> 16) : (w & 0xffffu));
                float v = sc[e] * c2;
                if (a.logit_softcap != 0.0f) v = a.logit_softcap * FM_LOG2E * tanhf(v);
                v = (hbits == 0xfc00u || !row_ok) ? -INFINITY : (cls == 1 ? v : v + slope2 * h2f(hbits));
                sc[e] = v;
                tmax = fmaxf(tmax, v);
            }
        }
        tmax = fd_max_halves(tmax);
        if (__any(tmax > M + FD_THR)) { Mn = fmaxf(M, tmax); Mu = Mn == -INFINITY ? 0.0f : Mn; alpha = __builtin_amdgcn_exp2f(M - Mu); }
        else                          { Mn = M; Mu = M == -INFINITY ? 0.0f : M; alpha = 1.0f; }
        if (plain) {                                                  // pairs: v_pk_fma_f32 for the exponents' arguments, v_pk_add_f32 for the row sums
            const f32x2 c2v = { c2, c2 }, nMu = { -Mu, -Mu };
            f32x2 ps2 = { 0.0f, 0.0f };
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 x = { sc[e], sc[e + 1] };
                const f32x2 y = __builtin_elementwise_fma(x, c2v, nMu);
                const f32x2 pp = { __builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1]) };
                ps2 += pp;
                pf[e >> 3].h2[(e & 7) >> 1] = __builtin_convertvector(pp, h2v);
            }
            psum = ps2[0] + ps2[1];
        } else {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const float p0 = __builtin_amdgcn_exp2f(sc[e] - Mu), p1 = __builtin_amdgcn_exp2f(sc[e + 1] - Mu);   // masked -> 0
                psum += p0 + p1;
                const f32x2 pp = { p0, p1 };
                pf[e >> 3].h2[(e & 7) >> 1] = __builtin_convertvector(pp, h2v);
            }
        }
        S = S * alpha + psum;
        M = Mn;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc_o[db][e] *= alpha;
        }
    };
    auto pv = [&](fd_vfrag & vf, const pfrag (&pf)[2]) {              // O^T += V^T . P^T (the four accumulators in turn: consecutive MFMAs are independent)
        if constexpr (D == 128) fd_tr_wait(vf); else fd_tr_wait64(vf);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                union { u32x4 u; h8v v; } vv;
                vv.u = u32x4{ vf.r[db][2 * s2][0], vf.r[db][2 * s2][1], vf.r[db][2 * s2 + 1][0], vf.r[db][2 * s2 + 1][1] };
                acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vv.v, pf[s2].v, acc_o[db], 0, 0, 0);
            }
    };
    // the two tiles of a pair: both score products first, alternating (consecutive MFMAs are independent), then per tile V^T requested | soft-max | P.V -- tile 0's
    // P.V executes while tile 1's soft-max issues.  (Measured against a hand-ordered form with the K fragments requested eight at a time by inline asm and waited for
    // in halves: 538 vs 577 TFLOP/s at 8 x (512 x 2048) -- the compiler's own ds_read / MFMA interleave is the better one.)
    auto pair_work = [&](const int cls0, const int cls1, const u32x2 (&mw0)[4], const u32x2 (&mw1)[4], const uint32_t so0, const uint32_t so1) {
        f16a s0, s1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s0[e] = 0.0f; s1[e] = 0.0f; }
        if (cls0 != 0 && cls1 != 0) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {                            // (fragments requested one k-step ahead under sched_group_barriers: 554 vs 568 TFLOP/s -- not kept)
                const h8v k0 = *(const h8v *) (krow + so0 + (((ks * 2 + hb) ^ ksw) << 4));
                const h8v k1 = *(const h8v *) (krow + so1 + (((ks * 2 + hb) ^ ksw) << 4));
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[ks], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[ks], s1, 0, 0, 0);
            }
        } else if (cls0 != 0 || cls1 != 0) {
            const uint32_t so = cls0 != 0 ? so0 : so1;
            f16a sb;
#pragma unroll
            for (int e = 0; e < 16; ++e) sb[e] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < NKS; ks += 2) {
                const h8v k0 = *(const h8v *) (krow + so + (((ks * 2 + hb) ^ ksw) << 4));
                const h8v k1 = *(const h8v *) (krow + so + ((((ks + 1) * 2 + hb) ^ ksw) << 4));
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[ks], s0, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[ks + 1], sb, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) s0[e] += sb[e];
            if (cls0 == 0) { s1 = s0; }
        }
        if (ABL & 4) {
            pfrag pf[2];
            if (cls0 != 0) soft_max(cls0, mw0, s0, pf);
            if (cls1 != 0) soft_max(cls1, mw1, s1, pf);
            return;
        }
        fd_vfrag v0, v1; pfrag pf0[2], pf1[2];
        auto v_issue = [&](const uint32_t so_, fd_vfrag & vf) {
            if constexpr (D == 128) { const uint32_t ad[4] = { vdb[0] + so_, vdb[1] + so_, vdb[2] + so_, vdb[3] + so_ }; fd_tr_issue(ad, vf); }
            else                    { const uint32_t ad[2] = { vdb[0] + so_, vdb[1] + so_ }; fd_tr_issue64(ad, vf); }
        };
        if (cls0 != 0) { v_issue(so0, v0); soft_max(cls0, mw0, s0, pf0); pv(v0, pf0); }
        if (cls1 != 0) { v_issue(so1, v1); soft_max(cls1, mw1, s1, pf1); pv(v1, pf1); }
    };
    // ---- two live tiles per barrier: the ring is two pairs of slots; while a pair is worked on, the next pair's rows are in flight (a pair of tiles of matrix work to arrive)
    int t = t_first;
    dma_tile(); dma_tile();
    int pair = 0;
    for (int step = 0; step < nsteps; ++step) {
        const int t1 = t < ntile ? next_live(t + 1) : ntile;
        const int cls0 = tile_class(t), cls1 = t1 < ntile ? tile_class(t1) : 0;
        u32x2 mw0[4] = {}, mw1[4] = {};
        if (cls0 == 2) load_mask(t, mw0);
        if (cls1 == 2) load_mask(t1, mw1);
        // own requests of this pair have landed; after the barrier everybody's have, and everybody is past the previous pair, whose slots take the next one
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (!(ABL & 1)) { dma_tile(); dma_tile(); }
        const uint32_t so = (uint32_t) (pair * 2 * STAGEB);
        pair ^= 1;
        pair_work(cls0, cls1, mw0, mw1, so, so + STAGEB);
        t = t1 < ntile ? (KS == 2 ? skip_live(t1, 3) : next_live(t1 + 1)) : ntile;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (requests past the last live tile)

    S += __shfl_xor(S, 32, 64);
    if constexpr (KS == 2) {                                         // fold the second group's state into the first's (exponents in the base-2 domain, as inside the loop)
        __syncthreads();                                             // everybody is past the ring: it becomes the staging area
        float * const mg = (float *) lds + wq * (64 * (NDB * 16 + 2));
        if (grp == 1) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) mg[(db * 16 + e) * 64 + lane] = acc_o[db][e];
            mg[(NDB * 16) * 64 + lane] = M; mg[(NDB * 16 + 1) * 64 + lane] = S;
        }
        __syncthreads();
        if (grp == 1) return;
        const float M1 = mg[(NDB * 16) * 64 + lane], S1 = mg[(NDB * 16 + 1) * 64 + lane];
        const float Mn = fmaxf(M, M1), Mu = Mn == -INFINITY ? 0.0f : Mn;
        const float a0 = __builtin_amdgcn_exp2f(M - Mu), a1 = __builtin_amdgcn_exp2f(M1 - Mu);
        S = S * a0 + S1 * a1; M = Mn;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc_o[db][e] = acc_o[db][e] * a0 + mg[(db * 16 + e) * 64 + lane] * a1;
    }
    float osc = 1.0f;
    if (a.sinks) {
        const float sk = a.sinks[h] * FM_LOG2E;
        if (sk > M) { const float f = __builtin_amdgcn_exp2f(M - sk); S = S * f + 1.0f; osc = f; }
        else S += __builtin_amdgcn_exp2f(sk - M);
    }
    const float inv = S == 0.0f ? 0.0f : osc / S;
    // (per-lane 16-byte stores: two workgroups share a CU here and the other one's matrix work covers them -- staging the rows through LDS as k_fattn_mma does, behind two more
    // barriers, measured the same within the box-to-box spread: 562-563 TFLOP/s either way at 8 x 512 x 2048)
    if (row_ok) {
        char * out = a.dst + h * a.dnb1 + q * a.dnb2 + is3 * a.dnb3;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o4;
#pragma unroll
                for (int i = 0; i < 4; ++i) o4[i] = acc_o[db][4 * g + i] * inv;
                if (a.write_f32) *(f32x4 *) (out + (db * 32 + 8 * g + 4 * hb) * 4) = o4;
                if (a.out16) {
                    u32x2 hw;
                    hw[0] = (uint32_t) f2h(o4[0]) | ((uint32_t) f2h(o4[1]) << 16); hw[1] = (uint32_t) f2h(o4[2]) | ((uint32_t) f2h(o4[3]) << 16);
                    *(u32x2 *) (a.out16 + ((int64_t) is3 * a.nq + q) * a.out16_rs + ((int64_t) h * D + db * 32 + 8 * g + 4 * hb) * 2) = hw;
                }
            }
    }
}

// ---- decode shape on the matrix cores (a few query tokens; fattn.hip routes them here): the 32 query columns of a wave tile are up to
// qpw * gq <= 32 (token, head) pairs that share ONE KV head -- column j = token j / gq, head ikv*gq + j % gq -- so a GQA group reads its
// K / V rows once.  Every wave works alone on its own 32-row KV tiles (tile t of the slice goes to wave t % NW): no workgroup barrier
// in the loop.
//   K : straight from memory into the A-operand registers.  The contraction index is only a label, so K-slot (ks, hb, e) is declared to be
//       d = hb*D/2 + 8*ks + e: a lane reads D/2 CONTIGUOUS halfs of its row (Q is loaded with the same labelling).
//   V : the wave's 32 rows, transposed through its private LDS region exactly as the prefill kernel does.
//   next tile's K / V / mask words are requested right after the score product freed the registers, and land under softmax + P.V.
// The KV range is cut into a.nsplit slices (grid = slices x token chunks x KV heads x sequences); the NW waves of a workgroup fold their
// states through LDS.  nsplit == 1: the workgroup finishes the rows itself (sinks, 1/S, store, optional Q8_K image of the output row for
// the wo mat-vec); nsplit > 1: ONE partial (O, M, S) row per (token, head) goes to a.part for k_fattn_merge (fattn.hip), natural-log
// convention.  PRE: the layer's q / k chains (RMS_NORM -> MUL -> ROPE) and the k / v cache stores of the one new token run first, in
// this launch (fa_pre, fattn_dev.hpp); with a KV split only the slice that owns the new cache row stores it -- the other slices never
// read that row.
template <int D, int NW, bool PRE>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2))) k_fattn_gqa(const fa_dev a) {
    constexpr int VLD = fm_cfg<D>::VLD;
    constexpr int NKS = D / 16, NDB = D / 32;
    constexpr int VW  = D * VLD / 2;                      // 32-bit words of one V^T tile
    constexpr int MS  = NDB * 16 + 4;                     // floats of one lane's parked state: M, S, 2 pad, O (16-byte vectors)
    constexpr int MW  = 64 * MS;                          // floats of one parked wave state
    constexpr int LW  = NW * VW > (NW / 2) * MW ? NW * VW : (NW / 2) * MW;       // the merge area and the final rows re-use the V tiles
    static_assert((NW & (NW - 1)) == 0 && NW >= 2, "NW is a power of two");
    static_assert(LW >= 32 * D, "final rows fit");
    __shared__ __attribute__((aligned(16))) uint32_t Vt[LW];
    __shared__ __attribute__((aligned(16))) _Float16 qs[PRE ? 32 * D : 8];     // pre-stage: the GQA group's q heads after norm + rope

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lq = lane & 31, hb = lane >> 5;
    int b = (int) blockIdx.x;
    const int nqc = (a.nq + a.qpw - 1) / a.qpw;
    const int sp  = b % a.nsplit; b /= a.nsplit;
    const int qc  = b % nqc;      b /= nqc;
    const int ikv = b % a.nhkv;   const int is3 = b / a.nhkv;
    const int tok = qc * a.qpw + lq / a.gq;
    const bool row_ok = lq < a.qpw * a.gq && tok < a.nq;
    const int q = row_ok ? tok : 0;
    const int h = ikv * a.gq + (row_ok ? lq % a.gq : 0);

    const int ntile = (a.nkv + FM_KT - 1) / FM_KT;
    const int tps = (ntile + a.nsplit - 1) / a.nsplit;
    const int t_hi = (sp + 1) * tps < ntile ? (sp + 1) * tps : ntile;

    const uint32_t hu = (uint32_t) h;
    const float slope = a.max_bias > 0.0f ? (hu < a.n_head_log2 ? powf(a.m0, (float) (hu + 1)) : powf(a.m1, (float) (2 * (hu - a.n_head_log2) + 1))) : 1.0f;
    const float slope2 = slope * FM_LOG2E;
    const float c2 = a.logit_softcap != 0.0f ? a.scale : a.scale * FM_LOG2E;
    const uint16_t * mrow = a.mask ? (const uint16_t *) (a.mask + q * a.mnb1 + (is3 % (int) a.mne3) * a.mnb3) : nullptr;   // (launcher: mask ne2 == 1)
    const bool mask_vec = a.mask && (a.mnb1 % 8 == 0) && (((uintptr_t) mrow) % 8 == 0);

    f16a acc_o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[db][e] = 0.0f;
    float M = -INFINITY, S = 0.0f;

    const char * kbase = a.k + ikv * a.knb2 + is3 * a.knb3;
    const char * vbase = a.v + ikv * a.vnb2 + is3 * a.vnb3;
    uint32_t * vt = Vt + wave * VW;

    u32x4 kreg[NKS], vreg[NDB][2]; u32x2 mwn[4];
    auto load_tile = [&](int t) {
        const int kv0 = t * FM_KT;
        const bool kok = kv0 + lq < a.nkv;
        const char * kr = kbase + (int64_t) (kok ? kv0 + lq : 0) * a.knb1 + hb * D;        // hb * (D/2) halfs
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) { kreg[ks] = *(const u32x4 *) (kr + ks * 16); if (!kok) kreg[ks] = u32x4{ 0u, 0u, 0u, 0u }; }
#pragma unroll
        for (int i = 0; i < NDB; ++i) {
            const int c = lane + i * 64;
            const int o = c % (D / 8), p = c / (D / 8);
            vreg[i][0] = vreg[i][1] = u32x4{ 0u, 0u, 0u, 0u };
            if (kv0 + 2 * p     < a.nkv) vreg[i][0] = *(const u32x4 *) (vbase + (int64_t) (kv0 + 2 * p)     * a.vnb1 + o * 16);
            if (kv0 + 2 * p + 1 < a.nkv) vreg[i][1] = *(const u32x4 *) (vbase + (int64_t) (kv0 + 2 * p + 1) * a.vnb1 + o * 16);
        }
        // mask words of this lane's (token) row: word pair g = halfs kv0 + 4*hb + 8*g + {0..3}; cells past nkv read as -inf
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kvb = kv0 + 4 * hb + 8 * g;
            if (kvb + 3 < a.nkv && (!mrow || mask_vec)) {
                if (mrow) mwn[g] = *(const u32x2 *) (mrow + kvb); else { mwn[g][0] = 0u; mwn[g][1] = 0u; }
            } else {
                uint32_t hv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[i] = kvb + i < a.nkv ? (mrow ? (uint32_t) mrow[kvb + i] : 0u) : 0xfc00u;
                mwn[g][0] = hv[0] | (hv[1] << 16); mwn[g][1] = hv[2] | (hv[3] << 16);
            }
        }
    };

    // the first tile's K / V / mask words are requested before anything else: their latency hides the pre-stage / the Q loads
    int t = sp * tps + wave;
    if (t < t_hi) load_tile(t);

    h8v qf[NKS];
    if (PRE) {
        // one token, one sequence (launcher-checked): tasks 0 = k head (norm, rope, store), 1 = v head (store), 2 + r = q head r
        const fa_pre & P = a.pre;
        const float posf = (float) P.pos[0];
        const int64_t krow = P.idx_is64 ? *(const int64_t *) P.kidx : (int64_t) *(const int32_t *) P.kidx;
        const int64_t vrow = P.idx_is64 ? *(const int64_t *) P.vidx : (int64_t) *(const int32_t *) P.vidx;
        const bool own_k = (int) (krow / FM_KT) / tps == sp, own_v = (int) (vrow / FM_KT) / tps == sp;
        for (int task = wave; task < a.gq + 2; task += NW) {
            if (task == 1) {
                if (!own_v) continue;
                uint16_t * vr = (uint16_t *) (P.vcache + vrow * P.vc_rs) + ikv * D;
                const float * xv = (const float *) (P.vraw + ikv * P.v_hs);
                for (int e = lane; e < D; e += 64) vr[e] = f2h(xv[e]);
            } else {
                const bool isk = task == 0;
                if (isk && !own_k) continue;
                const int  r   = task - 2;
                const char * xr = isk ? P.kraw + ikv * P.k_hs : P.qraw + (ikv * a.gq + r) * P.q_hs;
                float r0[1], r1[1]; int e0[1], e1[1]; bool act[1];
                norm_rope_wave<1>(xr, isk ? P.kw : P.qw, D, P.eps, posf, P.ff, P.rd, lane, r0, r1, e0, e1, act);
                if (act[0]) {
                    if (isk) {
                        uint16_t * kr = (uint16_t *) (P.kcache + krow * P.kc_rs) + ikv * D;
                        kr[e0[0]] = f2h(r0[0]); kr[e1[0]] = f2h(r1[0]);
                    } else {
                        qs[r * D + e0[0]] = (_Float16) r0[0]; qs[r * D + e1[0]] = (_Float16) r1[0];    // q_to_vec_dot rounding (ops.cpp:8040)
                    }
                }
            }
        }
        __syncthreads();                                  // (also drains the k / v cache stores before any wave reads the cache)
        if (t < t_hi && (t == (int) (krow / FM_KT) || t == (int) (vrow / FM_KT))) load_tile(t);     // the tile with the row just stored: again
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const h8v *) &qs[(row_ok ? lq : 0) * D + hb * (D / 2) + ks * 8];
    } else {
        const char * qr = a.q + q * a.qnb1 + h * a.qnb2 + is3 * a.qnb3;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const f32x4 v0 = *(const f32x4 *) (qr + (hb * (D / 2) + ks * 8) * 4);
            const f32x4 v1 = *(const f32x4 *) (qr + (hb * (D / 2) + ks * 8 + 4) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qf[ks][e] = (_Float16) v0[e]; qf[ks][4 + e] = (_Float16) v1[e]; }   // q_to_vec_dot: f32 -> f16 (ops.cpp:8040)
        }
    }
    while (t < t_hi) {
        // ---- S^T = K . Q^T (frees kreg)
        f16a sc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            union { u32x4 u; h8v v; } kf; kf.u = kreg[ks];
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf.v, qf[ks], sc, 0, 0, 0);
        }
        u32x2 mw[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) mw[g] = mwn[g];
        // ---- V rows -> transposed tile in this wave's LDS region (frees vreg); wave-private, so only a wave-level ordering point
#pragma unroll
        for (int i = 0; i < NDB; ++i) {
            const int c = lane + i * 64;
            const int o = c % (D / 8), p = c / (D / 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vt[(8 * o + 2 * e)     * (VLD / 2) + p] = (vreg[i][0][e] & 0xffffu) | (vreg[i][1][e] << 16);
                vt[(8 * o + 2 * e + 1) * (VLD / 2) + p] = (vreg[i][0][e] >> 16)     | (vreg[i][1][e] & 0xffff0000u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int tn = t + NW;
        if (tn < t_hi) load_tile(tn);                                  // in flight under the softmax and P.V below

        // ---- scale / softcap / mask, base-2 online softmax (as k_fattn_mma)
        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t w = mw[e >> 2][(e >> 1) & 1];
            const uint16_t hbits = (uint16_t) ((e & 1) ? (w >> 16) : (w & 0xffffu));
            float v = sc[e] * c2;
            if (a.logit_softcap != 0.0f) v = a.logit_softcap * FM_LOG2E * tanhf(v);
            v = (hbits == 0xfc00u || !row_ok) ? -INFINITY : v + slope2 * h2f(hbits);
            sc[e] = v;
            tmax = fmaxf(tmax, v);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float Mn = fmaxf(M, tmax);
        const float Mu = Mn == -INFINITY ? 0.0f : Mn;
        const float alpha = __builtin_amdgcn_exp2f(M - Mu);
        float psum = 0.0f;
        union { h2v h2[4]; h8v v; } pf[2];
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const float p0 = __builtin_amdgcn_exp2f(sc[e] - Mu), p1 = __builtin_amdgcn_exp2f(sc[e + 1] - Mu);
            psum += p0 + p1;
            const f32x2 pp = { p0, p1 };
            pf[e >> 3].h2[(e & 7) >> 1] = __builtin_convertvector(pp, h2v);
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
                const uint32_t * vr = &vt[(db * 32 + lq) * (VLD / 2) + 8 * s2 + 2 * hb];
                union { uint32_t u[4]; h8v v; } vf;
                vf.u[0] = vr[0]; vf.u[1] = vr[1]; vf.u[2] = vr[4]; vf.u[3] = vr[5];
                acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf.v, pf[s2].v, acc_o[db], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        t = tn;
    }

    // ---- fold the NW wave states pairwise through LDS (the V tiles are dead): waves [s, 2s) park, waves [0, s) fold, s = NW/2 .. 1
    float * mrg = (float *) Vt;
    __syncthreads();
#pragma unroll
    for (int s = NW / 2; s >= 1; s >>= 1) {
        if (wave >= s && wave < 2 * s && row_ok) {              // (dead columns carry nothing)
            float * mine = mrg + (size_t) (wave - s) * MW + lane * MS;
            *(f32x2 *) mine = f32x2{ M, S };
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) *(f32x4 *) (mine + 4 + db * 16 + 4 * g) = f32x4{ acc_o[db][4 * g], acc_o[db][4 * g + 1], acc_o[db][4 * g + 2], acc_o[db][4 * g + 3] };
        }
        __syncthreads();
        if (wave < s && row_ok) {
            const float * oth = mrg + (size_t) wave * MW + lane * MS;
            const f32x2 ms = *(const f32x2 *) oth;
            const float Mo = ms[0], So = ms[1];
            const float Mn = fmaxf(M, Mo);
            const float Mu = Mn == -INFINITY ? 0.0f : Mn;
            const float f0 = __builtin_amdgcn_exp2f(M - Mu), f1 = __builtin_amdgcn_exp2f(Mo - Mu);
            S = S * f0 + So * f1; M = Mn;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 o4 = *(const f32x4 *) (oth + 4 + db * 16 + 4 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc_o[db][4 * g + i] = acc_o[db][4 * g + i] * f0 + o4[i] * f1;
                }
        }
        __syncthreads();
    }
    if (a.nsplit > 1) {
        if (wave != 0 || !row_ok) return;
        S += __shfl_xor(S, 32, 64);                                    // (row_ok is the same in both lane halves)
        float * pr = a.part + ((((int64_t) is3 * a.nq + q) * a.nh + h) * a.nsplit + sp) * (D + 2);
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o4;
#pragma unroll
                for (int i = 0; i < 4; ++i) o4[i] = acc_o[db][4 * g + i];
                *(f32x4 *) (pr + db * 32 + 8 * g + 4 * hb) = o4;
            }
        if (hb == 0) { pr[D] = M * 0.6931471805599453f; pr[D + 1] = S; }
        return;
    }
    // ---- single slice: finish here -- sinks (ops.cpp:8116-8130), normalise, store permuted, optional Q8_K image of the rows
    float * fin = mrg;                                                 // [32 columns][D] finals (image epilogue)
    if (wave == 0) {
        S += __shfl_xor(S, 32, 64);
        float osc = 1.0f;
        if (a.sinks) {
            const float sk = a.sinks[h] * FM_LOG2E;
            if (sk > M) { const float f = __builtin_amdgcn_exp2f(M - sk); S = S * f + 1.0f; osc = f; }
            else S += __builtin_amdgcn_exp2f(sk - M);
        }
        const float inv = S == 0.0f ? 0.0f : osc / S;
        char * out = a.dst + h * a.dnb1 + q * a.dnb2 + is3 * a.dnb3;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o4;
#pragma unroll
                for (int i = 0; i < 4; ++i) o4[i] = acc_o[db][4 * g + i] * inv;
                if (row_ok) *(f32x4 *) (out + (db * 32 + 8 * g + 4 * hb) * 4) = o4;
                if (a.img) *(f32x4 *) (fin + lq * D + db * 32 + 8 * g + 4 * hb) = row_ok ? o4 : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
            }
    }
    if (a.img) {                                                       // (launcher: gq * D % 256 == 0; column lq = token lq / gq, head lq % gq)
        __syncthreads();
        const int per_q = a.gq * D / 256, nblk = a.qpw * per_q;
        const int64_t Kimg = (int64_t) a.nh * D;
        for (int bq = wave; bq < nblk; bq += NW) {
            const int qq = bq / per_q, bb = bq % per_q;
            const int64_t qrow = qc * a.qpw + qq;
            if (qrow >= a.nq) continue;
            const f32x4 v = *(const f32x4 *) (fin + (qq * a.gq) * D + bb * 256 + 4 * lane);
            char * im = a.img + (is3 * a.nq + qrow) * a.img_bytes;
            const int64_t ib = ((int64_t) ikv * a.gq * D) / 256 + bb;
            q8k_block_from_regs(v, lane, (int8_t *) im + ib * 256, (int16_t *) (im + Kimg) + ib * 16, (float *) (im + Kimg + Kimg / 8) + ib);
        }
    }
}

void flash_attn_ext_gqa(const fa_dev & a, int D, int nw, hipStream_t st) {
    const int nqc = (a.nq + a.qpw - 1) / a.qpw;
    const dim3 grid((unsigned) (a.nsplit * nqc * a.nhkv * a.ns));
    const bool pre = a.pre.qraw != nullptr;
#define GQA_GO(DD, NWW) do { if (pre) k_fattn_gqa<DD, NWW, true><<<grid, dim3(64 * NWW), 0, st>>>(a); else k_fattn_gqa<DD, NWW, false><<<grid, dim3(64 * NWW), 0, st>>>(a); } while (0)
    if (D == 64) { if (nw == 8) GQA_GO(64, 8); else GQA_GO(64, 4); }
    else         { if (nw == 8) GQA_GO(128, 8); else GQA_GO(128, 4); }
#undef GQA_GO
}

// ---- mask tile map: class of every 32 (q) x 32 (kv) tile of the f16 mask; one wave per tile
__global__ void __launch_bounds__(256) k_fattn_mask_map(const char * __restrict__ mask, int64_t mnb1, int64_t mnb2, int64_t mnb3, int mne2, int mne3,
                                                        int nq, int nkv, int nqb, int ntile, uint8_t * __restrict__ map) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (int64_t) mne3 * mne2 * nqb * ntile) return;
    int64_t r = wid;
    const int t  = (int) (r % ntile); r /= ntile;
    const int qb = (int) (r % nqb);   r /= nqb;
    const int i2 = (int) (r % mne2);  const int i3 = (int) (r / mne2);
    const int q = qb * 32 + (lane >> 1), kv0 = t * FM_KT + (lane & 1) * 16;
    bool live = false, nz = false;
    if (q < nq) {
        const uint16_t * row = (const uint16_t *) (mask + q * mnb1 + i2 * mnb2 + i3 * mnb3);
        if (kv0 + 15 < nkv && (((uintptr_t) (row + kv0)) & 15) == 0) {
            const u32x4 w0 = *(const u32x4 *) (row + kv0), w1 = *(const u32x4 *) (row + kv0 + 8);
            uint32_t d = 0, o = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { d |= (w0[i] ^ 0xfc00fc00u) | (w1[i] ^ 0xfc00fc00u); o |= w0[i] | w1[i]; }
            live = d != 0; nz = o != 0;
        } else {
            for (int i = 0; i < 16; ++i) {
                const uint32_t hv = kv0 + i < nkv ? (uint32_t) row[kv0 + i] : 0xfc00u;
                live |= hv != 0xfc00u; nz |= hv != 0;
            }
        }
    }
    const bool any_live = __any(live), any_nz = __any(nz);
    if (lane == 0) map[wid] = any_live ? (any_nz ? 2 : 1) : 0;
}

size_t fattn_map_bytes(int64_t nq, int64_t nkv, int64_t mne2, int64_t mne3) {
    return (size_t) (mne3 * mne2 * ((nq + 31) / 32) * ((nkv + FM_KT - 1) / FM_KT));
}
bool fattn_mma_ok(int64_t nkv) { return (nkv + FM_KT - 1) / FM_KT <= FM_MAXT; }

void fattn_mask_map(const fa_dev & a, uint8_t * map, hipStream_t st) {
    const int nqb = (a.nq + 31) / 32, ntile = (a.nkv + FM_KT - 1) / FM_KT;
    const int64_t nw = (int64_t) a.mne3 * a.mne2 * nqb * ntile;
    k_fattn_mask_map<<<dim3((unsigned) ((nw + 3) / 4)), dim3(256), 0, st>>>(a.mask, a.mnb1, a.mnb2, a.mnb3, (int) a.mne2, (int) a.mne3, a.nq, a.nkv, nqb, ntile, map);
}

static int  g_fd_mode = -1;                                        // -1: MI355X_FA_NO_DMA decides, 0 off, 1 on
static long g_fd_launches = 0;
void fattn_set_dma(int m) { g_fd_mode = m; }
long fattn_dma_launches() { return g_fd_launches; }

static int64_t fa_dma_min_wgs64() { static const int64_t v = getenv("MI355X_FA_BIG_MIN_WGS64") ? atoll(getenv("MI355X_FA_BIG_MIN_WGS64")) : 192; return v; }
static int64_t fa_dma_min_wgs() { static const int64_t v = getenv("MI355X_FA_BIG_MIN_WGS") ? atoll(getenv("MI355X_FA_BIG_MIN_WGS")) : 512; return v; }

template <int D>
static void launch_fm(const fa_dev & a, hipStream_t st) {
    const int nqt4 = (a.nq + 127) / 128;
    static const bool no_split = getenv("MI355X_FA_NO_KVSPLIT") != nullptr;
    static const int sq_env = getenv("MI355X_FA_SQ") ? atoi(getenv("MI355X_FA_SQ")) : 2;      // (tuning: tiles per wave between two barriers, 1 or 2)
    const int64_t blocks32 = (int64_t) ((a.nq + 31) / 32) * a.nh * a.ns;          // one wave each without a KV split
    if (a.nq <= 32) {
        k_fattn_mma<D, 1, 1, 1><<<dim3((unsigned) (a.nh * a.ns)), dim3(64), 0, st>>>(a, 1);
    } else if ((int64_t) nqt4 * a.nh * a.ns >= (D == 64 ? fa_dma_min_wgs64() : fa_dma_min_wgs())) {
        static const int abl = getenv("MI355X_FA_ABL") ? atoi(getenv("MI355X_FA_ABL")) : 0;
        static const bool env_no_dma = getenv("MI355X_FA_NO_DMA") != nullptr;
        const bool no_dma = g_fd_mode >= 0 ? g_fd_mode == 0 : env_no_dma;
        if constexpr (D == 64) {                                     // the encoders' shapes (Whisper 1500 x 1500 x 16 heads: 192 workgroups): the LDS-DMA ring form, one head per workgroup
            static const bool no_dma64 = getenv("MI355X_FA_NO_DMA64") != nullptr;
            if (!no_dma && !no_dma64 && !a.vt && a.nkv >= 128 && a.knb1 % 16 == 0 && a.vnb1 % 16 == 0 && a.knb2 % 16 == 0 && a.vnb2 % 16 == 0 && a.knb3 % 16 == 0 && a.vnb3 % 16 == 0 &&
                (((uintptr_t) a.k | (uintptr_t) a.v) & 15) == 0) {
                constexpr int lds64 = FD_NST * 2 * FM_KT * 2 * 64 + FM_MAXT / 8 + 4 * (FM_MAXT / 16) * 4, lds64x2 = lds64 + FD_NST * 2 * FM_KT * 2 * 64;
                static bool attr[64] = {};
                int dev = 0; HIP_CHECK(hipGetDevice(&dev));
                if (dev < 0 || dev >= 64 || !attr[dev]) {
                    HIP_CHECK(hipFuncSetAttribute((const void *) k_fattn_dma<64, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds64));
                    HIP_CHECK(hipFuncSetAttribute((const void *) k_fattn_dma<64, 0, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds64x2));
                    if (dev >= 0 && dev < 64) attr[dev] = true;
                }
                ++g_fd_launches;
                static const bool no_ks2 = getenv("MI355X_FA_DMA64_NO_KS2") != nullptr;
                const dim3 grid((unsigned) (nqt4 * a.nh * a.ns));
                static const int64_t ks2_max = getenv("MI355X_FA_DMA64_KS2_MAX_WGS") ? atoll(getenv("MI355X_FA_DMA64_KS2_MAX_WGS")) : 257;
                if (!no_ks2 && (int64_t) grid.x < ks2_max && a.nkv >= 512) k_fattn_dma<64, 0, 1, 2><<<grid, dim3(512), lds64x2, st>>>(a, nqt4);      // fewer workgroups than CUs: two waves per SIMD from a split of the keys
                else                                                   k_fattn_dma<64, 0, 1><<<grid, dim3(256), lds64, st>>>(a, nqt4);
                return;
            }
        }
        if constexpr (D == 128)
        if (!no_dma && !a.vt && a.nkv >= 128 && a.knb1 % 16 == 0 && a.vnb1 % 16 == 0 && a.knb2 % 16 == 0 && a.vnb2 % 16 == 0 && a.knb3 % 16 == 0 && a.vnb3 % 16 == 0 &&
            (((uintptr_t) a.k | (uintptr_t) a.v) & 15) == 0) {
            static bool attr[64] = {};
            int dev = 0; HIP_CHECK(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !attr[dev]) {
#define FD_ALL(F) F(0, 1) F(1, 1) F(4, 1) F(0, 2) F(1, 2) F(4, 2)
#define FD_ATTR(A, H) HIP_CHECK(hipFuncSetAttribute((const void *) k_fattn_dma<128, A, H>, hipFuncAttributeMaxDynamicSharedMemorySize, FD_LDS));
                FD_ALL(FD_ATTR)
                if (dev >= 0 && dev < 64) attr[dev] = true;
            }
            static const int hw_env = getenv("MI355X_FA_HW") ? atoi(getenv("MI355X_FA_HW")) : 2;
            const int hw = (hw_env >= 2 && a.gq % 2 == 0 && a.mne2 <= 1 && (int64_t) nqt4 * (a.nh / 2) * a.ns >= 256) ? 2 : 1;      // (fewer workgroups than CUs otherwise)      // pairs of heads of one KV head, one mask for every head
            const dim3 grid((unsigned) (nqt4 * (a.nh / hw) * a.ns));
            bool done = false;
#define FD_GO(A, H) if (!done && abl == A && hw == H) { k_fattn_dma<128, A, H><<<grid, dim3(256 * H), FD_LDS, st>>>(a, nqt4); done = true; }
            ++g_fd_launches;
            FD_ALL(FD_GO)
            if (!done) { if (hw == 2) k_fattn_dma<128, 0, 2><<<grid, dim3(512), FD_LDS, st>>>(a, nqt4); else k_fattn_dma<128, 0, 1><<<grid, dim3(256), FD_LDS, st>>>(a, nqt4); }
#undef FD_GO
#undef FD_ATTR
#undef FD_ALL
            return;
        }
        if (abl == 1)      k_fattn_mma<D, 4, 1, 2, 1><<<dim3((unsigned) (nqt4 * a.nh * a.ns)), dim3(256), 0, st>>>(a, nqt4);
        else if (abl == 2) k_fattn_mma<D, 4, 1, 2, 2><<<dim3((unsigned) (nqt4 * a.nh * a.ns)), dim3(256), 0, st>>>(a, nqt4);
        else if (abl == 3) k_fattn_mma<D, 4, 1, 2, 3><<<dim3((unsigned) (nqt4 * a.nh * a.ns)), dim3(256), 0, st>>>(a, nqt4);
        else if (abl == 6) k_fattn_mma<D, 4, 1, 2, 6><<<dim3((unsigned) (nqt4 * a.nh * a.ns)), dim3(256), 0, st>>>(a, nqt4);
        else if (abl == 7) k_fattn_mma<D, 4, 1, 2, 7><<<dim3((unsigned) (nqt4 * a.nh * a.ns)), dim3(256), 0, st>>>(a, nqt4);
        else if (sq_env >= 2 && a.nkv >= 128) k_fattn_mma<D, 4, 1, 2><<<dim3((unsigned) (nqt4 * a.nh * a.ns)), dim3(256), 0, st>>>(a, nqt4);
        else                             k_fattn_mma<D, 4, 1, 1><<<dim3((unsigned) (nqt4 * a.nh * a.ns)), dim3(256), 0, st>>>(a, nqt4);
    } else {
        const int nqt = (a.nq + 63) / 64;
        static const int ks_env = getenv("MI355X_FA_KS") ? atoi(getenv("MI355X_FA_KS")) : 4;
        static const bool want_stamps = getenv("MI355X_FA_STAMPS") != nullptr;
        if (want_stamps) {
            static unsigned long long * dst = nullptr; static int shown = 0;
            if (!dst) HIP_CHECK(hipMalloc(&dst, 32 * 8));
            fa_dev b2 = a; b2.stamps = dst;
            HIP_CHECK(hipMemsetAsync(dst, 0, 32 * 8, st));
            if (ks_env >= 4) k_fattn_mma<D, 2, 4, 1><<<dim3((unsigned) (nqt * a.nh * a.ns)), dim3(512), 0, st>>>(b2, nqt); else k_fattn_mma<D, 2, 2, 1><<<dim3((unsigned) (nqt * a.nh * a.ns)), dim3(256), 0, st>>>(b2, nqt);
            ++shown; if (shown == 100 || shown == 101 || shown == 230 || shown == 231) {
                unsigned long long h[32]; HIP_CHECK(hipStreamSynchronize(st)); HIP_CHECK(hipMemcpy(h, dst, sizeof(h), hipMemcpyDeviceToHost));
                fprintf(stderr, "[fa stamps] n=%llu (10 ns ticks after the first):", h[31]); for (int i = 1; i < (int) h[31] && i < 30; ++i) fprintf(stderr, " %.2f", (double) (h[i] - h[0]) / 100.0); fprintf(stderr, " us\n");
            }
            return;
        }
        static const int64_t ks4_max = getenv("MI355X_FA_KS4_MAX_BLOCKS") ? atoll(getenv("MI355X_FA_KS4_MAX_BLOCKS")) : 512;
        if (!no_split && ks_env >= 4 && blocks32 <= ks4_max && a.nkv >= 256) k_fattn_mma<D, 2, 4, 1><<<dim3((unsigned) (nqt * a.nh * a.ns)), dim3(512), 0, st>>>(a, nqt);   // at most one wave per two SIMDs otherwise: four waves per query block, a quarter of the KV range each
        else if (!no_split && blocks32 <= 768 && a.nkv >= 128) k_fattn_mma<D, 2, 2, 1><<<dim3((unsigned) (nqt * a.nh * a.ns)), dim3(256), 0, st>>>(a, nqt);   // fewer waves than SIMDs
        else if (sq_env >= 2 && a.nkv >= 128)            k_fattn_mma<D, 2, 1, 2><<<dim3((unsigned) (nqt * a.nh * a.ns)), dim3(128), 0, st>>>(a, nqt);
        else                                             k_fattn_mma<D, 2, 1, 1><<<dim3((unsigned) (nqt * a.nh * a.ns)), dim3(128), 0, st>>>(a, nqt);
    }
}

void flash_attn_ext_mma(const fa_dev & a, int D, hipStream_t st) {
    if (D == 64) launch_fm<64>(a, st); else launch_fm<128>(a, st);
}

} // namespace mi

// mmq_tile.hip -- MUL_MAT of Q4_K weights against a PREFILL ubatch (> 64 activation columns) on the int8 matrix cores of gfx950, straight from the
// quantised blocks: the tiled sibling of mmq.hip (<= 64 columns).
//
// reference arithmetic: ggml_compute_forward_mul_mat (ggml-cpu/ggml-cpu.c:1210-1402) with src1 quantised by quantize_row_q8_K (ggml-quants.c:2555-2592)
// and ggml_vec_dot_q4_K_q8_K (ggml-cpu/quants.c:550-623) per (row, column): per 256-block
//     d * yd * sum_j sc_j * (q4 . q8)_j  -  dmin * yd * sum_j m_j * bsum_j                     (8 sub-blocks of 32; get_scale_min_k4, ggml-quants.c:703-710)
// Every integer of that expression is reproduced exactly; only the f32 accumulation over the K/256 blocks is re-associated (as in every mat-vec kernel here).
// What the reference's GPU backend launches for the same node: mul_mat_q (ggml-cuda/mmq.cuh:3136, chosen at mmq.cu:402-470).
//
// How the 6-bit sub-block scale gets onto an int8 matrix core without a VALU pass over the accumulators (mmq.hip spends 16 v_mad_i32_i24 per MFMA on it,
// which is fine for <= 64 columns and VALU-bound beyond): sc_j = 8 * hi_j + lo_j with hi_j, lo_j in 0..7, and q4 * lo_j, q4 * hi_j <= 105 still fit a signed
// byte.  The weight operand of a sub-block is unpacked ONCE per wave (v_and / v_lshr), multiplied by lo_j and by hi_j as packed 16-bit lanes
// (v_pk_mul_lo_u16: no carry between the bytes) and fed to TWO v_mfma_i32_32x32x32_i8 that accumulate over the whole 256-block:
//     sum_j sc_j (q4 . q8)_j = 8 * ACC_hi + ACC_lo.
// That is the MFMA time of the F16 path (two 32x32x16 f16 MFMAs per 32 k) at 0.5625 instead of 2 bytes per weight and 1 instead of 2 bytes per activation
// through LDS, with nothing left of the resident F16 image for these tensors.
// The mins:  sum_j m_j * bsum_j  is ONE v_mfma_f32_32x32x16_f16 per 256-block: bsum_j = 64 h_j + l_j (l_j in 0..63), the activation image carries
// (64 h_0 .. 64 h_7, l_0 .. l_7) as f16 (exact: |64 h| <= 4096, a power-of-two multiple), the weight side m_0..m_7 twice; all products and partial sums are
// integers below 2^24, so the f32 result IS the integer.
//
// Staging: both operands by LDS-DMA (global_load_lds_dwordx4), one 256-block per stage, two stages:
//     W   the tile's 128 rows x ONE raw 144-byte block each, rows packed back to back in LDS ([row][144 B]: a row's 9 x 16-byte pieces are 9 consecutive
//         DMA lanes; fragment reads at a 36-dword row stride are conflict-free for ds_read_b128's 16-lane groups),
//     X   the tile's 128 tokens of the block-major activation image (quantize_q8k_tile_image): int8 [tok][256] with the 16-byte chunk c of a token stored
//         at chunk c ^ (tok & 15) (conflict-free ds_read_b128 of 32 tokens at one k), the f16 mins operand [tok][16] and the f32 block scale [tok];
//         all three are contiguous per (block, token tile): a linear copy.
// One __syncthreads() per 256-block publishes stage s and frees stage s - 1 for the DMA of block s + 1, which then has the 34 MFMAs per wave of a whole
// block (> 1 us) to land: twice the reach of the F16 kernels' one-K-step-ahead ring at a third of the bytes.
#include "../kernels.hpp"
#include <type_traits>

namespace mi {

typedef int   i32x4t  __attribute__((ext_vector_type(4)));
typedef int   i32x16t __attribute__((ext_vector_type(16)));
typedef float f32x16t __attribute__((ext_vector_type(16)));
typedef unsigned short u16x2t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void * lds_ptr_q;
typedef const __attribute__((address_space(1))) void * gbl_ptr_q;

constexpr int QT_ROWS = 128, QT_TOKS = 128;                       // QT_TOKS: the image's token padding; a workgroup tile is QT_ROWS x TT tokens, TT = 128 (8 waves) or 64 (4 waves, two workgroups per CU)
constexpr int QT_WB = QT_ROWS * 144, QT_XDB = 1024;               // (the scale DMA is one whole wave instruction: 256 floats, TT used)
constexpr int qt_lds(int TT) { return 2 * (QT_WB + TT * 256) + 3 * (TT * 32 + QT_XDB); }      // 117760 / 75776 bytes

struct mmqt_dev {
    const char * W[3]; size_t w_rs[3]; char * dst[3]; size_t dst_cs[3]; const char * resid[3]; size_t resid_cs[3]; int M[3]; int tm_end[3];
    int nmat;
    const char * xq; const char * xm; const char * xd;            // image sections (mmqt_image_*)
    int N, Npad, K, tiles_m, tiles_n, blocks_per_split; size_t split_stride;
    unsigned long long * dbg;                                     // ABL & 32: cycle stamps of workgroup 0 (tools/mmq_tile_bench.py)
};

extern __shared__ __attribute__((aligned(16))) char mmqt_lds[];

static __device__ __forceinline__ uint32_t pk_mul_u16(uint32_t a, uint32_t b) {
    u16x2t x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
    const u16x2t r = x * y; uint32_t o; __builtin_memcpy(&o, &r, 4); return o;
}

// ABL (MI355X_MMQT_ABL, tools/mmq_tile_bench.py): 1 = no DMA beyond the first two stages (results WRONG: what the kernel costs without its memory side), 32 = cycle stamps
// Two wave groups per workgroup (waves w and w + 4 share a SIMD): group 0 finishes a block -- mins MFMA, 8 hi + lo, scales -- right behind its MFMAs, group 1 at the START of
// the next block interval, so that one wave of every SIMD is in its VALU part (epilogue + header unpack, ~240 instructions) while the other one feeds the matrix core
// (32 MFMAs + the operand unpack): the interval is bound by the 2 x 34 MFMAs per SIMD instead of the sum of both parts.  The mins operand and the block scales
// therefore ride a ring of their own, three deep (group 1 reads block b - 1's while block b + 1's land).
template <int ABL, int TT>
__global__ void __launch_bounds__(TT * 4) k_mmq_tile_q4k(const mmqt_dev g) {
    constexpr int NW = TT / 16, QT_XQB = TT * 256, QT_XMB = TT * 32, QT_MAIN = QT_WB + QT_XQB, QT_AUX = QT_XMB + QT_XDB;      // main stage: W blocks + X quants (x 2); aux stage: mins operand + scales (x 3)
    char * const lds = mmqt_lds;
    // ---- which tile: consecutive tile ids (token tile fastest) go to the same XCD, so a W row tile is fetched once per XCD
    const int nt    = g.tiles_m * g.tiles_n;
    const int split = blockIdx.x / nt;
    const int bid   = blockIdx.x % nt;
    const int q8 = nt / 8, r8 = nt % 8, xcd = bid % 8, idx = bid / 8;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    int tm = tile / g.tiles_n; const int tn = tile % g.tiles_n;
    int mi = 0;
    if (g.nmat > 1 && tm >= g.tm_end[0]) { mi = 1; if (g.nmat > 2 && tm >= g.tm_end[1]) mi = 2; }
    tm -= mi == 0 ? 0 : g.tm_end[mi - 1];
    const char * const W = mi == 0 ? g.W[0] : (mi == 1 ? g.W[1] : g.W[2]);
    const size_t w_rs = mi == 0 ? g.w_rs[0] : (mi == 1 ? g.w_rs[1] : g.w_rs[2]);
    const int M = mi == 0 ? g.M[0] : (mi == 1 ? g.M[1] : g.M[2]);
    const int m0 = tm * QT_ROWS, n0 = tn * TT;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave & 3, wt = wave >> 2;                         // this wave: weight rows [32 wr, +32) x tokens [64 wt, +64) of the tile; wt = its group (TT = 64: one group)
    const int fr = lane & 31, hb = lane >> 5;

    // ---- DMA: per block NX + 18 + NM + 1 wave instructions of 1 KiB (X quants, W blocks, mins operand, scales), dealt round-robin to the waves.
    //      W: chunk c = 16-byte piece c % 9 of tile row c / 9
    constexpr int NX = TT / 4, NWI = QT_WB / 1024, NM = TT / 32, NDMA = NX + NWI + NM + 1, NU = (NDMA + NW - 1) / NW;
    const char * wsrc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int it = u * NW + wave;
        wsrc[u] = nullptr;
        if (it >= NX && it < NX + NWI) {
            const int c = (it - NX) * 64 + lane;
            int row = m0 + c / 9; row = row < M ? row : M - 1;
            wsrc[u] = W + (size_t) row * w_rs + (c % 9) * 16;
        }
    }
    const size_t tile_tok = (size_t) n0;
    char * const aux0 = lds + 2 * QT_MAIN;
    auto stage = [&](int buf, int abuf, int b) {
        char * const sb = lds + buf * QT_MAIN; char * const ab = aux0 + abuf * QT_AUX;
        const size_t xrow = (size_t) b * g.Npad + tile_tok;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int it = u * NW + wave;                          // (wave-uniform: scalar branches)
            if (it < NX)                 __builtin_amdgcn_global_load_lds((gbl_ptr_q) (g.xq + xrow * 256 + it * 1024 + lane * 16), (lds_ptr_q) (sb + QT_WB + it * 1024), 16, 0, 0);
            else if (it < NX + NWI)      __builtin_amdgcn_global_load_lds((gbl_ptr_q) (wsrc[u] + (size_t) b * 144), (lds_ptr_q) (sb + (it - NX) * 1024), 16, 0, 0);
            else if (it < NX + NWI + NM) __builtin_amdgcn_global_load_lds((gbl_ptr_q) (g.xm + xrow * 32 + (it - NX - NWI) * 1024 + lane * 16), (lds_ptr_q) (ab + (it - NX - NWI) * 1024), 16, 0, 0);
            else if (it < NDMA)          __builtin_amdgcn_global_load_lds((gbl_ptr_q) (g.xd + xrow * 4 + lane * 16), (lds_ptr_q) (ab + QT_XMB), 16, 0, 0);
        }
    };

    f32x16t out[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) out[a][e] = 0.0f;

    const int nblk = g.K >> 8;
    const int b_lo = split * g.blocks_per_split;
    const int b_hi = b_lo + g.blocks_per_split < nblk ? b_lo + g.blocks_per_split : nblk;
    const int wrow = wr * 32 + fr;                                   // this lane's weight row of the tile (B operand: column fr of the MFMA tile)
    const int tok0 = wt * 64 + fr;                                   // this lane's token of A tile 0 (A operand: row fr); tile 1: + 32 (same tok & 15)
    const int tsw = tok0 & 15;

    i32x16t alo[2], ahi[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) { alo[a][e] = 0; ahi[a][e] = 0; }
    // finish a block: out += yd * (d * (8 ACC_hi + ACC_lo) - dmin * mins), mins = ONE f16 MFMA on the mins operand of aux stage `abuf`
    auto finish = [&](int abuf, float d, float ndmin, const f16x8 bm) {
        const char * const xmb = aux0 + abuf * QT_AUX + tok0 * 32 + hb * 16;
        const float * const xdb = (const float *) (aux0 + abuf * QT_AUX + QT_XMB) + wt * 64 + 4 * hb;
        f16x8 am[2]; f32x4 y4r[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            am[a] = *(const f16x8 *) (xmb + a * 32 * 32);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) y4r[a][gq] = *(const f32x4 *) (xdb + a * 32 + 8 * gq);      // tokens (e & 3) + 8 (e >> 2) + 4 hb of A tile a
        }
        f32x16t mins[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            f32x16t z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.0f;
            mins[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[a], bm, z, 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float tf = (float) ((ahi[a][e] << 3) + alo[a][e]);
                out[a][e] = fmaf(y4r[a][e >> 2][e & 3], fmaf(ndmin, mins[a][e], d * tf), out[a][e]);
            }
    };

    unsigned long long tacc[5] = { 0, 0, 0, 0, 0 }, tprev = 0;
    auto stamp = [&](int k) { if (ABL & 32) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tacc[k] += t - tprev; tprev = t; } };
    if (ABL & 32) tprev = __builtin_amdgcn_s_memtime();
    float d_s = 0.0f, ndmin_s = 0.0f; f16x8 bm_s; int abuf_s = 0;    // group 1: the block it still owes its finish
#pragma unroll
    for (int j = 0; j < 8; ++j) bm_s[j] = (_Float16) 0.0f;
    if (b_lo < b_hi) stage(0, 0, b_lo);
    int abuf = 0;
    for (int b = b_lo; b < b_hi; ++b) {
        const int cur = (b - b_lo) & 1;
        stamp(4);
        __syncthreads();                                   // block b has landed (the fence drains the DMA queue), the other main stage and the aux stage of block b - 2 are free
        stamp(0);
        const int abuf_n = abuf == 2 ? 0 : abuf + 1;
        if (b + 1 < b_hi && (!(ABL & 1) || b < b_lo + 1)) stage(cur ^ 1, abuf_n, b + 1);
        if (wt == 1 && b > b_lo) finish(abuf_s, d_s, ndmin_s, bm_s);
        const char * const wb = lds + cur * QT_MAIN + wrow * 144;
        const char * const xqb = lds + cur * QT_MAIN + QT_WB + tok0 * 256;

        // ---- the row's block header: d, dmin, 12 bytes of 6-bit scales / mins (get_scale_min_k4)
        const u32x4 hdr = *(const u32x4 *) wb;
        const uint32_t s0 = hdr[1], s1 = hdr[2], s2 = hdr[3];
        // sc[0..3] = s0 bytes & 63, sc[4..7] = (s2 bytes & 15) | (s0 bytes >> 6) << 4 ; mn[0..3] = s1 bytes & 63, mn[4..7] = (s2 bytes >> 4) | (s1 bytes >> 6) << 4
        const uint32_t scA = s0 & 0x3f3f3f3fu, scB = (s2 & 0x0f0f0f0fu) | ((s0 >> 2) & 0x30303030u);
        const uint32_t mnA = s1 & 0x3f3f3f3fu, mnB = ((s2 >> 4) & 0x0f0f0f0fu) | ((s1 >> 2) & 0x30303030u);
        uint32_t lo2[8], hi2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t sc = ((j < 4 ? scA : scB) >> (8 * (j & 3))) & 0xffu;
            lo2[j] = (sc & 7u) * 0x00010001u; hi2[j] = (sc >> 3) * 0x00010001u;
        }
        f16x8 bm;                                                    // m_0 .. m_7 as f16 (both k halves of the mins MFMA take the same eight)
#pragma unroll
        for (int j = 0; j < 8; ++j) bm[j] = (_Float16) (float) (((j < 4 ? mnA : mnB) >> (8 * (j & 3))) & 0xffu);
        const float d = h2f((uint16_t) (hdr[0] & 0xffff)), ndmin = -h2f((uint16_t) (hdr[0] >> 16));

#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) { alo[a][e] = 0; ahi[a][e] = 0; }
        if (ABL & 32) { asm volatile("" :: "v"(lo2[7]), "v"(hi2[7]), "v"(bm)); stamp(1); }
        // A wave issues in order: four MFMAs back to back park it for 3 x 32 cycles, and the operand unpack behind them then runs with the matrix core idle.  So sub-block
        // j + 1's weight operand (unpack + the two scale multiplies, ~14 VALU) and its activation fragments are produced BETWEEN the four MFMAs of sub-block j
        // (sched_group_barrier: one MFMA, four VALU, one LDS read, four times), into the other half of two-deep register sets.
        u32x4 qsr[2], avr[2][2]; i32x4t blr[2], bhr[2];
        auto prep = [&](int j) {                                     // weight operand of sub-block j from qsr[(j >> 1) & 1]
            const u32x4 qs = qsr[(j >> 1) & 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t w4 = (qs[e] >> (4 * (j & 1))) & 0x0f0f0f0fu;
                blr[j & 1][e] = (int) pk_mul_u16(w4, lo2[j]); bhr[j & 1][e] = (int) pk_mul_u16(w4, hi2[j]);
            }
        };
        qsr[0] = *(const u32x4 *) (wb + 16 + 16 * hb);
#pragma unroll
        for (int a = 0; a < 2; ++a) avr[0][a] = *(const u32x4 *) (xqb + a * 32 * 256 + ((hb ^ tsw) << 4));
        prep(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j + 1 < 8) {
                if ((j & 1) == 0 && j + 2 < 8) qsr[((j >> 1) + 1) & 1] = *(const u32x4 *) (wb + 16 + ((j >> 1) + 1) * 32 + 16 * hb);      // (even j: the next pair's nibbles; sub-block j + 1 still reads this pair's)
#pragma unroll
                for (int a = 0; a < 2; ++a) avr[(j + 1) & 1][a] = *(const u32x4 *) (xqb + a * 32 * 256 + (((2 * (j + 1) + hb) ^ tsw) << 4));
                prep(j + 1);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const u32x4 av = avr[j & 1][a];
                i32x4t aa; aa[0] = (int) av[0]; aa[1] = (int) av[1]; aa[2] = (int) av[2]; aa[3] = (int) av[3];
                alo[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, blr[j & 1], alo[a], 0, 0, 0);
                ahi[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, bhr[j & 1], ahi[a], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);     // four VALU
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one LDS read
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ABL & 32) { asm volatile("" :: "v"(alo[1][15]), "v"(ahi[1][15])); stamp(2); }
        if (wt == 0) finish(abuf, d, ndmin, bm);
        else { d_s = d; ndmin_s = ndmin; bm_s = bm; abuf_s = abuf; }
        abuf = abuf_n;
    }
    if (wt == 1 && b_lo < b_hi) finish(abuf_s, d_s, ndmin_s, bm_s);

    if (ABL & 32) { asm volatile("" :: "v"(out[1][15])); stamp(3); if (blockIdx.x == 0 && lane == 0) for (int k = 0; k < 5; ++k) g.dbg[wave * 5 + k] = tacc[k]; }
    // ---- store: lane = weight row (col fr of the tile), registers = tokens; one 128-byte run per (token, half wave)
    char * dst = (mi == 0 ? g.dst[0] : (mi == 1 ? g.dst[1] : g.dst[2]));
    dst += (size_t) split * g.split_stride;
    const size_t dst_cs = mi == 0 ? g.dst_cs[0] : (mi == 1 ? g.dst_cs[1] : g.dst_cs[2]);
    const char * resid = mi == 0 ? g.resid[0] : (mi == 1 ? g.resid[1] : g.resid[2]);
    const size_t resid_cs = mi == 0 ? g.resid_cs[0] : (mi == 1 ? g.resid_cs[1] : g.resid_cs[2]);
    const int m = m0 + wrow;
    if (m >= M) return;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = n0 + wt * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * hb;
            if (n >= g.N) continue;
            float v = out[a][e];
            if (resid) v += *(const float *) (resid + (size_t) n * resid_cs + (size_t) m * 4);
            *(float *) (dst + (size_t) n * dst_cs + (size_t) m * 4) = v;
        }
}

// ------------------------------------------------------------------------------------------------ the block-major activation image
//   xq : [K/256][Npad][256] int8   -- the Q8_K quants of (block, token); 16-byte chunk c of a token row stored at chunk c ^ (token & 15)
//   xm : [K/256][Npad][16]  f16    -- (64 h_0 .. 64 h_7, l_0 .. l_7) with bsum_j = sum of the 32 quants of sub-block j = 64 h_j + l_j, 0 <= l_j < 64
//   xd : [K/256][Npad]      f32    -- the block scale d (+ 1 KiB of slack behind the last block: the scale DMA always moves 256 floats)
// Npad = N rounded up to 128 tokens (the rows past N are never written and never stored: whatever they hold only reaches outputs nobody keeps).
static inline int64_t qt_npad(int64_t N) { return (N + QT_TOKS - 1) / QT_TOKS * QT_TOKS; }
size_t mmqt_image_bytes(int64_t K, int64_t N) { return (size_t) (K / 256) * (size_t) qt_npad(N) * (256 + 32 + 4) + 1024; }

// one wave per (token, 256-block), lane l owns elements 4l .. 4l+3: quantize_row_q8_K_ref (ggml-quants.c:2555-2592) as q8k_block_from_regs (common.hpp) computes it
__global__ void __launch_bounds__(256) k_quantize_q8k_tile(const char * x, size_t xs, char * xq, char * xm, char * xd, int nblk, int N, int Npad) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= N * nblk) return;
    const int tok = wid / nblk, b = wid % nblk;
    const f32x4 v = *(const f32x4 *) (x + (size_t) tok * xs + (size_t) b * 1024 + lane * 16);
    float amax = fabsf(v[0]); float mval = v[0]; int idx = 4 * lane;
#pragma unroll
    for (int i = 1; i < 4; ++i) { const float a = fabsf(v[i]); if (a > amax) { amax = a; mval = v[i]; idx = 4 * lane + i; } }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float a2 = __shfl_xor(amax, o, 64); const float m2 = __shfl_xor(mval, o, 64); const int i2 = __shfl_xor(idx, o, 64);
        if (a2 > amax || (a2 == amax && i2 < idx)) { amax = a2; mval = m2; idx = i2; }
    }
    uint32_t packed = 0; int s = 0; float dd = 0.0f;
    if (amax != 0.0f) {
        const float iscale = -127.0f / mval;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float p = iscale * v[i];
            int r = (int) __builtin_rintf(p); r = r > 127 ? 127 : r;
            packed |= (uint32_t) (r & 0xff) << (8 * i); s += r;
        }
        dd = 1.0f / iscale;
    }
    const size_t row = (size_t) b * Npad + tok;
    *(uint32_t *) (xq + row * 256 + (((lane >> 2) ^ (tok & 15)) << 4) + 4 * (lane & 3)) = packed;
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);      // the 32 quants of sub-block lane / 8
    if ((lane & 7) == 0) {
        const int j = lane >> 3, h = s >> 6, l = s & 63;
        uint16_t * m16 = (uint16_t *) (xm + row * 32);
        m16[j] = f2h((float) (h * 64)); m16[8 + j] = f2h((float) l);
    }
    if (lane == 0) *(float *) (xd + row * 4) = dd;
}

void quantize_q8k_tile_image(const float * x, size_t xs, void * img, int64_t K, int64_t N, hipStream_t st) {
    const int nblk = (int) (K / 256); const int64_t Npad = qt_npad(N);
    char * xq = (char *) img, * xm = xq + (size_t) nblk * Npad * 256, * xd = xm + (size_t) nblk * Npad * 32;
    const int64_t waves = N * nblk;
    if (waves == 0) return;
    k_quantize_q8k_tile<<<dim3((unsigned) ((waves + 3) / 4)), dim3(256), 0, st>>>((const char *) x, xs, xq, xm, xd, nblk, (int) N, (int) Npad);
}

bool mmq_tile_ok(int type, int64_t K, const void * W, size_t w_rs) {
    return type == GGML_TYPE_Q4_K && K % 256 == 0 && K >= 256 && w_rs % 16 == 0 && ((uintptr_t) W & 15) == 0;
}

static int mmqt_tt() { return 128; }                          // token tile of a workgroup (8 waves: 4 row groups x 2 token groups)
// split the K blocks over workgroups when the tile grid leaves CUs without one (one workgroup per CU: 110 KB of LDS): wo / ffn_down at ubatch 512 are 128 tiles
int mmq_tile_ksplit(int64_t tiles, int64_t nblk) {
    static const int force = getenv("MI355X_MMQT_KSPLIT") ? atoi(getenv("MI355X_MMQT_KSPLIT")) : 0;
    if (force > 0) return force <= nblk ? force : (int) nblk;
    const int64_t slots = (int64_t) device_cu_count();      // resident workgroups: one per CU (115 KB of LDS)
    int s = 1;
    while (s < 8 && tiles * (s * 2) <= slots && nblk / (s * 2) >= 4) s *= 2;
    return s;
}

size_t mmq_tile_split_scratch_bytes(int64_t m_sum, int64_t N, int64_t K) {      // slabs a launch over m_sum rows (all matrices of a group) may ask for
    const int64_t tn = (N + mmqt_tt() - 1) / mmqt_tt(), tm = (m_sum + QT_ROWS - 1) / QT_ROWS;      // (a group's matrices round their own rows up: a few tiles more, a split at most as deep)
    const int s = mmq_tile_ksplit(tm * tn, K / 256);
    return s > 1 ? (size_t) s * (size_t) m_sum * (size_t) N * 4 : 0;
}

template <int ABL, int TT> static void mmqt_go_abl(const mmqt_dev & g, int grid, hipStream_t st) {
    static bool done[64] = {};
    int dev = 0; HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !done[dev]) { HIP_CHECK(hipFuncSetAttribute((const void *) k_mmq_tile_q4k<ABL, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, qt_lds(TT))); if (dev >= 0 && dev < 64) done[dev] = true; }
    k_mmq_tile_q4k<ABL, TT><<<dim3((unsigned) grid), dim3(TT * 4), qt_lds(TT), st>>>(g);
}
template <int TT> static void mmqt_go_tt(const mmqt_dev & g, int grid, hipStream_t st) {
    static const int abl = getenv("MI355X_MMQT_ABL") ? atoi(getenv("MI355X_MMQT_ABL")) : 0;
    switch (abl) {
        case 0:  mmqt_go_abl<0, TT>(g, grid, st); break;
        case 1:  mmqt_go_abl<1, TT>(g, grid, st); break;
        case 32: {
            static unsigned long long * dbg = nullptr; static int shown = 0;
            if (!dbg) HIP_CHECK(hipMalloc(&dbg, 40 * 8));
            mmqt_dev g2 = g; g2.dbg = dbg;
            mmqt_go_abl<32, TT>(g2, grid, st);
            if (shown++ < 3) {
                unsigned long long h[40]; HIP_CHECK(hipStreamSynchronize(st)); HIP_CHECK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
                const double nb = (double) g.blocks_per_split;
                for (int w = 0; w < TT / 16; w += 3) fprintf(stderr, "[mmqt stamps] wave %d, cycles per block: barrier wait %.0f | DMA issue + header + scales %.0f | 32 MFMAs + operands %.0f | mins + epilogue %.0f\n", w,
                                                   h[w * 5 + 0] / nb, h[w * 5 + 1] / nb, h[w * 5 + 2] / nb, h[w * 5 + 4] / nb);
            }
        } break;
        default: fprintf(stderr, "[mi355x] MI355X_MMQT_ABL=%d has no instance\n", abl); abort();
    }
}
static void mmqt_go(const mmqt_dev & g, int grid, hipStream_t st) { mmqt_go_tt<128>(g, grid, st); }
static long g_mmqt_launches = 0;
long mmq_tile_launches() { return g_mmqt_launches; }

void mmq_tile(const mmqt_args & a, hipStream_t st) {
    if (a.nmat < 1 || a.nmat > 3 || a.N < 1 || a.K % 256 != 0) { fprintf(stderr, "[mi355x] mmq_tile: bad shape\n"); abort(); }
    mmqt_dev g;
    const int nblk = (int) (a.K / 256); const int64_t Npad = qt_npad(a.N);
    g.xq = (const char *) a.img; g.xm = g.xq + (size_t) nblk * Npad * 256; g.xd = g.xm + (size_t) nblk * Npad * 32;
    g.N = (int) a.N; g.Npad = (int) Npad; g.K = (int) a.K; g.nmat = a.nmat;
    int tm = 0; int64_t m_sum = 0;
    for (int i = 0; i < 3; ++i) {
        const mmqt_mat & m = a.m[i < a.nmat ? i : 0];
        g.W[i] = (const char *) m.W; g.w_rs[i] = m.w_rs; g.dst[i] = (char *) m.dst; g.dst_cs[i] = m.dst_cs; g.resid[i] = (const char *) m.resid; g.resid_cs[i] = m.resid_cs; g.M[i] = (int) m.M;
        if (i < a.nmat) { tm += (int) ((m.M + QT_ROWS - 1) / QT_ROWS); m_sum += m.M; }
        g.tm_end[i] = tm;
    }
    g.tiles_m = tm; g.tiles_n = (int) ((a.N + mmqt_tt() - 1) / mmqt_tt()); g.dbg = nullptr;
    if (tm == 0) return;
    ++g_mmqt_launches;
    int ksplit = 1;
    bool m4 = true;
    for (int i = 0; i < a.nmat; ++i) m4 = m4 && a.m[i].M % 4 == 0 && a.m[i].dst_cs % 16 == 0;
    if (a.partial && m4) {
        ksplit = mmq_tile_ksplit((int64_t) tm * g.tiles_n, nblk);
        while (ksplit > 1 && (size_t) ksplit * (size_t) m_sum * (size_t) a.N * 4 > a.partial_bytes) ksplit /= 2;
    }
    g.blocks_per_split = (nblk + ksplit - 1) / ksplit; g.split_stride = 0;
    if (a.deferred_split) *a.deferred_split = 0;
    if (ksplit > 1) {
        // slab s of the scratch holds, matrix after matrix, the dense [N][M_i] partial sums of K range s (gemm_f16_multi's layout: the same reductions take them)
        const size_t slab = (size_t) m_sum * (size_t) a.N;
        size_t off = 0;
        for (int i = 0; i < a.nmat; ++i) { g.dst[i] = (char *) (a.partial + off); g.dst_cs[i] = (size_t) a.m[i].M * 4; g.resid[i] = nullptr; off += (size_t) a.m[i].M * (size_t) a.N; }
        g.split_stride = slab * 4;
        mmqt_go(g, tm * g.tiles_n * ksplit, st);
        if (a.deferred_split && (a.nmat == 1 || a.defer_multi)) { *a.deferred_split = ksplit; return; }
        if (a.nmat == 1) { gemm_reduce(a.partial, ksplit, a.m[0].resid, a.m[0].resid_cs, a.m[0].dst, a.m[0].dst_cs, a.m[0].M, a.N, st); return; }
        size_t offs[3] = { 0, 0, 0 }; int64_t Ms[3] = { 0, 0, 0 }; float * dsts[3] = { nullptr, nullptr, nullptr }; size_t css[3] = { 0, 0, 0 };
        off = 0;
        for (int i = 0; i < a.nmat; ++i) {
            if (a.m[i].resid) { fprintf(stderr, "[mi355x] mmq_tile: a grouped launch takes no residual\n"); abort(); }
            offs[i] = off; Ms[i] = a.m[i].M; dsts[i] = a.m[i].dst; css[i] = a.m[i].dst_cs; off += (size_t) a.m[i].M * (size_t) a.N;
        }
        gemm_reduce_group(a.partial, ksplit, slab, a.nmat, offs, Ms, a.N, dsts, css, st);
        return;
    }
    mmqt_go(g, tm * g.tiles_n, st);
}

} // namespace mi

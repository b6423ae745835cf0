// tools/lab/mmq_tile_w1.hip (LAB, not in the product build): the one-wave-per-SIMD organisation of mmq_tile.hip's kernel as measured in round 5 -- exact like the product
// kernel, slower (4096 x 4096 x 512: 36 us against 30; profiles/r05_mmq_tile.txt).  It uses mmq_tile.hip's mmqt_dev, constants and helpers: to revive it paste it back
// in front of the activation-image section (git show 01c97cf:llama.cpp-omni_amd/csrc/kernels/mmq_tile.hip has the launcher switch MI355X_MMQT_V).
// ---- the one-wave-per-SIMD form: 4 waves, each 32 weight rows x ALL 128 tokens of the tile (one weight operand per sub-block feeds 8 MFMAs instead of 4), 512
// registers per lane: two sets of block accumulators, so a block's finish (mins MFMA, 8 hi + lo, the three scales: ~60 VALU per 32 x 32 tile) is spread over the
// NEXT block's eight sub-blocks, between its MFMAs, next to the unpack of the following weight operand.  A wave issues in order, so everything that is not an MFMA has
// to sit in the 32-cycle gaps between two of them (sched_group_barrier: one MFMA, six VALU, one LDS read).
template <int ABL>
__global__ void __launch_bounds__(256) k_mmq_tile_q4k_w1(const mmqt_dev g) {
    constexpr int TT = 128, NW = 4, QT_XQB = TT * 256, QT_XMB = TT * 32, QT_MAIN = QT_WB + QT_XQB, QT_AUX = QT_XMB + QT_XDB;
    char * const lds = mmqt_lds;
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
    const int fr = lane & 31, hb = lane >> 5;

    // DMA per block: 32 wave instructions of X quants (8 per wave), 18 of W blocks (instruction 4 u + wave, u < 5), 4 of the mins operand (one per wave), 1 of scales (wave 0)
    uint32_t woff[5];                                              // (32-bit offsets from the matrix base: the launcher keeps M * w_rs below 4 GB for this form)
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int wi = u * 4 + wave, c = (wi < 18 ? wi : 17) * 64 + lane;
        int row = m0 + c / 9; row = row < M ? row : M - 1;
        woff[u] = (uint32_t) ((size_t) row * w_rs + (c % 9) * 16);
    }
    const size_t tile_tok = (size_t) n0;
    char * const aux0 = lds + 2 * QT_MAIN;
    auto stage = [&](int buf, int abuf, int b) {
        char * const sb = lds + buf * QT_MAIN; char * const ab = aux0 + abuf * QT_AUX;
        const size_t xrow = (size_t) b * g.Npad + tile_tok;
#pragma unroll
        for (int u = 0; u < 8; ++u) __builtin_amdgcn_global_load_lds((gbl_ptr_q) (g.xq + xrow * 256 + (u * 4 + wave) * 1024 + lane * 16), (lds_ptr_q) (sb + QT_WB + (u * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int u = 0; u < 5; ++u)
            if (u < 4 || wave < 2) __builtin_amdgcn_global_load_lds((gbl_ptr_q) (W + (size_t) b * 144 + woff[u]), (lds_ptr_q) (sb + (u * 4 + wave) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_q) (g.xm + xrow * 32 + wave * 1024 + lane * 16), (lds_ptr_q) (ab + wave * 1024), 16, 0, 0);
        if (wave == 0) __builtin_amdgcn_global_load_lds((gbl_ptr_q) (g.xd + xrow * 4 + lane * 16), (lds_ptr_q) (ab + QT_XMB), 16, 0, 0);
    };

    f32x16t out[4];
    i32x16t alo[4], ahi[4], tp[4];                                  // tp: 8 ACC_hi + ACC_lo of the block before, waiting for its scales in the gaps of the current one
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) { out[a][e] = 0.0f; alo[a][e] = 0; ahi[a][e] = 0; tp[a][e] = 0; }

    const int nblk = g.K >> 8;
    const int b_lo = split * g.blocks_per_split;
    const int b_hi = b_lo + g.blocks_per_split < nblk ? b_lo + g.blocks_per_split : nblk;
    const int wrow = wave * 32 + fr;
    const int tsw = fr & 15;

    // what the finish of the block before needs: its scales, its mins operand, its aux stage
    float d_p = 0.0f, ndmin_p = 0.0f; f16x8 bm_p; int abuf_p = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) bm_p[j] = (_Float16) 0.0f;
    f32x16t mins_p;                                                  // the mins tile being consumed
#pragma unroll
    for (int e = 0; e < 16; ++e) mins_p[e] = 0.0f;
    f32x4 y4p[2];

    // one eighth of the finish of the block before: tile a = sl >> 1, elements 8 (sl & 1) .. + 8
    auto finish_slice = [&](int sl) {
        const int a = sl >> 1, h = sl & 1;
        const float * const xdb = (const float *) (aux0 + abuf_p * QT_AUX + QT_XMB) + a * 32 + 4 * hb + 16 * h;      // tokens (e & 3) + 8 (e >> 2) + 4 hb, e = 8 h ..
        y4p[0] = *(const f32x4 *) (xdb); y4p[1] = *(const f32x4 *) (xdb + 8);
        if (h == 0) {
            const f16x8 am = *(const f16x8 *) (aux0 + abuf_p * QT_AUX + (a * 32 + fr) * 32 + hb * 16);
            f32x16t z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.0f;
            mins_p = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bm_p, z, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = 8 * h + i;
            const float tf = (ABL & 4) ? __int_as_float(tp[a][e] & 0x3fffffff) : (float) tp[a][e];
            out[a][e] = fmaf(y4p[i >> 2][i & 3], fmaf(ndmin_p, mins_p[e], d_p * tf), out[a][e]);
        }
    };

    unsigned long long tacc[5] = { 0, 0, 0, 0, 0 }, tprev = 0;
    auto stamp = [&](int k) { if (ABL & 32) { const unsigned long long tt = __builtin_amdgcn_s_memtime(); tacc[k] += tt - tprev; tprev = tt; } };
    if (ABL & 32) tprev = __builtin_amdgcn_s_memtime();

    int abuf = 0;
    // one block; the finish of the block before rides in its gaps when `owe`
    auto block = [&](int b, bool owe) {
        const int cur = (b - b_lo) & 1;
        stamp(4);
        __syncthreads();                                   // block b has landed, the other main stage and the aux stage of block b - 2 are free
        stamp(0);
        const int abuf_n = abuf == 2 ? 0 : abuf + 1;
        if (b + 1 < b_hi && (!(ABL & 1) || b < b_lo + 1)) stage(cur ^ 1, abuf_n, b + 1);
        const char * const wb = lds + cur * QT_MAIN + wrow * 144;
        const char * const xqb = lds + cur * QT_MAIN + QT_WB + fr * 256;

        const u32x4 hdr = *(const u32x4 *) wb;
        const uint32_t s0 = hdr[1], s1 = hdr[2], s2 = hdr[3];
        const uint32_t scA = s0 & 0x3f3f3f3fu, scB = (s2 & 0x0f0f0f0fu) | ((s0 >> 2) & 0x30303030u);
        const uint32_t mnA = s1 & 0x3f3f3f3fu, mnB = ((s2 >> 4) & 0x0f0f0f0fu) | ((s1 >> 2) & 0x30303030u);
        // lo / hi three bits of every scale, replicated into both 16-bit halves: one v_perm_b32 each (byte j of the source into bytes 0 and 2, zero elsewhere)
        const uint32_t loA = scA & 0x07070707u, hiA = (scA >> 3) & 0x07070707u, loB = scB & 0x07070707u, hiB = (scB >> 3) & 0x07070707u;
        uint32_t lo2[8], hi2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t sel = 0x0c000c00u | (uint32_t) (j & 3) | ((uint32_t) (j & 3) << 16);
            lo2[j] = __builtin_amdgcn_perm(0u, j < 4 ? loA : loB, sel); hi2[j] = __builtin_amdgcn_perm(0u, j < 4 ? hiA : hiB, sel);
        }
        f16x8 bm;
#pragma unroll
        for (int j = 0; j < 8; ++j) bm[j] = (_Float16) (float) (((j < 4 ? mnA : mnB) >> (8 * (j & 3))) & 0xffu);
        const float d = h2f((uint16_t) (hdr[0] & 0xffff)), ndmin = -h2f((uint16_t) (hdr[0] >> 16));

        u32x4 qsr[2], avr[4]; i32x4t blr[2], bhr[2];
        auto prep = [&](int j) {
            const u32x4 qs = qsr[(j >> 1) & 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t w4 = (qs[e] >> (4 * (j & 1))) & 0x0f0f0f0fu;
                blr[j & 1][e] = (int) pk_mul_u16(w4, lo2[j]); bhr[j & 1][e] = (int) pk_mul_u16(w4, hi2[j]);
            }
        };
        qsr[0] = *(const u32x4 *) (wb + 16 + 16 * hb);
#pragma unroll
        for (int a = 0; a < 4; ++a) avr[a] = *(const u32x4 *) (xqb + a * 32 * 256 + ((hb ^ tsw) << 4));
        prep(0);
        if (ABL & 32) { asm volatile("" :: "v"(blr[0]), "v"(bhr[0])); stamp(1); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j + 1 < 8) {
                if ((j & 1) == 0 && j + 2 < 8) qsr[((j >> 1) + 1) & 1] = *(const u32x4 *) (wb + 16 + ((j >> 1) + 1) * 32 + 16 * hb);
                if (!(ABL & 16)) prep(j + 1); else { blr[(j + 1) & 1] = blr[j & 1]; bhr[(j + 1) & 1] = bhr[j & 1]; }
            }
            if (owe && !(ABL & 2)) finish_slice(j);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                i32x4t aa; aa[0] = (int) avr[a][0]; aa[1] = (int) avr[a][1]; aa[2] = (int) avr[a][2]; aa[3] = (int) avr[a][3];
                if (j == 0) {                                        // (a fresh block: C = 0)
                    i32x16t z;
#pragma unroll
                    for (int e = 0; e < 16; ++e) z[e] = 0;
                    alo[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, blr[0], z, 0, 0, 0);
                    ahi[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, bhr[0], z, 0, 0, 0);
                } else {
                    alo[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, blr[j & 1], alo[a], 0, 0, 0);
                    ahi[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, bhr[j & 1], ahi[a], 0, 0, 0);
                }
                if (j + 1 < 8 && !(ABL & 8)) avr[a] = *(const u32x4 *) (xqb + a * 32 * 256 + (((2 * (j + 1) + hb) ^ tsw) << 4));      // (behind the two MFMAs that read the old one: six MFMAs ahead of its use)
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);     // six VALU
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one LDS read
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ABL & 32) { asm volatile("" :: "v"(alo[3][15]), "v"(ahi[3][15])); stamp(2); }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) tp[a][e] = (ahi[a][e] << 3) + alo[a][e];
        d_p = d; ndmin_p = ndmin; bm_p = bm; abuf_p = abuf;
        abuf = abuf_n;
    };

    if (b_lo < b_hi) stage(0, 0, b_lo);
    for (int b = b_lo; b < b_hi; ++b) block(b, b > b_lo);
    if (b_lo < b_hi) { for (int sl = 0; sl < 8; ++sl) finish_slice(sl); }      // the last block's finish, nothing to hide it under
    if (ABL & 32) { asm volatile("" :: "v"(out[3][15])); stamp(3); if (blockIdx.x == 0 && lane == 0) for (int k = 0; k < 5; ++k) g.dbg[wave * 5 + k] = tacc[k]; }

    char * dst = (mi == 0 ? g.dst[0] : (mi == 1 ? g.dst[1] : g.dst[2]));
    dst += (size_t) split * g.split_stride;
    const size_t dst_cs = mi == 0 ? g.dst_cs[0] : (mi == 1 ? g.dst_cs[1] : g.dst_cs[2]);
    const char * resid = mi == 0 ? g.resid[0] : (mi == 1 ? g.resid[1] : g.resid[2]);
    const size_t resid_cs = mi == 0 ? g.resid_cs[0] : (mi == 1 ? g.resid_cs[1] : g.resid_cs[2]);
    const int m = m0 + wrow;
    if (m >= M) return;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = n0 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * hb;
            if (n >= g.N) continue;
            float v = out[a][e];
            if (resid) v += *(const float *) (resid + (size_t) n * resid_cs + (size_t) m * 4);
            *(float *) (dst + (size_t) n * dst_cs + (size_t) m * 4) = v;
        }
}


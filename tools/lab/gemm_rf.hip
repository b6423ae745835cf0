// tools/lab/gemm_rf.hip (LAB, not in the product build since round 5: bit-identical to k_gemm_f16_glds<2> and measured not faster, profiles/r04_gemm_ring.txt).
// The kernel as it stood inside csrc/kernels/gemm.hip at commit 5985770 (it uses that file's gemm_dev, H_* constants and LDS helpers; to revive it paste it back
// below k_gemm_f16_glds and re-add the launcher branch `use_rf`, see git show 5985770:llama.cpp-omni_amd/csrc/kernels/gemm.hip).
// ---- the 128 x 128 x 64 tile with BOTH operands staged through REGISTERS, D K-steps ahead (k_gemm_f16_rf<D>).  What a CU can pull in per unit of time is
// bytes in flight / round trip (profiles/r04_gemm_streamk.txt): the LDS-DMA kernels above can only keep what their LDS ring holds in flight -- one K-step per workgroup
// here (2 x 32 KB with two workgroups per CU), against a ~2 us round trip for weight lines that come from HBM.  The register file is the bigger store: 512 KB per CU
// against 160 KB of LDS.  Every thread asks for its 8 x 16 bytes of K-step s + D with plain loads into a ring of D x 8 vector registers (D = 4: 128 KB per workgroup
// in flight, 256 KB per CU), and writes the registers of step s + 1 into the other LDS buffer after the MFMAs of step s (ds_write_b128 at the swizzled chunk position;
// loads return in order, so the wait the compiler places there leaves the younger D - 1 steps in flight).  One barrier per K-step, LDS double-buffered as before.
// Tile, wave layout, MFMA order, split-K slabs and epilogue are k_gemm_f16_glds<2>'s: results are bit-identical.
typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
template <int D>
__global__ void __launch_bounds__(256) k_gemm_f16_rf(const gemm_dev g) {
    static_assert(D == 2 || D == 4, "the ring depth is even (LDS buffer = ring slot & 1)");
    constexpr int MB = 2, BM = 64 * MB, WTILEB = BM * H_ROWB, BUFB = WTILEB + H_TILEB;
    char * const lds = gemm_lds;

    const int nt    = g.tiles_m * g.tiles_n;
    const int split = blockIdx.x / nt;
    const int bid   = blockIdx.x % nt;
    const int q = nt / 8, r = nt % 8, xcd = bid % 8, idx = bid / 8;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    int tm = tile / g.tiles_n; const int tn = tile % g.tiles_n;
    int mi = 0;
    if (g.nmat > 1 && tm >= g.tm_end[0]) { mi = 1; if (g.nmat > 2 && tm >= g.tm_end[1]) mi = 2; }
    tm -= mi == 0 ? 0 : g.tm_end[mi - 1];
    const char * const W = mi == 0 ? g.W[0] : (mi == 1 ? g.W[1] : g.W[2]);
    const size_t w_rs = mi == 0 ? g.w_rs[0] : (mi == 1 ? g.w_rs[1] : g.w_rs[2]);
    const int M = mi == 0 ? g.M[0] : (mi == 1 ? g.M[1] : g.M[2]);
    const int N = g.N;
    const int m0 = tm * BM, n0 = tn * G_BN;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    // staging: instruction j of wave w covers rows 32 w + 8 j + [0, 8) of the W tile and of the X tile, lane l the 16-byte chunk l & 7 of row l >> 3 (8 whole 128-byte
    // lines per instruction); in LDS the chunk sits at position (l & 7) ^ ((row >> 1) & 7) of its row (the fragment reads' swizzle)
    const int r8 = lane >> 3, c8 = lane & 7;
    uint32_t woff[4], xoff[4], loff[4];                                // (32-bit offsets from wave-uniform bases; launcher: M * w_rs, N * x_rs < 4 GB)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + j * 8 + r8;
        int mr = m0 + row; mr = mr < M ? mr : M - 1;
        int nr = n0 + row; nr = nr < N ? nr : N - 1;
        woff[j] = (uint32_t) ((size_t) mr * w_rs + c8 * 16);
        xoff[j] = (uint32_t) ((size_t) nr * g.x_rs + c8 * 16);
        loff[j] = (uint32_t) (row * H_ROWB + ((c8 ^ ((row >> 1) & 7)) << 4));
    }
    const int nk_all = g.K / H_BK;
    const int k_lo = split * g.ksteps_per_split;
    const int k_hi = k_lo + g.ksteps_per_split < nk_all ? k_lo + g.ksteps_per_split : nk_all;
    const int nsteps = k_hi - k_lo;
    u32x4g fw[D][4], fx[D][4];
    auto load = [&](auto SLc, int ks) {
        constexpr int SL = decltype(SLc)::value;
        const bool in = ks < k_hi;
        const char * const wb = in ? W + (size_t) ks * H_ROWB : W, * const xb = in ? g.X + (size_t) ks * H_ROWB : g.X;
#pragma unroll
        for (int j = 0; j < 4; ++j) { fx[SL][j] = *(const u32x4g *) (xb + (size_t) (in ? xoff[j] : 0u)); fw[SL][j] = *(const u32x4g *) (wb + (size_t) (in ? woff[j] : 0u)); }
    };
    auto store = [&](auto SLc) {                                       // ring slot SL -> LDS buffer SL & 1
        constexpr int SL = decltype(SLc)::value;
        char * const b = lds + (SL & 1) * BUFB;
#pragma unroll
        for (int j = 0; j < 4; ++j) { *(u32x4g *) (b + WTILEB + loff[j]) = fx[SL][j]; *(u32x4g *) (b + loff[j]) = fw[SL][j]; }
    };

    f16v acc[2][MB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    const int fr = lane & 31, hb = lane >> 5, sw = (fr >> 1) & 7;
    auto compute = [&](int buf) {
        const char * wb = lds + buf * BUFB; const char * xb = wb + WTILEB;
#pragma unroll
        for (int kk = 0; kk < H_BK / 16; ++kk) {
            const int co = ((kk * 2 + hb) ^ sw) << 4;
            h8 af[2], bf[MB];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = *(const h8 *) (xb + (wn * 64 + a * 32 + fr) * H_ROWB + co);
#pragma unroll
            for (int b = 0; b < MB; ++b) bf[b] = *(const h8 *) (wb + (wm * 32 * MB + b * 32 + fr) * H_ROWB + co);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < MB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    };
    // step i (ring slot i % D, LDS buffer i & 1): barrier -- its tile is visible and the other buffer is free --, ask for step i + D into the slot just consumed, the MFMAs,
    // then step i + 1's registers into the other buffer.  The loop is branch-free (launcher: the K range of a workgroup is a multiple of D steps): with conditions inside
    // -- or two forms of the prologue -- the compiler's s_waitcnt in front of the ds_writes must be safe on every path and becomes vmcnt(0): the ring drained once per
    // round.  Requests past the range's end go, all lanes alike, to the first 16 bytes of the operands (one line per instruction), their registers are never written out.
    auto step = [&](auto SLc, int i) {
        constexpr int SL = decltype(SLc)::value;
        __syncthreads();
        load(SLc, k_lo + i + D);
        __builtin_amdgcn_sched_barrier(0);                             // (the requests first: left alone, the scheduler puts the ds_writes -- and their wait -- in front of them)
        compute(SL & 1);
        __builtin_amdgcn_sched_barrier(0);
        store(std::integral_constant<int, (SL + 1) % D>());
    };
    load(std::integral_constant<int, 0>(), k_lo);
    load(std::integral_constant<int, 1>(), k_lo + 1);
    if (D > 2) { load(std::integral_constant<int, 2 % D>(), k_lo + 2); load(std::integral_constant<int, 3 % D>(), k_lo + 3); }
    __builtin_amdgcn_sched_barrier(0);
    store(std::integral_constant<int, 0>());
    for (int i = 0; i < nsteps; i += D) {
        step(std::integral_constant<int, 0>(), i);
        step(std::integral_constant<int, 1>(), i + 1);
        if (D > 2) { step(std::integral_constant<int, 2 % D>(), i + 2); step(std::integral_constant<int, 3 % D>(), i + 3); }
    }

    char * dst = (mi == 0 ? g.dst[0] : (mi == 1 ? g.dst[1] : g.dst[2])) + (size_t) split * g.split_stride;
    const size_t dst_cs = mi == 0 ? g.dst_cs[0] : (mi == 1 ? g.dst_cs[1] : g.dst_cs[2]);
    const char * resid = mi == 0 ? g.resid[0] : (mi == 1 ? g.resid[1] : g.resid[2]);
    const size_t resid_cs = mi == 0 ? g.resid_cs[0] : (mi == 1 ? g.resid_cs[1] : g.resid_cs[2]);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            const int m = m0 + wm * 32 * MB + b * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m < M && n < N) {
                    float v = acc[a][b][e];
                    if (resid) v += *(const float *) (resid + (size_t) n * resid_cs + (size_t) m * 4);
                    *(float *) (dst + (size_t) n * dst_cs + (size_t) m * 4) = v;
                }
            }
        }
}

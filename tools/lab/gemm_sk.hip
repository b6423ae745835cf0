// tools/lab/gemm_sk.hip (LAB, not in the product build since round 5: the persistent stream-K form of the F16 GEMM measured slower than the launch form inside the model,
// profiles/r04_gemm_streamk.txt).  The file as it stood at csrc/kernels/gemm_sk.hip in commit 5985770 (it needs that tree's kernels.hpp: gemm_multi_args::sk_part / sk_cnt and the
// gemm_f16_sk_ok() / gemm_f16_sk() hooks in gemm_f16_multi(); tools/lab/gemm_sk_check.py drove it through the option "gemm_sk").
// gemm_sk.hip -- the F16 GEMM on 256 (m) x 128 (n) tiles as ONE persistent, stream-K launch: a workgroup per CU walks an equal share of
// the launch's (tile, K-step) units, whatever the tile count is.
//
// Same product as gemm.hip (reference: ggml_compute_forward_mul_mat for F16 weights, src1 rounded to the F16 vec_dot_type, f32
// accumulation, ggml-cpu.c:1245-1268; what the reference's GPU backend hands to the vendor BLAS, ggml-cuda.cu:1211-1355):
//     dst[n][m] = sum_k W[m][k] * X[n][k] (+ resid[n][m])       W: M x K f16 rows, X: N x K f16 rows, dst f32
// STATUS: a measured lab, OFF by default (option "gemm_sk" / MI355X_GEMM_SK=1 turns it on wherever it is legal): parity with the launch form stand-alone,
// slower inside the model -- profiles/r04_gemm_streamk.txt has the numbers, the knock-outs and what was tried; DESIGN.md section 7 the reading.
// Why: a prefill ubatch of 512 tokens makes wq/wk/wv, wo and ffn_down launches of 128-192 tiles of 128 x 128 -- less than one round
// of the 512 workgroup slots, so gemm.hip split K 2-4 ways through a scratch of f32 slabs and a reduction launch (33 MB written and
// read back per wo / ffn_down).  Here the launch is cut into (tile row, 64-deep K-step) UNITS; the tiles_n workgroups of a GROUP walk
// the same run of units in lock step, one column tile each (they share every W line through their XCD's L2; ranges that cut the column
// tiles of a W panel at different K phases lose that and were 40 % slower), group q of Q = CUs / tiles_n takes units [q * upw, (q + 1) * upw):
// every CU does the same number of K-steps, a tile is shared by the few groups whose ranges meet inside it (four at wo, ubatch 512), and
// only those exchange a 128 KB partial tile.
//   * tile: what a CU can pull through its vector memory path into LDS is the ceiling here (measured: ~60 GB/s per CU, the MFMAs of a
//     128 x 128 x 64 step need 32 KB); 256 x 128 needs 48 KB for twice the MFMA work.  Eight waves, 4 (m) x 2 (n), 64 x 64 each.
//   * pipeline: a 3-slot ring of 48 KB stages (W tile + X tile of one K-step) filled by LDS-DMA (global_load_lds_dwordx4, the
//     source-side bank swizzle of gemm.hip) two K-steps ahead; one barrier per K-step behind a COUNTED s_waitcnt vmcnt(6) --
//     a K-step stays in flight across it, and the stream does not stop at tile or segment boundaries (the epilogue of a
//     finished tile runs under the next tile's loads).
//   * fold: a workgroup that holds only part of a tile's K range stores its accumulators to its own slot of `part` (16-byte
//     write-through stores, sc1), drains them and takes a ticket on the tile's counter; whoever arrives LAST adds the contributors'
//     tiles in contributor order (its own from registers -- the same bits it stored; the others by sc1 loads), so the result does
//     not depend on the arrival order, and resets the counter for the next launch.  No workgroup ever waits for another one: no
//     co-residency assumption, nothing to deadlock; no cache-wide release / acquire fence either (cdna_hip_programming.md G16).
//   * MFMA order inside a K-step and the C layout are those of k_gemm_f16_glds<2> (v_mfma_f32_32x32x16_f16, X supplies the rows
//     i = n, W the columns j = m of a 32 x 32 block); results differ from that kernel only by where a tile's K range is cut.
#include "../kernels.hpp"

namespace mi {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float    f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void * lds_ptr_t;
typedef const __attribute__((address_space(1))) void * gbl_ptr_t;

constexpr int SK_BM = 256, SK_BN = 128, SK_NT = 512;
constexpr int SK_BK = 64, SK_ROWB = SK_BK * 2, SK_WTILEB = SK_BM * SK_ROWB, SK_XTILEB = SK_BN * SK_ROWB, SK_BUFB = SK_WTILEB + SK_XTILEB, SK_NST = 3;
constexpr int SK_PART_FLOATS = SK_BM * SK_BN;                   // one partial tile

struct gemm_sk_dev {
    const char * W[3]; size_t w_rs[3]; char * dst[3]; size_t dst_cs[3]; const char * resid[3]; size_t resid_cs[3]; int M[3]; int tm_end[3];
    int nmat; const char * X; size_t x_rs; int N, K, tiles_n, tiles_m, nk, upw, total;      // nk: K-steps per tile; a unit = (tile row, K-step); upw: units per group of tiles_n workgroups; total = tiles_m * nk
    float * part; unsigned * cnt;
};

extern __shared__ __attribute__((aligned(16))) char sk_lds[];

// ABL (measurement only, MI355X_GEMM_SK_ABL): 1 no DMA inside the loop (stale operands), 2 no LDS reads / MFMAs, 64 every workgroup of an XCD streams the same tile (all L2 hits)
template <int ABL>
__global__ void __launch_bounds__(SK_NT) k_gemm_f16_sk(const gemm_sk_dev g) {
    char * const lds = sk_lds;
    unsigned * const s_ticket = (unsigned *) (lds + SK_NST * SK_BUFB);      // (behind the ring, in the dynamic region: a static would shift its 16-byte alignment)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave & 3, wn = wave >> 2;                        // (waves w and w + 4 share a SIMD: same W rows, the two X halves)
    const int G = (int) gridDim.x, b = (int) blockIdx.x;
    // workgroups b, b + 8, b + 16 .. share an XCD (round-robin placement; for speed only): give an XCD a contiguous run of units,
    // so that the workgroups streaming the same W panel (the tiles_n tiles of one tile row) sit behind one L2
    const int p = G % 8 == 0 ? (b % 8) * (G / 8) + b / 8 : b;
    const int grp = p / g.tiles_n, tn = p % g.tiles_n;              // the tiles_n workgroups of a group walk the same (tile row, K-step) units in step, one column tile each: they share every W line
    const int u_lo = grp * g.upw, u_hi = u_lo + g.upw < g.total ? u_lo + g.upw : g.total;
    if (u_lo >= u_hi) return;
    const int ns = u_hi - u_lo;

    // ---- DMA side: lane pointers of the tile the stream is in (wave w fills rows [32w, 32w + 32) of the W tile and [16w, 16w + 16) of the X
    // tile, 8 rows per instruction; 16w and 32w are multiples of 16, so the swizzle term (row >> 1) & 7 only depends on j and r8)
    const int r8 = lane >> 3;
    const char * wp[4]; const char * xp[2];
    auto tile_ptrs = [&](int tile) {
        if (ABL & 64) tile = b % 8;                                 // (64: every workgroup of an XCD streams the same tile row -- all L2 hits; request-path experiment)
        int tm = tile;
        int mi = 0;
        if (g.nmat > 1 && tm >= g.tm_end[0]) { mi = 1; if (g.nmat > 2 && tm >= g.tm_end[1]) mi = 2; }
        tm -= mi == 0 ? 0 : g.tm_end[mi - 1];
        const char * W = mi == 0 ? g.W[0] : (mi == 1 ? g.W[1] : g.W[2]);
        const size_t w_rs = mi == 0 ? g.w_rs[0] : (mi == 1 ? g.w_rs[1] : g.w_rs[2]);
        const int M = mi == 0 ? g.M[0] : (mi == 1 ? g.M[1] : g.M[2]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gc = (lane & 7) ^ (((j * 8 + r8) >> 1) & 7);      // source chunk that lands in LDS chunk (lane & 7) of that row
            int mr = tm * SK_BM + wave * 32 + j * 8 + r8; mr = mr < M ? mr : M - 1;
            wp[j] = W + (size_t) mr * w_rs + gc * 16;
            if (j < 2) {
                int nr = tn * SK_BN + wave * 16 + j * 8 + r8; nr = nr < g.N ? nr : g.N - 1;
                xp[j] = g.X + (size_t) nr * g.x_rs + gc * 16;
            }
        }
    };
    int ud = u_lo, tile_d = u_lo / g.nk, kd = u_lo % g.nk;
    tile_ptrs(tile_d);
    // one K-step of the stream into ring slot `slot` (6 instructions per wave).  Past the end of the range the last unit is loaded again,
    // into a slot nobody reads any more: the counted waits below stay the same for every step
    // ... issued in three parts (2 instructions each) between the k-sub-steps of the current K-step: a wave that issues its six requests back to back sits in
    // the vector-memory queue while its matrix pipe idles
    int dslot = 0;
    auto dma_part = [&](int part) {
        char * const sb = lds + dslot * SK_BUFB;
        const size_t ko = (size_t) kd * SK_ROWB;
        if (part < 2) {
#pragma unroll
            for (int j = 2 * part; j < 2 * part + 2; ++j) __builtin_amdgcn_global_load_lds((gbl_ptr_t) (wp[j] + ko), (lds_ptr_t) (sb + (wave * 32 + j * 8) * SK_ROWB), 16, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) __builtin_amdgcn_global_load_lds((gbl_ptr_t) (xp[j] + ko), (lds_ptr_t) (sb + SK_WTILEB + (wave * 16 + j * 8) * SK_ROWB), 16, 0, 0);
            if (ud + 1 < u_hi) { ++ud; if (++kd == g.nk) { kd = 0; ++tile_d; tile_ptrs(tile_d); } }
            dslot = dslot == SK_NST - 1 ? 0 : dslot + 1;
        }
    };
    auto dma_step = [&]() { dma_part(0); dma_part(1); dma_part(2); };

    f16v acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][bb][e] = 0.0f;
    };
    zero_acc();

    // ---- one finished 32 x 32 block (a, bb) of `tile` -> dst (+ resid); its addends are all read before its first store: dst may be the residual's own memory
    struct tile_out { int M, mb, nb; char * dst; size_t dst_cs; const char * resid; size_t resid_cs; };
    auto tile_out_of = [&](int tile) {
        int tm = tile;
        int mi = 0;
        if (g.nmat > 1 && tm >= g.tm_end[0]) { mi = 1; if (g.nmat > 2 && tm >= g.tm_end[1]) mi = 2; }
        tm -= mi == 0 ? 0 : g.tm_end[mi - 1];
        tile_out o;
        o.M = mi == 0 ? g.M[0] : (mi == 1 ? g.M[1] : g.M[2]);
        o.dst = mi == 0 ? g.dst[0] : (mi == 1 ? g.dst[1] : g.dst[2]);
        o.dst_cs = mi == 0 ? g.dst_cs[0] : (mi == 1 ? g.dst_cs[1] : g.dst_cs[2]);
        o.resid = mi == 0 ? g.resid[0] : (mi == 1 ? g.resid[1] : g.resid[2]);
        o.resid_cs = mi == 0 ? g.resid_cs[0] : (mi == 1 ? g.resid_cs[1] : g.resid_cs[2]);
        o.mb = tm * SK_BM + wm * 64 + (lane & 31); o.nb = tn * SK_BN + wn * 64 + 4 * (lane >> 5);
        return o;
    };
    auto store_block = [&](const tile_out & o, int a, int bb, const f16v & v) {
        const int m = o.mb + bb * 32;
        f16v r;
        if (o.resid) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = o.nb + a * 32 + (e & 3) + 8 * (e >> 2);
                r[e] = (m < o.M && n < g.N) ? *(const float *) (o.resid + (size_t) n * o.resid_cs + (size_t) m * 4) : 0.0f;
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = o.nb + a * 32 + (e & 3) + 8 * (e >> 2);
            if (m < o.M && n < g.N) *(float *) (o.dst + (size_t) n * o.dst_cs + (size_t) m * 4) = o.resid ? v[e] + r[e] : v[e];
        }
    };

    // ---- a segment [k_first, k_end) of `tile` is complete in acc
    auto finish = [&](int tile, int k_first, int k_end) {
        const tile_out o = tile_out_of(tile);
        if (k_first == 0 && k_end == g.nk) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) store_block(o, a, bb, acc[a][bb]);
            return;
        }
        const int u_t0 = tile * g.nk;
        const int p_first = u_t0 / g.upw, p_last = (u_t0 + g.nk - 1) / g.upw;      // (groups)
        auto slot_of = [&](int q) { return (uint32_t) (2 * (q * g.tiles_n + tn) + (q * g.upw < u_t0 ? 1 : 0)) * (uint32_t) (SK_PART_FLOATS * 4); };   // a range that begins before the tile: its last segment
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void *) g.part, (short) 0, (int) (2u * gridDim.x * (uint32_t) (SK_PART_FLOATS * 4)), 0x00020000);
        const uint32_t mine = slot_of(grp);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    union { f32x4 f; u32x4 u; } v; v.f = f32x4{ acc[a][bb][4 * q4], acc[a][bb][4 * q4 + 1], acc[a][bb][4 * q4 + 2], acc[a][bb][4 * q4 + 3] };
                    __builtin_amdgcn_raw_buffer_store_b128(v.u, prs, mine + (uint32_t) (((a * 2 + bb) * 4 + q4) * SK_NT + t) * 16u, 0, 16);      // sc1: written through
                }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every storing wave drains its stores, then one lane takes the ticket
        if (t == 0) *s_ticket = __hip_atomic_fetch_add(g.cnt + p_first * g.tiles_n + tn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned ticket = *s_ticket;
        if (ticket != (unsigned) (p_last - p_first)) return;
        // last to arrive: the contributors' tiles added in contributor order (deterministic whoever is last), a 32 x 32 block at a time; the other contributors'
        // blocks by sc1 loads (past the non-coherent caches), up to three contributors' requests in flight together
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                f16v tot;
                for (int q0 = p_first; q0 <= p_last; q0 += 3) {
                    u32x4 ld[3][4];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int q = q0 + j <= p_last ? q0 + j : p_last;
                        const uint32_t so = slot_of(q);
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) ld[j][q4] = __builtin_amdgcn_raw_buffer_load_b128(prs, so + (uint32_t) (((a * 2 + bb) * 4 + q4) * SK_NT + t) * 16u, 0, 16);
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int q = q0 + j;
                        if (q > p_last) break;
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            union { f32x4 f; u32x4 u; } v; v.u = ld[j][q4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float x = q == grp ? acc[a][bb][4 * q4 + i] : v.f[i];      // (its own: from registers -- the bits it stored)
                                tot[4 * q4 + i] = q == p_first ? x : tot[4 * q4 + i] + x;
                            }
                        }
                    }
                }
                store_block(o, a, bb, tot);
            }
        if (t == 0) __hip_atomic_store(g.cnt + p_first * g.tiles_n + tn, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next launch
    };

    const int fr = lane & 31, hb = lane >> 5, sw = (fr >> 1) & 7;
    const char * const xrow = lds + SK_WTILEB + (wn * 64 + fr) * SK_ROWB;
    const char * const wrow = lds + (wm * 64 + fr) * SK_ROWB;

    dma_step(); dma_step();
    int tile_c = u_lo / g.nk, kc = u_lo % g.nk, k_first = kc, slot_c = 0;
    for (int i = 0; i < ns; ++i) {
        // own requests of step i have landed (step i + 1 may still fly); after the barrier everybody's have, and everybody is past
        // the reads of step i - 1, whose slot takes step i + 2
        asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        const int so = slot_c * SK_BUFB;
        slot_c = slot_c == SK_NST - 1 ? 0 : slot_c + 1;
#pragma unroll
        for (int kk = 0; kk < ((ABL & 2) ? 0 : SK_BK / 16); ++kk) {
            const int co = ((kk * 2 + hb) ^ sw) << 4;
            h8 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = *(const h8 *) (xrow + so + a * 32 * SK_ROWB + co);
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) bf[bb] = *(const h8 *) (wrow + so + bb * 32 * SK_ROWB + co);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[bb], acc[a][bb], 0, 0, 0);
            if (!(ABL & 1) && kk < 3) { __builtin_amdgcn_sched_barrier(0); dma_part(kk); __builtin_amdgcn_sched_barrier(0); }
        }
        if ((ABL & 2) && !(ABL & 1)) dma_step();
        if (++kc == g.nk || i == ns - 1) {
            finish(tile_c, k_first, kc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the epilogue's own loads / stores are out of the counted window again
            zero_acc();
            kc = 0; k_first = 0; ++tile_c;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the dummy requests past the end of the range)
}

// ---- host side
static int g_sk_mode = -1;
void gemm_sk_set_mode(int m) { g_sk_mode = m; }
static int sk_cus() {
    static int cus[64] = {};
    int dev = 0; HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) dev = 0;
    if (!cus[dev]) { hipDeviceProp_t pr; HIP_CHECK(hipGetDeviceProperties(&pr, dev)); cus[dev] = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 1; }
    return cus[dev];
}
int gemm_sk_groups() { return sk_cus(); }
size_t gemm_sk_part_bytes()  { return (size_t) 2 * sk_cus() * SK_PART_FLOATS * 4; }      // two slots per workgroup (first / last segment of its range)
size_t gemm_sk_count_bytes() { return (size_t) sk_cus() * sizeof(unsigned); }             // a tile's counter is indexed by its first contributor

// the shapes the launch form handles badly: more than one column tile, fewer 128 x 128 tiles than two rounds of workgroup slots or a ragged last round
bool gemm_f16_sk_ok(const gemm_multi_args & a) {
    // OFF unless asked for: measured slower than the launch form at every prefill shape tried (profiles/r04_gemm_streamk.txt) --
    // MI355X_GEMM_SK=1 / option "gemm_sk": wherever it is legal (tests, measurements), 2: by the shape rule below
    static const int env = getenv("MI355X_GEMM_SK") ? atoi(getenv("MI355X_GEMM_SK")) : 0;
    const int force = g_sk_mode >= 0 ? g_sk_mode : env;
    if (force <= 0 || !a.sk_part || !a.sk_cnt || a.nmat <= 0 || a.nbatch > 1 || a.glu_out16 || a.K % SK_BK != 0 || a.N <= 0) return false;
    int64_t tm = 0;
    for (int i = 0; i < a.nmat; ++i) { if (a.m[i].qtype != 0 || a.m[i].resid2 || a.m[i].M <= 0) return false; tm += (a.m[i].M + SK_BM - 1) / SK_BM; }
    const int64_t tiles_n = (a.N + SK_BN - 1) / SK_BN, tiles = tm * tiles_n, nk = a.K / SK_BK, G = sk_cus();
    if (tiles * nk >= (int64_t) 1 << 30 || tiles_n > G) return false;
    if (force == 1) return true;
    if (a.N <= 128) return false;                                // (one column tile: the split-K launches with their two-addend reduction)
    return tiles * nk >= G * 8;                                  // at least eight K-steps per workgroup
}

void gemm_f16_sk(const gemm_multi_args & a, hipStream_t st) {
    gemm_sk_dev g;
    int tm = 0;
    for (int i = 0; i < 3; ++i) {
        const gemm_mat & m = a.m[i < a.nmat ? i : 0];
        g.W[i] = (const char *) m.W; g.w_rs[i] = m.w_rs; g.dst[i] = (char *) m.dst; g.dst_cs[i] = m.dst_cs;
        g.resid[i] = (const char *) m.resid; g.resid_cs[i] = m.resid_cs; g.M[i] = (int) m.M;
        if (i < a.nmat) tm += (int) ((m.M + SK_BM - 1) / SK_BM);
        g.tm_end[i] = tm;
    }
    g.nmat = a.nmat; g.X = (const char *) a.X; g.x_rs = a.x_rs; g.N = (int) a.N; g.K = (int) a.K;
    g.tiles_n = (int) ((a.N + SK_BN - 1) / SK_BN); g.nk = (int) (a.K / SK_BK);
    g.total = tm * g.nk; g.tiles_m = tm;
    int Q = sk_cus() / g.tiles_n;                                 // groups of tiles_n workgroups (gemm_f16_sk_ok: at least one)
    if (Q > g.total) Q = g.total;
    g.upw = (g.total + Q - 1) / Q;
    Q = (g.total + g.upw - 1) / g.upw;                            // (no empty groups)
    const int G = Q * g.tiles_n;
    g.part = a.sk_part; g.cnt = a.sk_cnt;
    constexpr int lds = SK_NST * SK_BUFB + 16;                     // 144 KB ring + the ticket word: one workgroup per CU
    static bool attr[64] = {};
    int dev = 0; HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr[dev]) {
#define SK_ALL(F) F(0) F(1) F(2) F(66)
#define SK_ATTR(A) HIP_CHECK(hipFuncSetAttribute((const void *) k_gemm_f16_sk<A>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        SK_ALL(SK_ATTR)
        if (dev >= 0 && dev < 64) attr[dev] = true;
    }
    static const int abl = getenv("MI355X_GEMM_SK_ABL") ? atoi(getenv("MI355X_GEMM_SK_ABL")) : 0;
    const dim3 block((unsigned) SK_NT);
    bool launched = false;
#define SK_GO(A) if (!launched && abl == A) { k_gemm_f16_sk<A><<<dim3((unsigned) G), block, lds, st>>>(g); launched = true; }
    SK_ALL(SK_GO)
    if (!launched) k_gemm_f16_sk<0><<<dim3((unsigned) G), block, lds, st>>>(g);
#undef SK_GO
#undef SK_ATTR
#undef SK_ALL
}

} // namespace mi

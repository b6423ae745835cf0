// gemm.hip -- dense F16 x F16 -> F32 GEMM on the CDNA4 matrix cores (prefill path, batches of more than 8 columns).
//
// Replaces what the reference sends to the vendor BLAS: ggml_cuda_op_mul_mat_cublas, ggml-cuda.cu:1211-1355 (F16 weights,
// activations cast F32 -> F16, FP32 accumulate and FP32 output on CDNA, :1293-1303), and is what the CPU oracle computes in
// ggml_compute_forward_mul_mat for F16 weights (src1 rounded to the F16 vec_dot_type, f32 accumulation; ggml-cpu.c:1245-1268).
//     dst[n][m] = sum_k W[m][k] * X[n][k]          W: M x K f16 rows, X: N x K f16 rows (K contiguous in both), dst f32
// Quantised weights reach these kernels as resident F16 images (shadow.cpp), or -- without an image -- as the K-quant blocks themselves,
// de-quantised inside the LDS staging (k_gemm_kq_glds).  Kernels in this file, by shape:
//     k_gemm_f16           K % 64 != 0 (K % 32 == 0): 128 x 128 x 32, register staging                       (below)
//     k_gemm_f16_glds<MB>  the general prefill kernel: 64 MB x 128 x 64, LDS-DMA staging, up to 3 matrices, residual / bias epilogue,
//                          deterministic split-K (grouped launches too at <= 256 columns), batch over heads
//     k_gemm_kq_glds       the same tile on Q4_K / Q6_K blocks de-quantised in the staging (weights without a resident image)
//     k_gemm_f16_glds256   256 x 256, one barrier per K-step (kept for A/B)
//     k_gemm_f16_ph8       256 x 256, eight-phase never-draining pipeline; <.., true>: ffn_gate / ffn_up + SWIGLU in one launch
//
// v_mfma_f32_32x32x16_f16: the A fragment of lane l is 8 consecutive k of row l%32 (k-octet l/32), the B fragment likewise --
// both operands are K-contiguous rows here, so each fragment is ONE 16-byte LDS read.  X supplies the rows (i) and W the columns
// (j) of every 32x32 C tile so that, per accumulator register, 32 lanes store 32 consecutive m (128 B, coalesced).
// Workgroup = 4 waves, tile 128 (m) x 128 (n) x 32 (k), each wave 64 x 64 = 2x2 MFMA tiles; LDS rows padded to 80 B
// (5 x 16 B: conflict-free for ds_read_b128); global -> register -> LDS staging, double-buffered (one barrier per K-step),
// next tile's global loads issued before the current tile's MFMAs.  blockIdx is remapped so that the workgroups sharing a
// weight panel run on the same XCD (private L2 per XCD).
#include "../kernels.hpp"
#include "act_dev.hpp"

namespace mi {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float    f16v __attribute__((ext_vector_type(16)));

constexpr int G_BM = 128, G_BN = 128, G_BK = 32, G_LD = G_BK + 8;    // LDS row = 40 halfs = 80 bytes

__global__ void __launch_bounds__(256) k_gemm_f16(const char * __restrict__ W, size_t w_rs, const char * __restrict__ X, size_t x_rs,
                                                  char * __restrict__ dst, size_t dst_cs, int M, int N, int K, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) _Float16 Ws[2][G_BM * G_LD];
    __shared__ __attribute__((aligned(16))) _Float16 Xs[2][G_BN * G_LD];

    // XCD-aware tile order: consecutive block ids go round-robin over the 8 XCDs; give each XCD a contiguous run of tiles
    const int nt  = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int q = nt / 8, r = nt % 8, xcd = bid % 8, idx = bid / 8;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;      // bijective for any nt
    const int tm = tile / tiles_n, tn = tile % tiles_n;                                   // n fastest: neighbours share the W panel
    const int m0 = tm * G_BM, n0 = tn * G_BN;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    // staging map: 128 rows x 4 chunks of 16 B per operand tile = 512 chunks, 2 per thread
    int srow[2], scol[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int c = t + 256 * i; srow[i] = c >> 2; scol[i] = c & 3; }
    const char * wp[2]; const char * xp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int mr = m0 + srow[i]; mr = mr < M ? mr : M - 1;
        int nr = n0 + srow[i]; nr = nr < N ? nr : N - 1;
        wp[i] = W + (size_t) mr * w_rs + scol[i] * 16;
        xp[i] = X + (size_t) nr * x_rs + scol[i] * 16;
    }

    f16v acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    u32x4 wreg[2], xreg[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { wreg[i] = *(const u32x4 *) wp[i]; xreg[i] = *(const u32x4 *) xp[i]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        *(u32x4 *) &Ws[0][srow[i] * G_LD + scol[i] * 8] = wreg[i];
        *(u32x4 *) &Xs[0][srow[i] * G_LD + scol[i] * 8] = xreg[i];
    }
    __syncthreads();

    const int nk = K / G_BK;
    const int fr = lane & 31, fk = (lane >> 5) * 8;
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wreg[i] = *(const u32x4 *) (wp[i] + (size_t) (ks + 1) * (G_BK * 2));
                xreg[i] = *(const u32x4 *) (xp[i] + (size_t) (ks + 1) * (G_BK * 2));
            }
        }
#pragma unroll
        for (int kk = 0; kk < G_BK / 16; ++kk) {
            h8 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = *(const h8 *) &Xs[cur][(wn * 64 + a * 32 + fr) * G_LD + kk * 16 + fk];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = *(const h8 *) &Ws[cur][(wm * 64 + b * 32 + fr) * G_LD + kk * 16 + fk];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (ks + 1 < nk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *(u32x4 *) &Ws[cur ^ 1][srow[i] * G_LD + scol[i] * 8] = wreg[i];
                *(u32x4 *) &Xs[cur ^ 1][srow[i] * G_LD + scol[i] * 8] = xreg[i];
            }
        }
        __syncthreads();
    }

    // C/D layout of the 32x32 tile: col j = lane%32 (-> m), row i = (reg&3) + 8*(reg>>2) + 4*(lane/32) (-> n)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + wm * 64 + b * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m < M && n < N) *(float *) (dst + (size_t) n * dst_cs + (size_t) m * 4) = acc[a][b][e];
            }
        }
}

// ---- K % 64 == 0: direct global -> LDS staging (global_load_lds_dwordx4), BK = 64, one barrier per K-step.
// An LDS-DMA instruction writes lane l's 16 bytes at (wave-uniform base) + 16*l, so a wave fills 8 consecutive unpadded 128-byte
// rows per instruction; bank conflicts are avoided on the SOURCE side instead of by padding: LDS chunk c of row r holds global
// chunk c ^ ((r >> 1) & 7), and the fragment reads apply the same XOR (a ds_read_b128 is served in groups of 16 lanes over 64 banks:
// two 128-byte rows, so the row's parity picks the bank half and (r >> 1) & 7 spreads the 8 chunk slots -- conflict-free).  Tile t+1's DMA is issued right after the barrier that publishes
// tile t, so it runs under tile t's 16 MFMAs per wave.
typedef __attribute__((address_space(3))) void * lds_ptr_t;
typedef const __attribute__((address_space(1))) void * gbl_ptr_t;
constexpr int H_BK = 64, H_ROWB = H_BK * 2, H_TILEB = 128 * H_ROWB;          // 16 KB per operand tile

// One launch serves up to three matrices that share the activation X (wq / wk / wv, ffn_gate / ffn_up): their M tiles are simply
// concatenated, which is what fills the 256 CUs at ubatch 512 (wk alone is 32 tiles).  Optional epilogue: dst = acc + resid (the
// residual ADD that follows wo / ffn_down).  Split-K: blockIdx / n_tiles selects a K range and the partial sums go to a dense
// scratch slab per split (k_gemm_reduce adds the slabs -- and the residual -- in a fixed order: deterministic, no atomics).
struct gemm_dev {
    const char * W[3]; size_t w_rs[3]; char * dst[3]; size_t dst_cs[3]; const char * resid[3]; size_t resid_cs[3]; int M[3]; int tm_end[3];
    int nmat; const char * X; size_t x_rs; int N, K, tiles_m, tiles_n, ksteps_per_split; size_t split_stride;
    // broadcast batch over blockIdx.y (attention without FLASH_ATTN_EXT: one K / V^T matrix per KV head, one activation block per query
    // head): batch b = i13 * ne12 + i12 uses W + (i12 / r2) * w_nb2 + (i13 / r3) * w_nb3, X + b * x_bs, dst + i12 * dst_nb2 + i13 * dst_nb3
    int ne12, r2, r3; size_t w_nb2, w_nb3, x_bs, dst_nb2, dst_nb3;
    unsigned long long * dbg;
    int wtype[3];                                    // k_gemm_kq_glds: GGML_TYPE_Q4_K / GGML_TYPE_Q6_K per matrix (W points at the block rows)
    char * out16; size_t out16_rs; int glu_gate;      // k_gemm_f16_ph8<.., true>: f16 rows of silu(W[glu_gate].x) * (W[1 - glu_gate].x)
    char * y16[3] = { nullptr, nullptr, nullptr }; size_t y16_rs[3] = { 0, 0, 0 }, y16_ms[3] = { 2, 2, 2 };   // k_gemm_f16_glds<MB>, un-split: f16 copy of matrix i's result, element (m, n) at y16 + m * ms + n * rs (dst null: only that)
};

// MB = 32-row MFMA tiles per wave along m: the workgroup tile is (64*MB) x 128.  MB = 2 is the default; MB = 3 (192 x 128) is chosen
// by the launcher when it removes a partially filled round of workgroups (ffn_gate+ffn_up at ubatch 512: 768 tiles of 128 rows are
// 1.5 rounds of the 512 resident workgroups, 512 tiles of 192 rows are exactly one).
extern __shared__ __attribute__((aligned(16))) char gemm_lds[];
template <int MB>
__global__ void __launch_bounds__(256) k_gemm_f16_glds(const gemm_dev g) {
    constexpr int BM = 64 * MB, WTILEB = BM * H_ROWB, BUFB = WTILEB + H_TILEB;  // per buffer: W tile then X tile
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
    const int bi12 = blockIdx.y % g.ne12, bi13 = blockIdx.y / g.ne12;
    const char * W = (mi == 0 ? g.W[0] : (mi == 1 ? g.W[1] : g.W[2])) + (size_t) (bi12 / g.r2) * g.w_nb2 + (size_t) (bi13 / g.r3) * g.w_nb3;
    const size_t w_rs = mi == 0 ? g.w_rs[0] : (mi == 1 ? g.w_rs[1] : g.w_rs[2]);
    const int M = mi == 0 ? g.M[0] : (mi == 1 ? g.M[1] : g.M[2]);
    const int N = g.N;
    const int m0 = tm * BM, n0 = tn * G_BN;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    // staging: wave w fills rows [16*MB*w, +16*MB) of the W tile and [32w, 32w+32) of the X tile, 8 rows per instruction
    // (16*MB*w and 32w are multiples of 16, so the swizzle term (row >> 1) & 7 only depends on j and r8)
    const int r8 = lane >> 3;
    const char * wp[2 * MB]; const char * xp[4];
#pragma unroll
    for (int j = 0; j < 2 * MB; ++j) {
        const int gc = (lane & 7) ^ (((j * 8 + r8) >> 1) & 7);       // source chunk that lands in LDS chunk (lane & 7) of that row
        int mr = m0 + wave * 16 * MB + j * 8 + r8; mr = mr < M ? mr : M - 1;
        wp[j] = W + (size_t) mr * w_rs + gc * 16;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gc = (lane & 7) ^ (((j * 8 + r8) >> 1) & 7);
        int nr = n0 + wave * 32 + j * 8 + r8; nr = nr < N ? nr : N - 1;
        xp[j] = g.X + (size_t) blockIdx.y * g.x_bs + (size_t) nr * g.x_rs + gc * 16;
    }
    auto stage_w = [&](int buf, int ks, int j) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t) (wp[j] + (size_t) ks * H_ROWB), (lds_ptr_t) (lds + buf * BUFB + (wave * 16 * MB + j * 8) * H_ROWB), 16, 0, 0);
    };
    auto stage_x = [&](int buf, int ks, int j) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t) (xp[j] + (size_t) ks * H_ROWB), (lds_ptr_t) (lds + buf * BUFB + WTILEB + (wave * 32 + j * 8) * H_ROWB), 16, 0, 0);
    };
    // quarter q (0..3) of the next tile's DMA: issued between the k-sub-steps so that the LDS reads and MFMAs of the current tile start
    // right after the barrier instead of behind all of the wave's DMA issues
    auto stage_part = [&](int buf, int ks, int q) {
        stage_x(buf, ks, q);
        if (q < 2 * MB) stage_w(buf, ks, q);
        if (4 + q < 2 * MB) stage_w(buf, ks, 4 + q);
    };
    auto stage = [&](int buf, int ks) {
#pragma unroll
        for (int q = 0; q < 4; ++q) stage_part(buf, ks, q);
    };

    f16v acc[2][MB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    const int nk_all = g.K / H_BK;
    const int k_lo = split * g.ksteps_per_split;
    const int k_hi = k_lo + g.ksteps_per_split < nk_all ? k_lo + g.ksteps_per_split : nk_all;
    const int fr = lane & 31, hb = lane >> 5, sw = (fr >> 1) & 7;
    stage(0, k_lo);
    for (int ks = k_lo; ks < k_hi; ++ks) {
        const int cur = (ks - k_lo) & 1;
        __syncthreads();                                   // tile ks has landed (the fence drains the DMA), buffer cur^1 is free
        if (ks + 1 < k_hi) stage(cur ^ 1, ks + 1);          // (interleaving the DMA issues with the k-sub-steps pays on the 256-square kernel only: two workgroups per CU already overlap here)
        const char * wb = lds + cur * BUFB; const char * xb = wb + WTILEB;
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
    }

    char * dst = (mi == 0 ? g.dst[0] : (mi == 1 ? g.dst[1] : g.dst[2]));
    if (dst) dst += (size_t) split * g.split_stride + (size_t) bi12 * g.dst_nb2 + (size_t) bi13 * g.dst_nb3;
    const size_t dst_cs = mi == 0 ? g.dst_cs[0] : (mi == 1 ? g.dst_cs[1] : g.dst_cs[2]);
    const char * resid = mi == 0 ? g.resid[0] : (mi == 1 ? g.resid[1] : g.resid[2]);
    const size_t resid_cs = mi == 0 ? g.resid_cs[0] : (mi == 1 ? g.resid_cs[1] : g.resid_cs[2]);
    // (un-split launches: an f16 copy of the rows -- the CAST behind a K / V projection of an encoder -- element (m, n) at y16 + m * ms + n * rs; dst null: nobody reads the f32 rows)
    char * const y16 = mi == 0 ? g.y16[0] : (mi == 1 ? g.y16[1] : g.y16[2]);
    const size_t y16_rs = mi == 0 ? g.y16_rs[0] : (mi == 1 ? g.y16_rs[1] : g.y16_rs[2]), y16_ms = mi == 0 ? g.y16_ms[0] : (mi == 1 ? g.y16_ms[1] : g.y16_ms[2]);
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
                    if (y16) *(uint16_t *) (y16 + (size_t) n * y16_rs + (size_t) m * y16_ms) = f2h(v);
                    if (dst) *(float *) (dst + (size_t) n * dst_cs + (size_t) m * 4) = v;
                }
            }
        }
}


// ---- K-quant weights de-quantised INSIDE the LDS staging (few columns: short prompts, omni stream_prefill chunks).  At N <= 256 the F16 GEMM
// above is bound by streaming the resident F16 weight images (2 B per weight: 386 MB per Qwen3-8B layer); here the workgroup reads the Q4_K /
// Q6_K blocks themselves (0.5625 / 0.8203 B per weight) and turns each K-step's slice into the same f16 tile the F16 kernel would have staged:
// thread (row r = t / 2, half h = t % 2) fetches its 16-byte piece(s) and the block header one K-step ahead (registers), and after the step's
// MFMAs de-quantises them -- the reference's arithmetic (dequantize_row_q4_K / _q6_K, ggml-quants.c:1352-1374 / 1762-1791: d * sc * q - dmin * m,
// d * sc * q, then the f32 -> f16 rounding of the weight image) -- into 32 halves = four ds_write_b128 at the swizzled chunk positions.
//   Q4_K, K-step ks: super-block ks / 4, 64-weight group j = ks % 4: bytes qs[32 j + 16 h .. +16); low nibbles = weights 16 h + i of the step
//                    (sub-block 2 j), high nibbles = weights 32 + 16 h + i (sub-block 2 j + 1)
//   Q6_K, K-step ks: half n = (ks / 2) % 2, e = ks % 2: ql[64 n + 16 h + i], ql[64 n + 32 + 16 h + i], qh[32 n + 16 h + i]; e picks the nibble and
//                    the qh bit pair; scales[8 n + h + 4 e] and [.. + 2]  (blocks are only 2-byte aligned: dword loads, the hardware handles the phase)
// The activation tile still arrives by LDS-DMA.  Same tile, wave layout, MFMA order and split-K as k_gemm_f16_glds<2>: results are bit-identical
// to the F16-image path.
static __device__ __forceinline__ uint32_t ld_u32_a2(const char * p) { typedef uint32_t __attribute__((aligned(2))) u32a2; return *(const u32a2 *) p; }
static __device__ __forceinline__ void kq_scale_min(int s, const uint8_t * q, int & sc, int & m) {   // get_scale_min_k4, ggml-quants.c:703-710
    if (s < 4) { sc = q[s] & 63; m = q[s + 4] & 63; }
    else       { sc = (q[s + 4] & 0xF) | ((q[s - 4] >> 6) << 4); m = (q[s + 4] >> 4) | ((q[s] >> 6) << 4); }
}

__global__ void __launch_bounds__(256) k_gemm_kq_glds(const gemm_dev g) {
    constexpr int BM = 128, WTILEB = BM * H_ROWB, BUFB = WTILEB + H_TILEB;
    char * const lds = gemm_lds;

    const int nt    = g.tiles_m * g.tiles_n;
    const int split = blockIdx.x / nt;
    const int bid   = blockIdx.x % nt;
    const int q = nt / 8, rr = nt % 8, xcd = bid % 8, idx = bid / 8;
    const int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    int tm = tile / g.tiles_n; const int tn = tile % g.tiles_n;
    int mi = 0;
    if (g.nmat > 1 && tm >= g.tm_end[0]) { mi = 1; if (g.nmat > 2 && tm >= g.tm_end[1]) mi = 2; }
    tm -= mi == 0 ? 0 : g.tm_end[mi - 1];
    const char * W = mi == 0 ? g.W[0] : (mi == 1 ? g.W[1] : g.W[2]);
    const size_t w_rs = mi == 0 ? g.w_rs[0] : (mi == 1 ? g.w_rs[1] : g.w_rs[2]);
    const int M = mi == 0 ? g.M[0] : (mi == 1 ? g.M[1] : g.M[2]);
    const bool q4 = (mi == 0 ? g.wtype[0] : (mi == 1 ? g.wtype[1] : g.wtype[2])) == GGML_TYPE_Q4_K;      // (workgroup-uniform)
    const int N = g.N;
    const int m0 = tm * BM, n0 = tn * G_BN;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    // activation tile: LDS-DMA exactly as k_gemm_f16_glds
    const int r8 = lane >> 3;
    const char * xp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gc = (lane & 7) ^ (((j * 8 + r8) >> 1) & 7);
        int nr = n0 + wave * 32 + j * 8 + r8; nr = nr < N ? nr : N - 1;
        xp[j] = g.X + (size_t) nr * g.x_rs + gc * 16;
    }
    auto stage_x = [&](int buf, int ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (xp[j] + (size_t) ks * H_ROWB), (lds_ptr_t) (lds + buf * BUFB + WTILEB + (wave * 32 + j * 8) * H_ROWB), 16, 0, 0);
    };
    // weight tile: this thread's row and half
    const int r = t >> 1, h = t & 1;
    const char * wrow = W + (size_t) (m0 + r < M ? m0 + r : M - 1) * w_rs;
    uint32_t qreg[12], hreg[4];
    auto fetch_w = [&](int ks) {
        if (q4) {
            const char * blk = wrow + (size_t) (ks >> 2) * 144;
            const u32x4 hd = *(const u32x4 *) blk, qq = *(const u32x4 *) (blk + 16 + 32 * (ks & 3) + 16 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) { hreg[i] = hd[i]; qreg[i] = qq[i]; }
        } else {
            const char * blk = wrow + (size_t) (ks >> 2) * 210;
            const int n = (ks >> 1) & 1, e = ks & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                qreg[i]     = ld_u32_a2(blk + 64 * n + 16 * h + 4 * i);
                qreg[4 + i] = ld_u32_a2(blk + 64 * n + 32 + 16 * h + 4 * i);
                qreg[8 + i] = ld_u32_a2(blk + 128 + 32 * n + 16 * h + 4 * i);
            }
            hreg[0] = (uint32_t) (uint8_t) blk[192 + 8 * n + h + 4 * e];
            hreg[1] = (uint32_t) (uint8_t) blk[192 + 8 * n + h + 4 * e + 2];
            hreg[2] = (uint32_t) *(const uint16_t *) (blk + 208);
        }
    };
    auto park_w = [&](int buf, int ks) {
        float a[16], b[16];                                  // weights 16 h + i and 32 + 16 h + i of this K-step
        if (q4) {
            uint8_t sc12[12];
            __builtin_memcpy(sc12, &hreg[1], 12);
            int sca, ma, scb, mb;
            kq_scale_min(2 * (ks & 3), sc12, sca, ma); kq_scale_min(2 * (ks & 3) + 1, sc12, scb, mb);
            const float d = h2f((uint16_t) (hreg[0] & 0xffffu)), dmin = h2f((uint16_t) (hreg[0] >> 16));
            const float d1 = d * (float) sca, m1 = dmin * (float) ma, d2 = d * (float) scb, m2 = dmin * (float) mb;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t by = (qreg[i >> 2] >> (8 * (i & 3))) & 0xffu;
                a[i] = d1 * (float) (int) (by & 0xFu) - m1;
                b[i] = d2 * (float) (int) (by >> 4) - m2;
            }
        } else {
            const int e = ks & 1;
            const float d = h2f((uint16_t) hreg[2]);
            const float s1 = d * (float) (int) (int8_t) hreg[0], s2 = d * (float) (int) (int8_t) hreg[1];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t la = (qreg[i >> 2] >> (8 * (i & 3))) & 0xffu, lb = (qreg[4 + (i >> 2)] >> (8 * (i & 3))) & 0xffu, hq = (qreg[8 + (i >> 2)] >> (8 * (i & 3))) & 0xffu;
                const int q1 = (int) (e ? (la >> 4) | (((hq >> 4) & 3u) << 4) : (la & 0xFu) | ((hq & 3u) << 4)) - 32;
                const int q2 = (int) (e ? (lb >> 4) | (((hq >> 6) & 3u) << 4) : (lb & 0xFu) | (((hq >> 2) & 3u) << 4)) - 32;
                a[i] = s1 * (float) q1;
                b[i] = s2 * (float) q2;
            }
        }
        char * row = lds + buf * BUFB + r * H_ROWB;
        const int sw = (r >> 1) & 7;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            u32x4 pa, pb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pa[j] = (uint32_t) f2h(a[8 * c + 2 * j]) | ((uint32_t) f2h(a[8 * c + 2 * j + 1]) << 16);
                pb[j] = (uint32_t) f2h(b[8 * c + 2 * j]) | ((uint32_t) f2h(b[8 * c + 2 * j + 1]) << 16);
            }
            *(u32x4 *) (row + (((2 * h + c) ^ sw) << 4)) = pa;
            *(u32x4 *) (row + (((4 + 2 * h + c) ^ sw) << 4)) = pb;
        }
    };

    f16v acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    const int nk_all = g.K / H_BK;
    const int k_lo = split * g.ksteps_per_split;
    const int k_hi = k_lo + g.ksteps_per_split < nk_all ? k_lo + g.ksteps_per_split : nk_all;
    const int fr = lane & 31, hb = lane >> 5, sw = (fr >> 1) & 7;
    if (k_lo < k_hi) { stage_x(0, k_lo); fetch_w(k_lo); park_w(0, k_lo); }
    for (int ks = k_lo; ks < k_hi; ++ks) {
        const int cur = (ks - k_lo) & 1;
        __syncthreads();                                   // tile ks is complete (DMA drained by the fence, every thread's weight slice parked); buffer cur^1 is free
        if (ks + 1 < k_hi) { stage_x(cur ^ 1, ks + 1); fetch_w(ks + 1); }
        const char * wb = lds + cur * BUFB; const char * xb = wb + WTILEB;
#pragma unroll
        for (int kk = 0; kk < H_BK / 16; ++kk) {
            const int co = ((kk * 2 + hb) ^ sw) << 4;
            h8 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = *(const h8 *) (xb + (wn * 64 + a * 32 + fr) * H_ROWB + co);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = *(const h8 *) (wb + (wm * 64 + b * 32 + fr) * H_ROWB + co);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (ks + 1 < k_hi) park_w(cur ^ 1, ks + 1);
    }

    char * dst = (mi == 0 ? g.dst[0] : (mi == 1 ? g.dst[1] : g.dst[2])) + (size_t) split * g.split_stride;
    const size_t dst_cs = mi == 0 ? g.dst_cs[0] : (mi == 1 ? g.dst_cs[1] : g.dst_cs[2]);
    const char * resid = mi == 0 ? g.resid[0] : (mi == 1 ? g.resid[1] : g.resid[2]);
    const size_t resid_cs = mi == 0 ? g.resid_cs[0] : (mi == 1 ? g.resid_cs[1] : g.resid_cs[2]);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + wm * 64 + b * 32 + (lane & 31);
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

// ---- 256 x 256 tile, 8 waves (2 along m x 4 along n, 128 x 64 per wave = 4 x 2 MFMA tiles, 128 accumulator registers), same staging
// scheme: 64 KB of LDS per K-step buffer, one workgroup per CU.  Twice the operand reuse of the 128 x 128 kernel (0.75 LDS reads and
// half the global bytes per MFMA); chosen by the launcher only where 256-square tiles fill the chip in whole rounds.
__global__ void __launch_bounds__(512) k_gemm_f16_glds256(const gemm_dev g) {
    constexpr int WTILEB = 256 * H_ROWB, BUFB = 2 * WTILEB;
    char * const lds = gemm_lds;

    const int nt  = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int q = nt / 8, r = nt % 8, xcd = bid % 8, idx = bid / 8;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    int tm = tile / g.tiles_n; const int tn = tile % g.tiles_n;
    int mi = 0;
    if (g.nmat > 1 && tm >= g.tm_end[0]) { mi = 1; if (g.nmat > 2 && tm >= g.tm_end[1]) mi = 2; }
    tm -= mi == 0 ? 0 : g.tm_end[mi - 1];
    const char * W = mi == 0 ? g.W[0] : (mi == 1 ? g.W[1] : g.W[2]);
    const size_t w_rs = mi == 0 ? g.w_rs[0] : (mi == 1 ? g.w_rs[1] : g.w_rs[2]);
    const int M = mi == 0 ? g.M[0] : (mi == 1 ? g.M[1] : g.M[2]);
    const int N = g.N;
    const int m0 = tm * 256, n0 = tn * 256;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    // staging: wave w fills rows [32w, 32w+32) of both 256-row operand tiles, 8 rows per instruction
    const int r8 = lane >> 3;
    const char * wp[4]; const char * xp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gc = (lane & 7) ^ (((j * 8 + r8) >> 1) & 7);
        int mr = m0 + wave * 32 + j * 8 + r8; mr = mr < M ? mr : M - 1;
        int nr = n0 + wave * 32 + j * 8 + r8; nr = nr < N ? nr : N - 1;
        wp[j] = W + (size_t) mr * w_rs + gc * 16;
        xp[j] = g.X + (size_t) nr * g.x_rs + gc * 16;
    }
    // one quarter of the next tile's DMA (2 of the wave's 8 instructions): issued between the k-sub-steps so that the LDS reads and
    // MFMAs of the current tile start right after the barrier instead of behind eight DMA issues
    auto stage_part = [&](int buf, int ks, int j) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t) (wp[j] + (size_t) ks * H_ROWB), (lds_ptr_t) (lds + buf * BUFB + (wave * 32 + j * 8) * H_ROWB), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t) (xp[j] + (size_t) ks * H_ROWB), (lds_ptr_t) (lds + buf * BUFB + WTILEB + (wave * 32 + j * 8) * H_ROWB), 16, 0, 0);
    };
    auto stage = [&](int buf, int ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) stage_part(buf, ks, j);
    };

    f16v acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    const int nk = g.K / H_BK;
    const int fr = lane & 31, hb = lane >> 5, sw = (fr >> 1) & 7;
    stage(0, 0);
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        __syncthreads();                                   // tile ks has landed (the fence drains the DMA), buffer cur^1 is free
        const char * wb = lds + cur * BUFB; const char * xb = wb + WTILEB;
#pragma unroll
        for (int kk = 0; kk < H_BK / 16; ++kk) {
            const int co = ((kk * 2 + hb) ^ sw) << 4;
            h8 af[2], bf[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = *(const h8 *) (xb + (wn * 64 + a * 32 + fr) * H_ROWB + co);
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = *(const h8 *) (wb + (wm * 128 + b * 32 + fr) * H_ROWB + co);
            if (ks + 1 < nk) stage_part(cur ^ 1, ks + 1, kk);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }

    char * dst = mi == 0 ? g.dst[0] : (mi == 1 ? g.dst[1] : g.dst[2]);
    const size_t dst_cs = mi == 0 ? g.dst_cs[0] : (mi == 1 ? g.dst_cs[1] : g.dst_cs[2]);
    const char * resid = mi == 0 ? g.resid[0] : (mi == 1 ? g.resid[1] : g.resid[2]);
    const size_t resid_cs = mi == 0 ? g.resid_cs[0] : (mi == 1 ? g.resid_cs[1] : g.resid_cs[2]);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int m = m0 + wm * 128 + b * 32 + (lane & 31);
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

// ---- 256 x 256 tile, 8 waves, EIGHT-PHASE schedule: the kernel above drains every LDS-DMA at its one barrier per K-step, so a tile's
// operands have at most one K-step (~0.4 us of MFMA work) to arrive -- less than an HBM round trip under load.  Here the DMA never drains
// inside the loop:
//   * a K-step's operands are four 16-KB "items", in the order they are needed: X-h0, W-h0, X-h1, W-h1.  Wave (wm, wn) owns rows
//     wm*128 + [0,128) x columns wn*64 + [0,64); W-h0 / W-h1 hold the first / second 64 rows of BOTH wm, X-h0 / X-h1 the first / second 32
//     columns of all four wn, so every wave walks its C quadrants (64 x 32, 8 MFMAs) in the same item order, one per phase:
//         phase 0: reads W-h0 (8 x ds_read_b128), acc[0][0..1] (X-h0 is in registers already)     phase 1: reads X-h1 (4), acc[1][0..1]
//         phase 2: reads W-h1 (8), acc[1][2..3]              phase 3: reads X-h0 of the NEXT K-step (4, into the registers X-h1 left), acc[0][2..3]
//   * the 8 DMA instructions per thread and K-step are issued 1 / 3 / 1 / 3 per phase -- most where the fewest LDS reads are -- five to six
//     phases before the phase that reads them, each into the slot the same item of two K-steps earlier occupied; every phase ends its
//     issue part with a counted s_waitcnt vmcnt(9 or 10): what the NEXT phase reads has landed, more than a K-step stays in flight across
//     the barrier
//   * waves 0-3 (wm = 0) and 4-7 (wm = 1) run half a phase apart (one extra barrier up front for the second group): wave w and w + 4 share
//     a SIMD, so while one issues LDS reads and DMA the other runs its 8 MFMAs (ping-pong); two raw s_barrier per phase keep the lock step
//   RAW: an item is read one phase after the vmcnt wait + barrier that retire it.  WAR: an item overwrites the slot of the same item two
//   K-steps back, whose last ds_read was retired by an lgkmcnt(0) at least three barriers before the issue (lead <= 6 phases).
// GLU = true (ffn_gate / ffn_up -> SWIGLU -> ffn_down's activation): the tile is 128 rows of BOTH matrices -- W-h0 comes from the gate matrix,
// W-h1 from the same rows of the up matrix -- so a lane's acc[a][0..1] and acc[a][2..3] are gate and up of the same output element:
// silu(gate) * up (vec.h:958 arithmetic, f32) is rounded to f16 straight into the activation image of the next GEMM; the two f32
// [n_ff x n_tokens] intermediates and the GLU launch never exist.
// MI355X_GEMM_ABL (measurement only): 1 no DMA inside the loop (stale operands), 8 cycle stamps per phase part.
// R96 (GLU only): the tile is 96 rows of both matrices instead of 128 -- the waves of the second group (wm = 1) own ONE 32-row fragment of gate and of up instead of two.
// Waves w and w + 4 share a SIMD, so every SIMD has three quarters of the matrix work per K-step, and a grid that filled 192 of the 256 CUs (n_ff 12288 x 512 tokens:
// 96 x 2 tiles) becomes 128 x 2 = 256 workgroups.  The LDS rows 96..127 of the W items are not used; the DMA instructions that would fill them stay (the counted
// vmcnt waits rest on every wave issuing the same number) but all their lanes ask for one and the same 16 bytes.
template <int ABL, bool GLU, bool R96 = false>
__global__ void __launch_bounds__(512) k_gemm_f16_ph8(const gemm_dev g) {
    static_assert(GLU || !R96, "96-row tiles exist for the GLU form only");
    constexpr int HALFB = 128 * H_ROWB, BUFB = 4 * HALFB;          // 16 KB per item, 64 KB per buffer: slots W-h0, W-h1, X-h0, X-h1
    constexpr int S_W0 = 0, S_W1 = 1, S_X0 = 2, S_X1 = 3;
    char * const lds = gemm_lds;

    const int nt  = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int q = nt / 8, r = nt % 8, xcd = bid % 8, idx = bid / 8;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // an XCD's 32 resident workgroups are consecutive tiles: walk the tile grid in bands of 8 tile rows, column-major inside a band, so that
    // they form an 8 x 4 block (8 W panels + 4 X panels through that XCD's L2) instead of a 1 x 32 strip (1 + 32 panels)
    const int band = tile / (8 * g.tiles_n), inb = tile % (8 * g.tiles_n), bh = g.tiles_m - band * 8 < 8 ? g.tiles_m - band * 8 : 8;
    int tm = band * 8 + inb % bh; const int tn = inb / bh;
    int mi = 0;
    if (!GLU && g.nmat > 1 && tm >= g.tm_end[0]) { mi = 1; if (g.nmat > 2 && tm >= g.tm_end[1]) mi = 2; }
    tm -= mi == 0 ? 0 : g.tm_end[mi - 1];
    const char * W = mi == 0 ? g.W[0] : (mi == 1 ? g.W[1] : g.W[2]);
    const size_t w_rs = mi == 0 ? g.w_rs[0] : (mi == 1 ? g.w_rs[1] : g.w_rs[2]);
    const int M = mi == 0 ? g.M[0] : (mi == 1 ? g.M[1] : g.M[2]);
    const int N = g.N;
    const int m0 = tm * (GLU ? (R96 ? 96 : 128) : 256), n0 = tn * 256;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;                       // waves w and w + 4 (one SIMD) are in different phase groups

    // staging: an item is 128 LDS rows; instruction i of wave w fills local rows i*64 + w*8 + [0,8), lane l the 16-B chunk l & 7 of row l >> 3
    // from source chunk (l & 7) ^ ((row >> 1) & 7) (the bank swizzle of the kernels above, applied on the source side)
    const int r8 = lane >> 3;
    // (32-bit offsets from wave-uniform bases: 8 registers instead of the 32 of sixteen pointers -- the kernel sits at the 256-register line)
    uint32_t soff[4][2];                                           // [slot][instruction]
    const char * const wb0 = GLU ? g.W[g.glu_gate] : W, * const wb1 = GLU ? g.W[1 - g.glu_gate] : W;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = i * 64 + wave * 8 + r8;
        const int gc = (lane & 7) ^ ((lr >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int mr = GLU ? m0 + lr : m0 + (lr >> 6) * 128 + h * 64 + (lr & 63); mr = mr < M ? mr : M - 1;
            int nr = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);  nr = nr < N ? nr : N - 1;
            soff[S_W0 + h][i] = (uint32_t) ((size_t) mr * w_rs + gc * 16);                          // (GLU: M and the row stride are checked equal; launcher: M * w_rs < 4 GB)
            if (R96 && lr >= 96) soff[S_W0 + h][i] = (uint32_t) ((size_t) (m0 < M ? m0 : M - 1) * w_rs);   // (rows nobody reads: one request for the whole wave)
            soff[S_X0 + h][i] = (uint32_t) ((size_t) nr * g.x_rs + gc * 16);
        }
    }
    const int nk = g.K / H_BK;
    bool in_loop = false;
    auto dma = [&](int buf, int slot, int i, int kt) {             // one DMA instruction: half i of an item
        if ((ABL & 1) && in_loop) return;
        const size_t ko = (size_t) (kt < nk ? kt : nk - 1) * H_ROWB; // past the end: the last K-step again, into a slot nobody reads any more
        const char * const base = (slot == S_W0 ? wb0 : slot == S_W1 ? wb1 : g.X) + ko;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t) (base + (size_t) soff[slot][i]), (lds_ptr_t) (lds + buf * BUFB + slot * HALFB + (i * 64 + wave * 8) * H_ROWB), 16, 0, 0);
    };

    f16v acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    const int fr = lane & 31, hb = lane >> 5, sw = (fr >> 1) & 7;
    int co[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) co[kk] = ((kk * 2 + hb) ^ sw) << 4;
    const char * const wrow = lds + (wm * 64 + fr) * H_ROWB;        // + buffer, slot, (second 32-row fragment) 32 * H_ROWB
    const char * const xrow = lds + (wn * 32 + fr) * H_ROWB;
    h8 wr[2][4], x0[4], x1[4];
    // (SH: the short form of a 96-row tile's second wave group -- one W fragment per item; the whole loop exists once per form, straight-line: a test per fragment
    // inside one loop made the register allocator spill)
    auto read_w = [&](auto SHc, int buf, int slot) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wr[0][kk] = *(const h8 *) (wrow + buf * BUFB + slot * HALFB + co[kk]);
        if (!(R96 && wm == 1)) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wr[1][kk] = *(const h8 *) (wrow + buf * BUFB + slot * HALFB + 32 * H_ROWB + co[kk]);
        }
    };
    auto read_x = [&](int buf, int slot, h8 (&x)[4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) x[kk] = *(const h8 *) (xrow + buf * BUFB + slot * HALFB + co[kk]);
    };
    auto mma = [&](auto SHc, int a, int bh, const h8 (&x)[4]) {
        __builtin_amdgcn_s_setprio(1);
        if (R96) {                                                 // (fragment 0, then -- first wave group only -- fragment 1: one wave-uniform branch per phase)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[a][2 * bh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[kk], wr[0][kk], acc[a][2 * bh], 0, 0, 0);
            if (wm == 0) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc[a][2 * bh + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[kk], wr[1][kk], acc[a][2 * bh + 1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) acc[a][2 * bh + bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[kk], wr[bb][kk], acc[a][2 * bh + bb], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
#define PH8_BAR()     asm volatile("s_barrier" ::: "memory")
#define PH8_ISSUED(n) asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier" ::: "memory")
#define PH8_READ()    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
    unsigned long long st[12] = {}, tc = 0;
#define PH8_T(i) do { if (ABL & 8) { const unsigned long long n = __builtin_readcyclecounter(); st[i] += n - tc; tc = n; } } while (0)

    // prologue: K-step 0 whole, X-h0 / W-h0 / X-h1 of K-step 1 (14 instructions, the state the loop's phase 3 leaves), X-h0 of K-step 0 into x0
#pragma unroll
    for (int i = 0; i < 2; ++i) dma(0, S_X0, i, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma(0, S_W0, i, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma(0, S_X1, i, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma(0, S_W1, i, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma(1, S_X0, i, 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma(1, S_W0, i, 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma(1, S_X1, i, 1);
    PH8_ISSUED(12);
    read_x(0, S_X0, x0);
    PH8_ISSUED(10);
    PH8_READ();
    if (wm == 1) PH8_BAR();
    in_loop = true;
    if (ABL & 8) tc = __builtin_readcyclecounter();
    auto kstep = [&](auto PARc, auto SHc, int kt) {
        constexpr int P = decltype(PARc)::value;
        h8 (&xa)[4] = P ? x1 : x0; h8 (&xb)[4] = P ? x0 : x1;      // X-h0 of this K-step sits where the previous K-step's X-h1 was
        read_w(SHc, P, S_W0);  dma(P ^ 1, S_W1, 0, kt + 1);                                                                        PH8_T(0); PH8_ISSUED(9);  PH8_T(1);  PH8_READ(); mma(SHc, 0, 0, xa); PH8_BAR(); PH8_T(2);
        read_x(P, S_X1, xb);   dma(P ^ 1, S_W1, 1, kt + 1); dma(P, S_X0, 0, kt + 2); dma(P, S_X0, 1, kt + 2);                      PH8_T(3); PH8_ISSUED(10); PH8_T(4);  PH8_READ(); mma(SHc, 1, 0, xb); PH8_BAR(); PH8_T(5);
        read_w(SHc, P, S_W1);  dma(P, S_W0, 0, kt + 2);                                                                            PH8_T(6); PH8_ISSUED(9);  PH8_T(7);  PH8_READ(); mma(SHc, 1, 1, xb); PH8_BAR(); PH8_T(8);
        read_x(P ^ 1, S_X0, xb); dma(P, S_W0, 1, kt + 2); dma(P, S_X1, 0, kt + 2); dma(P, S_X1, 1, kt + 2);                        PH8_T(9); PH8_ISSUED(10); PH8_T(10); PH8_READ(); mma(SHc, 0, 1, xa); PH8_BAR(); PH8_T(11);
    };
    auto loop = [&](auto SHc) {
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) { kstep(std::integral_constant<int, 0>(), SHc, kt); kstep(std::integral_constant<int, 1>(), SHc, kt + 1); }
        if (kt < nk) kstep(std::integral_constant<int, 0>(), SHc, kt);
    };
    loop(std::false_type());
    if (wm == 0) PH8_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((ABL & 8) && g.dbg && blockIdx.x == 300 && lane == 0 && (wave & 3) == 0)
        for (int i = 0; i < 12; ++i) g.dbg[wm * 12 + i] = st[i];
#undef PH8_BAR
#undef PH8_ISSUED
#undef PH8_READ
#undef PH8_T

    if (GLU) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                if (R96 && wm == 1 && bb == 1) continue;
                const int m = m0 + wm * 64 + bb * 32 + (lane & 31);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = n0 + wn * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    const float x = acc[a][bb][e], u = acc[a][2 + bb][e];
                    const float v = (x / (1.0f + expf(-x))) * u;
                    if (m < M && n < N) *(uint16_t *) (g.out16 + (size_t) n * g.out16_rs + (size_t) m * 2) = f2h(v);
                }
            }
        return;
    }
    char * dst = mi == 0 ? g.dst[0] : (mi == 1 ? g.dst[1] : g.dst[2]);
    const size_t dst_cs = mi == 0 ? g.dst_cs[0] : (mi == 1 ? g.dst_cs[1] : g.dst_cs[2]);
    const char * resid = mi == 0 ? g.resid[0] : (mi == 1 ? g.resid[1] : g.resid[2]);
    const size_t resid_cs = mi == 0 ? g.resid_cs[0] : (mi == 1 ? g.resid_cs[1] : g.resid_cs[2]);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int m = m0 + wm * 128 + b * 32 + (lane & 31);
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

// dst[n][m] = sum_s part[s][n][m] (+ resid[n][m]), fixed summation order
__global__ void __launch_bounds__(256) k_gemm_reduce(const float * __restrict__ part, int nsplit, size_t split_elems, const char * __restrict__ resid, size_t resid_cs,
                                                     char * __restrict__ dst, size_t dst_cs, int M, int N) {
    const int64_t i = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (int64_t) M * N) return;
    const int n = (int) (i / M), m = (int) (i % M);                       // M % 4 == 0 (checked by the launcher)
    f32x4 v = *(const f32x4 *) (part + i);
    for (int s = 1; s < nsplit; ++s) { const f32x4 w = *(const f32x4 *) (part + s * split_elems + i); v += w; }
    if (resid) { const f32x4 w = *(const f32x4 *) (resid + (size_t) n * resid_cs + (size_t) m * 4); v += w; }
    *(f32x4 *) (dst + (size_t) n * dst_cs + (size_t) m * 4) = v;
}

// the same for every matrix of a grouped launch in ONE launch (wq / wk / wv of a short prompt or a streaming encoder chunk: three reductions of a few us each were three
// dependent launches), with an optional second addend per matrix.  Slab s holds the matrices back to back as dense [N][M_i] blocks; quads never straddle (M_i % 4 == 0).
struct gemm_reduce_multi_dev { int nmat, nsplit, N; size_t split_elems; size_t off[3]; int M[3]; const char * resid[3]; size_t resid_cs[3]; const char * resid2[3]; size_t resid2_cs[3]; char * dst[3]; size_t dst_cs[3];
                               int unary[3] = { -1, -1, -1 }; char * y16[3] = { nullptr, nullptr, nullptr }; size_t y16_rs[3] = { 0, 0, 0 }; size_t y16_ms[3] = { 2, 2, 2 }; };      // (unary >= 0: GELU / GELU_QUICK of the value, f16 rows to y16, dst optional)
__global__ void __launch_bounds__(256) k_gemm_reduce_multi(const float * __restrict__ part, const gemm_reduce_multi_dev g) {
    int64_t i = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 4;
    int q = 0;
    for (; q < g.nmat; ++q) { const int64_t n = (int64_t) g.M[q] * g.N; if (i < n) break; i -= n; }
    if (q >= g.nmat) return;
    const int M = g.M[q], n = (int) (i / M), m = (int) (i % M);
    const float * p = part + g.off[q] + i;
    f32x4 sl[8];                                                   // (all requests first, additions in slab order: see k_gemm_reduce_rms_norm)
#pragma unroll
    for (int s = 0; s < 8; ++s) sl[s] = s < g.nsplit ? *(const f32x4 *) (p + s * g.split_elems) : f32x4{0, 0, 0, 0};
    f32x4 r1 = f32x4{0, 0, 0, 0}, r2 = f32x4{0, 0, 0, 0};
    if (g.resid[q])  r1 = *(const f32x4 *) (g.resid[q]  + (size_t) n * g.resid_cs[q]  + (size_t) m * 4);
    if (g.resid2[q]) r2 = *(const f32x4 *) (g.resid2[q] + (size_t) n * g.resid2_cs[q] + (size_t) m * 4);
    f32x4 v = sl[0];
#pragma unroll
    for (int s = 1; s < 8; ++s) if (s < g.nsplit) v += sl[s];
    if (g.resid[q])  v += r1;
    if (g.resid2[q]) v += r2;
    if (g.y16[q]) {
        // f16 rows out: behind an activation (an encoder's fc1 -> + bias -> GELU -> fc2: the image fc2 reads), or plain (the CPY of a streaming encoder's K / V rows into
        // its f16 cache; a transposed V cache has the rows a cache pitch apart: four 2-byte stores)
        if (g.unary[q] >= 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = g.unary[q] == GGML_UNARY_OP_GELU ? op_gelu(v[e]) : op_gelu_quick(v[e]);
        }
        char * const o = g.y16[q] + (size_t) n * g.y16_rs[q] + (size_t) m * g.y16_ms[q];
        if (g.y16_ms[q] == 2) { u32x2 h; h[0] = (uint32_t) f2h(v[0]) | ((uint32_t) f2h(v[1]) << 16); h[1] = (uint32_t) f2h(v[2]) | ((uint32_t) f2h(v[3]) << 16); *(u32x2 *) o = h; }
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) *(uint16_t *) (o + (size_t) e * g.y16_ms[q]) = f2h(v[e]);
        }
        if (!g.dst[q]) return;
    }
    *(f32x4 *) (g.dst[q] + (size_t) n * g.dst_cs[q] + (size_t) m * 4) = v;
}
// dst[n][m] += r[n][m] (rows of M % 4 == 0 floats): the second addend of a launch that did not split K
__global__ void __launch_bounds__(256) k_gemm_add_rows(char * __restrict__ dst, size_t dst_cs, const char * __restrict__ r, size_t r_cs, int M, int N) {
    const int64_t i = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (int64_t) M * N) return;
    const int n = (int) (i / M), m = (int) (i % M);
    f32x4 * d = (f32x4 *) (dst + (size_t) n * dst_cs + (size_t) m * 4);
    *d = *d + *(const f32x4 *) (r + (size_t) n * r_cs + (size_t) m * 4);
}

bool gemm_f16_ok(const void * W, size_t w_rs, const void * X, size_t x_rs, int64_t K) {
    return K % G_BK == 0 && K >= G_BK && w_rs % 16 == 0 && x_rs % 16 == 0 && ((uintptr_t) W & 15) == 0 && ((uintptr_t) X & 15) == 0;
}

// split-K reduction fused with the RMS_NORM -> MUL(w) that follows the residual ADD (ops.cpp:3517-3566 arithmetic: sum of squares in
// double): x = sum_s part[s] + resid -> dst (f32, the next residual); y = (x * scale) * w -> y32 (optional) and / or f16 rows y16
// (the activation image of the next GEMM).  One workgroup per row, the row stays in registers (M <= 16384, M % 4 == 0).
template <int MAXV>
__global__ void __launch_bounds__(256) k_gemm_reduce_rms_norm(const float * __restrict__ part, int nsplit, size_t split_elems, const char * __restrict__ resid, size_t resid_cs,
                                                              char * __restrict__ dst, size_t dst_cs, const float * __restrict__ w, float eps,
                                                              char * __restrict__ y32, size_t y32_cs, char * __restrict__ y16, size_t y16_rs, int M, int y16_q8) {
    __shared__ double red[4];
    const int n = blockIdx.x;
    f32x4 v[MAXV];
    double ss = 0.0;
    if (MAXV <= 4) {
        // rows of at most 4096: EVERY piece of every slab, and the residual's, requested before the first addition -- a thread's four quads one after the other were four
        // dependent round trips (the store of a quad sits between the loads of two quads), 13.9 us for 52 MB; the additions keep their order: slab 0 + slab 1 + ... + residual
        f32x4 sl[MAXV][8], rr[MAXV];
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int i = (threadIdx.x + k * 256) * 4;
#pragma unroll
            for (int s = 0; s < 8; ++s) sl[k][s] = (i < M && s < nsplit) ? *(const f32x4 *) (part + s * split_elems + (size_t) n * M + i) : f32x4{0, 0, 0, 0};
            rr[k] = (i < M && resid) ? *(const f32x4 *) (resid + (size_t) n * resid_cs + (size_t) i * 4) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int i = (threadIdx.x + k * 256) * 4;
            v[k] = f32x4{0, 0, 0, 0};
            if (i < M) {
                f32x4 a = sl[k][0];
#pragma unroll
                for (int s = 1; s < 8; ++s) if (s < nsplit) a += sl[k][s];
                if (resid) a += rr[k];
                *(f32x4 *) (dst + (size_t) n * dst_cs + (size_t) i * 4) = a;
                v[k] = a;
#pragma unroll
                for (int e = 0; e < 4; ++e) ss += (double) (a[e] * a[e]);
            }
        }
    } else {
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = (threadIdx.x + k * 256) * 4;
        v[k] = f32x4{0, 0, 0, 0};
        if (i < M) {
            // every slab's piece and the residual's requested before the first addition (a run-time loop of load -> add serialises the round trips: 14.3 us for 52 MB);
            // the additions keep their order: slab 0 + slab 1 + ... + residual
            f32x4 sl[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) sl[s] = s < nsplit ? *(const f32x4 *) (part + s * split_elems + (size_t) n * M + i) : f32x4{0, 0, 0, 0};
            f32x4 rr = f32x4{0, 0, 0, 0};
            if (resid) rr = *(const f32x4 *) (resid + (size_t) n * resid_cs + (size_t) i * 4);
            f32x4 a = sl[0];
#pragma unroll
            for (int s = 1; s < 8; ++s) if (s < nsplit) a += sl[s];
            if (resid) a += rr;
            *(f32x4 *) (dst + (size_t) n * dst_cs + (size_t) i * 4) = a;
            v[k] = a;
#pragma unroll
            for (int e = 0; e < 4; ++e) ss += (double) (a[e] * a[e]);
        }
    }
    }
    ss = block_sum<double>(ss, red);
    const float mean  = (float) (ss / (double) M);
    const float scale = 1.0f / sqrtf(mean + eps);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = (threadIdx.x + k * 256) * 4;
        if (i >= M) break;
        const f32x4 ww = *(const f32x4 *) (w + i);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (v[k][e] * scale) * ww[e];
        if (y32) *(f32x4 *) (y32 + (size_t) n * y32_cs + (size_t) i * 4) = y;
        if (y16) {
            if (y16_q8) y = q8k_requant4(y, threadIdx.x & 63);      // (M % 256 == 0: a wave holds whole 256-blocks, lane l elements 4l..4l+3 -- the image carries the Q8_K-quantised values)
            u32x2 h;
            h[0] = (uint32_t) f2h(y[0]) | ((uint32_t) f2h(y[1]) << 16); h[1] = (uint32_t) f2h(y[2]) | ((uint32_t) f2h(y[3]) << 16);
            *(u32x2 *) (y16 + (size_t) n * y16_rs + (size_t) i * 2) = h;
        }
    }
}

// the reduction kernels that read the slabs into registers (k_gemm_reduce_multi, k_gemm_reduce_rms_norm; k_norm_rows / k_norm_rope_v4 with slab sources) hold at most
// this many -- pick_ksplit never asks for more; a caller that does is a programming error, not a silently short sum
constexpr int GEMM_MAX_SPLIT = 8;
static void check_nsplit(int nsplit, const char * who) {
    if (nsplit < 1 || nsplit > GEMM_MAX_SPLIT) { fprintf(stderr, "[mi355x] %s: %d K slabs, the reduction kernels sum at most %d\n", who, nsplit, GEMM_MAX_SPLIT); abort(); }
}
void gemm_reduce(const float * partial, int nsplit, const float * resid, size_t resid_cs, float * dst, size_t dst_cs, int64_t M, int64_t N, hipStream_t st) {
    const int64_t quads = M * N / 4;
    if (quads == 0) return;
    k_gemm_reduce<<<dim3((unsigned) ((quads + 255) / 256)), dim3(256), 0, st>>>(partial, nsplit, (size_t) M * (size_t) N, (const char *) resid, resid_cs, (char *) dst, dst_cs, (int) M, (int) N);
}
// the reduction of a grouped launch whose caller deferred it (defer_multi) and could not fold it into its next kernel after all
void gemm_reduce_group(const float * partial, int nsplit, size_t slab_elems, int nmat, const size_t * off, const int64_t * M, int64_t N, float * const * dst, const size_t * dst_cs, hipStream_t st) {
    check_nsplit(nsplit, "gemm_reduce_group");
    gemm_reduce_multi_dev r; r.nmat = nmat; r.nsplit = nsplit; r.N = (int) N; r.split_elems = slab_elems;
    int64_t quads = 0;
    for (int i = 0; i < 3; ++i) {
        const int k = i < nmat ? i : 0;
        r.off[i] = off[k]; r.M[i] = i < nmat ? (int) M[k] : 0; r.resid[i] = nullptr; r.resid_cs[i] = 0; r.resid2[i] = nullptr; r.resid2_cs[i] = 0; r.dst[i] = (char *) dst[k]; r.dst_cs[i] = dst_cs[k];
        if (i < nmat) quads += M[k] * N / 4;
    }
    if (quads > 0) k_gemm_reduce_multi<<<dim3((unsigned) ((quads + 255) / 256)), dim3(256), 0, st>>>(partial, r);
}
// the same with a second addend (k_gemm_reduce_multi's order: slabs, addend 1, addend 2)
void gemm_reduce2(const float * partial, int nsplit, const float * resid, size_t resid_cs, const float * resid2, size_t resid2_cs, float * dst, size_t dst_cs, int64_t M, int64_t N, hipStream_t st) {
    if (!resid2) { gemm_reduce(partial, nsplit, resid, resid_cs, dst, dst_cs, M, N, st); return; }
    const int64_t quads = M * N / 4;
    if (quads == 0) return;
    check_nsplit(nsplit, "gemm_reduce2");
    gemm_reduce_multi_dev r; r.nmat = 1; r.nsplit = nsplit; r.N = (int) N; r.split_elems = (size_t) M * (size_t) N;
    for (int i = 0; i < 3; ++i) { r.off[i] = 0; r.M[i] = i == 0 ? (int) M : 0; r.resid[i] = (const char *) resid; r.resid_cs[i] = resid_cs; r.resid2[i] = (const char *) resid2; r.resid2_cs[i] = resid2_cs; r.dst[i] = (char *) dst; r.dst_cs[i] = dst_cs; }
    k_gemm_reduce_multi<<<dim3((unsigned) ((quads + 255) / 256)), dim3(256), 0, st>>>(partial, r);
}
bool gemm_reduce_rms_norm_ok(int64_t M) { return M % 4 == 0 && M <= 16384; }
void gemm_reduce_rms_norm(const float * partial, int nsplit, const float * resid, size_t resid_cs, float * dst, size_t dst_cs, const float * w, float eps,
                          float * y32, size_t y32_cs, uint16_t * y16, size_t y16_rs, int64_t M, int64_t N, hipStream_t st, bool y16_q8) {
    if (M == 0 || N == 0) return;
    if (y16_q8 && M % 256 != 0) { fprintf(stderr, "[mi355x] gemm_reduce_rms_norm: the Q8_K image needs rows of whole 256-blocks\n"); abort(); }
    check_nsplit(nsplit, "gemm_reduce_rms_norm");
    if (M <= 4096) k_gemm_reduce_rms_norm<4><<<dim3((unsigned) N), dim3(256), 0, st>>>(partial, nsplit, (size_t) M * (size_t) N, (const char *) resid, resid_cs, (char *) dst, dst_cs, w, eps,
                                                                                   (char *) y32, y32_cs, (char *) y16, y16_rs, (int) M, y16_q8 ? 1 : 0);
    else           k_gemm_reduce_rms_norm<16><<<dim3((unsigned) N), dim3(256), 0, st>>>(partial, nsplit, (size_t) M * (size_t) N, (const char *) resid, resid_cs, (char *) dst, dst_cs, w, eps,
                                                                                    (char *) y32, y32_cs, (char *) y16, y16_rs, (int) M, y16_q8 ? 1 : 0);
}

// dynamic LDS above 64 KB needs a function attribute, once per (kernel, device): a process may drive several GPUs
static void allow_big_lds(const void * kernel, int bytes, int slot) {
    static bool done[8][64] = {};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !done[slot][dev]) {
        HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        if (dev >= 0 && dev < 64) done[slot][dev] = true;
    }
}

// choose a K split that brings an under-filled launch up to about two workgroups per CU.  Few columns (N <= 256: short prompts, omni chunks):
// a launch is then 32 .. 192 tiles whose 64 .. 192-step K loops run one after the other on a fraction of the CUs and the time is that loop's
// latency, not bandwidth or flops -- split up to 8 ways, down to 8 K-steps per workgroup
static int pick_ksplit(int64_t tiles, int64_t nk, int64_t N) {
    if (tiles >= 256) return 1;
    if (nk < 32) {
        // short K (the omni encoders: 1024 / 1152): worth splitting only when the launch is a handful of workgroups -- a streaming audio chunk is 50 columns,
        // wq at 128 x 128 is 8 workgroups, each bound by what ONE CU can pull (~1.5 us per 64-deep K-step: measured 25 us for 2 MB of weights)
        static const bool off = getenv("MI355X_NO_SHORT_K_SPLIT") != nullptr;
        if (off || tiles > 64 || nk < 4 || N > 128) return 1;
        int s = (int) (nk / 2); if (s > 8) s = 8;
        while (s > 1 && tiles * s > 512) --s;
        return s < 1 ? 1 : s;
    }
    const int smax = N <= 256 ? 8 : 4, min_steps = N <= 256 ? 8 : 16;
    static const int64_t target = getenv("MI355X_GEMM_SPLIT_TARGET") ? atoll(getenv("MI355X_GEMM_SPLIT_TARGET")) : 512;
    int s = (int) (target / tiles);
    if (s > smax) s = smax;
    while (s > 1 && nk / s < min_steps) --s;
    return s < 1 ? 1 : s;
}
// M: rows of the launch (the sum over its matrices); an upper bound (the launcher's own tile count decides the split)
size_t gemm_split_scratch_bytes(int64_t M, int64_t N, int64_t K) {
    if (K % H_BK != 0 || M % 4 != 0) return 0;
    const int64_t tiles = ((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
    int s = pick_ksplit(tiles, K / H_BK, N);
    if (N <= 256 && tiles < 256 && K / H_BK >= 32) s = 8;     // (a group's tile count may differ by a few tiles from this estimate)
    if (N <= 128 && tiles <= 72 && K / H_BK >= 4 && K / H_BK < 32) s = 8;
    return s > 1 ? (size_t) s * (size_t) M * (size_t) N * 4 : 0;
}

int device_cu_count() {                                        // CUs of the current device (cached per device)
    static int cus[64] = {};
    int dev = 0; HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) dev = 0;
    if (!cus[dev]) { hipDeviceProp_t pr; HIP_CHECK(hipGetDeviceProperties(&pr, dev)); cus[dev] = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 1; }
    return cus[dev];
}

// launches per tile variant (tests assert that a shape really selected the kernel it is meant to cover): 0 = 256 x 256, 1 = 192-row
static long g_gemm_variant_launches[7] = { 0, 0, 0, 0, 0, 0, 0 };  // ... 2 = gate / up + SWIGLU, 3 = K-quant staging, 4 = unused (the stream-K lab form, tools/lab), 5 = the 96-row tiles among 2, 6 = unused (the register-ring lab form)
long gemm_variant_launches(int v) { return v >= 0 && v < 7 ? g_gemm_variant_launches[v] : 0; }
// gate / up + SWIGLU in one launch: equal shapes and row strides, whole 128-row blocks, enough tiles to occupy the chip
bool gemm_glu_ok(const gemm_multi_args & a) {
    static const bool off = getenv("MI355X_NO_GEMM_GLU") != nullptr;
    if (off || a.nmat != 2 || a.nbatch > 1 || a.K % H_BK != 0 || a.m[0].M != a.m[1].M || a.m[0].w_rs != a.m[1].w_rs || a.m[0].M % 128 != 0 || a.m[0].resid || a.m[1].resid) return false;
    if ((size_t) a.m[0].M * a.m[0].w_rs >= (1ull << 32) || (size_t) a.N * ((size_t) a.K * 2 + 256) >= (1ull << 32)) return false;   // (32-bit operand offsets in the kernel; the caller has not laid the activation image out yet: its row is K halfs plus padding)
    return (a.m[0].M / 128) * ((a.N + 255) / 256) >= 128;
}

// K split gemm_f16_multi will choose for a launch of at most 128 columns (one column tile: neither the 192-row nor the 256 x 256 kernels apply); > 1 means the
// result goes through the reduction, whose epilogue takes two addends per matrix.  Callers that want the second addend fused ask first.
int gemm_f16_small_n_ksplit(const gemm_multi_args & a) {
    if (a.N <= 0 || a.N > 128 || a.nmat <= 0 || a.nbatch > 1 || !a.partial || a.glu_out16 || a.K % H_BK != 0) return 1;
    int64_t tm = 0, m_sum = 0; bool m4 = true;
    for (int i = 0; i < a.nmat; ++i) { if (a.m[i].qtype != 0) return 1; tm += (a.m[i].M + G_BM - 1) / G_BM; m_sum += a.m[i].M; m4 = m4 && a.m[i].M % 4 == 0 && a.m[i].dst_cs % 16 == 0; }
    if (!m4 || tm == 0) return 1;
    int ksplit = pick_ksplit(tm, a.K / H_BK, a.N);
    while (ksplit > 1 && (size_t) ksplit * (size_t) m_sum * (size_t) a.N * 4 > a.partial_bytes) --ksplit;
    return ksplit;
}

void gemm_f16_multi(const gemm_multi_args & a, hipStream_t st) {
    if (a.N == 0 || a.nmat == 0) return;
    const int tiles_n = (int) ((a.N + G_BN - 1) / G_BN);
    if (a.probe_path && (a.K % H_BK != 0 || a.glu_out16)) { *a.probe_path = 0; return; }
    if (a.K % H_BK != 0) {                                    // padded register-staged kernel, one matrix at a time
        if (a.nbatch > 1) { fprintf(stderr, "[mi355x] gemm: batched launch needs K %% 64 == 0\n"); abort(); }
        for (int i = 0; i < a.nmat; ++i) {
            const gemm_mat & m = a.m[i];
            if (m.M == 0) continue;
            if (m.resid) { fprintf(stderr, "[mi355x] gemm: residual epilogue needs K %% 64 == 0\n"); abort(); }
            const int tiles_m = (int) ((m.M + G_BM - 1) / G_BM);
            k_gemm_f16<<<dim3((unsigned) (tiles_m * tiles_n)), dim3(256), 0, st>>>((const char *) m.W, m.w_rs, (const char *) a.X, a.x_rs, (char *) m.dst, m.dst_cs,
                                                                                  (int) m.M, (int) a.N, (int) a.K, tiles_m, tiles_n);
        }
        return;
    }
    if (a.qt_img) {                                           // Q4_K blocks x the block-major Q8_K image: mmq_tile.hip (same slab layout, same deferred reductions)
        if (a.probe_path) { *a.probe_path = 0; return; }
        mmqt_args q; q.nmat = a.nmat; q.img = a.qt_img; q.N = a.N; q.K = a.K;
        q.partial = a.partial; q.partial_bytes = a.partial_bytes; q.deferred_split = a.deferred_split; q.defer_multi = a.defer_multi;
        for (int i = 0; i < a.nmat; ++i) {
            const gemm_mat & m = a.m[i];
            if (m.qtype != GGML_TYPE_Q4_K || m.resid2 || m.y16 || m.unary >= 0 || a.nbatch > 1 || a.glu_out16) { fprintf(stderr, "[mi355x] gemm: the tiled int8 kernel takes Q4_K blocks, one addend, f32 rows out\n"); abort(); }
            q.m[i] = { m.W, m.w_rs, m.dst, m.dst_cs, m.M, m.resid, m.resid_cs };
        }
        mmq_tile(q, st);
        return;
    }
    if (a.deferred_split) *a.deferred_split = 0;
    gemm_dev g;
    if (a.glu_out16) {                                        // ffn_gate / ffn_up + SWIGLU in one launch (gemm_glu_ok() said yes)
        // 96-row tiles when they take fewer rounds of the CUs, a tile of theirs counted as three quarters of a 128-row one (n_ff 12288 x 512 tokens: 192 tiles -> 256)
        static const int r96_env = getenv("MI355X_GEMM_GLU96") ? atoi(getenv("MI355X_GEMM_GLU96")) : -1;
        if ((size_t) a.N * a.x_rs >= (1ull << 32)) { fprintf(stderr, "[mi355x] gemm: activation image of %lld rows x %zu bytes is past the 32-bit offsets of the GLU kernel\n", (long long) a.N, a.x_rs); abort(); }
        const int tiles_n256 = (int) ((a.N + 255) / 256);
        const int64_t cus = (int64_t) device_cu_count();                  // (CUs of the current device)
        const int64_t t128 = (a.m[0].M / 128) * tiles_n256, t96 = ((a.m[0].M + 95) / 96) * tiles_n256;
        const bool r96 = r96_env >= 0 ? r96_env != 0 : ((t96 + cus - 1) / cus) * 3 < ((t128 + cus - 1) / cus) * 4;
        const int tm128 = r96 ? (int) ((a.m[0].M + 95) / 96) : (int) (a.m[0].M / 128);
        for (int i = 0; i < 3; ++i) {
            const gemm_mat & m = a.m[i < 2 ? i : 0];
            g.W[i] = (const char *) m.W; g.w_rs[i] = m.w_rs; g.dst[i] = nullptr; g.dst_cs[i] = 0; g.resid[i] = nullptr; g.resid_cs[i] = 0; g.M[i] = (int) m.M; g.tm_end[i] = tm128;
        }
        g.nmat = 1; g.X = (const char *) a.X; g.x_rs = a.x_rs; g.N = (int) a.N; g.K = (int) a.K; g.tiles_m = tm128; g.tiles_n = tiles_n256;
        g.ksteps_per_split = (int) (a.K / H_BK); g.split_stride = 0;
        g.ne12 = g.r2 = g.r3 = 1; g.w_nb2 = g.w_nb3 = g.x_bs = g.dst_nb2 = g.dst_nb3 = 0;
        g.dbg = nullptr; g.out16 = (char *) a.glu_out16; g.out16_rs = a.glu_out16_rs; g.glu_gate = a.glu_gate;
        constexpr int lds256 = 2 * 2 * 256 * H_ROWB;
        if (r96) {
            allow_big_lds((const void *) k_gemm_f16_ph8<0, true, true>, lds256, 6);
            k_gemm_f16_ph8<0, true, true><<<dim3((unsigned) (tm128 * tiles_n256)), dim3(512), lds256, st>>>(g);
            ++g_gemm_variant_launches[5];
        } else {
            allow_big_lds((const void *) k_gemm_f16_ph8<0, true>, lds256, 2);
            k_gemm_f16_ph8<0, true><<<dim3((unsigned) (tm128 * tiles_n256)), dim3(512), lds256, st>>>(g);
        }
        ++g_gemm_variant_launches[2];
        return;
    }
    // workgroup tile height: 192 rows when that removes a partially filled round of the 512 resident workgroups
    auto count_tm = [&](int bm) { int t = 0; for (int i = 0; i < a.nmat; ++i) t += (int) ((a.m[i].M + bm - 1) / bm); return t; };
    int BM = G_BM;
    {
        static const bool no192 = getenv("MI355X_GEMM_NO_192") != nullptr;
        const int64_t t128 = (int64_t) count_tm(128) * tiles_n, t192 = (int64_t) count_tm(192) * tiles_n;
        const int64_t c128 = ((t128 + 511) / 512) * 128, c192 = ((t192 + 511) / 512) * 192;
        if (!no192 && a.nbatch <= 1 && t128 > 512 && c192 < c128) BM = 192;
        // (64-row tiles for launches with fewer 128-row tiles than CUs -- qkv at ubatch 512 is 192 -- measured neutral; tuning override:)
        if (const char * e = getenv("MI355X_GEMM_BM")) { const int f = atoi(e); if (f == 64 || f == 128 || f == 192) BM = f; }
    }
    // 256 x 256 workgroup tiles (one workgroup per CU, ~1.2x the per-CU rate of the 128-row kernels: 1015 vs 851 TFLOP/s at 8192^3) when they come in
    // whole rounds of the 256 CUs: time ~ rounds * tile area / rate, 128-row kernels run two workgroups per CU at half rate each
    bool big = false;
    {
        static const int force = getenv("MI355X_GEMM_256") ? atoi(getenv("MI355X_GEMM_256")) : -1;      // 1 force on, 0 off (tuning)
        const int tiles_n256 = (int) ((a.N + 255) / 256);
        const int64_t t256 = (int64_t) count_tm(256) * tiles_n256, tsel = (int64_t) count_tm(BM) * tiles_n;
        const double c256 = (double) ((t256 + 255) / 256) * 65536.0 / 1.2, csel = (double) ((tsel + 511) / 512) * 2.0 * 128.0 * BM;
        big = a.nbatch <= 1 && a.N >= 256 && t256 >= 256 && c256 < csel;
        if (force == 0) big = false;
        if (force == 1 && a.nbatch <= 1) big = true;
    }
    bool any_q = false;
    for (int i = 0; i < a.nmat; ++i) any_q = any_q || a.m[i].qtype != 0;
    for (int i = 0; i < a.nmat; ++i) if ((size_t) a.m[i].M * a.m[i].w_rs >= (1ull << 32)) big = false;     // (k_gemm_f16_ph8 addresses its operands with 32-bit offsets)
    if ((size_t) a.N * a.x_rs >= (1ull << 32)) big = false;
    if (any_q) { BM = G_BM; big = false; }                    // K-quant blocks de-quantised in the staging: the 128 x 128 kernel (few columns by construction)
    if (a.probe_path) { *a.probe_path = (!big && !any_q && a.nbatch <= 1) ? 1 : 0; return; }
    if (big) BM = 256;
    int tm = 0;
    for (int i = 0; i < 3; ++i) {
        const gemm_mat & m = a.m[i < a.nmat ? i : 0];
        g.W[i] = (const char *) m.W; g.w_rs[i] = m.w_rs; g.dst[i] = (char *) m.dst; g.dst_cs[i] = m.dst_cs;
        g.resid[i] = (const char *) m.resid; g.resid_cs[i] = m.resid_cs; g.M[i] = (int) m.M;
        if (i < a.nmat) tm += (int) ((m.M + BM - 1) / BM);
        g.tm_end[i] = tm;
    }
    if (big) {
        for (int i = 0; i < a.nmat; ++i) if (a.m[i].y16) { fprintf(stderr, "[mi355x] gemm: f16 rows out of the epilogue are not a feature of the 256 x 256 kernels (ask with probe_path first)\n"); abort(); }
        if (tm == 0) return;
        const int tiles_n256 = (int) ((a.N + 255) / 256);
        g.nmat = a.nmat; g.X = (const char *) a.X; g.x_rs = a.x_rs; g.N = (int) a.N; g.K = (int) a.K; g.tiles_m = tm; g.tiles_n = tiles_n256;
        g.ksteps_per_split = (int) (a.K / H_BK); g.split_stride = 0;
        g.ne12 = g.r2 = g.r3 = 1; g.w_nb2 = g.w_nb3 = g.x_bs = g.dst_nb2 = g.dst_nb3 = 0;
        constexpr int lds256 = 2 * 2 * 256 * H_ROWB;               // 128 KB
        static const bool one_barrier = getenv("MI355X_GEMM_PH8") && atoi(getenv("MI355X_GEMM_PH8")) == 0;      // (A/B: the one-barrier-per-K-step kernel)
        g.dbg = nullptr; g.out16 = nullptr; g.out16_rs = 0; g.glu_gate = 0;
        if (one_barrier) {
            allow_big_lds((const void *) k_gemm_f16_glds256, lds256, 0);
            k_gemm_f16_glds256<<<dim3((unsigned) (tm * tiles_n256)), dim3(512), lds256, st>>>(g);
        } else {
            static const int abl = getenv("MI355X_GEMM_ABL") ? atoi(getenv("MI355X_GEMM_ABL")) : 0;
            auto go = [&](auto kern, int slot) { allow_big_lds((const void *) kern, lds256, slot); kern<<<dim3((unsigned) (tm * tiles_n256)), dim3(512), lds256, st>>>(g); };
            if (abl == 8) {
                static unsigned long long * dbg = nullptr; static int shown = 0;
                if (!dbg) HIP_CHECK(hipMalloc(&dbg, 24 * 8));
                g.dbg = dbg;
                go(k_gemm_f16_ph8<8, false>, 5);
                if (shown++ < 2) {
                    unsigned long long h[24]; HIP_CHECK(hipStreamSynchronize(st)); HIP_CHECK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
                    for (int w = 0; w < 2; ++w) { fprintf(stderr, "[ph8 stamps] group %d, cycles per K-step:", w); for (int i = 0; i < 12; ++i) fprintf(stderr, " %s%.0f", i % 3 == 0 ? "| " : "", (double) h[w * 12 + i] / (a.K / 64)); fprintf(stderr, "\n"); }
                }
                return;
            }
            if (abl == 1) go(k_gemm_f16_ph8<1, false>, 4); else go(k_gemm_f16_ph8<0, false>, 3);
        }
        ++g_gemm_variant_launches[0];
        return;
    }
    g.nmat = a.nmat; g.X = (const char *) a.X; g.x_rs = a.x_rs; g.N = (int) a.N; g.K = (int) a.K; g.tiles_m = tm; g.tiles_n = tiles_n;
    const int nbatch = a.nbatch > 1 ? a.nbatch : 1;
    g.ne12 = nbatch > 1 ? a.ne12 : 1; g.r2 = nbatch > 1 ? a.r2 : 1; g.r3 = nbatch > 1 ? a.r3 : 1;
    g.w_nb2 = a.w_nb2; g.w_nb3 = a.w_nb3; g.x_bs = a.x_bs; g.dst_nb2 = a.dst_nb2; g.dst_nb3 = a.dst_nb3;
    const int nk = (int) (a.K / H_BK);
    int ksplit = 1;
    bool kq = a.nmat > 0;                                      // every matrix given as K-quant blocks: de-quantise inside the staging
    for (int i = 0; i < a.nmat; ++i) kq = kq && a.m[i].qtype != 0;
    for (int i = 0; i < 3; ++i) g.wtype[i] = a.m[i < a.nmat ? i : 0].qtype;
    if (any_q && (!kq || BM != G_BM || nbatch != 1 || a.K % 256 != 0)) { fprintf(stderr, "[mi355x] gemm: K-quant staging needs every matrix of the launch as K-quant blocks, K %% 256 == 0, no batch\n"); abort(); }
    int64_t m_sum = 0; bool m4 = true;
    for (int i = 0; i < a.nmat; ++i) { m_sum += a.m[i].M; m4 = m4 && a.m[i].M % 4 == 0 && a.m[i].dst_cs % 16 == 0; }
    if (BM == G_BM && nbatch == 1 && a.partial && m4 && (a.nmat == 1 || a.N <= 512)) {      // (the caller hands grouped launches a scratch only up to its own column limit)
        ksplit = pick_ksplit((int64_t) tm * tiles_n, nk, a.N);
        while (ksplit > 1 && (size_t) ksplit * (size_t) m_sum * (size_t) a.N * 4 > a.partial_bytes) --ksplit;
    }
    // a second addend rides in the REDUCTION's epilogue only (it reads both addends before it writes; ggml-alloc usually gives the second ADD's result the memory of
    // its residual operand, which the tile epilogue of an un-split launch would overwrite before k_gemm_add_rows reads it): when the split the caller counted on did
    // not happen (a tuning override of the tile height), split in two anyway
    bool has_r2 = false, r2_alias = false;
    for (int i = 0; i < a.nmat; ++i) { has_r2 = has_r2 || a.m[i].resid2; r2_alias = r2_alias || (a.m[i].resid2 && (const void *) a.m[i].resid2 == (const void *) a.m[i].dst); }
    if (has_r2 && ksplit == 1 && BM == G_BM && nbatch == 1 && a.partial && m4 && nk >= 2 && (size_t) 2 * (size_t) m_sum * (size_t) a.N * 4 <= a.partial_bytes) ksplit = 2;
    if (r2_alias && ksplit == 1) { fprintf(stderr, "[mi355x] gemm: a second addend in the result's own memory needs the split-K reduction (no scratch for it)\n"); abort(); }
    g.ksteps_per_split = (nk + ksplit - 1) / ksplit; g.split_stride = 0;
    if (tm == 0) return;
    for (int i = 0; i < a.nmat; ++i)
        if ((a.m[i].unary >= 0 || a.m[i].y16) && ((ksplit == 1 && (a.m[i].unary >= 0 || kq)) || a.deferred_split || (a.m[i].unary >= 0 && a.m[i].unary != GGML_UNARY_OP_GELU && a.m[i].unary != GGML_UNARY_OP_GELU_QUICK) || !a.m[i].y16 ||
                                                  (a.m[i].y16_ms == 2 && (a.m[i].y16_rs % 8 != 0 || ((uintptr_t) a.m[i].y16 & 7) != 0)) || a.m[i].y16_ms % 2 != 0 || a.m[i].y16_rs % 2 != 0 || a.m[i].M % 4 != 0)) {
            fprintf(stderr, "[mi355x] gemm: f16 rows / an activation out of the epilogue need the split-K reduction launch (ask gemm_f16_small_n_ksplit first), GELU / GELU_QUICK and aligned f16 rows\n"); abort();
        }
    if (ksplit == 1)
        for (int i = 0; i < a.nmat; ++i) if (a.m[i].y16) { g.y16[i] = (char *) a.m[i].y16; g.y16_rs[i] = a.m[i].y16_rs; g.y16_ms[i] = a.m[i].y16_ms; if (!a.m[i].y32) g.dst[i] = nullptr; }
    if (ksplit > 1) {
        // slab s of the scratch holds, matrix after matrix, the dense [N][M_i] partial sums of K range s
        const size_t slab = (size_t) m_sum * (size_t) a.N;
        size_t off = 0;
        for (int i = 0; i < a.nmat; ++i) {
            g.dst[i] = (char *) (a.partial + off); g.dst_cs[i] = (size_t) a.m[i].M * 4; g.resid[i] = nullptr;
            off += (size_t) a.m[i].M * (size_t) a.N;
        }
        g.split_stride = slab * 4;
        if (kq) { k_gemm_kq_glds<<<dim3((unsigned) (tm * tiles_n * ksplit)), dim3(256), 2 * (128 * H_ROWB + H_TILEB), st>>>(g); ++g_gemm_variant_launches[3]; }
        else    k_gemm_f16_glds<2><<<dim3((unsigned) (tm * tiles_n * ksplit)), dim3(256), 2 * (128 * H_ROWB + H_TILEB), st>>>(g);
        if (a.deferred_split && (a.nmat == 1 || a.defer_multi)) { *a.deferred_split = ksplit; return; }          // the caller fuses the reduction into its next kernel
        check_nsplit(ksplit, "gemm_f16_multi");
        gemm_reduce_multi_dev r; r.nmat = a.nmat; r.nsplit = ksplit; r.N = (int) a.N; r.split_elems = slab;
        off = 0; int64_t quads = 0;
        for (int i = 0; i < 3; ++i) {
            const gemm_mat & m = a.m[i < a.nmat ? i : 0];
            r.off[i] = off; r.M[i] = i < a.nmat ? (int) m.M : 0; r.resid[i] = (const char *) m.resid; r.resid_cs[i] = m.resid_cs; r.resid2[i] = (const char *) m.resid2; r.resid2_cs[i] = m.resid2_cs;
            r.dst[i] = (char *) m.dst; r.dst_cs[i] = m.dst_cs;
            if (i < a.nmat && m.y16) { r.unary[i] = m.unary; r.y16[i] = (char *) m.y16; r.y16_rs[i] = m.y16_rs; r.y16_ms[i] = m.y16_ms; if (!m.y32) r.dst[i] = nullptr; }
            if (i < a.nmat) { off += (size_t) m.M * (size_t) a.N; quads += m.M * a.N / 4; }
        }
        if (quads > 0) k_gemm_reduce_multi<<<dim3((unsigned) ((quads + 255) / 256)), dim3(256), 0, st>>>(a.partial, r);
        return;
    }
    if (BM == 64) {
        k_gemm_f16_glds<1><<<dim3((unsigned) (tm * tiles_n), (unsigned) nbatch), dim3(256), 2 * (64 * H_ROWB + H_TILEB), st>>>(g);
    } else if (BM == 192) {
        constexpr int lds192 = 2 * (192 * H_ROWB + H_TILEB);        // 80 KB: two workgroups per CU
        allow_big_lds((const void *) k_gemm_f16_glds<3>, lds192, 1);
        k_gemm_f16_glds<3><<<dim3((unsigned) (tm * tiles_n), (unsigned) nbatch), dim3(256), lds192, st>>>(g);
        ++g_gemm_variant_launches[1];
    } else if (kq) {
        k_gemm_kq_glds<<<dim3((unsigned) (tm * tiles_n)), dim3(256), 2 * (128 * H_ROWB + H_TILEB), st>>>(g);
        ++g_gemm_variant_launches[3];
    } else {
        k_gemm_f16_glds<2><<<dim3((unsigned) (tm * tiles_n), (unsigned) nbatch), dim3(256), 2 * (128 * H_ROWB + H_TILEB), st>>>(g);
    }
    for (int i = 0; i < a.nmat; ++i)                               // (the tile epilogue takes one addend; a second one of an un-split launch goes on top)
        if (a.m[i].resid2 && a.m[i].M * a.N > 0)
            k_gemm_add_rows<<<dim3((unsigned) ((a.m[i].M * a.N / 4 + 255) / 256)), dim3(256), 0, st>>>((char *) a.m[i].dst, a.m[i].dst_cs, (const char *) a.m[i].resid2, a.m[i].resid2_cs, (int) a.m[i].M, (int) a.N);
}

void gemm_f16_mfma(const uint16_t * W, size_t w_rs, const uint16_t * X, size_t x_rs, float * dst, size_t dst_cs,
                   int64_t M, int64_t N, int64_t K, hipStream_t st) {
    gemm_multi_args a;
    a.nmat = 1; a.m[0] = { W, w_rs, dst, dst_cs, M, nullptr, 0 };
    a.X = X; a.x_rs = x_rs; a.N = N; a.K = K; a.partial = nullptr;
    gemm_f16_multi(a, st);
}

} // namespace mi

// mmv2.hip -- the batch-1 decode mat-vec of K-quant weights with DENSE weight loads: every wave streams its contiguous piece of the matrix
// into a wave-private LDS ring by LDS-DMA (`buffer_load_dwordx4 ... lds`: 1 KiB of consecutive bytes per wave instruction, every 128-B line
// requested by exactly one instruction, no VGPRs) and reads whole 144-B / 210-B super-blocks back from LDS in the layout the arithmetic wants.
//
// What is computed is what mmv1.hip computes (reference: ggml_compute_forward_mul_mat ne11 == 1, ggml-cpu/ggml-cpu.c:1210-1402;
// quantize_row_q8_K_ref ggml-quants.c:2555-2592; ggml_vec_dot_q4_K_q8_K / _q6_K_q8_K ggml-cpu/quants.c:550-623 / 705-758; RMS norm
// ggml-cpu/ops.cpp:3517-3566), the integer sub-block sums exact, one f32 partial sum per lane folded on the DPP network.
//
// Why (profiles/r02_microbench.txt, tools/mmv_lab.hip knock-outs): mmv1's 4-lanes-per-block register loads touch every 128-B line of a step with
// three wave instructions (16 blocks = 2.3 KB of lines per 1 KB of data): 11.3 us loads-only for the 56.6 MB gate/up pair against 10.3 us with
// dense loads.  Here the redistribution happens in LDS, which has the bandwidth to spare (a step costs ~40 LDS cycles against ~460 cycles of
// HBM time per CU), and the ring -- not VGPRs -- holds what is in flight (up to 9 KB per wave, 147 KB per CU).
//
// Order of memory operations of a wave (all VMEM of the kernel is inline asm: hipcc drains LDS-DMA with vmcnt(0) at the next use of any
// ordinary load and before __syncthreads, MI355X guide "Pipelining across barriers"; completion is counted by hand -- a wave's VMEM returns
// in order):   activation row (+ norm weights, + residual rows)  ->  DMA of the first S steps  ->  s_waitcnt vmcnt(DMA issued): the row is
// here  ->  norm / Q8_K image (raw s_barrier, no vmcnt drain)  ->  per step: vmcnt(all but the oldest step), ds_read, re-issue the slot.
#include "../llama.cpp-omni_amd/csrc/kernels.hpp"
#include "../llama.cpp-omni_amd/csrc/kernels/mv_dev.hpp"

namespace mi {

typedef __attribute__((address_space(3))) void * mv2_lds_ptr;
static __device__ __forceinline__ uint32_t mv2_lds_addr(const void * p) { return (uint32_t) (uintptr_t) (mv2_lds_ptr) p; }

#ifdef MV2_TRACE          // measurement builds (tools/mmv2_lab.hip): per-wave time stamps (100 MHz s_memrealtime) of the stages of a launch, kept in
// SGPRs and written once at the very end (a store per stamp would sit in the same memory queue as what is being timed)
__device__ unsigned long long * mv2_trace_buf = nullptr;
#define MV2_STAMP_DECL uint32_t tr_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }
#define MV2_STAMP(i) do { tr_[i] = (uint32_t) __builtin_amdgcn_s_memrealtime(); asm volatile("" : "+s"(tr_[i]) :: "memory"); } while (0)
#define MV2_STAMP_FLUSH do { if (mv2_trace_buf) { const int l_ = threadIdx.x & 63; uint32_t v_ = 0; _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) if (l_ == i_) v_ = tr_[i_]; \
    if (l_ < 8) mv2_trace_buf[(size_t) (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + l_] = v_; } } while (0)
#else
#define MV2_STAMP_DECL
#define MV2_STAMP(i) do { } while (0)
#define MV2_STAMP_FLUSH do { } while (0)
#endif
#define MV2_LGKM0()   asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define MV2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int N> static __device__ __forceinline__ void mv2_vmcnt() { static_assert(N >= 0 && N < 64, "vmcnt is 6 bits"); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// lane -> (blk, q): every cycle group of a wave64 ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32; MI355X_MICROARCH.md
// "LDS") holds the 16 blocks of a step with ONE q, so its sixteen 16-byte pieces (header 9 blk, nibbles 9 blk + 1 + 2q (+1), activations
// 17 blk + 4q + k: odd strides) fall on sixteen different bank quads
static __device__ __forceinline__ void mv2_lane_map(int lane, int & blk, int & q) {
    const int l5 = lane & 31;
    const bool ga = l5 < 4 || (l5 >= 12 && l5 < 16) || (l5 >= 20 && l5 < 28);
    blk = ga ? (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12)) : (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16));
    q = (ga ? 0 : 1) + 2 * (lane >> 5);
}

// one piece of PIECE bytes (a step of one row: 16 super-blocks) from `rs` at byte offset soff (wave-uniform) into LDS at lds (wave-uniform),
// lane-linear: PIECE / 1024 instructions of 1 KiB + one dword instruction of 256 B.  v16 = 16 * lane, v4 = 4 * lane (| kill: out of the
// descriptor's range -> no memory traffic, zeros)
template <int PIECE, bool NT>
static __device__ __forceinline__ void mv2_dma_piece(const mv1_rsrc rs, uint32_t soff, uint32_t lds, uint32_t v16, uint32_t v4) {
    static_assert(PIECE == 2304 || PIECE == 3360, "16 Q4_K / Q6_K super-blocks");
    if constexpr (PIECE == 2304) {
        if constexpr (NT)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen nt lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:1024 nt lds\n\t"
                         "buffer_load_dword %2, %3, %4 offen offset:2048 nt lds"
                         :: "s"(lds), "v"(v16), "v"(v4), "s"(rs), "s"(soff) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:1024 lds\n\t"
                         "buffer_load_dword %2, %3, %4 offen offset:2048 lds"
                         :: "s"(lds), "v"(v16), "v"(v4), "s"(rs), "s"(soff) : "memory", "m0");
    } else {                                            // 3360 = 3 x 1024 + 288: the tail as 288 B = 72 dwords -> dwordx4 of 18 lanes; issued as a full 1 KiB
        // instruction whose lanes past byte 3360 of the piece belong to the NEXT step's bytes (harmless: the slot is 3 x 1024 + 1024 wide)
        if constexpr (NT)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen nt lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:1024 nt lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:2048 nt lds\n\t"
                         "buffer_load_dwordx4 %2, %3, %4 offen offset:3072 nt lds"
                         :: "s"(lds), "v"(v16), "v"(v4), "s"(rs), "s"(soff) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:1024 lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:2048 lds\n\t"
                         "buffer_load_dwordx4 %2, %3, %4 offen offset:3072 lds"
                         :: "s"(lds), "v"(v16), "v"(v4), "s"(rs), "s"(soff) : "memory", "m0");
    }
}

// ------------------------------------------------------------------------------------------------ activation prologue, asm loads
// (same arithmetic as mv1_act_issue / mv1_act_finish, mv_dev.hpp)
template <int NW, int XB>
static __device__ __forceinline__ void mv2_rows_issue(const mv1_src s, int K, mv1_act_regs<XB> & r) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *) s.x, (short) 0, K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *) s.nw, (short) 0, s.nw ? K * 4 : 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < XB; ++c) {
        const uint32_t off = (uint32_t) ((wave + c * NW) * 1024 + 16 * lane);
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r.x[c]) : "v"(off), "s"(xr) : "memory");
    }
#pragma unroll
    for (int c = 0; c < XB; ++c) {
        const uint32_t off = (uint32_t) ((wave + c * NW) * 1024 + 16 * lane);
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r.w[c]) : "v"(off), "s"(wr) : "memory");
    }
}
// the wait for the row: ONE statement that also names every destination register, so no copy of a not-yet-written register can be scheduled
// in front of it (N = VMEM instructions issued after the row loads)
template <int N, int XB> static __device__ __forceinline__ void mv2_rows_wait(mv1_act_regs<XB> & r, float & resid) {
    static_assert(XB >= 1 && XB <= 3, "image blocks per wave");
    if constexpr (XB == 1) asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(r.x[0]), "+v"(r.w[0]), "+v"(resid) : [n] "n"(N) : "memory");
    if constexpr (XB == 2) asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(r.x[0]), "+v"(r.w[0]), "+v"(r.x[1]), "+v"(r.w[1]), "+v"(resid) : [n] "n"(N) : "memory");
    if constexpr (XB == 3) asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(r.x[0]), "+v"(r.w[0]), "+v"(r.x[1]), "+v"(r.w[1]), "+v"(r.x[2]), "+v"(r.w[2]), "+v"(resid) : [n] "n"(N) : "memory");
}
template <int NW, int XB>
static __device__ __forceinline__ void mv2_image(const mv1_src s, int K, const mv1_act_regs<XB> & r, char * im, double * red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nb = K >> 8;
    float scale = 1.0f;
    if (s.nw) {
        double ss = 0.0;
#pragma unroll
        for (int c = 0; c < XB; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) ss += (double) (r.x[c][i] * r.x[c][i]);
        ss = wave_sum_f64(ss);
        if (lane == 0) red[wave] = ss;
        MV2_BARRIER();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[w];
        const float mean = (float) (tot / (double) K);
        scale = 1.0f / sqrtf(mean + s.eps);
    }
#pragma unroll
    for (int c = 0; c < XB; ++c) {
        const int ib = wave + c * NW;
        if (ib < nb) {
            f32x4 y = r.x[c];
            if (s.nw) {
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = (r.x[c][i] * scale) * r.w[c][i];
            }
            q8k_block_fast(y, lane, (int8_t *) im + ib * 272, (int16_t *) (im + mv1_img_bs(nb)) + ib * 8, (int16_t *) (im + mv1_img_b16(nb)) + ib * 16, (float *) (im + mv1_img_d(nb)) + ib);
        }
    }
    MV2_BARRIER();
}
// ready-made image (common.hpp layout) -> this file's layout; ordinary loads, complete before any DMA is issued
template <int NW>
static __device__ __forceinline__ void mv2_image_copy(const char * img, int K, char * im) {
    const int nb = K >> 8;
    for (int i = threadIdx.x; i < nb * 16; i += 64 * NW) *(u32x4 *) (im + (i >> 4) * 272 + (i & 15) * 16) = ((const u32x4 *) img)[i];
    for (int i = threadIdx.x; i < nb * 8; i += 64 * NW) {
        const uint32_t p = *(const uint32_t *) (img + K + i * 4);
        *(int16_t *) (im + mv1_img_bs(nb) + i * 2) = (int16_t) ((int) (int16_t) (p & 0xffff) + (int) (int16_t) (p >> 16));
        const int b = i >> 3, s0 = (i & 7) * 2;
        *(int16_t *) (im + mv1_img_b16(nb) + b * 32 + mv1_b16_pos(s0) * 2)     = (int16_t) (p & 0xffff);
        *(int16_t *) (im + mv1_img_b16(nb) + b * 32 + mv1_b16_pos(s0 + 1) * 2) = (int16_t) (p >> 16);
    }
    for (int i = threadIdx.x; i < nb; i += 64 * NW) *(float *) (im + mv1_img_d(nb) + i * 4) = *(const float *) (img + K + (K >> 3) + i * 4);
}

// ================================================================================================= Q4_K body
// per step and row: header + 2 x 16 B of nibbles per lane from the ring slot -> one f32 contribution per lane (see mmv1.hip mv1_q4k for the unpack)
struct mv2_q4k_act { u32x4 a[4]; uint32_t bsw; float yd; };
static __device__ __forceinline__ void mv2_q4k_load_act(const char * im, int nb, int blk, int q, int so, mv2_q4k_act & A) {
    const char * la = im + (so + blk) * 272 + 64 * q;
#pragma unroll
    for (int k = 0; k < 4; ++k) A.a[k] = *(const u32x4 *) (la + 16 * k);
    A.bsw = *(const uint32_t *) (im + mv1_img_bs(nb) + (so + blk) * 16 + 4 * q);
    A.yd  = *(const float *) (im + mv1_img_d(nb) + (so + blk) * 4);
}
static __device__ __forceinline__ void mv2_q4k_dot(const u32x4 H, const u32x4 Q, const u32x4 P, const mv2_q4k_act & A, uint32_t sel, float & acc, float & accm) {
    const uint32_t s_lo = H[1] & 0x3f3f3f3fu, s_hi = (H[3] & 0x0f0f0f0fu) | ((H[1] >> 2) & 0x30303030u);
    const uint32_t m_lo = H[2] & 0x3f3f3f3fu, m_hi = ((H[3] >> 4) & 0x0f0f0f0fu) | ((H[2] >> 2) & 0x30303030u);
    const uint32_t sw = __builtin_amdgcn_perm(s_hi, s_lo, sel), mw = __builtin_amdgcn_perm(m_hi, m_lo, sel);
    const int sc0 = sw & 0xff, sc1 = sw >> 8;
    const int mn0 = mw & 0xff, mn1 = mw >> 8;
    const float dx = h2f((uint16_t) (H[0] & 0xffff)), dmin = h2f((uint16_t) (H[0] >> 16));
    const int bs0 = (int) (int16_t) (A.bsw & 0xffff), bs1 = (int) (int16_t) (A.bsw >> 16);
    int dl = 0, dh = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        dl = dot4(Q[k] & 0x0f0f0f0fu, A.a[0][k], dl); dh = dot4((Q[k] >> 4) & 0x0f0f0f0fu, A.a[2][k], dh);
        dl = dot4(P[k] & 0x0f0f0f0fu, A.a[1][k], dl); dh = dot4((P[k] >> 4) & 0x0f0f0f0fu, A.a[3][k], dh);
    }
    const int isum = mad24(sc0, dl, mul24(sc1, dh));
    const int msum = mad24(mn0, bs0, mul24(mn1, bs1));
    acc  = fmaf(dx * A.yd, (float) isum, acc);
    accm = fmaf(dmin * A.yd, (float) msum, accm);
}

// NW waves per workgroup, XB image blocks per wave, R rows per task (PAIR: row t of the gate and of the up matrix), S ring slots of R x 2304 B
// per wave, NIT = K / 4096 (> 0: the activation fragments of a lane live in registers for the whole kernel; the step loop is unrolled S x NIT so
// slot and fragment indices are literals).  Requires K % 4096 == 0 (launcher).
template <int NW, int XB, int R, int S, int PRE, int NIT, bool PAIR, bool NT>
__global__ void __launch_bounds__(64 * NW) k_mv2_q4k(const mv1_dev a) {
    __shared__ double red[NW];
    MV2_STAMP_DECL;
    MV2_STAMP(0);
    constexpr int PIECE = 2304, SLOTB = R * PIECE, RR = PAIR ? 1 : R;
    const int lane = threadIdx.x & 63;
    const int wiw  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * NW + wiw);
    int mi_ = 0, w0 = 0;
    if (!PAIR) {
#pragma unroll
        for (int i = 0; i < 2; ++i) if (i + 1 < a.nmat && wave >= a.m[i].wave_end) { mi_ = i + 1; w0 = a.m[i].wave_end; }
    }
    const mv1_mat M = mi_ == 0 ? a.m[0] : (mi_ == 1 ? a.m[1] : a.m[2]);
    const int K = a.K, nb = K >> 8;
    const int ntask = PAIR ? M.nrows : M.nrows / R;
    int g0, g1; mv1_range(ntask, wave - w0, M.wave_end - w0, g0, g1);
    g0 = __builtin_amdgcn_readfirstlane(g0); g1 = __builtin_amdgcn_readfirstlane(g1);
    const int nsteps = (g1 - g0) * NIT;

    char * im = mv1_lds;
    const uint32_t ring = mv2_lds_addr(mv1_lds) + (uint32_t) mv1_image_bytes(K) + (uint32_t) wiw * (S * SLOTB);
    const mv1_rsrc rs0 = mv1_make_rsrc(M.W, (size_t) M.nrows * M.w_rs), rs1 = mv1_make_rsrc(PAIR ? a.W1 : M.W, (size_t) M.nrows * M.w_rs);
    const uint32_t rs32 = (uint32_t) M.w_rs;
    const uint32_t v16 = 16u * (uint32_t) lane, v4 = 4u * (uint32_t) lane;

    auto dma = [&](int step, int slot) {                // step of this wave's stream -> ring slot (steps past the end: nothing is fetched)
        const bool live = step < nsteps;
        const int task = g0 + step / NIT, it = step % NIT;
        const uint32_t kill = live ? 0u : MV1_KILL;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = PAIR ? task : task * R + r;
            const uint32_t so = __builtin_amdgcn_readfirstlane(live ? (uint32_t) row * rs32 + (uint32_t) it * (uint32_t) PIECE : 0u);
            mv2_dma_piece<PIECE, NT>((PAIR && r == 1) ? rs1 : rs0, so, ring + (uint32_t) (slot * SLOTB + r * PIECE), v16 | kill, v4 | kill);
        }
    };

    // ---- requests: activation row, residual rows, then the first S steps of the weight stream
    mv1_act_regs<XB> rows;
    float resid = 0.0f;
    if (a.src.img) {
        mv2_image_copy<NW>(a.src.img, K, im);
#pragma unroll
        for (int c = 0; c < XB; ++c) { rows.x[c] = f32x4{0, 0, 0, 0}; rows.w[c] = f32x4{0, 0, 0, 0}; }
    } else mv2_rows_issue<NW, XB>(a.src, K, rows);
    if (!PAIR) {                                        // the residual of this wave's rows, one per lane (launcher: at most 64 rows per wave when resid is given)
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *) M.resid, (short) 0, M.resid ? M.nrows * 4 : 0, 0x00020000);
        const uint32_t off = (uint32_t) (g0 * RR + lane) * 4u;
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(resid) : "v"(off), "s"(rr) : "memory");
    }
    MV2_STAMP(1);
    // A wave's VMEM instructions ISSUE only as fast as the CU's memory queue drains (measured with the stamps: a wave that puts its whole ring
    // in front of the prologue is still issuing at 2.5 - 7 us), so only PRE steps go in before the row is waited for: enough to keep the
    // memory pipe fed while the image is built; the rest of the ring is requested when the image is done.
#pragma unroll
    for (int d = 0; d < PRE; ++d) dma(d, d);
    MV2_STAMP(2);
    mv2_rows_wait<3 * R * PRE, XB>(rows, resid);
    MV2_STAMP(3);
    if (!a.src.img) mv2_image<NW, XB>(a.src, K, rows, im, red);
    else MV2_BARRIER();
    MV2_STAMP(4);
#pragma unroll
    for (int d = PRE; d < S; ++d) dma(d, d);
    MV2_STAMP(5);

    int blk, q; mv2_lane_map(lane, blk, q);
    const uint32_t sel = 0x0c0c0000u | (uint32_t) ((q < 2 ? 0 : 4) + 2 * (q & 1)) | ((uint32_t) ((q < 2 ? 0 : 4) + 2 * (q & 1) + 1) << 8);
    mv2_q4k_act A[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) mv2_q4k_load_act(im, nb, blk, q, it * 16, A[it]);

    const char * wl = mv1_lds + mv1_image_bytes(K) + wiw * (S * SLOTB) + blk * 144;      // lane base inside slot 0, row 0: header; nibbles at + 16 + 32 q
    float acc[R], accm[R], res = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) { acc[r] = 0.0f; accm[r] = 0.0f; }
    int step = 0, nres = 0, res0 = g0 * RR;
    if (nsteps > 0) for (;;) {
        bool done = false;
#pragma unroll
        for (int u = 0; u < S * NIT; ++u) {
            constexpr int dummy = 0; (void) dummy;
            const int slot = u % S, it = u % NIT;
            mv2_vmcnt<3 * R * (S - 1)>();                                   // everything but the S - 1 youngest steps has landed
            if (step == 0) MV2_STAMP(6);
            u32x4 H[R], Q[R], P[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const char * p = wl + slot * SLOTB + r * PIECE;
                H[r] = *(const u32x4 *) p; Q[r] = *(const u32x4 *) (p + 16 + 32 * q); P[r] = *(const u32x4 *) (p + 32 + 32 * q);
            }
            MV2_LGKM0();                                                    // the slot is in registers: refill it
            dma(step + S, slot);
#pragma unroll
            for (int r = 0; r < R; ++r) mv2_q4k_dot(H[r], Q[r], P[r], A[it], sel, acc[r], accm[r]);
            if (it == NIT - 1) {                                            // task finished: fold, keep the result in lane `nres` until the wave stores
                if (PAIR) {
                    const float gsum = wave_sum_f32(acc[0] - accm[0]), usum = wave_sum_f32(acc[R - 1] - accm[R - 1]);
                    if (lane == nres) res = mv1_silu(gsum) * usum;
                    ++nres;
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) { const float s = wave_sum_f32(acc[r] - accm[r]); if (lane == nres) res = s; ++nres; }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) { acc[r] = 0.0f; accm[r] = 0.0f; }
                if (nres + RR > 64) {                                       // (only matrices with more than 64 rows per wave)
                    if (lane < nres) *(float *) (M.dst + (size_t) (res0 + lane) * 4) = res + resid;
                    res0 += nres; nres = 0;
                }
            }
            if (++step >= nsteps) { done = true; break; }
        }
        if (done) break;
    }
    if (lane < nres) *(float *) (M.dst + (size_t) (res0 + lane) * 4) = res + resid;
    mv2_vmcnt<0>();
    MV2_STAMP(7);
    MV2_STAMP_FLUSH;
                                                        // no LDS-DMA may land after the workgroup's LDS is released
}

} // namespace mi

// ------------------------------------------------------------------------------------------------ host side
namespace mi {

bool mmv2_ok(const mv1_args & a) {
    if (!mmv1_ok(a)) return false;
    if (a.nmat < 1 || (a.m[0].type != GGML_TYPE_Q4_K && a.m[0].type != GGML_TYPE_Q6_K)) return false;
    if (a.K != 4096 && a.K != 12288) return false;
    for (int i = 0; i < a.nmat; ++i) {
        if (a.m[i].type != GGML_TYPE_Q4_K) return false;                                       // (Q6_K: mmv1 for now)
        if ((uint64_t) a.m[i].nrows * a.m[i].w_rs >= (uint64_t) MV1_KILL) return false;
        if (((uintptr_t) a.m[i].W & 15) != 0 || a.m[i].w_rs % 16 != 0) return false;
    }
    return true;
}

// ring + image must fit the 160 KiB of a CU (minus the static reduction scratch)
template <int NW, int XB, int R, int S, int PRE, int NIT, bool PAIR, bool NT>
static void mv2_launch(const mv1_dev & d, int grid, hipStream_t st) {
    const size_t lds = mv1_image_bytes(d.K) + (size_t) NW * S * R * 2304;
    static bool attr = false;
    if (!attr) { HIP_CHECK(hipFuncSetAttribute((const void *) k_mv2_q4k<NW, XB, R, S, PRE, NIT, PAIR, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)); attr = true; }
    k_mv2_q4k<NW, XB, R, S, PRE, NIT, PAIR, NT><<<dim3(grid), dim3(64 * NW), lds, st>>>(d);
}

} // namespace mi

// mmv2.hip -- the batch-1 decode mat-vec of K-quant weights as a loader / consumer engine: ONE workgroup of 16 waves per CU; its loader wave(s)
// stream the workgroup's contiguous piece of the matrix into an LDS ring by LDS-DMA (`buffer_load_dwordx4 ... lds`: 1 KiB of consecutive bytes
// per wave instruction, every 128-B line requested by exactly one instruction, no VGPRs) from the first cycle of the launch, while the consumer
// waves build the Q8_K image of the activation row (RMS norm + quantiser) and then read whole 144-B / 210-B super-blocks back from the ring in
// the layout the arithmetic wants.
//
// What is computed is what mmv1.hip computes (reference: ggml_compute_forward_mul_mat ne11 == 1, ggml-cpu/ggml-cpu.c:1210-1402;
// quantize_row_q8_K_ref ggml-quants.c:2555-2592; ggml_vec_dot_q4_K_q8_K / _q6_K_q8_K ggml-cpu/quants.c:550-623 / 705-758; RMS norm
// ggml-cpu/ops.cpp:3517-3566), the integer sub-block sums exact, one f32 partial sum per lane folded on the DPP network.
//
// Why this shape (tools/mmv2_lab.hip -- and a symmetric every-wave-loads-and-consumes variant, removed in round 4: 12.0 vs 11.1 us --, per-wave s_memrealtime stamps; profiles/r03_microbench.txt):
//   * mmv1's 4-lanes-per-block register loads touch every 128-B line of a step with three wave instructions; dense 1 KiB instructions stream
//     a 350 MB Q4_K matrix at 6.4 TB/s against 5.0 TB/s (non-temporal policy, aux nt: +12 %).
//   * a wave's VMEM instructions ISSUE only as fast as its CU's memory queue drains (~25 KB deep): a symmetric kernel whose waves put their ring
//     in front of the prologue is still issuing at 2.5 - 7 us, and its activation row -- queued behind other waves' weight requests -- arrives at
//     4 - 7 us; one that issues only one step first leaves the memory pipe idle for most of a 3 - 5 us prologue.  Separating the roles lets the
//     weight stream run at full depth from the start without delaying the prologue: the loader blocks in issue, the consumers do arithmetic.
//   * the consumers request the activation row before the loader's first weight request (rows_issued counter), so the row is at the head of the queue.
// All synchronisation is inside the CU: LDS counters (ring steps landed / tasks consumed, prologue stages), no s_barrier after the first one -- the
// loader must never stand at a barrier with requests to issue.  The loader's VMEM is inline asm and its completion is counted by hand
// (hipcc drains LDS-DMA with vmcnt(0) at the next ordinary load / barrier: MI355X guide "Pipelining across barriers").
#include "../kernels.hpp"
#include "mv_dev.hpp"
#include "mv2_dev.hpp"

namespace mi {

#ifndef MV2_BLOCK_LANE
#define MV2_BLOCK_LANE 1
#endif
// ================================================================================================= kernel
// mv1_dev as in mmv1.hip, with wave_end = the first WORKGROUP past the matrix's range.  TM: bit0 = Q4_K body compiled in, bit1 = Q6_K.
// NIT = K / 4096.  PAIR: m[0] = gate, W1 = up (same type / shape), epilogue silu(g) * u (Q4_K only).  NL loader waves, 16 - NL consumers.
// The first ten arguments -- 14 dwords, no padding between them: what the loader and the row waves need to put their first requests out -- are marked for KERNARG PRELOAD
// (-mllvm -amdgpu-kernarg-preload-count=14, csrc/Makefile; a hole in the pre-loaded region -- an int in front of a pointer -- faulted in round 6, keep pointers first): the command processor writes them into SGPRs at wave launch, so the launch's critical
// waves do not begin with a ~0.3 us scalar-load round trip.  (Firmware without the feature runs the compiler's compatibility prologue: the same
// loads, as before.)  Everything else (`rest`: destination pointers, the second and third matrix of a grouped launch, a ready-made image) is read
// through the kernarg segment pointer where it is first needed -- never by name, or hipcc hoists its loads to the entry and every early
// s_waitcnt lgkmcnt(0) waits for them.
#define MV2_WGT_PACKED 0x80000000u
static __device__ __forceinline__ int mv2_wgt_type(uint32_t wgt, int i) { const uint32_t c = (wgt >> (24 + 2 * i)) & 3u; return c == 1u ? GGML_TYPE_Q4_K : c == 2u ? GGML_TYPE_Q6_K : GGML_TYPE_Q8_0; }
typedef const __attribute__((address_space(4))) mv2_dev * mv2_karg;
#define MV2_REST_OFFSET 56
template <int TM, int NIT, bool PAIR, bool NT, int NW = MV2_WAVES, bool PARTS = false, int RWK = MV2_ROW_WAVES>
__global__ void __launch_bounds__(64 * NW) k_mv2(const char * W0, const float * x, const float * nw, const char * aux /* pair: the up matrix; one matrix: its residual; a group: the offsets of W1 / W2 */,
                                                         uint32_t w_rs0, uint32_t qr0, uint32_t qr1, uint32_t qr2, float eps, uint32_t wgt /* MV2_WGT: first workgroups of m[1] / m[2], their types */, const mv2_dev rest) {
    __shared__ mv2_flags F;
    __shared__ double red[16];
    static_assert(!PARTS || (!PAIR && NIT == 1), "PARTS: one matrix of K = 4096 = n_head x 128 on the attention slices' partial states (x = the parts buffer)");
    constexpr int PBYTES = MV2_PARTS_NSL * 4096 * 4 + MV2_PARTS_NSL * 32 * 8;                          // the parts buffer (K = 4096)
    constexpr int XS = PARTS ? PBYTES + 512 - 32768 : 0;            // staging: the parts buffer + the fold's coefficient table instead of row + norm weights
    constexpr int RWN = PARTS ? MV2_PARTS_RW : RWK;                   // row waves (RWK: lab sweeps; 4 in the product)
    static_assert(RWN <= NW - 1, "row waves are consumers");
    // Q4_K launches at K = 4096: one super-block per lane (mv2_consume_q4k_b); mixed / Q6_K / Q8_0 launches: the sub-block-pair form.  (K = 12288: a group of 4 rows is 12 of
    // the ring's 29 slots and is released whole -- the loader stalls, ffn_down 7.6 -> 17.9 us; that launch is bound by its stream, not by the consumers' instructions.)
    constexpr bool BL = MV2_BLOCK_LANE && (TM & 1) != 0 && TM != 4 + 1 && NIT == 1;      // (a mixed launch: its Q4_K workgroups)
#ifndef MV2_HALF_BLOCK
#define MV2_HALF_BLOCK 1
#endif
    // one matrix of a few thousand rows at K = 4096 (wo: 16 rows per workgroup, ten waves): half a super-block per lane, two rows per wave step (mv2_consume_q4k_h)
    constexpr bool HBL = MV2_HALF_BLOCK && BL && !PAIR && TM == 1 && (NW == 10 || (MV2_HALF_BLOCK > 1 && NW == 9));
    constexpr int GRB = BL ? (PAIR || HBL ? 2 : 4) : 1;
    constexpr int C = NW - 1;
    constexpr int PW = 4 * NIT < C ? 4 * NIT : C;   // prologue waves
    typedef mv2_geo<2304, 1, NIT, XS> geo0;            // (IMG and STG do not depend on the weight type)
    MV2_STAMP_DECL;
    MV2_STAMP(0);
#ifdef MV2_TRACE
    tr_[6] = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    if (threadIdx.x < 64) tr_[5] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
#endif
    if (threadIdx.x < sizeof(mv2_flags) / 4) ((uint32_t *) &F)[threadIdx.x] = 0u;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int wiw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg  = blockIdx.x;
    // The scalar unit is one per CU: sixteen waves running their set-up at once take ~0.5 us of it.  The loader and the row waves are on the
    // launch's critical path and go first (and at raised priority); the waves that only build the image or only consume stay out of the way.
#ifndef MV2_NO_STAGGER
    if (wiw == 0 || wiw >= NW - RWN) __builtin_amdgcn_s_setprio(3);
    else if (wiw <= PW) __builtin_amdgcn_s_sleep(12);
    else { __builtin_amdgcn_s_sleep(40); }
#endif
    uint64_t kp = (uint64_t) (uintptr_t) __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));                        // (laundered: loads through it stay where they are written)
    const mv2_karg R = (mv2_karg) (uintptr_t) (kp + MV2_REST_OFFSET);
    (void) rest;
    const int K = 4096 * NIT;
    mv2_mat M;
    M.W = W0; M.w_rs = w_rs0; M.nrows = 0; M.q = (int) (qr0 & 0xffffu); M.r = (int) (qr0 >> 16); M.wg0 = 0; M.resid = PAIR ? nullptr : aux; M.dst = nullptr;
    M.type = TM == 2 ? GGML_TYPE_Q6_K : (TM == 4 ? GGML_TYPE_Q8_0 : (TM == 3 ? mv2_wgt_type(wgt, 0) : GGML_TYPE_Q4_K));
    int mi_ = 0;
    const int wg1 = (int) (wgt & 0xfffu);
    if (!PAIR && wg >= wg1) {                           // a workgroup of the second / third matrix
        const int wg2 = (int) ((wgt >> 12) & 0xfffu);
        mi_ = wg >= wg2 ? 2 : 1;
        if (wgt & MV2_WGT_PACKED) {
            // its description is in the pre-loaded scalars too (round 6: as two dependent scalar loads of the argument block -- the matrix index, then the
            // matrix -- these workgroups' first weight request went out 0.9 us after everybody else's and the launch ended with them, tools/mmv2_lab.hip):
            // W = W0 + 16 x a signed 32-bit offset (the two halves of `aux`: a group has neither a pair matrix nor residuals), rows tightly packed
            const uint64_t offs = (uint64_t) (uintptr_t) aux;
            const int32_t off16 = (int32_t) (mi_ == 2 ? (uint32_t) (offs >> 32) : (uint32_t) offs);
            const uint32_t qr = mi_ == 2 ? qr2 : qr1;
            M.W = W0 + (int64_t) off16 * 16; M.q = (int) (qr & 0xffffu); M.r = (int) (qr >> 16); M.wg0 = mi_ == 2 ? wg2 : wg1; M.resid = nullptr;
            M.type = mv2_wgt_type(wgt, mi_);
            M.w_rs = (uint32_t) (K / 256) * (M.type == GGML_TYPE_Q4_K ? 144u : M.type == GGML_TYPE_Q6_K ? 210u : 272u);
        } else {                                        // (a group the scalars cannot describe -- residuals, padded rows, matrices > 32 GB apart: from the argument block)
            const __attribute__((address_space(4))) uint32_t * wsrc = (const __attribute__((address_space(4))) uint32_t *) &R->m[mi_];
            uint32_t wbuf[sizeof(mv2_mat) / 4];
#pragma unroll
            for (size_t k = 0; k < sizeof(mv2_mat) / 4; ++k) wbuf[k] = wsrc[k];
            __builtin_memcpy(&M, wbuf, sizeof M);
        }
    }
    if (!PAIR && (wgt & MV2_WGT_PACKED)) M.resid = nullptr;      // (`aux` holds offsets)
    const char * img = nullptr;
    if (x == nullptr) { asm volatile("" ::: "memory"); img = R->src.img; }      // (a real branch: a speculated scalar load would put a round trip in front of every wave)
    const mv1_src src = { x, nw, eps, img };
    const char * W1 = aux;
    const int lw = wg - M.wg0;
    const int G0 = __builtin_amdgcn_readfirstlane(lw * M.q + (lw < M.r ? lw : M.r));
    const int ntask = __builtin_amdgcn_readfirstlane(M.q + (lw < M.r ? 1 : 0));
    char * im = mv1_lds;
    char * stg = mv1_lds + geo0::IMG;
    char * rstg = stg + geo0::STG;
    char * ringp = rstg + geo0::RSTG;
    constexpr bool Q80 = TM == 4;                        // (Q8_0 matrices never share a launch with K-quants: the activation image differs)
    const bool q4 = TM == 1 || ((TM & 1) && M.type == GGML_TYPE_Q4_K);
    MV2_STAMP(1);
    if (wiw == 0) {
        const size_t wbytes = (size_t) (G0 + ntask) * M.w_rs;      // (the workgroup's rows end here: nothing of the launch is requested past them)
        const mv1_rsrc rs0 = mv1_make_rsrc(M.W, wbytes), rs1 = mv1_make_rsrc(PAIR ? W1 : M.W, wbytes);
        if constexpr (Q80) mv2_loader<4352, PAIR ? 2 : 1, NIT, C, NT, XS, RWN>(rs0, rs1, (uint32_t) M.w_rs, G0, ntask * NIT, mv2_lds_addr(ringp), &F MV2_TR_ARG);
        else if (q4) mv2_loader<2304, PAIR ? 2 : 1, NIT, C, NT, XS, RWN, GRB>(rs0, rs1, (uint32_t) M.w_rs, G0, ntask * NIT, mv2_lds_addr(ringp), &F MV2_TR_ARG);
        else if constexpr ((TM & 2) != 0 && !PAIR) mv2_loader<3360, 1, NIT, C, NT, XS, RWN>(rs0, rs1, (uint32_t) M.w_rs, G0, ntask * NIT, mv2_lds_addr(ringp), &F MV2_TR_ARG);
        MV2_STAMP(7);
    } else {
        const int c = wiw - 1, lane = threadIdx.x & 63;
        const char * resid_p = PAIR ? nullptr : M.resid;
        // roles before the stream is consumed: the last 4 consumers fetch the row, consumers 0 .. 4 NIT - 1 (NIT per SIMD) build the image, the rest wait
        if constexpr (PARTS) { if (c >= C - RWN) mv2_parts_loader((const char *) x, K, c - (C - RWN), resid_p, G0, ntask, mv2_lds_addr(stg), mv2_lds_addr(rstg), &F MV2_TR_ARG); }
        else if (c >= C - RWN) mv2_row_loader<RWN>(src, K, c - (C - RWN), resid_p, G0, ntask, mv2_lds_addr(stg), mv2_lds_addr(rstg), &F MV2_TR_ARG);
        uint32_t img_need = 4 * NIT;
        if constexpr (PARTS) { if (c < 4) mv2_prologue_parts<TM == 4>(K, c, im, stg, (float *) (stg + PBYTES), &F MV2_TR_ARG); }
        else if (src.img) { if constexpr (Q80) mv2_image_copy_q80<C>(src.img, K, c, im, &F); else mv2_image_copy<C>(src.img, K, c, im, &F); img_need = C; }
        else if (c < PW) mv2_prologue<NIT, Q80, PW, RWN>(src, K, c, im, stg, red, &F MV2_TR_ARG);
        { uint32_t spins = 0; while (mv2_peek(MV2_FLAG(F.img_cnt)) < img_need) { __builtin_amdgcn_s_sleep(4); if (++spins > MV2_SPIN_MAX) __builtin_trap(); } asm volatile("" ::: "memory"); }
        // the residual of this consumer's tasks, one per lane (task k of the consumer is row G0 + c + k C), from the staging area: the consumers
        // issue NO vector-memory loads -- one would wait for a place in the CU's memory queue behind the loader's stream
        float resid = 0.0f;
        if (resid_p) { mv2_await(MV2_FLAG(F.x_landed), RWN); const int r = c + lane * C; if (r < ntask) resid = *(const float *) (rstg + r * 4); }
        MV2_STAMP(4);
        char * dst = R->m[mi_].dst;                        // (needed when the first results are stored)
        if constexpr (Q80) mv2_consume_q80<PAIR ? 2 : 1, NIT, C, PAIR, XS>(im, ringp, K, c, ntask, dst, G0 + c, resid, &F);
        else if (q4) {
            if constexpr (HBL) mv2_consume_q4k_h<NIT, C, XS>(im, ringp, K, c, ntask, dst, G0, rstg, resid_p != nullptr, &F);
            else if constexpr (BL) mv2_consume_q4k_b<NIT, C, PAIR, XS>(im, ringp, K, c, ntask, dst, G0, rstg, resid_p != nullptr, &F);
            else mv2_consume_q4k<PAIR ? 2 : 1, NIT, C, PAIR, XS>(im, ringp, K, c, ntask, dst, G0 + c, resid, &F);
        }
        else if constexpr ((TM & 2) != 0 && !PAIR) mv2_consume_q6k<NIT, C, XS>(im, ringp, K, c, ntask, dst, G0 + c, resid, &F);
        MV2_STAMP(7);
    }
    MV2_STAMP_FLUSH;
}

} // namespace mi

// ------------------------------------------------------------------------------------------------ host side
namespace mi {

// per-device state (the omni pinning map puts the LLM on a device other than the first one that launched this kernel): CU count and the
// dynamic-LDS attribute of each instantiation are tracked per device ordinal
static int mv2_dev_ordinal() { int dev = 0; HIP_CHECK(hipGetDevice(&dev)); return dev >= 0 && dev < 64 ? dev : 0; }
int mv2_cus() {
    static int n[64] = { 0 };
    const int dev = mv2_dev_ordinal();
    if (!n[dev]) { hipDeviceProp_t p; HIP_CHECK(hipGetDeviceProperties(&p, dev)); n[dev] = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256; }
    return n[dev];
}

bool mmv2_ok(const mv1_args & a) {
    if (a.nmat < 1 || a.nmat > 3 || (a.K != 4096 && a.K != 12288)) return false;
    if (a.K == 12288 && a.norm_w && !a.img) return false;                                    // (the staging area holds the row OR row + norm weights of 4096)
    const bool q80 = a.m[0].type == GGML_TYPE_Q8_0;                                          // all-Q8_0 launches (the 8B LLM of BASELINE configs[4]): their own activation image
    if (a.W_up && (a.nmat != 1 || (a.m[0].type != GGML_TYPE_Q4_K && !q80) || a.K != 4096 || a.m[0].resid || ((uintptr_t) a.W_up & 15) != 0)) return false;
    const int cus = mv2_cus();
    for (int i = 0; i < a.nmat; ++i) {
        const mmv_mat & m = a.m[i];
        if (q80 ? m.type != GGML_TYPE_Q8_0 : (m.type != GGML_TYPE_Q4_K && m.type != GGML_TYPE_Q6_K)) return false;
        if (m.nrows <= 0 || (uint64_t) m.nrows * m.w_rs > 0xffffffffull) return false;
        if (((uintptr_t) m.W & 15) != 0 || m.w_rs % 16 != 0) return false;                   // LDS-DMA pieces are 16-byte (the last of a Q4_K / Q8_0 step 4-byte) requests
        if (((uintptr_t) m.dst & 3) != 0 || ((uintptr_t) m.resid & 3) != 0) return false;
        if (m.resid && m.nrows > (int64_t) cus * 256) return false;                           // the residual staging area holds 256 rows per workgroup
    }
    if (a.parts)                                                                             // attention slices' partial states: folded in the prologue (k_mv2 PARTS)
        return a.nslice == MV2_PARTS_NSL && a.K == 4096 && a.nmat == 1 && !a.W_up && !a.norm_w && !a.img && !a.x && ((uintptr_t) a.parts & 15) == 0;
    if (a.img) return ((uintptr_t) a.img & 15) == 0;
    return a.x && ((uintptr_t) a.x & 15) == 0 && ((uintptr_t) a.norm_w & 15) == 0;
}

template <int TM, int NIT, bool PAIR, bool NT, int NW = MV2_WAVES, bool PARTS = false, int RWK = MV2_ROW_WAVES>
static void mv2_launch(const mv2_dev & d, int grid, hipStream_t st, const float * parts = nullptr) {
    const size_t lds = 160 * 1024 - 512;                                                       // image + staging + ring: the whole CU (mv2_geo)
    static bool attr[64] = { false };
    const int dev = mv2_dev_ordinal();
    if (!attr[dev]) { HIP_CHECK(hipFuncSetAttribute((const void *) k_mv2<TM, NIT, PAIR, NT, NW, PARTS, RWK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); attr[dev] = true; }
    static_assert(offsetof(mv2_dev, src) % 8 == 0, "argument block layout");
    // the scalars of the launch (see k_mv2): q | r << 16 per matrix; wgt = first workgroup of m[1] | of m[2] << 12 | the three types << 24 | MV2_WGT_PACKED
    auto qr = [&](int i) { return (uint32_t) d.m[i].q | ((uint32_t) d.m[i].r << 16); };
    auto tcode = [&](int i) { return d.m[i].type == GGML_TYPE_Q4_K ? 1u : d.m[i].type == GGML_TYPE_Q6_K ? 2u : 3u; };
    const int wg1 = d.nmat > 1 ? d.m[1].wg0 : grid, wg2 = d.nmat > 2 ? d.m[2].wg0 : grid;
    uint32_t wgt = (uint32_t) wg1 | ((uint32_t) wg2 << 12) | (tcode(0) << 24) | (tcode(d.nmat > 1 ? 1 : 0) << 26) | (tcode(d.nmat > 2 ? 2 : 0) << 28);
    const char * aux = PAIR ? d.W1 : d.m[0].resid;
    bool packed = !PAIR && d.nmat > 1 && grid < 4096;
    int64_t off[3] = { 0, 0, 0 };
    for (int i = 0; i < d.nmat && packed; ++i) {
        const uint32_t tight = (uint32_t) (d.K / 256) * (d.m[i].type == GGML_TYPE_Q4_K ? 144u : d.m[i].type == GGML_TYPE_Q6_K ? 210u : 272u);
        off[i] = ((int64_t) (intptr_t) d.m[i].W - (int64_t) (intptr_t) d.m[0].W) / 16;
        if (d.m[i].resid || d.m[i].q > 0xffff || d.m[i].r > 0xffff || (i > 0 && (d.m[i].w_rs != tight || off[i] != (int64_t) (int32_t) off[i] || (((intptr_t) d.m[i].W - (intptr_t) d.m[0].W) & 15) != 0))) packed = false;
    }
    if (packed) { wgt |= MV2_WGT_PACKED; aux = (const char *) (uintptr_t) ((uint64_t) (uint32_t) (int32_t) off[1] | ((uint64_t) (uint32_t) (int32_t) off[2] << 32)); }
    if (d.m[0].q > 0xffff || d.m[0].r > 0xffff || grid > 4095) { fprintf(stderr, "[mi355x] mmv2: %d rows per workgroup\n", d.m[0].q); abort(); }
    k_mv2<TM, NIT, PAIR, NT, NW, PARTS, RWK><<<dim3(grid), dim3(64 * NW), lds, st>>>(d.m[0].W, PARTS ? parts : (d.src.img ? nullptr : d.src.x), PARTS ? nullptr : d.src.nw, aux, d.m[0].w_rs, qr(0), qr(d.nmat > 1 ? 1 : 0), qr(d.nmat > 2 ? 2 : 0), d.src.eps, wgt, d);
}

// workgroup ranges of the matrices of a launch: by bytes -- a Q6_K matrix beside Q4_K ones counted MV2_Q6_SHARE times (its rows cost more instructions per byte and its
// workgroups run the sub-block-pair consumer: the launch ended with them) --, every matrix at least one workgroup
#ifndef MV2_Q6_SHARE
#define MV2_Q6_SHARE 1.7                      // tools/mmv3_lab.hip, qkv with a Q6_K v: 1.0 6.46 us, 1.3 5.97, 1.6 5.95, 2.0 5.85, 2.5 6.59
#endif
static int mv2_plan(const mv1_args & a, mv2_dev & d, int & tm) {
    const int cus = mv2_cus();
    double bytes[3], total = 0; int64_t tasks = 0; tm = 0;
    for (int i = 0; i < a.nmat; ++i) {
        bytes[i] = (double) a.m[i].nrows * (double) (a.m[i].type == GGML_TYPE_Q4_K ? 144 : a.m[i].type == GGML_TYPE_Q6_K ? 210 : 272) * (double) (a.K / 256);
#ifdef MV2_LAB_NW
        { static const double q6w = getenv("MV2_Q6_WEIGHT") ? atof(getenv("MV2_Q6_WEIGHT")) : MV2_Q6_SHARE; if (a.nmat > 1 && a.m[i].type == GGML_TYPE_Q6_K) bytes[i] *= q6w; }
#else
        if (a.nmat > 1 && a.m[i].type == GGML_TYPE_Q6_K) bytes[i] *= MV2_Q6_SHARE;
#endif
        total += bytes[i]; tasks += a.m[i].nrows;
        tm |= a.m[i].type == GGML_TYPE_Q4_K ? 1 : a.m[i].type == GGML_TYPE_Q6_K ? 2 : 4;
    }
    int grid = (int) (tasks < cus ? tasks : cus);
    if (grid < a.nmat) grid = a.nmat;
    d.nmat = a.nmat; d.K = (int) a.K; d.W1 = (const char *) a.W_up;
    d.src = { a.img ? nullptr : a.x, a.img ? nullptr : a.norm_w, a.eps, (const char *) a.img };
    int acc_w = 0; double acc_b = 0;
    for (int i = 0; i < 3; ++i) {
        if (i >= a.nmat) { d.m[i] = d.m[0]; d.m[i].wg0 = grid; continue; }
        acc_b += bytes[i];
        int end = i == a.nmat - 1 ? grid : (int) (grid * (acc_b / total) + 0.5);
        if (end <= acc_w) end = acc_w + 1;
        if (end > grid - (a.nmat - 1 - i)) end = grid - (a.nmat - 1 - i);
        const int nwg = end - acc_w;
        d.m[i] = { (const char *) a.m[i].W, (char *) a.m[i].dst, (const char *) a.m[i].resid, (uint32_t) a.m[i].w_rs, (int) a.m[i].nrows, a.m[i].type, acc_w, (int) (a.m[i].nrows / nwg), (int) (a.m[i].nrows % nwg) };
        acc_w = end;
    }
    return grid;
}

void mmv2(const mv1_args & a, hipStream_t st) {
    if (!mmv2_ok(a)) { fprintf(stderr, "[mi355x] mmv2: unsupported arguments (K=%lld)\n", (long long) a.K); abort(); }
    mv2_dev d; int tm;
    const int grid = mv2_plan(a, d, tm);
    const bool pair = a.W_up != nullptr;
    // waves per workgroup (1 loader + the consumers), by launch shape: tools/mmv2_lab.hip sweep (profiles/r06_mv2_waves.txt).  The long pair launch keeps sixteen; the short
    // ones run faster with fewer waves contending for the CU's issue slots through the prologue and the tail: the three-matrix group with twelve (6.4 -> 6.1 us; nine with
    // the block-per-lane Q4_K consumer, whose groups of four rows need fewer consumers: 6.1 -> 5.55), one
    // matrix of a few thousand rows with ten (ffn_down Q4_K 8.1 -> 7.5, Q6_K 11.4 -> 9.6, wo 5.0 -> 4.8); the lm-head (hundreds of steps per workgroup) keeps sixteen
    static const bool nw16 = getenv("MI355X_MV2_NW16") != nullptr;                        // A/B: sixteen waves everywhere (the round-5 form)
    const bool small = !nw16 && a.nmat == 1 && a.m[0].nrows <= 16384;
#ifdef MV2_LAB_NW       // lab builds (tools/mmv3_lab.hip: the dependent chain of a layer's four mat-vec launches): waves per workgroup by shape class from the environment
    {
        auto nw_env = [](const char * k, int dflt) { const char * e = getenv(k); return e ? atoi(e) : dflt; };
        static const int nw_pair = nw_env("MV2_NW_PAIR", 16), nw_grp = nw_env("MV2_NW_GRP", 12), nw_s4 = nw_env("MV2_NW_SMALL4", 10), nw_s12 = nw_env("MV2_NW_SMALL12", 10);
#define MV2_LAB_GO(TMv, NITv, PAIRv, nw) do { switch (nw) { case 9: mv2_launch<TMv, NITv, PAIRv, true, 9>(d, grid, st); break; case 10: mv2_launch<TMv, NITv, PAIRv, true, 10>(d, grid, st); break; \
        case 12: mv2_launch<TMv, NITv, PAIRv, true, 12>(d, grid, st); break; case 13: mv2_launch<TMv, NITv, PAIRv, true, 13>(d, grid, st); break; default: mv2_launch<TMv, NITv, PAIRv, true, 16>(d, grid, st); } return; } while (0)
        if (tm != 4 && !a.parts) {
            if (pair) MV2_LAB_GO(1, 1, true, nw_pair);
            if (a.K == 4096 && a.nmat > 1) { if (tm == 1) MV2_LAB_GO(1, 1, false, nw_grp); if (tm == 2) MV2_LAB_GO(2, 1, false, nw_grp); MV2_LAB_GO(3, 1, false, nw_grp); }
            if (a.K == 4096 && small) { if (tm == 1) MV2_LAB_GO(1, 1, false, nw_s4); MV2_LAB_GO(2, 1, false, nw_s4); }
            if (a.K == 12288 && small) { if (tm == 1) MV2_LAB_GO(1, 3, false, nw_s12); MV2_LAB_GO(2, 3, false, nw_s12); }
        }
    }
#endif
    if (tm == 4) {                                                                           // Q8_0 (the 8B LLM of BASELINE configs[4]): the same rule, its own measurements
        static const int q80_nw = getenv("MI355X_MV2_Q80_NW") ? atoi(getenv("MI355X_MV2_Q80_NW")) : -1;      // A/B: 16 = the round-5 form everywhere
        static const int q80_grp = getenv("MI355X_MV2_Q80_NW_GRP") ? atoi(getenv("MI355X_MV2_Q80_NW_GRP")) : -1;
        // (tools/q80_decode_bench.py, three alternating repeats: one matrix 16 / q-k-v group 10: 552-553 tok/s; 10 / 10: 548-549; 16 / 16: 543-554, bimodal; 10 / 16: 545-547)
        const int nwq = nw16 ? 16 : small ? (q80_nw > 0 ? q80_nw : 16) : (a.nmat > 1 && a.K == 4096) ? (q80_grp > 0 ? q80_grp : 10) : 16;
        if (pair)              mv2_launch<4, 1, true, true>(d, grid, st);
        else if (a.parts)      mv2_launch<4, 1, false, true, 10, true>(d, grid, st, a.parts);
        else if (a.K == 4096)  { if (nwq == 10) mv2_launch<4, 1, false, true, 10>(d, grid, st); else if (nwq == 12) mv2_launch<4, 1, false, true, 12>(d, grid, st); else mv2_launch<4, 1, false, true>(d, grid, st); }
        else                   { if (nwq == 10) mv2_launch<4, 3, false, true, 10>(d, grid, st); else if (nwq == 12) mv2_launch<4, 3, false, true, 12>(d, grid, st); else mv2_launch<4, 3, false, true>(d, grid, st); }
        return;
    }
    if (pair)               { mv2_launch<1, 1, true, true>(d, grid, st); return; }
    if (a.parts) {
        if (tm == 1) mv2_launch<1, 1, false, true, 10, true>(d, grid, st, a.parts);
        else         mv2_launch<2, 1, false, true, 10, true>(d, grid, st, a.parts);
        return;
    }
    if (a.K == 4096) {
        if (a.nmat > 1 && !nw16) {
            if (tm == 1)      mv2_launch<1, 1, false, true, 9>(d, grid, st);
            else if (tm == 2) mv2_launch<2, 1, false, true, 12>(d, grid, st);
            else              mv2_launch<3, 1, false, true, 9>(d, grid, st);
        } else if (small) {
            if (tm == 1)      mv2_launch<1, 1, false, true, 10>(d, grid, st);
            else              mv2_launch<2, 1, false, true, 10>(d, grid, st);
        } else {
            if (tm == 1)      mv2_launch<1, 1, false, true>(d, grid, st);
            else if (tm == 2 && a.nmat == 1 && !nw16) mv2_launch<2, 1, false, true, 9>(d, grid, st);      // the Q6_K lm-head (151936 rows): 78.9 us at sixteen waves, 76.4 at nine (tools/mmv2_lab.hip shape 8)
            else if (tm == 2) mv2_launch<2, 1, false, true>(d, grid, st);
            else              mv2_launch<3, 1, false, true>(d, grid, st);
        }
    } else {
        if (a.nmat > 1) {
            if (tm == 1)      mv2_launch<1, 3, false, true>(d, grid, st);
            else if (tm == 2) mv2_launch<2, 3, false, true>(d, grid, st);
            else              mv2_launch<3, 3, false, true>(d, grid, st);
        } else if (small) {
            if (tm == 1)      mv2_launch<1, 3, false, true, 10>(d, grid, st);
            else              mv2_launch<2, 3, false, true, 10>(d, grid, st);
        } else {
            if (tm == 1)      mv2_launch<1, 3, false, true>(d, grid, st);
            else              mv2_launch<2, 3, false, true>(d, grid, st);
        }
    }
}

} // namespace mi

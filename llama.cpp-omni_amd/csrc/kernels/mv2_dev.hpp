// mv2_dev.hpp -- device side of the loader / consumer LDS-DMA mat-vec engine shared by mmv2.hip (one launch per graph node group) and mmv3.hip
// (several dependent launches as stages of ONE persistent launch): LDS flags, the LDS-DMA piece loader, the Q8_K row quantiser on DPP rows,
// the Q4_K / Q6_K consumers.  See mmv2.hip for the design notes and measurements.
#pragma once
#include "../kernels.hpp"
#include "mv_dev.hpp"

namespace mi {

typedef __attribute__((address_space(3))) void * mv2_lds_ptr;
static __device__ __forceinline__ uint32_t mv2_lds_addr(const void * p) { return (uint32_t) (uintptr_t) (mv2_lds_ptr) p; }

#ifdef MV2_TRACE          // measurement builds (tools/mmv2_lab.hip): per-wave time stamps (100 MHz s_memrealtime) of the stages of a launch, kept in
// SGPRs and written once at the very end (a store per stamp would sit in the same memory queue as what is being timed)
__device__ unsigned long long * mv2_trace_buf = nullptr;
#define MV2_STAMP_DECL uint32_t tr_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }
#define MV2_TR_PARAM , uint32_t (&tr_)[8]
#define MV2_TR_ARG , tr_
#define MV2_STAMP(i) do { tr_[i] = (uint32_t) __builtin_amdgcn_s_memrealtime(); asm volatile("" : "+s"(tr_[i]) :: "memory"); } while (0)
#define MV2_STAMP_FLUSH do { if (mv2_trace_buf) { const int l_ = threadIdx.x & 63; uint32_t v_ = 0; _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) if (l_ == i_) v_ = tr_[i_]; \
    if (l_ < 8) mv2_trace_buf[(size_t) (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + l_] = v_; } } while (0)
#else
#define MV2_STAMP_DECL
#define MV2_TR_PARAM
#define MV2_TR_ARG
#define MV2_STAMP(i) do { } while (0)
#define MV2_STAMP_FLUSH do { } while (0)
#endif

#define MV2_LGKM0()   asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int N> static __device__ __forceinline__ void mv2_vmcnt() { static_assert(N >= 0 && N < 64, "vmcnt is 6 bits"); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

#ifndef MV2_WAVES
#define MV2_WAVES 16           // waves per workgroup: 1 loader + MV2_WAVES - 1 consumers (lab builds: 8 / 12)
#endif
#ifndef MV2_ROW_WAVES
#define MV2_ROW_WAVES 4          // consumers that fetch the activation row before they consume
#endif

// kernel arguments: everything a wave needs is a load and an integer multiply-add away (no division, no search): workgroups [wg0, next wg0) stream
// matrix m; workgroup lw of them owns rows lw * q + min(lw, r) .. (q + 1 rows in the first r workgroups)
struct mv2_mat { const char * W; char * dst; const char * resid; uint32_t w_rs; int nrows; int type; int wg0; int q; int r; };
struct mv2_dev { mv2_mat m[3]; int nmat; const char * W1; mv1_src src; int K; };

// ---- LDS flags (all monotonic, zeroed before the launch's one s_barrier)
struct mv2_flags {
    uint32_t landed;                // steps of the workgroup's stream that have landed in the ring
    uint32_t rows_issued;           // row waves that have requested their part of the activation row -- the weight stream starts behind them
    uint32_t x_landed;              // row waves whose part of the activation row (and of the residual rows) is in the staging area
    uint32_t rows_landed;           // ... and of the norm weights
    uint32_t sum_cnt, img_cnt;      // prologue waves that have published their sum of squares / finished their image blocks
    uint32_t scale_ready; float scale;   // the RMS-norm scale, computed by the prologue wave that arrived last at sum_cnt
    uint32_t consumed[16];          // per consumer: tasks whose ring slots it has read into registers
};
// The flags are touched through address-space-3 pointers ONLY: through a generic pointer hipcc emits flat_load / flat_store + s_waitcnt
// vmcnt(0) for a volatile access, and a vmcnt(0) in the loader drains its whole DMA window every step (measured: 28 -> 8 GB/s per loader wave,
// tools/dma_bench.hip).  As DS instructions they count on lgkmcnt and leave the DMA queue alone.
typedef __attribute__((address_space(3))) uint32_t mv2_lds_u32;
#define MV2_FLAG(x) ((mv2_lds_u32 *) &(x))
static __device__ __forceinline__ uint32_t mv2_peek(const mv2_lds_u32 * p) { const uint32_t v = *(const volatile mv2_lds_u32 *) p; return __builtin_amdgcn_readfirstlane(v); }
static __device__ __forceinline__ void mv2_poke(mv2_lds_u32 * p, uint32_t v) { if ((threadIdx.x & 63) == 0) *(volatile mv2_lds_u32 *) p = v; }
static __device__ __forceinline__ void mv2_arrive(mv2_lds_u32 * p) {         // after this wave's LDS stores: the DS operations of a wave execute in order
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
// every spin of the engine is bounded: a protocol error must end in a trap, never in a hung GPU.  The consumers wait for a loader wave of the same
// resident workgroup, so forward progress is guaranteed and the bound only guards against protocol bugs: ~0.1 s in measurement / debug builds
// (MV2_TRACE, MV2_DEBUG), tens of seconds otherwise -- a slow but correct run (PC sampling, a debugger, a throttled clock) must not become a fault
#ifndef MV2_SPIN_MAX
#if defined(MV2_TRACE) || defined(MV2_DEBUG)
#define MV2_SPIN_MAX (1u << 21)
#else
#define MV2_SPIN_MAX (1u << 28)
#endif
#endif
#ifndef MV2_SLEEP
#define MV2_SLEEP 1
#endif
static __device__ __forceinline__ void mv2_await(const mv2_lds_u32 * p, uint32_t n) {
    uint32_t spins = 0;
    while (mv2_peek(p) < n) { __builtin_amdgcn_s_sleep(1); if (++spins > MV2_SPIN_MAX) __builtin_trap(); }
    asm volatile("" ::: "memory");
}

// lane -> (blk, q) of the Q4_K body: every cycle group of a wave64 ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32;
// MI355X_MICROARCH.md "LDS") holds the 16 blocks of a step with ONE q, so its sixteen 16-byte pieces (header 9 blk, nibbles 9 blk + 1 + 2q (+1);
// activations 17 blk + 4q + k: odd strides) fall on sixteen different bank quads
static __device__ __forceinline__ void mv2_lane_map(int lane, int & blk, int & q) {
    const int l5 = lane & 31;
    const bool ga = l5 < 4 || (l5 >= 12 && l5 < 16) || (l5 >= 20 && l5 < 28);
    blk = ga ? (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12)) : (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16));
    q = (ga ? 0 : 1) + 2 * (lane >> 5);
}

// one piece (a step of one row: 16 super-blocks, 2304 B of Q4_K / 3360 B of Q6_K) from `rs` at byte offset soff (wave-uniform) into LDS at lds
// (wave-uniform), lane-linear.  Q4_K: two 1 KiB instructions + one dword instruction of 256 B.  Q6_K: three 1 KiB instructions + 288 B as a
// 16-byte instruction of 18 lanes.  v16 = 16 * lane, v4 = 4 * lane.
template <int PIECE, bool NT>
static __device__ __forceinline__ void mv2_dma_piece(const mv1_rsrc rs, uint32_t soff, uint32_t lds, uint32_t v16, uint32_t v4) {
    static_assert(PIECE == 2304 || PIECE == 3360 || PIECE == 4352, "16 Q4_K / Q6_K super-blocks, or 128 Q8_0 blocks");
    if constexpr (PIECE == 4352) {
        // Q8_0: 4096 weights = 128 blocks of 34 B = four 1 KiB instructions + 256 B as a dword instruction.  The instruction offset field ends at 4095, and
        // it is the instruction offset that advances the LDS address: the fifth instruction gets its own m0 and scalar offset instead (s_mov, not s_add: an
        // s_add_u32 inside the statement would clobber SCC under the compiler's feet -- the loader's ring-wrap compare sat across it).
        if constexpr (NT)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen nt lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:1024 nt lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:2048 nt lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:3072 nt lds\n\t"
                         "s_mov_b32 m0, %6\n\ts_nop 0\n\t"
                         "buffer_load_dword %2, %3, %5 offen nt lds"
                         :: "s"(lds), "v"(v16), "v"(v4), "s"(rs), "s"(soff), "s"(soff + 4096u), "s"(lds + 4096u) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:1024 lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:2048 lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:3072 lds\n\t"
                         "s_mov_b32 m0, %6\n\ts_nop 0\n\t"
                         "buffer_load_dword %2, %3, %5 offen lds"
                         :: "s"(lds), "v"(v16), "v"(v4), "s"(rs), "s"(soff), "s"(soff + 4096u), "s"(lds + 4096u) : "memory", "m0");
    } else if constexpr (PIECE == 2304) {
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
    } else {
        uint64_t keep;
        if constexpr (NT)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %2, %3, %4 offen nt lds\n\t"
                         "buffer_load_dwordx4 %2, %3, %4 offen offset:1024 nt lds\n\t"
                         "buffer_load_dwordx4 %2, %3, %4 offen offset:2048 nt lds\n\t"
                         "s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0x3ffff\n\t"
                         "buffer_load_dwordx4 %2, %3, %4 offen offset:3072 nt lds\n\t"
                         "s_mov_b64 exec, %0"
                         : "=&s"(keep) : "s"(lds), "v"(v16), "s"(rs), "s"(soff) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                         "buffer_load_dwordx4 %2, %3, %4 offen offset:1024 lds\n\t"
                         "buffer_load_dwordx4 %2, %3, %4 offen offset:2048 lds\n\t"
                         "s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0x3ffff\n\t"
                         "buffer_load_dwordx4 %2, %3, %4 offen offset:3072 lds\n\t"
                         "s_mov_b64 exec, %0"
                         : "=&s"(keep) : "s"(lds), "v"(v16), "s"(rs), "s"(soff) : "memory", "m0");
        (void) v4;
    }
}

// ring geometry of a (type, rows per task, K) combination: as many slots of one step as the CU's LDS holds next to the image
template <int PIECE, int R, int NIT, int XS = 0> struct mv2_geo {       // XS: extra staging bytes (the attention slices' partial states, MV2 PARTS)
    static constexpr int VM    = (PIECE == 2304 ? 3 : PIECE == 3360 ? 4 : 5) * R;   // VMEM instructions per step
    static constexpr int D     = 60 / VM;                                     // steps the loader keeps in flight (vmcnt counts to 63)
    static constexpr int B     = R == 2 ? 2 : 4;                              // steps per loader round: one flag round trip per round
    static constexpr int SLOTB = PIECE * R;
    static constexpr int IMG   = NIT * 16 * 324 + 16;                         // mv1_image_bytes(4096 * NIT)
    static constexpr int STG   = (NIT == 1 ? 32768 : 49152) + XS;             // f32 row + norm weights (K = 4096), row only (K = 12288)
    static constexpr int RSTG  = 1024;                                        // the residual of the workgroup's rows (at most 256)
    static constexpr int NS    = (160 * 1024 - 512 - IMG - STG - RSTG) / SLOTB;
    static_assert(NS >= D + NIT + B, "ring too small");              // (+ a consumer sweep of slack where the loader uses the conservative test)
};

// ================================================================================================= loader
// Loader l of NL streams steps l, l + NL, ... of the workgroup's T steps (step t = task t / NIT, K-slice t % NIT; a task is one row, or row j of
// the gate and of the up matrix) into ring slot t % NS, keeps up to D of its steps in flight, publishes landed[l], and waits for the consumer of a
// slot's previous tenant before it overwrites the slot.
template <int VM, int K_> struct mv2_drain {            // at most K_ of my n steps outstanding -> n - K_ have landed; K_ down to 0 as literals
    static __device__ __forceinline__ void go(int n, mv2_lds_u32 * landed) {
        if (n > K_) { mv2_vmcnt<VM * K_>(); mv2_poke(landed, (uint32_t) (n - K_)); }
        if constexpr (K_ > 0) mv2_drain<VM, K_ - 1>::go(n, landed);
    }
};
template <int PIECE, int R, int NIT, int C, bool NT, int XS = 0, int RWN = MV2_ROW_WAVES, int GR = 1>
static __device__ __forceinline__ void mv2_loader(const mv1_rsrc rs0, const mv1_rsrc rs1, uint32_t rs32, int G0, int T, uint32_t ring, mv2_flags * F MV2_TR_PARAM) {
    typedef mv2_geo<PIECE, R, NIT, XS> geo;
    constexpr int VM = geo::VM, D = geo::D, B = geo::B, SLOTB = geo::SLOTB, NS = geo::NS;
    const int lane = threadIdx.x & 63;
    const uint32_t v16 = 16u * (uint32_t) lane, v4 = 4u * (uint32_t) lane;
    // the loader's per-step state lives in SGPRs and advances by additions: its instruction stream is what bounds a single wave's request rate
    uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t) G0 * rs32);          // byte offset of the next step in the matrix
    const uint32_t row_skip = __builtin_amdgcn_readfirstlane(rs32 - (uint32_t) (NIT * PIECE));   // 0 for tightly packed rows
    uint32_t la = __builtin_amdgcn_readfirstlane(ring);                          // LDS address of the next slot
    const uint32_t ring_end = __builtin_amdgcn_readfirstlane(ring + (uint32_t) (NS * SLOTB));
    int it = 0;
    auto issue = [&]() {
#pragma unroll
        for (int r = 0; r < R; ++r) mv2_dma_piece<PIECE, NT>(r == 1 ? rs1 : rs0, so, la + (uint32_t) (r * PIECE), v16, v4);
        so += (uint32_t) PIECE; la += (uint32_t) SLOTB;
        if (la == ring_end) la = __builtin_amdgcn_readfirstlane(ring);
        if (NIT > 1) { if (++it == NIT) { it = 0; so += row_skip; } } else so += row_skip;
    };
    // ring space: every task below free_tasks has been read by its consumer.  Consumer c reads its tasks c, c + C, ... in order, so the first task
    // not yet read is min over c of (c + consumed[c] * C): ONE LDS read + a 16-lane minimum, and only when the cached bound no longer covers the round
    int n = 0, free_tasks = 0;                          // steps issued; tasks known consumed
#ifndef MV2_PRE
#define MV2_PRE 0
#endif
    // the first step goes out at once -- it takes the loader's cold-start latency (address translation, first DRAM page) in parallel with the row
    // waves' -- the rest of the stream behind the row requests
    for (; n < MV2_PRE && n < T; ++n) issue();
    // (the gate / up launch is bound by this wave's stream -- its consumers start with > 1 us of slack -- so its stream does not wait for the row requests; the short
    //  launches are bound by when consumption can start, there the row goes first)
#ifndef MV2_PAIR_WAITS
    if (R != 2)                                         // (tried for ffn_down too -- 48 KB row: 8.1 -> 9.3 us; and for the 65 KB of attention slices, PARTS: wo 5.4 -> 6.8 us -- whatever is
#endif                                                  //  requested behind the weight stream arrives behind it)
    mv2_await(MV2_FLAG(F->rows_issued), RWN);
    MV2_STAMP(2);
    while (n < T) {
        const int nb_ = T - n < B ? T - n : B;
        const int last = n + nb_ - 1;                   // the round's last step overwrites the slot of step last - NS = task (last - NS) / NIT
        if (last >= NS) {
            const int need = (last - NS) / NIT + 1;     // tasks that must have been consumed
            uint32_t spins = 0;
            while (free_tasks < need) {
                uint32_t k = lane < C ? (uint32_t) lane + *(const volatile mv2_lds_u32 *) MV2_FLAG(F->consumed[lane & 15]) * (uint32_t) C : 0xffffffffu;
                // (each DPP value is taken ONCE, under the full exec mask: re-evaluated inside a select it runs under a partial mask and reads 0)
                { const uint32_t o = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) k, 0xB1, 0xf, 0xf, false);  k = __builtin_elementwise_min(k, o); }
                { const uint32_t o = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) k, 0x4E, 0xf, 0xf, false);  k = __builtin_elementwise_min(k, o); }
                { const uint32_t o = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) k, 0x141, 0xf, 0xf, false); k = __builtin_elementwise_min(k, o); }
                { const uint32_t o = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) k, 0x140, 0xf, 0xf, false); k = __builtin_elementwise_min(k, o); }
                free_tasks = (int) __builtin_amdgcn_readfirstlane(k) * GR;          // (GR > 1: consumers count GROUPS of GR rows, mv2_consume_q4k_b)
                if (free_tasks < need) { __builtin_amdgcn_s_sleep(1); if (++spins > MV2_SPIN_MAX) __builtin_trap(); }
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < B; ++i) if (i < nb_) issue();
        n += nb_;
        if (n > D - B) { mv2_vmcnt<VM * (D - B)>(); mv2_poke(MV2_FLAG(F->landed), (uint32_t) (n - (D - B))); }
    }
    mv2_drain<VM, D - B - 1>::go(n, MV2_FLAG(F->landed));     // ends with vmcnt(0): no LDS-DMA may land after the workgroup's LDS is released
}

// ================================================================================================= row loader (consumer 0, before it consumes)
// The activation row (f32), the norm weights and the residual of the workgroup's rows go into the staging area by LDS-DMA, requested before the
// loader's first weight request (rows_issued), so they are at the head of the CU's memory queue; nothing else of this wave is in flight, so
// vmcnt(0) is exactly "the row is here".
template <int RWN = MV2_ROW_WAVES>
static __device__ __forceinline__ void mv2_row_loader(const mv1_src src, int K, int rw, const char * resid, int G0, int ntask, uint32_t stg, uint32_t rstg, mv2_flags * F MV2_TR_PARAM) {
    const int lane = threadIdx.x & 63;
    const uint32_t v16 = 16u * (uint32_t) lane, v4 = 4u * (uint32_t) lane;
#ifdef MV2_ROWSTART
    mv2_arrive(MV2_FLAG(F->rows_issued));
#endif
    // Only the row itself (and the residual) is requested in front of the weight stream; the norm weights -- needed a microsecond later, when the
    // sum of squares is known -- go in behind the loader's first requests.  Piece b (1 KiB) of either belongs to row wave b % MV2_ROW_WAVES.
    const int nb = K >> 8;
    if (!src.img) {
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *) src.x, (short) 0, K * 4, 0x00020000);
        MV2_STAMP(2);
        for (int b = rw, i = 0; b < nb; b += RWN, ++i) {
            if (i == 1) MV2_STAMP(3);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(__builtin_amdgcn_readfirstlane(stg + (uint32_t) b * 1024u)), "v"(v16), "s"(xr), "s"(__builtin_amdgcn_readfirstlane((uint32_t) b * 1024u)) : "memory", "m0");
        }
    }
    if (resid && rw == 0) {                             // rows G0 .. G0 + ntask: 64 per instruction
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *) (resid + (size_t) G0 * 4), (short) 0, ntask * 4, 0x00020000);
        for (int b = 0; b * 64 < ntask; ++b)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" :: "s"(__builtin_amdgcn_readfirstlane(rstg + (uint32_t) b * 256u)), "v"(v4), "s"(rr), "s"(__builtin_amdgcn_readfirstlane((uint32_t) b * 256u)) : "memory", "m0");
    }
    MV2_STAMP(5);
    mv2_arrive(MV2_FLAG(F->rows_issued));             // (measured: letting the loader start before ALL of a 48 KB row is requested delays the row more than it gains)
    if (!src.img && src.nw) {                           // (K = 4096: the launcher refuses a norm at K = 12288) 4 pieces per wave
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *) src.nw, (short) 0, K * 4, 0x00020000);
        static_assert(16 % RWN == 0, "norm weights of K = 4096: 16 pieces over the row waves");
#pragma unroll
        for (int i = 0; i < 16 / RWN; ++i) {
            const int b = rw + i * RWN;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(__builtin_amdgcn_readfirstlane(stg + (uint32_t) (K * 4) + (uint32_t) b * 1024u)), "v"(v16), "s"(wr), "s"(__builtin_amdgcn_readfirstlane((uint32_t) b * 1024u)) : "memory", "m0");
        }
        mv2_vmcnt<16 / RWN>();                           // everything but the norm-weight pieces: the row is here
        mv2_arrive(MV2_FLAG(F->x_landed));
    } else {
        mv2_vmcnt<0>();
        mv2_arrive(MV2_FLAG(F->x_landed));
    }
    mv2_vmcnt<0>();
    mv2_arrive(MV2_FLAG(F->rows_landed));
}

// ================================================================================================= consumers: activation prologue
// Four prologue waves (one per SIMD: the arithmetic below is VALU-issue-bound, and sixteen waves doing it at once took 3 - 4 us of a 12 us launch)
// build the image; every DPP row of 16 lanes owns one 256-block: lane i of the row holds elements 64 m + 4 i .. + 3, m = 0..3.
// Reference arithmetic (quantize_row_q8_K_ref, ggml-quants.c:2555-2592): iscale = -127 / max, q = nearest_int(iscale x), d = 1 / iscale, bsums.
//   * `max` is AN element of largest magnitude (key = the float's bits rotated left by one: magnitude above sign, one v_max_u32 per element).  The
//     reference takes the first one; with the other sign iscale, every q, every bsum and d change sign together and every product the mat-vec forms
//     is the same number, so the launch's results are identical (the stand-alone quantiser, quantize.hip, keeps the reference's tie rule).
//   * nearest_int is the reference's own magic-number form: the low byte of bits(p + 12582912.f); its min(127, .) can never bind (|p| <= 127 (1 + 2^-22)).
template <int CTRL> static __device__ __forceinline__ uint32_t mv2_dpp_row(uint32_t v) { return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, CTRL, 0xf, 0xf, true); }
static __device__ __forceinline__ void mv2_q8k_rows(const f32x4 (&y)[4], int lane, int b, int nb, char * im) {
    const int i = lane & 15;
    uint32_t key = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const uint32_t u = __float_as_uint(y[m][e]); key = umax32(key, __builtin_amdgcn_alignbit(u, u, 31)); }
    key = umax32(key, mv2_dpp_row<0xB1>(key));          // quad_perm [1,0,3,2]
    key = umax32(key, mv2_dpp_row<0x4E>(key));          // quad_perm [2,3,0,1]
    key = umax32(key, mv2_dpp_row<0x141>(key));         // row_half_mirror
    key = umax32(key, mv2_dpp_row<0x140>(key));         // row_mirror: every lane of the row holds the row's key
    const bool zero = (key >> 1) == 0u;                 // all-zero block
    const float mval = __uint_as_float(__builtin_amdgcn_alignbit(key, key, 1));
    const float iscale = zero ? 0.0f : -127.0f / mval;
    char * qs = im + b * 272;
    int16_t * bs32 = (int16_t *) (im + mv1_img_bs(nb)) + b * 8;
    int16_t * b16  = (int16_t *) (im + mv1_img_b16(nb)) + b * 16;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        uint32_t t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float p = iscale * y[m][e]; t[e] = __float_as_uint(p + 12582912.0f); }
        const uint32_t lo = __builtin_amdgcn_perm(t[1], t[0], 0x0c0c0400u), hi = __builtin_amdgcn_perm(t[3], t[2], 0x04000c0cu);
        const uint32_t qd = lo | hi;
        *(uint32_t *) (qs + 64 * m + 4 * i) = qd;
        int s = dot4(qd, 0x01010101u, 0);
        s += (int) mv2_dpp_row<0xB1>((uint32_t) s);
        s += (int) mv2_dpp_row<0x4E>((uint32_t) s);      // sum of 16: bsums[4m + i / 4] in every lane of the quad
        if ((i & 3) == 0) b16[mv1_b16_pos(4 * m + (i >> 2))] = (int16_t) s;
        s += (int) mv2_dpp_row<0x141>((uint32_t) s);     // + the neighbouring quad: sum of 32
        if ((i & 7) == 0) bs32[2 * m + (i >> 3)] = (int16_t) s;
    }
    if (i == 0) *((float *) (im + mv1_img_d(nb)) + b) = zero ? 0.0f : 1.0f / iscale;
}

// The Q8_0 activation image ([qs : K int8][d : K / 32 f32], common.hpp q80_image_bytes) from the same register layout: a 32-element block is the 8 lanes
// i = 0..7 or 8..15 of the DPP row for one m.  Arithmetic of the compiled x86 quantiser (arch/x86/quants.c:290-345, as quantize.hip / mmv1q.hip): d = amax / 127
// stored as f16, id = 127 / amax, q = round-half-even(x * id).
static __device__ __forceinline__ void mv2_q80_rows(const f32x4 (&y)[4], int lane, int b, int K, char * im) {
    const int i = lane & 15;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        float amax = fmaxf(fmaxf(fabsf(y[m][0]), fabsf(y[m][1])), fmaxf(fabsf(y[m][2]), fabsf(y[m][3])));
        amax = fmaxf(amax, __uint_as_float(mv2_dpp_row<0xB1>(__float_as_uint(amax))));       // (non-negative floats: the bit patterns order like the values)
        amax = fmaxf(amax, __uint_as_float(mv2_dpp_row<0x4E>(__float_as_uint(amax))));
        amax = fmaxf(amax, __uint_as_float(mv2_dpp_row<0x141>(__float_as_uint(amax))));      // row_half_mirror: the 8 lanes of the block
        const float d = amax / 127.0f;
        const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
        uint32_t qd = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) qd |= ((uint32_t) (int) __builtin_rintf(y[m][e] * id) & 0xffu) << (8 * e);
        *(uint32_t *) (im + 256 * b + 64 * m + 4 * i) = qd;
        if ((i & 7) == 0) *(float *) (im + K + (8 * b + 2 * m + (i >> 3)) * 4) = h2f(f2h(d));
    }
}

// prologue wave pw of PW = min(4 NIT, C) (consumers 0 .. PW - 1): image block groups mw = pw, pw + PW, ... < 4 NIT, group mw = blocks 4 mw + row.
// With sixteen waves PW = 4 NIT and every wave owns one group (NIT per SIMD); narrower workgroups loop.  A wave publishes the sums of ALL its
// groups before it waits for the scale (the last arrival -- of 4 NIT -- computes it).
template <int NIT, bool Q80 = false, int PW = 4 * NIT, int RWN = MV2_ROW_WAVES>
static __device__ __forceinline__ void mv2_prologue(const mv1_src s, int K, int pw, char * im, const char * stg, double * red, mv2_flags * F MV2_TR_PARAM) {
    const int lane = threadIdx.x & 63, row = lane >> 4, i = lane & 15, nb = K >> 8;
    constexpr int NG = (4 * NIT + PW - 1) / PW;          // groups per wave (at most)
    mv2_await(MV2_FLAG(F->x_landed), RWN);
    MV2_STAMP(2);
    float scale = 1.0f;
    f32x4 x[NG][4];
    if (s.nw) {                                         // RMS norm: sum of squares in double like the reference
        bool last = false;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int mw = pw + g * PW;
            if (mw >= 4 * NIT) break;
            const char * xp = stg + (mw * 4 + row) * 1024 + 16 * i;
#pragma unroll
            for (int m = 0; m < 4; ++m) x[g][m] = *(const f32x4 *) (xp + 256 * m);
            double ss = 0.0;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) ss += (double) (x[g][m][e] * x[g][m][e]);
            ss = wave_sum_f64(ss);
            if (lane == 0) red[mw] = ss;
            asm volatile("" ::: "memory");
            uint32_t prev = 0;
            if (lane == 0) prev = __hip_atomic_fetch_add(MV2_FLAG(F->sum_cnt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            prev = __builtin_amdgcn_readfirstlane(prev);
            asm volatile("" ::: "memory");
            if (prev == (uint32_t) (4 * NIT - 1)) last = true;
        }
        if (last) {                                     // last to arrive: every partial sum is in LDS (the DS operations of a wave execute in order)
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < 4 * NIT; ++w) tot += *(const volatile double *) &red[w];
            const float mean = (float) ((K & (K - 1)) == 0 ? tot * (1.0 / (double) K) : tot / (double) K);     // (a power of two: the same double)
            const float sc = 1.0f / sqrtf(mean + s.eps);
            if (lane == 0) *(volatile __attribute__((address_space(3))) float *) &F->scale = sc;
            mv2_poke(MV2_FLAG(F->scale_ready), 1u);
            scale = sc;
        } else {
            mv2_await(MV2_FLAG(F->scale_ready), 1u);
            scale = *(const volatile __attribute__((address_space(3))) float *) &F->scale;
        }
    }
    MV2_STAMP(3);
    if (s.nw) mv2_await(MV2_FLAG(F->rows_landed), RWN);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int mw = pw + g * PW;
        if (mw >= 4 * NIT) break;
        const int b = mw * 4 + row;
        f32x4 y[4];
        if (s.nw) {
            const char * wp = stg + K * 4 + b * 1024 + 16 * i;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 w = *(const f32x4 *) (wp + 256 * m);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[m][e] = (x[g][m][e] * scale) * w[e];
            }
        } else {
            const char * xp = stg + b * 1024 + 16 * i;
#pragma unroll
            for (int m = 0; m < 4; ++m) y[m] = *(const f32x4 *) (xp + 256 * m);
        }
        if constexpr (Q80) mv2_q80_rows(y, lane, b, K, im); else mv2_q8k_rows(y, lane, b, nb, im);
        mv2_arrive(MV2_FLAG(F->img_cnt));
    }
    MV2_STAMP(5);
}
// ================================================================================================= PARTS: the activation row arrives as attention slices' partial states
// (fattn_one.hip k_fattn_gs: NSL x [K] unnormalised partial outputs O_s, then NSL x [K / 128 heads] x (M_s, S_s) -- one contiguous buffer of NSL K 4 + NSL K / 16 bytes,
// K = n_head x 128).  RWN row waves bring it into the staging area (1 KiB pieces), the prologue waves fold it
//     x[h, d] = sum_s f_s O_s[h, d] / sum_s f_s S_s[h],  f_s = exp(M_s[h] - max_s M_s[h])      (flash-decoding's merge, here in front of the quantiser every workgroup runs anyway)
// and quantise the folded row like any other.
#define MV2_PARTS_NSL 4
#define MV2_PARTS_RW  8
static __device__ __forceinline__ void mv2_parts_loader(const char * parts, int K, int rw, const char * resid, int G0, int ntask, uint32_t stg, uint32_t rstg, mv2_flags * F MV2_TR_PARAM) {
    const int lane = threadIdx.x & 63;
    const uint32_t v16 = 16u * (uint32_t) lane, v4 = 4u * (uint32_t) lane;
    const int bytes = MV2_PARTS_NSL * K * 4 + MV2_PARTS_NSL * (K / 128) * 8, np = (bytes + 1023) >> 10;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *) parts, (short) 0, bytes, 0x00020000);
    MV2_STAMP(2);
    for (int b = rw, i = 0; b < np; b += MV2_PARTS_RW, ++i) {
        if (i == 1) MV2_STAMP(3);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(__builtin_amdgcn_readfirstlane(stg + (uint32_t) b * 1024u)), "v"(v16), "s"(xr), "s"(__builtin_amdgcn_readfirstlane((uint32_t) b * 1024u)) : "memory", "m0");
    }
    if (resid && rw == 0) {                             // rows G0 .. G0 + ntask: 64 per instruction
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *) (resid + (size_t) G0 * 4), (short) 0, ntask * 4, 0x00020000);
        for (int b = 0; b * 64 < ntask; ++b)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" :: "s"(__builtin_amdgcn_readfirstlane(rstg + (uint32_t) b * 256u)), "v"(v4), "s"(rr), "s"(__builtin_amdgcn_readfirstlane((uint32_t) b * 256u)) : "memory", "m0");
    }
    MV2_STAMP(5);
    mv2_arrive(MV2_FLAG(F->rows_issued));
    mv2_vmcnt<0>();
    mv2_arrive(MV2_FLAG(F->x_landed));
    mv2_arrive(MV2_FLAG(F->rows_landed));
}
// prologue wave pw of 4 (K = 4096): blocks 4 pw + row; a 256-block is two heads (lane's m = 0, 1: head 2 b; m = 2, 3: head 2 b + 1)
template <bool Q80 = false>
static __device__ __forceinline__ void mv2_prologue_parts(int K, int pw, char * im, const char * stg, float * coef /* [16 blocks][8] */, mv2_flags * F MV2_TR_PARAM) {
    const int lane = threadIdx.x & 63, row = lane >> 4, i = lane & 15, nb = K >> 8, nh = K >> 7;
    constexpr int NSL = MV2_PARTS_NSL;
    mv2_await(MV2_FLAG(F->x_landed), MV2_PARTS_RW);
    MV2_STAMP(2);
    const int b = pw * 4 + row;
    // the block's 2 x NSL coefficients f_s / sum_s f_s S_s: lane i of the DPP row takes (head i >> 2 & 1, slice i & 3) -- lanes 8 .. 15 repeat 0 .. 7 --, the slices of a head are a quad
    {
        const int h = 2 * b + ((i >> 2) & 1), sl = i & 3;
        const u32x2 ms = *(const u32x2 *) (stg + (size_t) NSL * K * 4 + ((size_t) sl * nh + h) * 8);
        const float Ms = __uint_as_float(ms[0]), Ss = __uint_as_float(ms[1]);
        float M = fmaxf(Ms, __uint_as_float(mv2_dpp_row<0xB1>(__float_as_uint(Ms))));
        M = fmaxf(M, __uint_as_float(mv2_dpp_row<0x4E>(__float_as_uint(M))));
        const float f = Ms == -INFINITY ? 0.0f : expf(Ms - M);
        float den = Ss * f;
        den += __uint_as_float(mv2_dpp_row<0xB1>(__float_as_uint(den)));
        den += __uint_as_float(mv2_dpp_row<0x4E>(__float_as_uint(den)));
        const float cf = den == 0.0f ? 0.0f : f * (1.0f / den);
        if (i < 8) coef[b * 8 + i] = cf;                // (the wave's own LDS operations execute in order: the reads below see it)
    }
    asm volatile("" ::: "memory");
    const f32x4 c0 = *(const f32x4 *) (coef + b * 8), c1 = *(const f32x4 *) (coef + b * 8 + 4);
    f32x4 y[4];
    const char * xp = stg + b * 1024 + 16 * i;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4 cc = m < 2 ? c0 : c1;
        f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
            const f32x4 o = *(const f32x4 *) (xp + (size_t) sl * K * 4 + 256 * m);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(cc[sl], o[e], acc[e]);
        }
        y[m] = acc;
    }
    MV2_STAMP(3);
    if constexpr (Q80) mv2_q80_rows(y, lane, b, K, im); else mv2_q8k_rows(y, lane, b, nb, im);      // (the Q8_0 image of an all-Q8_0 model's wo: the same register layout)
    mv2_arrive(MV2_FLAG(F->img_cnt));
    MV2_STAMP(5);
}

// ready-made Q8_0 image (the same layout): a plain copy by all C consumers
template <int C>
static __device__ __forceinline__ void mv2_image_copy_q80(const char * img, int K, int c, char * im, mv2_flags * F) {
    const int lane = threadIdx.x & 63;
    for (int i = c * 64 + lane; i < (K + K / 8) / 16; i += 64 * C) ((u32x4 *) im)[i] = ((const u32x4 *) img)[i];
    mv2_arrive(MV2_FLAG(F->img_cnt));
}
// ready-made image (common.hpp layout) -> this family's layout, by all C consumers
template <int C>
static __device__ __forceinline__ void mv2_image_copy(const char * img, int K, int c, char * im, mv2_flags * F) {
    const int lane = threadIdx.x & 63, nb = K >> 8;
    for (int i = c * 64 + lane; i < nb * 16; i += 64 * C) *(u32x4 *) (im + (i >> 4) * 272 + (i & 15) * 16) = ((const u32x4 *) img)[i];
    for (int i = c * 64 + lane; i < nb * 8; i += 64 * C) {
        const uint32_t p = *(const uint32_t *) (img + K + i * 4);
        *(int16_t *) (im + mv1_img_bs(nb) + i * 2) = (int16_t) ((int) (int16_t) (p & 0xffff) + (int) (int16_t) (p >> 16));
        const int b = i >> 3, s0 = (i & 7) * 2;
        *(int16_t *) (im + mv1_img_b16(nb) + b * 32 + mv1_b16_pos(s0) * 2)     = (int16_t) (p & 0xffff);
        *(int16_t *) (im + mv1_img_b16(nb) + b * 32 + mv1_b16_pos(s0 + 1) * 2) = (int16_t) (p >> 16);
    }
    for (int i = c * 64 + lane; i < nb; i += 64 * C) *(float *) (im + mv1_img_d(nb) + i * 4) = *(const float *) (img + K + (K >> 3) + i * 4);
    mv2_arrive(MV2_FLAG(F->img_cnt));
}

// wait until step t of the workgroup's stream is in the ring
static __device__ __forceinline__ void mv2_wait_step(int t, uint32_t & seen, mv2_flags * F) {
    uint32_t spins = 0;
    while (seen <= (uint32_t) t) { seen = mv2_peek(MV2_FLAG(F->landed)); if (seen <= (uint32_t) t) { __builtin_amdgcn_s_sleep(MV2_SLEEP); if (++spins > MV2_SPIN_MAX) __builtin_trap(); } }
    asm volatile("" ::: "memory");
}

// results of up to 64 tasks sit one per lane until the wave stores them
struct mv2_out { float res; int nres, k0; };
template <int C>
static __device__ __forceinline__ void mv2_out_flush(mv2_out & o, char * dst, int row0, float resid) {
    const int lane = threadIdx.x & 63;
    if (lane < o.nres) *(float *) (dst + (size_t) (row0 + (o.k0 + lane) * C) * 4) = o.res + resid;
    o.k0 += o.nres; o.nres = 0;
}

// ================================================================================================= Q4_K consumer
// (the unpack: mmv1.hip mv1_q4k; 144-B super-block = 16-B header {d, dmin, 12 B of 6-bit scales / mins} + 128 B of nibbles, ggml-common.h:295-305)
struct mv2_q4k_act { u32x4 a[4]; uint32_t bsw; float yd; };
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
static __device__ __forceinline__ void mv2_q4k_act_load(const char * im, int nb, int ib, int q, mv2_q4k_act & A) {
    const char * la = im + ib * 272 + 64 * q;
#pragma unroll
    for (int k = 0; k < 4; ++k) A.a[k] = *(const u32x4 *) (la + 16 * k);
    A.bsw = *(const uint32_t *) (im + mv1_img_bs(nb) + ib * 16 + 4 * q);
    A.yd  = *(const float *) (im + mv1_img_d(nb) + ib * 4);
}
static __device__ __forceinline__ uint32_t mv2_q4k_sel(int q) { return 0x0c0c0000u | (uint32_t) ((q < 2 ? 0 : 4) + 2 * (q & 1)) | ((uint32_t) ((q < 2 ? 0 : 4) + 2 * (q & 1) + 1) << 8); }
template <int R, int NIT, int C, bool PAIR, int XS = 0>
static __device__ __forceinline__ void mv2_consume_q4k(const char * im, const char * ringp, int K, int c, int ntask, char * dst, int row0, float resid, mv2_flags * F) {
    typedef mv2_geo<2304, R, NIT, XS> geo;
    constexpr int SLOTB = geo::SLOTB, NS = geo::NS;
    const int lane = threadIdx.x & 63, nb = K >> 8;
    int blk, q; mv2_lane_map(lane, blk, q);
    const uint32_t sel = mv2_q4k_sel(q);
    mv2_q4k_act A[NIT];                                 // this lane's activation fragments, for the whole kernel
#pragma unroll
    for (int it = 0; it < NIT; ++it) mv2_q4k_act_load(im, nb, it * 16 + blk, q, A[it]);
    const char * wl = ringp + blk * 144;                // lane base inside a slot: header; nibbles at + 16 + 32 q
    uint32_t seen = 0;
    mv2_out o = { 0.0f, 0, 0 };
    int k = 0;
    for (int j = c; j < ntask; j += C, ++k) {
        float acc[R], accm[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { acc[r] = 0.0f; accm[r] = 0.0f; }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int t = __builtin_amdgcn_readfirstlane(j * NIT + it);
            mv2_wait_step(t, seen, F);
            const char * p = wl + (t % NS) * SLOTB;
            u32x4 H[R], Q[R], P[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { H[r] = *(const u32x4 *) (p + r * 2304); Q[r] = *(const u32x4 *) (p + r * 2304 + 16 + 32 * q); P[r] = *(const u32x4 *) (p + r * 2304 + 32 + 32 * q); }
            if (it == NIT - 1) { MV2_LGKM0(); mv2_poke(MV2_FLAG(F->consumed[c]), (uint32_t) (k + 1)); }      // the task's slots are in registers
#pragma unroll
            for (int r = 0; r < R; ++r) mv2_q4k_dot(H[r], Q[r], P[r], A[it], sel, acc[r], accm[r]);
        }
        if (PAIR) {
            const float gsum = wave_sum_f32(acc[0] - accm[0]), usum = wave_sum_f32(acc[R - 1] - accm[R - 1]);
            if (lane == o.nres) o.res = mv1_silu(gsum) * usum;
        } else {
            const float s = wave_sum_f32(acc[0] - accm[0]);
            if (lane == o.nres) o.res = s;
        }
        if (++o.nres == 64) mv2_out_flush<C>(o, dst, row0, resid);
    }
    mv2_out_flush<C>(o, dst, row0, resid);
}

// ================================================================================================= Q4_K consumer, one super-block per LANE (round 6)
// mv2_consume_q4k gives a (block, sub-block pair) to a lane: 64 lanes = the 16 blocks of ONE row's step, ~90 VALU instructions per row and step of which 40 are the
// nibble unpack + dot4 -- the 6-bit scale unpack is repeated by the four lanes of a block, and every row pays a 64-lane sum (17 instructions with their DPP wait states).
// Short launches are bound by exactly that instruction stream: their weights sit in the ring when the image is ready, and 16 .. 24 rows on a CU's four SIMDs took 0.7 ..
// 1.6 us (tools/mmv2_lab.hip time lines; the same times with the weights resident in L2, MV2_NROT=1).  Here a lane takes a WHOLE super-block: GR = 4 rows per wave step
// (lane = (row r of the group, block): a DPP row of 16 lanes is one row of the matrix) -- or GR = 2 row PAIRS (lane = (r, gate | up, block)) --, the scale unpack once per
// block, ONE integer sum and one fma per block, and a 16-lane DPP-row sum that serves the four rows at once: ~54 instructions per row and step.  Same integers, same
// products as the reference's vec_dot (quants.c:550-623); the float sums are taken in a different order (block sums exact in int32, then 16 per row).
// Consumers count GROUPS (consumed[c] = groups read; the loader's free-slot rule multiplies by GR).  A group's rows are consecutive tasks: lane row j = g GR + r reads
// ring slot (j NIT + it) % NS; rows past the workgroup's last task read a valid slot and are dropped.
template <int NIT, int C, bool PAIR, int XS = 0>
static __device__ __forceinline__ void mv2_consume_q4k_b(const char * im, const char * ringp, int K, int c, int ntask, char * dst, int G0, const char * rstg, bool has_resid, mv2_flags * F) {
    constexpr int R = PAIR ? 2 : 1, GR = PAIR ? 2 : 4;
    typedef mv2_geo<2304, R, NIT, XS> geo;
    constexpr int SLOTB = geo::SLOTB, NS = geo::NS;
    const int lane = threadIdx.x & 63, nb = K >> 8;
    const int blk = lane & 15, r = PAIR ? lane >> 5 : lane >> 4, mat = PAIR ? (lane >> 4) & 1 : 0;
    const int ng = (ntask + GR - 1) / GR;
    uint32_t seen = 0;
    int k = 0;
    for (int g = c; g < ng; g += C, ++k) {
        const int j = g * GR + r;                           // this lane's task (row, or gate / up row pair) of the workgroup
        const bool valid = j < ntask;
        const int jl = (g * GR + GR <= ntask ? g * GR + GR : ntask) - 1;        // the group's last existing task
        float accv = 0.0f, accm = 0.0f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            mv2_wait_step(__builtin_amdgcn_readfirstlane(jl * NIT + it), seen, F);      // (steps land in order: the earlier rows' slices are in)
            const int t = (valid ? j : jl) * NIT + it;
            const int b0 = __builtin_amdgcn_readfirstlane((g * GR * NIT + it) % NS);    // slot of the group's first row; this lane's: + (t - first) , wrapped
            int slot = b0 + (t - (g * GR * NIT + it));
            if (slot >= NS) slot -= NS;
            const char * p = ringp + slot * SLOTB + mat * 2304 + blk * 144;
            const u32x4 H = *(const u32x4 *) p;
            u32x4 Q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) Q[e] = *(const u32x4 *) (p + 16 + 16 * e);
            const int ib = it * 16 + blk;
            const char * la = im + ib * 272;
            const u32x4 bsv = *(const u32x4 *) (im + mv1_img_bs(nb) + ib * 16);            // bsums of the block's eight 32-element sub-blocks (int16)
            const float yd = *(const float *) (im + mv1_img_d(nb) + ib * 4);
            if (it == NIT - 1) { MV2_LGKM0(); mv2_poke(MV2_FLAG(F->consumed[c]), (uint32_t) (k + 1)); }      // the group's slots are in registers
            // 6-bit scales / mins of the eight sub-blocks, one byte each (get_scale_min_k4, ggml-quants.c:703-710)
            const uint32_t s_lo = H[1] & 0x3f3f3f3fu, s_hi = (H[3] & 0x0f0f0f0fu) | ((H[1] >> 2) & 0x30303030u);
            const uint32_t m_lo = H[2] & 0x3f3f3f3fu, m_hi = ((H[3] >> 4) & 0x0f0f0f0fu) | ((H[2] >> 2) & 0x30303030u);
            int isum = 0, msum = 0;
#pragma unroll
            for (int jg = 0; jg < 4; ++jg) {                // 64 weights: low nibbles = sub-block 2 jg, high nibbles = sub-block 2 jg + 1
                const u32x4 a0 = *(const u32x4 *) (la + 64 * jg), a1 = *(const u32x4 *) (la + 64 * jg + 16), a2 = *(const u32x4 *) (la + 64 * jg + 32), a3 = *(const u32x4 *) (la + 64 * jg + 48);
                int dl = 0, dh = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dl = dot4(Q[2 * jg][e] & 0x0f0f0f0fu, a0[e], dl); dh = dot4((Q[2 * jg][e] >> 4) & 0x0f0f0f0fu, a2[e], dh);
                    dl = dot4(Q[2 * jg + 1][e] & 0x0f0f0f0fu, a1[e], dl); dh = dot4((Q[2 * jg + 1][e] >> 4) & 0x0f0f0f0fu, a3[e], dh);
                }
                const uint32_t sw = jg < 2 ? s_lo : s_hi, mw = jg < 2 ? m_lo : m_hi;
                const int sh = 16 * (jg & 1);
                const int sc0 = (int) ((sw >> sh) & 0xffu), sc1 = (int) ((sw >> (sh + 8)) & 0xffu), mn0 = (int) ((mw >> sh) & 0xffu), mn1 = (int) ((mw >> (sh + 8)) & 0xffu);
                const uint32_t bw = bsv[jg];
                isum = mad24(sc0, dl, mad24(sc1, dh, isum));
                msum = mad24(mn0, (int) (int16_t) (bw & 0xffff), mad24(mn1, (int) (int16_t) (bw >> 16), msum));
            }
            const float dx = h2f((uint16_t) (H[0] & 0xffff)), dmin = h2f((uint16_t) (H[0] >> 16));
            accv = fmaf(dx * yd, (float) isum, accv);
            accm = fmaf(dmin * yd, (float) msum, accm);
        }
        float v = accv - accm;
        v += __uint_as_float(mv2_dpp_row<0xB1>(__float_as_uint(v))); v += __uint_as_float(mv2_dpp_row<0x4E>(__float_as_uint(v)));
        v += __uint_as_float(mv2_dpp_row<0x141>(__float_as_uint(v))); v += __uint_as_float(mv2_dpp_row<0x140>(__float_as_uint(v)));       // the 16 blocks of the row: every lane of the DPP row holds the sum
        if constexpr (PAIR) {
            const float o = __shfl_xor(v, 16, 64);              // gate rows (mat 0) receive up's sum
            if (valid && mat == 0 && blk == 0) *(float *) (dst + (size_t) (G0 + j) * 4) = mv1_silu(v) * o;
        } else {
            if (valid && blk == 0) *(float *) (dst + (size_t) (G0 + j) * 4) = v + (has_resid ? *(const float *) (rstg + j * 4) : 0.0f);
        }
    }
}

// ================================================================================================= Q4_K consumer, HALF a super-block per lane (round 6)
// mv2_consume_q4k_b keeps 4 rows x 16 blocks in a wave step: a workgroup's 16 rows of a 4096 x 4096 matrix (wo) are four such steps -- four of the nine consumers work for
// 1.0 - 1.4 us (~300 dependent-ish instructions at one wave per SIMD) while five have nothing to do.  Here a lane takes (row r of TWO, half h of the block's eight sub-blocks,
// block): eight wave steps of half the length for the same rows, one per consumer.  Same integers per sub-block; a block's two half sums are added after the 16-lane row sum
// (v_permlane16_swap), i.e. the float order is (sum over blocks of half 0) + (sum over blocks of half 1).  Groups are TWO rows (the loader's GR).
template <int NIT, int C, int XS = 0>
static __device__ __forceinline__ void mv2_consume_q4k_h(const char * im, const char * ringp, int K, int c, int ntask, char * dst, int G0, const char * rstg, bool has_resid, mv2_flags * F) {
    constexpr int GR = 2;
    typedef mv2_geo<2304, 1, NIT, XS> geo;
    constexpr int SLOTB = geo::SLOTB, NS = geo::NS;
    const int lane = threadIdx.x & 63, nb = K >> 8;
    const int blk = lane & 15, h = (lane >> 4) & 1, r = lane >> 5;
    const int ng = (ntask + GR - 1) / GR;
    uint32_t seen = 0;
    int k = 0;
    for (int g = c; g < ng; g += C, ++k) {
        const int j = g * GR + r;
        const bool valid = j < ntask;
        const int jl = (g * GR + GR <= ntask ? g * GR + GR : ntask) - 1;
        float accv = 0.0f, accm = 0.0f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            mv2_wait_step(__builtin_amdgcn_readfirstlane(jl * NIT + it), seen, F);
            const int t = (valid ? j : jl) * NIT + it;
            const int b0 = __builtin_amdgcn_readfirstlane((g * GR * NIT + it) % NS);
            int slot = b0 + (t - (g * GR * NIT + it));
            if (slot >= NS) slot -= NS;
            const char * p = ringp + slot * SLOTB + blk * 144;
            const u32x4 H = *(const u32x4 *) p;
            u32x4 Q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) Q[e] = *(const u32x4 *) (p + 16 + 64 * h + 16 * e);
            const int ib = it * 16 + blk;
            const char * la = im + ib * 272 + 128 * h;
            u32x4 a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = *(const u32x4 *) (la + 16 * e);
            const u32x2 bsv = *(const u32x2 *) (im + mv1_img_bs(nb) + ib * 16 + 8 * h);       // bsums of the half's four sub-blocks (int16)
            const float yd = *(const float *) (im + mv1_img_d(nb) + ib * 4);
            if (it == NIT - 1) { MV2_LGKM0(); mv2_poke(MV2_FLAG(F->consumed[c]), (uint32_t) (k + 1)); }
            // 6-bit scales / mins of this half's four sub-blocks, one byte each (get_scale_min_k4, ggml-quants.c:703-710): half 0 = sub-blocks 0..3, half 1 = 4..7
            const uint32_t sw = h ? ((H[3] & 0x0f0f0f0fu) | ((H[1] >> 2) & 0x30303030u)) : (H[1] & 0x3f3f3f3fu);
            const uint32_t mw = h ? (((H[3] >> 4) & 0x0f0f0f0fu) | ((H[2] >> 2) & 0x30303030u)) : (H[2] & 0x3f3f3f3fu);
            int isum = 0, msum = 0;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {                  // 64 weights: low nibbles = sub-block 2 jg, high nibbles = sub-block 2 jg + 1 (jg = 2 h + tj)
                int dl = 0, dh = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dl = dot4(Q[2 * tj][e] & 0x0f0f0f0fu, a[4 * tj][e], dl);         dh = dot4((Q[2 * tj][e] >> 4) & 0x0f0f0f0fu, a[4 * tj + 2][e], dh);
                    dl = dot4(Q[2 * tj + 1][e] & 0x0f0f0f0fu, a[4 * tj + 1][e], dl); dh = dot4((Q[2 * tj + 1][e] >> 4) & 0x0f0f0f0fu, a[4 * tj + 3][e], dh);
                }
                const int sh = 16 * tj;
                const int sc0 = (int) ((sw >> sh) & 0xffu), sc1 = (int) ((sw >> (sh + 8)) & 0xffu), mn0 = (int) ((mw >> sh) & 0xffu), mn1 = (int) ((mw >> (sh + 8)) & 0xffu);
                const uint32_t bw = bsv[tj];
                isum = mad24(sc0, dl, mad24(sc1, dh, isum));
                msum = mad24(mn0, (int) (int16_t) (bw & 0xffff), mad24(mn1, (int) (int16_t) (bw >> 16), msum));
            }
            const float dx = h2f((uint16_t) (H[0] & 0xffff)), dmin = h2f((uint16_t) (H[0] >> 16));
            accv = fmaf(dx * yd, (float) isum, accv);
            accm = fmaf(dmin * yd, (float) msum, accm);
        }
        float v = accv - accm;
        v += __uint_as_float(mv2_dpp_row<0xB1>(__float_as_uint(v))); v += __uint_as_float(mv2_dpp_row<0x4E>(__float_as_uint(v)));
        v += __uint_as_float(mv2_dpp_row<0x141>(__float_as_uint(v))); v += __uint_as_float(mv2_dpp_row<0x140>(__float_as_uint(v)));       // the 16 blocks of this (row, half)
        { const auto sw2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float((uint32_t) sw2[0]) + __uint_as_float((uint32_t) sw2[1]); }   // + the other half: DPP rows 0|1, 2|3
        if (valid && blk == 0 && h == 0) *(float *) (dst + (size_t) (G0 + j) * 4) = v + (has_resid ? *(const float *) (rstg + j * 4) : 0.0f);
    }
}

// ================================================================================================= Q6_K consumer
// 210-B super-block {ql[128], qh[64], int8 scales[16], f16 d} (ggml-common.h:330-335), 2-byte aligned in the ring as in memory (hardware-
// unaligned ds_read_b128).  FOUR lanes per super-block as in mmv1.hip mv1_q6k: lane (n, hf) owns l in [16hf, 16hf+16) of the 128-half n.
typedef u32x4 __attribute__((aligned(2))) mv2_u32x4_a2;
typedef u32x2 __attribute__((aligned(2))) mv2_u32x2_a2;
struct mv2_q6k_act { u32x4 a[4]; u32x2 bq; float yd; };
// lane constants of the Q6_K body: lane = (blk, n, hf); byte offsets of its pieces relative to its block (the slot is 16-byte aligned, so the
// alignment of every piece is a lane constant)
struct mv2_q6k_lane { int blk, n, hf, qoff, hoff, soff; uint32_t sh, sel; };
static __device__ __forceinline__ mv2_q6k_lane mv2_q6k_lane_of(int lane) {
    mv2_q6k_lane L;
    L.blk = lane >> 2; L.n = (lane >> 1) & 1; L.hf = lane & 1;
    L.sel = (uint32_t) L.hf | ((uint32_t) (L.hf + 2) << 8) | ((uint32_t) (4 + L.hf) << 16) | ((uint32_t) (6 + L.hf) << 24);
    const int qraw = L.blk * 210 + 64 * L.n + 16 * L.hf;
    L.qoff = (qraw & ~3) - L.blk * 210; L.hoff = 128 + 32 * L.n - 64 * L.n; L.soff = 192 + 8 * L.n - 64 * L.n - 16 * L.hf;
    L.sh = (qraw & 2) ? 16u : 0u;
    return L;
}
static __device__ __forceinline__ void mv2_q6k_act_load(const char * im, int nb, int ib, const mv2_q6k_lane & L, mv2_q6k_act & A) {
    const char * la = im + ib * 272 + 128 * L.n + 16 * L.hf;
#pragma unroll
    for (int m = 0; m < 4; ++m) A.a[m] = *(const u32x4 *) (la + 32 * m);
    A.bq = *(const u32x2 *) (im + mv1_img_b16(nb) + ib * 32 + (8 * L.n + 4 * L.hf) * 2);
    A.yd = *(const float *) (im + mv1_img_d(nb) + ib * 4);
}
// the lane's pieces of one super-block out of the ring (p = the lane's block inside the slot).  A misaligned DS read of any width takes the
// lane-serial path (~256 cycles, tools/lds_align.hip): every piece is fetched as aligned dwords (5 for 16 bytes) and funnel-shifted by the lane's
// half-word phase
struct mv2_q6k_regs { u32x4 qla, qlb, qh; u32x2 sc; uint16_t dw; };
static __device__ __forceinline__ void mv2_q6k_read(const char * p, const mv2_q6k_lane & L, mv2_q6k_regs & Rg) {
    const __attribute__((address_space(3))) char * pa = (const __attribute__((address_space(3))) char *) (p + L.qoff);      // (qoff: the lane's first piece rounded down to a dword, relative to its block)
    uint32_t d0[5], d1[5], d2[5], d3[3];
#pragma unroll
    for (int e = 0; e < 5; ++e) { d0[e] = *(const volatile mv2_lds_u32 *) (pa + 4 * e); d1[e] = *(const volatile mv2_lds_u32 *) (pa + 32 + 4 * e); d2[e] = *(const volatile mv2_lds_u32 *) (pa + L.hoff + 4 * e); }
#pragma unroll
    for (int e = 0; e < 3; ++e) d3[e] = *(const volatile mv2_lds_u32 *) (pa + L.soff + 4 * e);
#pragma unroll
    for (int e = 0; e < 4; ++e) { Rg.qla[e] = __builtin_amdgcn_alignbit(d0[e + 1], d0[e], L.sh); Rg.qlb[e] = __builtin_amdgcn_alignbit(d1[e + 1], d1[e], L.sh); Rg.qh[e] = __builtin_amdgcn_alignbit(d2[e + 1], d2[e], L.sh); }
    Rg.sc[0] = __builtin_amdgcn_alignbit(d3[1], d3[0], L.sh); Rg.sc[1] = __builtin_amdgcn_alignbit(d3[2], d3[1], L.sh);
    Rg.dw = *(const uint16_t *) (p + 208);
}
static __device__ __forceinline__ float mv2_q6k_dot(const mv2_q6k_regs & Rg, const mv2_q6k_act & A, uint32_t sel, float acc) {
    const uint32_t scw = __builtin_amdgcn_perm(Rg.sc[1], Rg.sc[0], sel);                  // scales[8n + hf + 2m], m = 0..3
    const int bs[4] = { (int) (int16_t) (A.bq[0] & 0xffff), (int) (int16_t) (A.bq[0] >> 16), (int) (int16_t) (A.bq[1] & 0xffff), (int) (int16_t) (A.bq[1] >> 16) };
    int d[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t Aq = Rg.qla[e], Bq = Rg.qlb[e], Hq = Rg.qh[e];
        d[0] = dot4((Aq & 0x0f0f0f0fu)        | ((Hq << 4) & 0x30303030u), A.a[0][e], d[0]);
        d[1] = dot4((Bq & 0x0f0f0f0fu)        | ((Hq << 2) & 0x30303030u), A.a[1][e], d[1]);
        d[2] = dot4(((Aq >> 4) & 0x0f0f0f0fu) | (Hq & 0x30303030u),        A.a[2][e], d[2]);
        d[3] = dot4(((Bq >> 4) & 0x0f0f0f0fu) | ((Hq >> 2) & 0x30303030u), A.a[3][e], d[3]);
    }
    int isum = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int scm = (int) (int8_t) ((scw >> (8 * m)) & 0xff);
        isum = mad24(scm, mad24(bs[m], -32, d[m]), isum);                           // sum((q - 32) a) = sum(q a) - 32 bsum: the same integers
    }
    return fmaf(h2f(Rg.dw) * A.yd, (float) isum, acc);
}
template <int NIT, int C, int XS = 0>
static __device__ __forceinline__ void mv2_consume_q6k(const char * im, const char * ringp, int K, int c, int ntask, char * dst, int row0, float resid, mv2_flags * F) {
    typedef mv2_geo<3360, 1, NIT, XS> geo;
    constexpr int SLOTB = geo::SLOTB, NS = geo::NS;
    const int lane = threadIdx.x & 63, nb = K >> 8;
    const mv2_q6k_lane L = mv2_q6k_lane_of(lane);
    mv2_q6k_act A[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) mv2_q6k_act_load(im, nb, it * 16 + L.blk, L, A[it]);
    const char * wl = ringp + L.blk * 210;             // (+ slot * 3360)
    uint32_t seen = 0;
    mv2_out o = { 0.0f, 0, 0 };
    int k = 0;
    for (int j = c; j < ntask; j += C, ++k) {
        float acc = 0.0f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int t = __builtin_amdgcn_readfirstlane(j * NIT + it);
            mv2_wait_step(t, seen, F);
            mv2_q6k_regs Rg;
            mv2_q6k_read(wl + (t % NS) * SLOTB, L, Rg);
            if (it == NIT - 1) { MV2_LGKM0(); mv2_poke(MV2_FLAG(F->consumed[c]), (uint32_t) (k + 1)); }
            acc = mv2_q6k_dot(Rg, A[it], L.sel, acc);
        }
        const float s = wave_sum_f32(acc);
        if (lane == o.nres) o.res = s;
        if (++o.nres == 64) mv2_out_flush<C>(o, dst, row0, resid);
    }
    mv2_out_flush<C>(o, dst, row0, resid);
}

// ================================================================================================= Q8_0 consumer
// 34-B blocks {f16 d, int8 qs[32]} (ggml-common.h:219-224), 128 per step.  Lane l owns blocks 2l and 2l + 1 of the step: 68 bytes at 68 l -- dword-aligned, an
// odd dword stride over the lanes (17): seventeen conflict-free ds_read_b32.  Block 2l: d = low half of dword 0, quants = dwords 0..8 shifted by 16 bits;
// block 2l + 1: d = high half of dword 8, quants = dwords 9..16 as they are.  Reference arithmetic: ggml_vec_dot_q8_0_q8_0 (ggml-cpu/quants.c:305-333):
// sumi over the 32 products, sumf += sumi * (d_w * d_a), blocks in ascending order per lane; the lanes fold on the DPP network.
struct mv2_q80_act { u32x4 a[4]; float yd[2]; };
static __device__ __forceinline__ void mv2_q80_act_load(const char * im, int K, int ib0 /* first of the lane's two blocks */, mv2_q80_act & A) {
#pragma unroll
    for (int k = 0; k < 4; ++k) A.a[k] = *(const u32x4 *) (im + ib0 * 32 + 16 * k);
    A.yd[0] = *(const float *) (im + K + ib0 * 4); A.yd[1] = *(const float *) (im + K + ib0 * 4 + 4);
}
static __device__ __forceinline__ float mv2_q80_dot(const uint32_t (&w)[17], const mv2_q80_act & A, float acc) {
    int s0 = 0, s1 = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s0 = dot4(__builtin_amdgcn_alignbit(w[e + 1], w[e], 16), A.a[e >> 2][e & 3], s0);
        s1 = dot4(w[9 + e], A.a[2 + (e >> 2)][e & 3], s1);
    }
    acc = fmaf((float) s0, h2f((uint16_t) (w[0] & 0xffff)) * A.yd[0], acc);
    acc = fmaf((float) s1, h2f((uint16_t) (w[8] >> 16)) * A.yd[1], acc);
    return acc;
}
template <int R, int NIT, int C, bool PAIR, int XS = 0>
static __device__ __forceinline__ void mv2_consume_q80(const char * im, const char * ringp, int K, int c, int ntask, char * dst, int row0, float resid, mv2_flags * F) {
    typedef mv2_geo<4352, R, NIT, XS> geo;
    constexpr int SLOTB = geo::SLOTB, NS = geo::NS;
    const int lane = threadIdx.x & 63;
    mv2_q80_act A[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) mv2_q80_act_load(im, K, it * 128 + 2 * lane, A[it]);
    const __attribute__((address_space(3))) char * wl = (const __attribute__((address_space(3))) char *) (ringp + lane * 68);
    uint32_t seen = 0;
    mv2_out o = { 0.0f, 0, 0 };
    int k = 0;
    for (int j = c; j < ntask; j += C, ++k) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int t = __builtin_amdgcn_readfirstlane(j * NIT + it);
            mv2_wait_step(t, seen, F);
            const __attribute__((address_space(3))) char * p = wl + (t % NS) * SLOTB;
            uint32_t w[R][17];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < 17; ++e) w[r][e] = *(const volatile mv2_lds_u32 *) (p + r * 4352 + 4 * e);
            if (it == NIT - 1) { MV2_LGKM0(); mv2_poke(MV2_FLAG(F->consumed[c]), (uint32_t) (k + 1)); }
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = mv2_q80_dot(w[r], A[it], acc[r]);
        }
        if (PAIR) {
            const float gsum = wave_sum_f32(acc[0]), usum = wave_sum_f32(acc[R - 1]);
            if (lane == o.nres) o.res = mv1_silu(gsum) * usum;
        } else {
            const float s = wave_sum_f32(acc[0]);
            if (lane == o.nres) o.res = s;
        }
        if (++o.nres == 64) mv2_out_flush<C>(o, dst, row0, resid);
    }
    mv2_out_flush<C>(o, dst, row0, resid);
}

} // namespace mi

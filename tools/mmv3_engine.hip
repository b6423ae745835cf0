// tools/mmv3_engine.hip (LAB, not in the product build: measured slower than the launch form, see profiles/r04_engine_lab.txt and DESIGN.md "Round 4") -- several DEPENDENT batch-1 decode mat-vec launches (mmv2.hip) as the stages of ONE persistent launch: the weight stream does not stop
// at the stage boundaries.  One workgroup of 16 waves per CU as in mmv2.hip; the loader wave walks the stages' matrices back to back and keeps
// this CU's share of the NEXT stage's weights landing in the LDS ring while the consumers are still finishing the current stage, handing its
// output vector over to every CU, and building the next Q8_K image.  The weights never depend on the activations, so the 1.3 - 1.9 us head and
// the boundary of every launch but the first hide under the stream (MI355X guide, "prefetch-credit").
//
// A decode layer of Qwen3 as the graph planner submits it (llm_build_qwen3, src/llama-model.cpp:9287-9406):
//     attention | wo + resid -> [ffn_norm] gate / up + SwiGLU -> down + resid -> [attn_norm of the next layer] wq / wk / wv | attention | ...
// becomes one k_fattn_one launch + ONE k_mv3 launch of four stages (the last layer's chain ends with [output_norm] lm-head).
//
// What is computed per stage is exactly what k_mv2 computes for the same arguments (the same functions from mv2_dev.hpp, the same summation
// orders): results are bit-identical to the launch form (tests/test_round4_gpu.py).  Reference arithmetic: ggml_compute_forward_mul_mat with
// ne11 == 1 (ggml-cpu/ggml-cpu.c:1210-1402), quantize_row_q8_K_ref (ggml-quants.c:2555-2592), ggml_vec_dot_q4_K_q8_K / _q6_K_q8_K
// (ggml-cpu/quants.c:550-623 / 705-758), RMS norm (ggml-cpu/ops.cpp:3517-3566), SwiGLU (ops.cpp:2934-2990); what the reference's GPU backend
// launches for the same nodes: ggml-cuda/mmvq.cu:142, vecdotq.cuh:461,580.
//
// Hand-off between stages (MI355X guide, Guideline 16, form R1 -- the payload is up to 48 KB per edge, so flags + one payload pass, not tagged
// granules whose every polling pass would move 2 x the payload through the L2s of all 256 CUs):
//   producer CU : its 15 consumers leave their rows in an LDS stash; the LAST of them to arrive writes the CU's rows to the real destination
//                 tensor (plain stores) and to the hand-off buffer (write-through, sc1), drains (s_waitcnt vmcnt(0)) and stores ONE flag word
//                 flag[cu] = epoch (relaxed, agent scope).
//   consumer CU : ONE wave polls the flag words of all CUs (one 1 KiB sc1 load covers 256 of them), then the gather waves read their 256-blocks
//                 of the vector with sc1 loads straight into the registers the row-parallel Q8_K quantiser wants (no LDS staging of the f32 row).
//   epochs      : epoch = base + stage + 1 with `base` a device word the launch itself advances at its end, so a replayed hipGraph (frozen kernel
//                 arguments) sees fresh epochs; two payload buffers alternate (a CU can only produce stage s + 2 after every CU gathered stage s).
// Every cross-CU wait is bounded: on a time-out the workgroup raises an error word, aborts all its waits and the launch ends with garbage --
// never with a hung GPU.  All workgroups must be resident at once: grid = CU count, one workgroup per CU by its LDS size (the host refuses
// anything else; two such launches from different streams / processes on ONE device could starve each other: the backend runs the engine only
// from the context that claimed the device, see graph.cpp).
//
// Ring: the CU's LDS ring is 120 quanta of 1152 B; a step (16 super-blocks of one row) takes 2 quanta (Q4_K), 4 (gate + up pair) or 3 (Q6_K:
// 3360 of 3456 B).  Positions are virtual and monotonic over the whole launch; a stage starts at the next multiple of its step size, and since
// 120 is a multiple of 2, 3 and 4 no step ever straddles the end of the ring.  The loader may write a step once every step that started more than
// a ring behind has been consumed; the consumers publish one word each -- the global index of their next unconsumed task -- and the first
// unconsumed step of the workgroup follows from the minimum of the fifteen.
#include "../llama.cpp-omni_amd/csrc/kernels.hpp"
#include "../llama.cpp-omni_amd/csrc/kernels/mv_dev.hpp"
#include "../llama.cpp-omni_amd/csrc/kernels/mv2_dev.hpp"

namespace mi {

#define MV3_MAX_STAGES 8
#define MV3_NREP 8
constexpr int MV3_QB   = 1152;                       // ring quantum (bytes)
constexpr int MV3_NQ   = 120;                        // quanta in the ring: 138 240 B
constexpr int MV3_C    = MV2_WAVES - 1;              // consumers per workgroup
constexpr int MV3_IMG  = 48 * 324 + 16 + 32;         // Q8_K image of a 12288-row (mv1_image_bytes) rounded up to 64 B
constexpr int MV3_RSTG = 512;                        // floats of external residual rows staged per workgroup (all stages together)
constexpr int MV3_STASH = 256;                       // rows a workgroup may produce in a stage whose output is handed over
constexpr int MV3_LDS  = MV3_IMG + MV3_RSTG * 4 + MV3_STASH * 4 + MV3_MAX_STAGES * 16 + MV3_NQ * MV3_QB;
enum { MV3_PUBLISH = 1, MV3_SAVE_KEEP = 2, MV3_USE_KEEP = 4 };

struct mv3_stage {
    mv2_mat m[3];                                    // W, dst, resid (external rows, or null), w_rs, nrows, type, wg0, q, r
    const char * W1;                                 // pair: the up matrix (m[0] = gate)
    const float * x;                                 // stage 0 only: the f32 activation row (later stages read the previous stage's output)
    const float * nw;                                // RMS-norm weights or null
    float eps;
    int nmat, K, pair;
    int flags;                                       // MV3_PUBLISH: a following stage gathers this one's output; MV3_SAVE_KEEP / MV3_USE_KEEP: residual = an earlier stage's own rows
    int rstg_off;                                    // floats: this stage's external residual rows in the staging area (-1: none)
    int pad_;
};
struct mv3_dev {
    int nstage, grid;
    uint32_t * epoch;                                // [0]: epoch base, advanced by the launch
    uint32_t * flag;                                 // [MV3_NREP][grid rounded up to 256]: every producer writes all copies, workgroup wg polls copy wg % MV3_NREP (256 pollers on the
                                                     // same eight 128-byte lines serialise on one memory channel: flag -> seen took 4.3 us)
    float * hbuf;                                    // [2][12288] hand-off payload
    uint32_t * err;                                  // [0] != 0: a bounded wait gave up (code << 16 | workgroup)
    uint32_t * trace;                                // measurement builds: [grid][16][64] stamps
    mv3_stage st[MV3_MAX_STAGES];
};

struct mv3_flags {                                   // LDS; zeroed before the launch's one s_barrier; everything monotonic over the launch
    uint32_t landed;                                 // steps (global index over the stages) that have landed in the ring
    uint32_t rows_issued;                            // stage-0 gather waves that have requested their part of the activation row
    uint32_t sum_cnt;                                // arrivals of partial sums of squares (4 per normed stage)
    uint32_t scale_epoch; float scale;               // stage + 1 whose RMS-norm scale is in `scale`
    uint32_t img_cnt;                                // gather waves that have finished their image blocks (cumulative)
    uint32_t go_epoch;                               // stage + 1 whose input vector is completely published (set by gather wave 0)
    uint32_t done_cnt;                               // consumers whose stage results are in the stash (cumulative: 15 per stage)
    uint32_t abort;                                  // a bounded wait gave up: every wait returns at once
    uint32_t rst_ready;                              // the external residual rows of all stages are in the staging area
    uint32_t pad_[2];
    uint32_t consumed[16];                           // per consumer: global index of its next unconsumed task
};

#ifndef MV3_SPIN_MAX
#define MV3_SPIN_MAX (1u << 22)                      // polls (each >= 64 clocks of s_sleep + a memory or LDS round trip): seconds
#endif
#ifdef MV3_TRACE
#define MV3_STAMP(slot) do { trv = mv3_writelane((uint32_t) __builtin_amdgcn_s_memrealtime(), (uint32_t) ((slot) & 63), trv); } while (0)
#else
#define MV3_STAMP(slot) do { } while (0)
#endif

// v[lane_sel] = val (no clang builtin for v_writelane_b32 in this toolchain); s_nop: a VALU-written SGPR used as lane select needs 4 wait states
static __device__ __forceinline__ uint32_t mv3_writelane(uint32_t val, uint32_t lane_sel, uint32_t v) {
    const uint32_t sv = __builtin_amdgcn_readfirstlane(val), sl = __builtin_amdgcn_readfirstlane(lane_sel);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(sv), "s"(sl) : "m0");     // (two SGPR operands would exceed the constant-bus limit: the lane select goes through m0)
    return v;
}
static __device__ __forceinline__ bool mv3_aborted(mv3_flags * F) { return mv2_peek(MV2_FLAG(F->abort)) != 0u; }
static __device__ __forceinline__ void mv3_give_up(mv3_flags * F, uint32_t * err, uint32_t code) {
    mv2_poke(MV2_FLAG(F->abort), 1u);
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, (code << 16) | (blockIdx.x + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until *p >= n (LDS, monotonic).  false: aborted
template <int SLEEP = 1>
static __device__ __forceinline__ bool mv3_await(const mv2_lds_u32 * p, uint32_t n, mv3_flags * F, uint32_t * err, uint32_t code) {
    uint32_t spins = 0;
    while ((int32_t) (mv2_peek(p) - n) < 0) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if ((++spins & 63u) == 0u) { if (mv3_aborted(F)) return false; if (spins > MV3_SPIN_MAX) { mv3_give_up(F, err, code); return false; } }
    }
    asm volatile("" ::: "memory");
    return true;
}

// what workgroup wg does in a stage
struct mv3_view { int mi; const char * W; const char * W1; char * dst; const char * resid; uint32_t w_rs; int nrows, type, G0, ntask, NIT, R, nq; };
static __device__ __forceinline__ mv3_view mv3_view_of(const mv3_stage & S, int wg) {
    mv3_view V;
    V.mi = (S.nmat > 1 && wg >= S.m[1].wg0) ? ((S.nmat > 2 && wg >= S.m[2].wg0) ? 2 : 1) : 0;
    const mv2_mat & M = S.m[V.mi];
    const int lw = wg - M.wg0;
    V.W = M.W; V.W1 = S.W1; V.dst = M.dst; V.resid = M.resid; V.w_rs = M.w_rs; V.nrows = M.nrows; V.type = M.type;
    V.G0 = lw * M.q + (lw < M.r ? lw : M.r);
    V.ntask = M.q + (lw < M.r ? 1 : 0);
    V.NIT = S.K >> 12; V.R = S.pair ? 2 : 1;
    V.nq = M.type == GGML_TYPE_Q4_K ? 2 * V.R : 3;
    V.G0 = __builtin_amdgcn_readfirstlane(V.G0); V.ntask = __builtin_amdgcn_readfirstlane(V.ntask); V.nq = __builtin_amdgcn_readfirstlane(V.nq);
    return V;
}
static __device__ __forceinline__ int mv3_align_up(int v, int nq) { const int m = nq == 3 ? v % 3 : (v & (nq - 1)); return m ? v + nq - m : v; }
static __device__ __forceinline__ int mv3_mod_nq(int v) { return v % MV3_NQ; }

// ================================================================================================= loader
struct mv3_lstate {
    int v, pos;                                      // virtual position (quanta) of the next step and v % NQ
    int safe_v;                                      // every step that started below this position has been consumed
    int ts;                                          // tail stage: the stage of the first unconsumed step (as of the last refresh)
    uint32_t t_g0, t_v0, t_nitq, t_nt;               // its st_tab entry (read back from LDS only when the tail moves to the next stage)
    uint32_t gtask, gstep;                           // global task / step index of the current stage's first
    uint32_t I_total, S_total;                       // VMEM instructions / steps issued so far
    uint32_t rounds, pub;                            // rounds issued; rounds known landed
    uint32_t recI, recS;                             // lane r & 63: I_total / S_total after round r   (VGPR lane arrays)
};
// st_tab[s] = { gtask0, v0, NIT * nq, ntask } of the stages the loader has entered (LDS; written and read by the loader only)
static __device__ __forceinline__ void mv3_refresh_safe(mv3_lstate & L, int s, const uint32_t * st_tab, mv3_flags * F) {
    const int lane = threadIdx.x & 63;
    uint32_t k = lane < MV3_C ? *(const volatile mv2_lds_u32 *) MV2_FLAG(F->consumed[lane & 15]) : 0xffffffffu;
    { const uint32_t o = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) k, 0xB1, 0xf, 0xf, false);  k = __builtin_elementwise_min(k, o); }
    { const uint32_t o = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) k, 0x4E, 0xf, 0xf, false);  k = __builtin_elementwise_min(k, o); }
    { const uint32_t o = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) k, 0x141, 0xf, 0xf, false); k = __builtin_elementwise_min(k, o); }
    { const uint32_t o = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) k, 0x140, 0xf, 0xf, false); k = __builtin_elementwise_min(k, o); }
    const uint32_t free_gtask = __builtin_amdgcn_readfirstlane(k);                 // the first unconsumed task of the workgroup (global index)
    while (L.ts < s && free_gtask >= L.t_g0 + L.t_nt) {
        ++L.ts;
        const volatile mv2_lds_u32 * t = (const volatile mv2_lds_u32 *) (const mv2_lds_u32 *) (st_tab + 4 * L.ts);
        L.t_g0 = __builtin_amdgcn_readfirstlane(t[0]); L.t_v0 = __builtin_amdgcn_readfirstlane(t[1]); L.t_nitq = __builtin_amdgcn_readfirstlane(t[2]); L.t_nt = __builtin_amdgcn_readfirstlane(t[3]);
    }
    const uint32_t done = free_gtask - L.t_g0 < L.t_nt ? free_gtask - L.t_g0 : L.t_nt;
    L.safe_v = (int) (L.t_v0 + done * L.t_nitq);
}
template <int PIECE, int R, bool NT>
static __device__ __forceinline__ void mv3_stream(mv3_lstate & L, int s, const mv3_view & V, uint32_t ring, const uint32_t * st_tab, mv3_flags * F, uint32_t * err) {
    constexpr int VM = (PIECE == 2304 ? 3 : 4) * R, B = 12 / VM, NQS = PIECE == 2304 ? 2 * R : 3;
    const int lane = threadIdx.x & 63;
    const uint32_t v16 = 16u * (uint32_t) lane, v4 = 4u * (uint32_t) lane;
    const mv1_rsrc rs0 = mv1_make_rsrc(V.W, (size_t) V.nrows * V.w_rs), rs1 = mv1_make_rsrc(R == 2 ? V.W1 : V.W, (size_t) V.nrows * V.w_rs);
    const int NIT = V.NIT;
    uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t) V.G0 * V.w_rs);
    const uint32_t row_skip = __builtin_amdgcn_readfirstlane(V.w_rs - (uint32_t) (NIT * PIECE));
    const int T = V.ntask * NIT;
    int it = 0, n = 0;
    while (n < T) {
        const int nb_ = T - n < B ? T - n : B;
        const int need_v = L.v + nb_ * NQS - MV3_NQ;
        if (need_v > L.safe_v) {
            uint32_t spins = 0;
            for (;;) {
                mv3_refresh_safe(L, s, st_tab, F);
                if (need_v <= L.safe_v) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 63u) == 0u) { if (mv3_aborted(F)) return; if (spins > MV3_SPIN_MAX) { mv3_give_up(F, err, 1u); return; } }
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < B; ++i) if (i < nb_) {
            const uint32_t la = __builtin_amdgcn_readfirstlane(ring + (uint32_t) L.pos * (uint32_t) MV3_QB);
#pragma unroll
            for (int r = 0; r < R; ++r) mv2_dma_piece<PIECE, NT>(r == 1 ? rs1 : rs0, so, la + (uint32_t) (r * PIECE), v16, v4);
            so += (uint32_t) PIECE;
            if (++it == NIT) { it = 0; so += row_skip; }
            L.pos += NQS; if (L.pos == MV3_NQ) L.pos = 0;
            L.v += NQS;
        }
        n += nb_;
        L.I_total += (uint32_t) (nb_ * VM); L.S_total += (uint32_t) nb_;
        L.recI = mv3_writelane(L.I_total, L.rounds & 63u, L.recI);
        L.recS = mv3_writelane(L.S_total, L.rounds & 63u, L.recS);
        ++L.rounds;
        if (L.I_total > 48u) {                       // at most 48 instructions stay outstanding: everything up to I_total - 48 has landed
            mv2_vmcnt<48>();
            bool moved = false;
            while (L.pub < L.rounds && (uint32_t) __builtin_amdgcn_readlane((int) L.recI, (int) (L.pub & 63u)) <= L.I_total - 48u) { ++L.pub; moved = true; }
            if (moved) mv2_poke(MV2_FLAG(F->landed), (uint32_t) __builtin_amdgcn_readlane((int) L.recS, (int) ((L.pub - 1u) & 63u)));
        }
    }
}
template <int W> static __device__ __forceinline__ void mv3_drain_to(mv3_lstate & L, mv3_flags * F) {
    if (L.I_total > (uint32_t) W) {
        mv2_vmcnt<W>();
        bool moved = false;
        while (L.pub < L.rounds && (uint32_t) __builtin_amdgcn_readlane((int) L.recI, (int) (L.pub & 63u)) <= L.I_total - (uint32_t) W) { ++L.pub; moved = true; }
        if (moved) mv2_poke(MV2_FLAG(F->landed), (uint32_t) __builtin_amdgcn_readlane((int) L.recS, (int) ((L.pub - 1u) & 63u)));
    }
}

// ================================================================================================= gather waves: the stage's Q8_K image
// Gather wave gw owns image blocks 4 gw .. 4 gw + 3; its DPP row `row` holds block b = 4 gw + row, lane i of the row elements 64 m + 4 i .. + 3,
// m = 0..3 (the layout of mv2_q8k_rows).  The summation order of the RMS norm is mv2_prologue's: per lane m, e ascending in double, wave_sum_f64,
// then the four waves' partials in wave order.
static __device__ __forceinline__ bool mv3_poll_flags(const uint32_t * flag, int grid, uint32_t want, mv3_flags * F, uint32_t * err) {
    const int lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void *) flag, (short) 0, grid * 4, 0x00020000);
    uint32_t spins = 0;
    for (;;) {
        bool ok = true;
        for (int b = 0; b * 256 < grid; ++b) {
            const u32x4 f = __builtin_amdgcn_raw_buffer_load_b128(fr, (uint32_t) (b * 1024 + 16 * lane), 0, 16);     // sc1: served at the coherence point
#pragma unroll
            for (int e = 0; e < 4; ++e) ok = ok && (b * 256 + 4 * lane + e >= grid || (int32_t) (f[e] - want) >= 0);
        }
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 15u) == 0u) { if (mv3_aborted(F)) return false; if (spins > (MV3_SPIN_MAX >> 2)) { mv3_give_up(F, err, 2u); return false; } }
    }
    asm volatile("" ::: "memory");
    return true;
}
static __device__ __forceinline__ bool mv3_gather(const mv3_dev & d, const mv3_stage & S, int s, int gw, uint32_t ebase, char * im, double * red, mv3_flags * F, uint32_t & trv) {
    int lane = threadIdx.x & 63; asm volatile("" : "+v"(lane));
    const int row = lane >> 4, i = lane & 15, K = S.K, nb = K >> 8, b = 4 * gw + row;
    const uint32_t off = (uint32_t) (256 * b + 4 * i) * 4u;
    f32x4 x[4], w[4];
    const float * nw = S.nw;
    if (nw) {
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *) nw, (short) 0, K * 4, 0x00020000);
#pragma unroll
        for (int m = 0; m < 4; ++m) { const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(wr, off + 256u * m, 0, 0); w[m] = f32x4{ __uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]) }; }
    }
    if (s == 0) {
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *) S.x, (short) 0, K * 4, 0x00020000);
#pragma unroll
        for (int m = 0; m < 4; ++m) { const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(xr, off + 256u * m, 0, 0); x[m] = f32x4{ __uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]) }; }
        asm volatile("" ::: "memory");
        mv2_arrive(MV2_FLAG(F->rows_issued));
    } else {
        if (gw == 0) {
            if (!mv3_poll_flags(d.flag + (size_t) (blockIdx.x % MV3_NREP) * ((d.grid + 255) & ~255), d.grid, ebase + (uint32_t) s, F, d.err)) return false;
            mv2_poke(MV2_FLAG(F->go_epoch), (uint32_t) (s + 1));
        } else if (!mv3_await<2>(MV2_FLAG(F->go_epoch), (uint32_t) (s + 1), F, d.err, 3u)) return false;
        MV3_STAMP(8 * s + 1);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *) (d.hbuf + (size_t) ((s - 1) & 1) * 12288), (short) 0, K * 4, 0x00020000);
#pragma unroll
        for (int m = 0; m < 4; ++m) { const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(xr, off + 256u * m, 0, 16); x[m] = f32x4{ __uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]) }; }
    }
    float scale = 1.0f;
    if (nw) {
        double ss = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) ss += (double) (x[m][e] * x[m][e]);
        MV3_STAMP(8 * s + 2);
        ss = wave_sum_f64(ss);
        if (lane == 0) red[gw] = ss;
        asm volatile("" ::: "memory");
        uint32_t prev = 0;
        if (lane == 0) prev = __hip_atomic_fetch_add(MV2_FLAG(F->sum_cnt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        prev = __builtin_amdgcn_readfirstlane(prev);
        asm volatile("" ::: "memory");
        if ((prev & 3u) == 3u) {                     // (K = 4096: four gather waves) the last to arrive: every partial sum is in LDS
            double tot = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) tot += *(const volatile double *) &red[q];
            const float mean = (float) (tot * (1.0 / 4096.0));
            const float sc = 1.0f / sqrtf(mean + S.eps);
            if (lane == 0) *(volatile __attribute__((address_space(3))) float *) &F->scale = sc;
            mv2_poke(MV2_FLAG(F->scale_epoch), (uint32_t) (s + 1));
            scale = sc;
        } else {
            if (!mv3_await(MV2_FLAG(F->scale_epoch), (uint32_t) (s + 1), F, d.err, 4u)) return false;
            scale = *(const volatile __attribute__((address_space(3))) float *) &F->scale;
        }
    } else MV3_STAMP(8 * s + 2);
    MV3_STAMP(8 * s + 3);
    f32x4 y[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) y[m][e] = nw ? (x[m][e] * scale) * w[m][e] : x[m][e];
    mv2_q8k_rows(y, lane, b, nb, im);
    mv2_arrive(MV2_FLAG(F->img_cnt));
    return true;
}

// ================================================================================================= consumers
struct mv3_cstate { uint32_t gtask0, gstep0, gtask_next; int v0, nq; };
static __device__ __forceinline__ bool mv3_wait_step(uint32_t g, uint32_t & seen, mv3_flags * F, uint32_t * err) {
    uint32_t spins = 0;
    while ((int32_t) (seen - g) <= 0) {
        seen = mv2_peek(MV2_FLAG(F->landed));
        if ((int32_t) (seen - g) <= 0) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0u) { if (mv3_aborted(F)) return false; if (spins > MV3_SPIN_MAX) { mv3_give_up(F, err, 5u); return false; } }
        }
    }
    asm volatile("" ::: "memory");
    return true;
}
// the mid-stage flush of a long final stage (the lm-head: hundreds of rows per workgroup): 64 results straight to the destination
static __device__ __forceinline__ void mv3_flush_direct(mv2_out & o, char * dst, int row0 /* G0 + c */, const float * rst /* LDS residual rows of the workgroup or null */, int c) {
    const int lane = threadIdx.x & 63;
    if (lane < o.nres) {
        const int rl = c + (o.k0 + lane) * MV3_C;
        *(float *) (dst + (size_t) (row0 - c + rl) * 4) = o.res + (rst ? rst[rl] : 0.0f);
    }
    o.k0 += o.nres; o.nres = 0;
}
template <int R, int NIT, bool PAIR>
static __device__ __forceinline__ bool mv3_consume_q4k(const char * im, const char * ring, int K, int c, const mv3_view & V, const mv3_cstate & Cs, bool direct, const float * rst, mv3_flags * F, uint32_t * err, mv2_out & o) {
    int lane = threadIdx.x & 63; asm volatile("" : "+v"(lane));       // (opaque: the lane constants of each body are rebuilt per stage instead of living across the whole stage loop)
    const int nb = K >> 8;
    int blk, q; mv2_lane_map(lane, blk, q);
    const uint32_t sel = mv2_q4k_sel(q);
    mv2_q4k_act A[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) mv2_q4k_act_load(im, nb, it * 16 + blk, q, A[it]);
    const char * wl = ring + blk * 144;
    uint32_t seen = 0;
    const int ntask = V.ntask;
    int pos = mv3_mod_nq(Cs.v0 + c * NIT * Cs.nq);                            // ring position (quanta) of this consumer's next task
    const int stride = mv3_mod_nq(MV3_C * NIT * Cs.nq);
    for (int j = c; j < ntask; j += MV3_C) {
        float acc[R], accm[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { acc[r] = 0.0f; accm[r] = 0.0f; }
        int p_ = pos;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const uint32_t g = __builtin_amdgcn_readfirstlane(Cs.gstep0 + (uint32_t) (j * NIT + it));
            if (!mv3_wait_step(g, seen, F, err)) return false;
            const char * p = wl + p_ * MV3_QB;
            u32x4 H[R], Q[R], P[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { H[r] = *(const u32x4 *) (p + r * 2304); Q[r] = *(const u32x4 *) (p + r * 2304 + 16 + 32 * q); P[r] = *(const u32x4 *) (p + r * 2304 + 32 + 32 * q); }
            if (it == NIT - 1) { MV2_LGKM0(); mv2_poke(MV2_FLAG(F->consumed[c]), j + MV3_C < ntask ? Cs.gtask0 + (uint32_t) (j + MV3_C) : Cs.gtask_next + (uint32_t) c); }
#pragma unroll
            for (int r = 0; r < R; ++r) mv2_q4k_dot(H[r], Q[r], P[r], A[it], sel, acc[r], accm[r]);
            p_ += Cs.nq; if (p_ >= MV3_NQ) p_ -= MV3_NQ;
        }
        pos += stride; if (pos >= MV3_NQ) pos -= MV3_NQ;
        if (PAIR) {
            const float gsum = wave_sum_f32(acc[0] - accm[0]), usum = wave_sum_f32(acc[R - 1] - accm[R - 1]);
            if (lane == o.nres) o.res = mv1_silu(gsum) * usum;
        } else {
            const float s = wave_sum_f32(acc[0] - accm[0]);
            if (lane == o.nres) o.res = s;
        }
        if (++o.nres == 64 && direct) mv3_flush_direct(o, V.dst, V.G0 + c, rst, c);
    }
    return true;
}
template <int NIT>
static __device__ __forceinline__ bool mv3_consume_q6k(const char * im, const char * ring, int K, int c, const mv3_view & V, const mv3_cstate & Cs, bool direct, const float * rst, mv3_flags * F, uint32_t * err, mv2_out & o) {
    int lane = threadIdx.x & 63; asm volatile("" : "+v"(lane));
    const int nb = K >> 8;
    const mv2_q6k_lane L = mv2_q6k_lane_of(lane);
    mv2_q6k_act A[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) mv2_q6k_act_load(im, nb, it * 16 + L.blk, L, A[it]);
    const char * wl = ring + L.blk * 210;
    uint32_t seen = 0;
    const int ntask = V.ntask;
    int pos = mv3_mod_nq(Cs.v0 + c * NIT * 3);
    const int stride = mv3_mod_nq(MV3_C * NIT * 3);
    for (int j = c; j < ntask; j += MV3_C) {
        float acc = 0.0f;
        int p_ = pos;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const uint32_t g = __builtin_amdgcn_readfirstlane(Cs.gstep0 + (uint32_t) (j * NIT + it));
            if (!mv3_wait_step(g, seen, F, err)) return false;
            mv2_q6k_regs Rg;
            mv2_q6k_read(wl + p_ * MV3_QB, L, Rg);
            if (it == NIT - 1) { MV2_LGKM0(); mv2_poke(MV2_FLAG(F->consumed[c]), j + MV3_C < ntask ? Cs.gtask0 + (uint32_t) (j + MV3_C) : Cs.gtask_next + (uint32_t) c); }
            acc = mv2_q6k_dot(Rg, A[it], L.sel, acc);
            p_ += 3; if (p_ >= MV3_NQ) p_ -= MV3_NQ;
        }
        pos += stride; if (pos >= MV3_NQ) pos -= MV3_NQ;
        const float s = wave_sum_f32(acc);
        if (lane == o.nres) o.res = s;
        if (++o.nres == 64 && direct) mv3_flush_direct(o, V.dst, V.G0 + c, rst, c);
    }
    return true;
}

// ================================================================================================= kernel
template <bool NT>
__global__ void __launch_bounds__(64 * MV2_WAVES) k_mv3(const mv3_dev d) {
    __shared__ mv3_flags F;
    __shared__ double red[16];
    if (threadIdx.x < sizeof(mv3_flags) / 4) ((uint32_t *) &F)[threadIdx.x] = threadIdx.x >= offsetof(mv3_flags, consumed) / 4 ? threadIdx.x - (uint32_t) (offsetof(mv3_flags, consumed) / 4) : 0u;   // consumed[c] = c: consumer c's first task
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int wiw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = blockIdx.x, lane = threadIdx.x & 63;
    uint32_t trv = 0; (void) trv;
    MV3_STAMP(63);
    if (wiw == 0) __builtin_amdgcn_s_setprio(3);
    // the epoch base is read by every consumer wave BEFORE its first arrival anywhere (pinned by the empty asm): the launch overwrites the word at its end
    uint32_t ebase_v = 0;
    if (wiw != 0) ebase_v = __hip_atomic_load(d.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ns = d.nstage;
    char * im = mv1_lds;
    float * rstg = (float *) (mv1_lds + MV3_IMG);
    float * stash = rstg + MV3_RSTG;
    uint32_t * st_tab = (uint32_t *) (stash + MV3_STASH);
    char * ringp = (char *) (st_tab + 4 * MV3_MAX_STAGES);

    if (wiw == 0) {
        // ------------------------------------------------------------------------------------------ the loader
        mv3_lstate L = { 0, 0, 0, 0, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
        const uint32_t ring = mv2_lds_addr(ringp);
        for (int s = 0; s < ns; ++s) {
            const mv3_stage & S = d.st[s];
            const mv3_view V = mv3_view_of(S, wg);
            const int nv = mv3_align_up(L.v, V.nq);
            L.pos += nv - L.v; if (L.pos >= MV3_NQ) L.pos -= MV3_NQ;
            L.v = nv;
            if (lane == 0) {
                volatile mv2_lds_u32 * t = (volatile mv2_lds_u32 *) (mv2_lds_u32 *) (st_tab + 4 * s);
                t[0] = L.gtask; t[1] = (uint32_t) L.v; t[2] = (uint32_t) (V.NIT * V.nq); t[3] = (uint32_t) V.ntask;
            }
            if (s == 0) { L.t_g0 = 0u; L.t_v0 = (uint32_t) L.v; L.t_nitq = (uint32_t) (V.NIT * V.nq); L.t_nt = (uint32_t) V.ntask; }
            asm volatile("" ::: "memory");
            if (s == 0) { if (!mv3_await(MV2_FLAG(F.rows_issued), (uint32_t) (S.K >> 10), &F, d.err, 6u)) break; }
            MV3_STAMP(8 * s + 0);
            if (V.type == GGML_TYPE_Q4_K) { if (S.pair) mv3_stream<2304, 2, NT>(L, s, V, ring, st_tab, &F, d.err); else mv3_stream<2304, 1, NT>(L, s, V, ring, st_tab, &F, d.err); }
            else mv3_stream<3360, 1, NT>(L, s, V, ring, st_tab, &F, d.err);
            MV3_STAMP(8 * s + 1);
            L.gtask += (uint32_t) V.ntask; L.gstep += (uint32_t) (V.ntask * V.NIT);
            if (mv3_aborted(&F)) break;
        }
        mv3_drain_to<36>(L, &F); mv3_drain_to<24>(L, &F); mv3_drain_to<12>(L, &F);
        mv2_vmcnt<0>();                              // no LDS-DMA may land after the workgroup's LDS is released
        mv2_poke(MV2_FLAG(F.landed), L.S_total);
        MV3_STAMP(62);
    } else {
        // ------------------------------------------------------------------------------------------ consumers (the first K / 1024 of them gather first)
        const int c = wiw - 1;
        // external residual rows of every stage -> LDS, by consumer 14 (not a gather wave), before the weight stream fills the memory queue
        if (c == MV3_C - 1) {
            for (int s = 0; s < ns; ++s) {
                const mv3_stage & S = d.st[s];
                if (S.rstg_off < 0) continue;
                const mv3_view V = mv3_view_of(S, wg);
                if (!V.resid) continue;
                for (int k = lane; k < V.ntask; k += 64) rstg[S.rstg_off + k] = *(const float *) (V.resid + (size_t) (V.G0 + k) * 4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mv2_poke(MV2_FLAG(F.rst_ready), 1u);
        }
        mv3_cstate Cs = { 0u, 0u, 0u, 0, 0 };
        int v_run = 0;
        uint32_t img_need = 0, done_need = 0;
        float keep = 0.0f;
        bool ok = true;
        for (int s = 0; s < ns && ok; ++s) {
            const mv3_stage & S = d.st[s];
            const mv3_view V = mv3_view_of(S, wg);
            Cs.v0 = mv3_align_up(v_run, V.nq); Cs.nq = V.nq;
            Cs.gtask_next = Cs.gtask0 + (uint32_t) V.ntask;
            const int NG = S.K >> 10;
            MV3_STAMP(8 * s + 0);
            if (c < NG) ok = mv3_gather(d, S, s, c, s > 0 ? __builtin_amdgcn_readfirstlane(ebase_v) : 0u, im, red, &F, trv);
            img_need += (uint32_t) NG;
            if (ok) ok = mv3_await<2>(MV2_FLAG(F.img_cnt), img_need, &F, d.err, 7u);
            MV3_STAMP(8 * s + 4);
            if (!ok) break;
            const bool publish = (S.flags & MV3_PUBLISH) != 0, direct = !publish;
            const float * rst = (V.resid && S.rstg_off >= 0) ? rstg + S.rstg_off : nullptr;
            if (rst && !mv3_await(MV2_FLAG(F.rst_ready), 1u, &F, d.err, 8u)) { ok = false; break; }
            mv2_out o = { 0.0f, 0, 0 };
            {
                if (V.type == GGML_TYPE_Q4_K) {
                    if (S.pair)            ok = mv3_consume_q4k<2, 1, true>(im, ringp, S.K, c, V, Cs, direct, rst, &F, d.err, o);
                    else if (V.NIT == 1)   ok = mv3_consume_q4k<1, 1, false>(im, ringp, S.K, c, V, Cs, direct, rst, &F, d.err, o);
                    else                   ok = mv3_consume_q4k<1, 3, false>(im, ringp, S.K, c, V, Cs, direct, rst, &F, d.err, o);
                } else {
                    if (V.NIT == 1)        ok = mv3_consume_q6k<1>(im, ringp, S.K, c, V, Cs, direct, rst, &F, d.err, o);
                    else                   ok = mv3_consume_q6k<3>(im, ringp, S.K, c, V, Cs, direct, rst, &F, d.err, o);
                }
            }
            if (V.ntask <= c) mv2_poke(MV2_FLAG(F.consumed[c]), Cs.gtask_next + (uint32_t) c);     // no task in this stage: the next one is in the next stage
            MV3_STAMP(8 * s + 5);
            if (!ok) break;
            if (direct) {
                if (S.flags & MV3_USE_KEEP) { if (lane < o.nres) o.res += keep; }
                mv3_flush_direct(o, V.dst, V.G0 + c, (S.flags & MV3_USE_KEEP) ? nullptr : rst, c);
            } else {
                // this wave's rows (+ residual) -> stash; the last consumer of the workgroup to arrive publishes the workgroup's rows
                if (lane < o.nres) {
                    const int rl = c + lane * MV3_C;
                    float v = o.res;
                    if (S.flags & MV3_USE_KEEP) v += keep; else if (rst) v += rst[rl];
                    if (S.flags & MV3_SAVE_KEEP) keep = v;
                    stash[rl] = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(ebase_v) : "memory");       // (the epoch word has been READ before this wave's first arrival)
                const uint32_t ebase = __builtin_amdgcn_readfirstlane(ebase_v);
                uint32_t prev = 0;
                if (lane == 0) prev = __hip_atomic_fetch_add(MV2_FLAG(F.done_cnt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                prev = __builtin_amdgcn_readfirstlane(prev);
                asm volatile("" ::: "memory");
                if (prev == done_need + (uint32_t) (MV3_C - 1)) {
                    const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc((void *) (d.hbuf + (size_t) (s & 1) * 12288 + V.G0), (short) 0, V.ntask * 4, 0x00020000);
                    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void *) (V.dst + (size_t) V.G0 * 4), (short) 0, V.ntask * 4, 0x00020000);
                    for (int k = lane; k < V.ntask; k += 64) {
                        const float v = *(const volatile float *) &stash[k];
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), hr, (uint32_t) k * 4u, 0, 16);       // write-through (sc1)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dr, (uint32_t) k * 4u, 0, 0);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane < MV3_NREP) __hip_atomic_store(d.flag + (size_t) lane * ((d.grid + 255) & ~255) + wg, ebase + (uint32_t) s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    MV3_STAMP(8 * s + 6);
                }
                done_need += (uint32_t) MV3_C;
            }
            v_run = Cs.v0 + V.ntask * V.NIT * V.nq;
            Cs.gtask0 = Cs.gtask_next; Cs.gstep0 += (uint32_t) (V.ntask * V.NIT);
        }
        // the launch advances the epoch base for the next one: by now every wave of every workgroup has read it (the last stage was entered only after
        // every workgroup had published the one before, i.e. after all its consumers had arrived at least once)
        if (ok && ns > 1 && wg == 0 && c == 0 && lane == 0) __hip_atomic_store(d.epoch, ebase_v + (uint32_t) ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        MV3_STAMP(62);
    }
#ifdef MV3_TRACE
    if (d.trace) d.trace[((size_t) wg * 16 + wiw) * 64 + lane] = trv;
#endif
}

} // namespace mi

// ------------------------------------------------------------------------------------------------ host side
namespace mi {

int mv2_cus();

// may `a` be a stage of a chain?  (what mmv2 takes, minus ready-made images; an RMS norm only at K = 4096)
bool mmv3_stage_ok(const mv1_args & a) {
    if (!mmv2_ok(a) || a.img) return false;
    if (a.K == 12288 && a.norm_w) return false;
    for (int i = 0; i < a.nmat; ++i) if (a.m[i].resid && a.m[i].nrows > (int64_t) mv2_cus() * MV3_STASH) return false;
    return true;
}
// may stage b follow stage a inside one launch?  b reads a's only output; a's rows per workgroup fit the stash
bool mmv3_link_ok(const mv1_args & a, const mv1_args & b) {
    if (a.nmat != 1 || b.x != a.m[0].dst || b.K != a.m[0].nrows) return false;
    return a.m[0].nrows <= (int64_t) mv2_cus() * MV3_STASH && a.m[0].nrows <= 12288;
}

struct mv3_ctx { uint32_t * epoch = nullptr; uint32_t * flag = nullptr; float * hbuf = nullptr; uint32_t * err = nullptr; uint32_t * trace = nullptr; int grid = 0; };
static mv3_ctx * mv3_ctx_of_device() {
    static mv3_ctx ctxs[64];
    int dev = 0; HIP_CHECK(hipGetDevice(&dev));
    mv3_ctx * c = &ctxs[dev >= 0 && dev < 64 ? dev : 0];
    if (!c->epoch) {
        c->grid = mv2_cus();
        char * p = nullptr;
        const size_t fl = (size_t) MV3_NREP * (((size_t) c->grid + 255) & ~(size_t) 255) * 4;
        const size_t bytes = 256 + fl + 2 * 12288 * 4;
        HIP_CHECK(hipMalloc((void **) &p, bytes));
        HIP_CHECK(hipMemset(p, 0, bytes));
        c->epoch = (uint32_t *) p; c->err = (uint32_t *) (p + 128); c->flag = (uint32_t *) (p + 256);
        c->hbuf = (float *) (p + 256 + fl);
    }
    return c;
}
void mmv3_set_trace(uint32_t * buf) { mv3_ctx_of_device()->trace = buf; }
uint32_t mmv3_error() {                               // after a synchronize: != 0 if a launch gave up a wait (and resets the word)
    mv3_ctx * c = mv3_ctx_of_device();
    uint32_t e = 0; HIP_CHECK(hipMemcpy(&e, c->err, 4, hipMemcpyDeviceToHost));
    if (e) HIP_CHECK(hipMemset(c->err, 0, 4));
    return e;
}

// workgroup ranges of the matrices of a stage: mmv2's plan (by bytes, every matrix at least one workgroup)
static void mv3_plan_stage(const mv1_args & a, int grid, mv3_stage & S) {
    double bytes[3], total = 0;
    for (int i = 0; i < a.nmat; ++i) { bytes[i] = (double) a.m[i].nrows * (double) (a.m[i].type == GGML_TYPE_Q4_K ? 144 : 210) * (double) (a.K / 256); total += bytes[i]; }
    int acc_w = 0; double acc_b = 0;
    for (int i = 0; i < 3; ++i) {
        if (i >= a.nmat) { S.m[i] = S.m[0]; S.m[i].wg0 = grid; continue; }
        acc_b += bytes[i];
        int end = i == a.nmat - 1 ? grid : (int) (grid * (acc_b / total) + 0.5);
        if (end <= acc_w) end = acc_w + 1;
        if (end > grid - (a.nmat - 1 - i)) end = grid - (a.nmat - 1 - i);
        const int nwg = end - acc_w;
        S.m[i] = { (const char *) a.m[i].W, (char *) a.m[i].dst, (const char *) a.m[i].resid, (uint32_t) a.m[i].w_rs, (int) a.m[i].nrows, a.m[i].type, acc_w, (int) (a.m[i].nrows / nwg), (int) (a.m[i].nrows % nwg) };
        acc_w = end;
    }
    S.W1 = (const char *) a.W_up; S.x = a.x; S.nw = a.norm_w; S.eps = a.eps; S.nmat = a.nmat; S.K = (int) a.K; S.pair = a.W_up ? 1 : 0;
    S.flags = 0; S.rstg_off = -1; S.pad_ = 0;
}

// n stages, stage i + 1 reading stage i's output (mmv3_link_ok), as one launch.  Returns false (nothing launched) when the chain does not fit.
bool mmv3(const mv1_args * st, int n, hipStream_t stream) {
    if (n < 1 || n > MV3_MAX_STAGES) return false;
    mv3_ctx * c = mv3_ctx_of_device();
    const int grid = c->grid;
    mv3_dev d;
    d.nstage = n; d.grid = grid; d.epoch = c->epoch; d.flag = c->flag; d.hbuf = c->hbuf; d.err = c->err; d.trace = c->trace;
    int rst = 0, keep_stage = -1;
    for (int i = 0; i < n; ++i) {
        if (!mmv3_stage_ok(st[i])) return false;
        if (i > 0 && !mmv3_link_ok(st[i - 1], st[i])) return false;
        mv3_stage & S = d.st[i];
        mv3_plan_stage(st[i], grid, S);
        if (i > 0) S.x = nullptr;
        if (i + 1 < n) S.flags |= MV3_PUBLISH;
        for (int q = 0; q < st[i].nmat; ++q) {
            if (!st[i].m[q].resid) continue;
            // the residual is an earlier stage's output: that stage's own rows, kept in the consumers' registers -- same row partition required
            int from = -1;
            for (int j = 0; j < i; ++j) if ((const void *) st[j].m[0].dst == (const void *) st[i].m[q].resid) from = j;
            if (from >= 0) {
                if (st[i].nmat != 1 || st[from].nmat != 1 || st[from].m[0].nrows != st[i].m[0].nrows || (keep_stage >= 0 && keep_stage != from) || !(d.st[from].flags & MV3_PUBLISH) ||
                    st[i].m[0].nrows > (int64_t) grid * 64 * MV3_C) return false;
                keep_stage = from; d.st[from].flags |= MV3_SAVE_KEEP; S.flags |= MV3_USE_KEEP; S.m[q].resid = nullptr;
            } else {
                // external rows: staged in LDS at the start of the launch; at most MV3_RSTG floats per workgroup over all stages, one matrix per stage
                if (st[i].nmat != 1) return false;
                const int per_wg = (int) ((st[i].m[0].nrows + grid - 1) / grid);
                if (rst + per_wg > MV3_RSTG) return false;
                S.rstg_off = rst; rst += per_wg;
            }
        }
        if ((S.flags & MV3_PUBLISH) && (st[i].m[0].nrows + grid - 1) / grid > MV3_STASH) return false;
    }
    static bool attr[64] = { false };
    int dev = 0; HIP_CHECK(hipGetDevice(&dev));
    if (!attr[dev & 63]) { HIP_CHECK(hipFuncSetAttribute((const void *) k_mv3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MV3_LDS)); attr[dev & 63] = true; }
    k_mv3<true><<<dim3(grid), dim3(64 * MV2_WAVES), MV3_LDS, stream>>>(d);
    return true;
}

} // namespace mi

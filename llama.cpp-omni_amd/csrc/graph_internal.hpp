// graph_internal.hpp -- what the three translation units of the graph executor share: graph_plan.cpp (supports_op, which kernel family a MUL_MAT takes, scratch
// sizing), graph_exec.cpp + graph_exec_llm.cpp + graph_exec_t2w.cpp (the node executors and their fusion matchers), graph.cpp (fingerprints, capture records, graph_compute / graph_optimize, options).
#pragma once
#include "graph.hpp"
#include <chrono>
#include "ggml_util.hpp"
#include "kernels.hpp"
#include "shadow.hpp"
#include "../../include/ggml-mi355x.h"
#include <unordered_map>
#include <unordered_set>

namespace mi {


// ------------------------------------------------------------------------------------------------ helpers
static inline tdesc td(const ggml_tensor * t) {
    tdesc d; d.p = t->data;
    for (int i = 0; i < 4; ++i) { d.ne[i] = t->ne[i]; d.nb[i] = t->nb[i]; }
    return d;
}
static inline bool is_noop(const ggml_tensor * t) {
    return t->op == GGML_OP_NONE || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE ||
           t->op == GGML_OP_TRANSPOSE || is_empty(t);
}

enum act_kind { ACT_NONE = 0, ACT_Q8K, ACT_Q80, ACT_F16, ACT_F32, ACT_Q8KT, ACT_F16Q };      // F16Q: f16 rows of the Q8_K-quantised values (what the F16-image GEMMs of K-quant weights multiply with)      // Q8KT: the block-major Q8_K image of a whole ubatch (mmq_tile.hip), not a per-row format
// block formats without integer-dot kernels of their own: every MUL_MAT runs on the F16 image of the weights (resident for model
// tensors, shadow.hpp; else de-quantised into scratch per call) with f16-rounded activations -- the arithmetic of the prefill GEMM
static inline bool is_image_quant(int t) {
    return t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_1 || t == GGML_TYPE_Q2_K || t == GGML_TYPE_Q3_K;
}
// Q4_0 / Q5_0: integer mat-vec kernels on Q8_0 activations up to 8 columns (mmvq.hip), the F16 image from 9 columns on
static inline bool is_q40_like(int t) { return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q5_0; }
static inline act_kind act_kind_for(int wtype) {
    if (is_image_quant(wtype)) return ACT_F16;
    switch (wtype) {
        case GGML_TYPE_Q4_K: case GGML_TYPE_Q5_K: case GGML_TYPE_Q6_K: return ACT_Q8K;
        case GGML_TYPE_Q8_0: case GGML_TYPE_Q4_0: case GGML_TYPE_Q5_0: return ACT_Q80;
        case GGML_TYPE_F16:  return ACT_F16;
        case GGML_TYPE_F32:  return ACT_F32;
        default: return ACT_NONE;
    }
}
static inline size_t act_image_bytes(act_kind k, int64_t K) {
    switch (k) {
        case ACT_Q8K: return q8k_image_bytes(K);
        case ACT_Q80: return q80_image_bytes(K);
        case ACT_F16: case ACT_F16Q: return ((size_t) K * 2 + 15) & ~(size_t) 15;
        default: return 0;
    }
}

struct exec_state {
    backend_ctx * c;
    hipStream_t   st;
    ggml_cgraph * g = nullptr;
    long          n_kernels = 0, n_fused = 0;
    std::vector<uint8_t> done;                                           // node already covered by a fused item
    std::unordered_map<const ggml_tensor *, int> index;                  // tensor -> node index
    std::unordered_map<const ggml_tensor *, std::vector<int>> users;     // tensor -> consumer node indices (ascending)
    std::unordered_set<const ggml_tensor *> external;                    // tensors with readers outside this cgraph (see is_out)
    const char * a_range_lo = nullptr; const char * a_range_hi = nullptr;
    // activation cache
    const void *  a_src = nullptr; act_kind a_kind = ACT_NONE; int64_t a_K = 0, a_ne[3] = {0, 0, 0}; size_t a_nb[3] = {0, 0, 0};
    bool          capturing = false;
    int           node_lo = 0, node_hi = -1;      // run_nodes walks [node_lo, node_hi) (-1: to the end) -- the slice timer of graph_compute (MI355X_GRAPH_SLICE)
    // deferred RMS_NORM -> MUL(w): not computed yet; its K-quant mat-vec consumers build the Q8_K image in-kernel (mmvk.hip act_norm)
    struct { const ggml_tensor * m = nullptr; const ggml_tensor * x = nullptr; const ggml_tensor * wt = nullptr; float eps = 0; int left = 0; } pn;
    // deferred q chain + k chain/store + v store of a decode layer: executed by the FLASH_ATTN_EXT node `fa` itself (fattn_pre)
    struct { int fa = -1; fattn_pre pre; int kst = -1, vst = -1;
             bool sm = false; int sm_soft = -1, sm_mm2 = -1, sm_cont = -1; attn_sm_args sma; } pq;   // sm: the flash-attention-off form, `fa` = its first MUL_MAT
    // deferred split-K reduction: `A` (the mat-mul + residual result) still lies as `nsplit` slabs in gemm_partial; the RMS_NORM that
    // reads it next folds the reduction in (gemm_reduce_rms_norm), anything else materialises it first
    struct { const ggml_tensor * A = nullptr; int nsplit = 0; const float * resid = nullptr; size_t resid_cs = 0; const float * resid2 = nullptr; size_t resid2_cs = 0; } pr;   // (resid2: only in front of a LayerNorm)
    // one-token attention left as slices' partial states in fa_scratch (fattn_one.hip k_fattn_gs): `n` = the FLASH_ATTN_EXT node whose f32 rows were NOT written, `consumer` = the
    // one MUL_MAT (wo) that folds them in its prologue (mv1_source); anything else that runs first materialises the rows (gs_materialise in graph_exec.cpp)
    struct { const ggml_tensor * n = nullptr; int consumer = -1; int nh = 0, D = 0; } gs;
    // (pos, rope parameters) whose (cos, sin) table currently sits in rope_scratch (prefill: shared by every layer of the graph)
    struct { const void * pos = nullptr; const void * ff = nullptr; int T = 0, D = 0; rope_params rp; } rt;
    // mask whose tile map currently sits in fa_scratch
    const void *  fa_mask = nullptr; int64_t fa_dims[4] = {0, 0, 0, 0}; size_t fa_mnb1 = 0;
    // a V^T that a fused soft-max attention will read where it LIES (a transposed V cache) instead of from the CONT + CAST copies the graph makes of it: `cast` = the CAST node the
    // attention's second mat-mul names, `v` = the same elements in the cache (try_alias_vt in graph_exec.cpp)
    struct { const ggml_tensor * cast = nullptr; tdesc v; int cont_i = -1, cast_i = -1; } va;
    // an encoder's V^T CAST tensor `t` whose bytes were written as V ROWS instead ([D, n_tokens, H] like K: the GEMM epilogue that makes the f16 copy chooses the layout): its one
    // reader, the flash-attention-off chain exec_attn_sm_prefill fuses, then runs as plain flash attention on the LDS-DMA ring kernel (head size 64); set by exec_gemm_group
    struct { const ggml_tensor * t = nullptr; tdesc v; } vplain;
    // deferred split-K reduction of a GROUPED launch (wq / wk / wv of a prefill ubatch): the results A[0..n) still lie as `nsplit` slabs in gemm_partial (slab = `slab` floats,
    // matrix q a dense [N][M[q]] block at + off[q]); the q / k norm + rope + store launch behind them sums the slabs itself (k_norm_rope_v4), anybody else gets materialise_group
    struct { int n = 0; const ggml_tensor * A[3] = { nullptr, nullptr, nullptr }; size_t off[3] = { 0, 0, 0 }; int64_t M[3] = { 0, 0, 0 }; int nsplit = 0; size_t slab = 0; int64_t N = 0; } prm;
    // CONT nodes that have NOT been run: copies of a view of a tensor from outside the graph (a persistent cache) whose readers may take the view itself (lazy_* in
    // graph_exec.cpp).  `src` = what the copy would read, `deadline` = the first node that writes over those bytes; any other reader materialises the copy first.
    struct lazy_ent { tdesc src; int deadline; };
    std::unordered_map<const ggml_tensor *, lazy_ent> lazy;
    std::unordered_map<const ggml_tensor *, int> lazy_base_deadline;
    // same-type strided copies (CONT / CONCAT / CPY / a lazy copy made real) that have been MET but not launched: mutually independent by byte ranges, they leave as one
    // k_copy_batch launch when a node of another kind is about to run, when a new copy touches bytes a pending one writes (or writes bytes one reads), or at COPY_BATCH_MAX
    // `group` / `Y` / `org`: the node the job belongs to, that node's output, the origin of the job's box in it (a CONCAT is two boxes); `same`: source and destination box have
    // the same shape.  A later copy that reads exactly a pending node's output is FORWARDED: it reads that node's sources instead (copy_queue), so chains of packs leave together
    // `node`: the tensor the group writes -- a group whose tensor has no reader left (every reader so far was forwarded, none comes later, nothing lazy looks at it, it is no
    // output and owns its bytes) is DROPPED instead of launched: the intermediate packs of a CONCAT chain are never written
    struct copy_pending { copy_pair job; const char * rlo, * rhi, * wlo, * whi; int group; tdesc Y; int64_t org[4]; bool same; const ggml_tensor * node; };
    int           cur_node = 0;
    const ggml_tensor * cq_owner = nullptr;          // a node run with its output re-pointed at a CONT sink's buffer (cont_sink): the bytes a queued copy writes belong to the SINK tensor
    struct dead_range { const char * lo, * hi; const ggml_tensor * node; int at; };
    std::vector<dead_range> cq_dead;                 // MI355X_COPY_PRUNE_VERIFY=1: outputs of dropped groups; a later read of one is reported
    int           cq_group = 0;
    std::vector<copy_pending> cq;
    long          n_copies_batched = 0;
};

// ------------------------------------------------------------------------------------------------ profiling
static inline hipEvent_t prof_event(backend_ctx * c) {
    if (!c->prof_event_pool.empty()) { hipEvent_t e = c->prof_event_pool.back(); c->prof_event_pool.pop_back(); return e; }
    hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return e;
}
struct prof_scope {
    backend_ctx * c; bool on; backend_ctx::pending_prof p;
    prof_scope(exec_state & s, const char * cls, double bytes) : c(s.c), on(s.c->opt_profile && !s.capturing) {
        if (!on) return;
        p.cls = cls; p.bytes = bytes; p.a = prof_event(c); p.b = prof_event(c);
        HIP_CHECK(hipEventRecord(p.a, s.st)); st = s.st;
    }
    ~prof_scope() { if (!on) return; HIP_CHECK(hipEventRecord(p.b, st)); c->prof_pending.push_back(p); }
    hipStream_t st = nullptr;
};
static inline void prof_drain(backend_ctx * c) {
    static const bool each = getenv("MI355X_PROFILE_EACH") != nullptr;            // one line per launch (class, the class's byte / flop figure, us) on stderr
    for (auto & p : c->prof_pending) {
        HIP_CHECK(hipEventSynchronize(p.b));
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
        if (each) fprintf(stderr, "[mi355x prof] %-18s %14.0f %9.2f us\n", p.cls.c_str(), p.bytes, ms * 1000.0);
        prof_class & pc = c->prof[p.cls];
        pc.us += ms * 1000.0; pc.bytes += p.bytes; pc.n += 1;
        c->prof_event_pool.push_back(p.a); c->prof_event_pool.push_back(p.b);
    }
    c->prof_pending.clear();
}


// ---- graph_plan.cpp
static const int64_t ROPE_TABLE_MIN_TOKENS = 32;
static const int64_t GEMM_MIN_COLS = MI_MMVQ_MAX_COLS + 1;
bool mm_uses_mmq(const ggml_tensor * n);
bool mm_uses_gemm(const ggml_tensor * n);
bool mm_uses_mmq_tile(const ggml_tensor * n);
act_kind gemm_act_kind(const ggml_tensor * n);          // ACT_F16, or ACT_F16Q for K-quant weights (option "prefill_q8k")
void prefill_q8k_set_mode(int m);
void mmq_tile_set_mode(int m);
int64_t mmq_max_cols();
bool mm_uses_gemm_any_f16(const ggml_tensor * n);
size_t graph_act_scratch_need(const ggml_cgraph * g);
size_t graph_w_scratch_need(const ggml_cgraph * g);
void fill_fattn_args(const ggml_tensor * n, fattn_args & f, tdesc & m);
size_t graph_fa_scratch_need(const ggml_cgraph * g);
size_t graph_rope_scratch_need(const ggml_cgraph * g);
int64_t gemm_group_split_max_cols();
size_t graph_gemm_partial_need(const ggml_cgraph * g);
bool ensure_scratch(backend_ctx * c, void ** p, size_t * have, size_t need);
// ---- graph_exec.cpp
bool mm_takes_gemm_any(const ggml_tensor * n);
void run_nodes(exec_state & s, ggml_cgraph * g);
} // namespace mi

// graph.cpp -- graph executor of the MI355X backend: what `ggml_backend_i::graph_compute` does.
//
// The scheduler hands over a ggml_cgraph whose nodes are all on this device
// (reference call site: ggml_backend_sched_compute_splits, ggml/src/ggml-backend.cpp:1553).  Nodes are executed in
// order on the backend's HIP stream; nothing here synchronises with the host, so graph_compute is
// asynchronous as the ABI allows (ggml-backend-impl.h:107-108) and `synchronize` drains the stream.
//
//  * node dispatch      : one hand-written gfx950 kernel per (op, type, layout) tuple that supports_op admits
//  * activation cache   : consecutive MUL_MATs that share src1 (wq/wk/wv, ffn_up/ffn_gate) quantise it once
//  * fusion             : RMS_NORM + MUL folded into one launch (the reference's GPU backend fuses the same pair,
//                         ggml-cuda.cu:3012-3022)
//  * hipGraph replay    : a cgraph submitted repeatedly with an identical fingerprint (decode: same topology,
//                         same buffers, only input *data* changes) is captured on its second submission and
//                         replayed afterwards -- ~10 us of host time per token instead of ~3.5 us per kernel
#include "graph.hpp"
#include "ggml_util.hpp"
#include "kernels.hpp"
#include "../../include/ggml-mi355x.h"

namespace mi {

// ------------------------------------------------------------------------------------------------ helpers
static tdesc td(const ggml_tensor * t) {
    tdesc d; d.p = t->data;
    for (int i = 0; i < 4; ++i) { d.ne[i] = t->ne[i]; d.nb[i] = t->nb[i]; }
    return d;
}
static bool is_noop(const ggml_tensor * t) {
    return t->op == GGML_OP_NONE || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE ||
           t->op == GGML_OP_TRANSPOSE || is_empty(t);
}

enum act_kind { ACT_NONE = 0, ACT_Q8K, ACT_Q80, ACT_F16, ACT_F32 };
static act_kind act_kind_for(int wtype) {
    switch (wtype) {
        case GGML_TYPE_Q4_K: case GGML_TYPE_Q6_K: return ACT_Q8K;
        case GGML_TYPE_Q8_0: return ACT_Q80;
        case GGML_TYPE_F16:  return ACT_F16;
        case GGML_TYPE_F32:  return ACT_F32;
        default: return ACT_NONE;
    }
}
static size_t act_image_bytes(act_kind k, int64_t K) {
    switch (k) {
        case ACT_Q8K: return q8k_image_bytes(K);
        case ACT_Q80: return q80_image_bytes(K);
        case ACT_F16: return ((size_t) K * 2 + 15) & ~(size_t) 15;
        default: return 0;
    }
}

struct exec_state {
    backend_ctx * c;
    hipStream_t   st;
    long          n_kernels = 0;
    // activation cache
    const void *  a_src = nullptr; act_kind a_kind = ACT_NONE; int64_t a_K = 0, a_ne[3] = {0, 0, 0}; size_t a_nb[3] = {0, 0, 0};
    bool          capturing = false;
};

// ------------------------------------------------------------------------------------------------ profiling
static hipEvent_t prof_event(backend_ctx * c) {
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
static void prof_drain(backend_ctx * c) {
    for (auto & p : c->prof_pending) {
        HIP_CHECK(hipEventSynchronize(p.b));
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
        prof_class & pc = c->prof[p.cls];
        pc.us += ms * 1000.0; pc.bytes += p.bytes; pc.n += 1;
        c->prof_event_pool.push_back(p.a); c->prof_event_pool.push_back(p.b);
    }
    c->prof_pending.clear();
}

// ------------------------------------------------------------------------------------------------ supports_op
bool supports_op(const ggml_tensor * op) {
    const ggml_tensor * s0 = op->src[0];
    const ggml_tensor * s1 = op->src[1];
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            return true;
        case GGML_OP_MUL_MAT: {
            if (!s0 || !s1) return false;
            const act_kind k = act_kind_for(s0->type);
            if (k == ACT_NONE || s1->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32) return false;
            if (s0->ne[0] % blck_size(s0->type) != 0) return false;
            if (s0->nb[0] != type_size(s0->type) || s1->nb[0] != sizeof(float) || op->nb[0] != sizeof(float)) return false;
            if (s0->nb[1] < row_size(s0->type, s0->ne[0])) return false;          // transposed weights: not handled
            if (s0->ne[2] == 0 || s0->ne[3] == 0 || s1->ne[2] % s0->ne[2] != 0 || s1->ne[3] % s0->ne[3] != 0) return false;
            if (k == ACT_Q8K || k == ACT_Q80) {
                // 16-B / 2-B vector paths assume block-aligned rows (always true for ggml-allocated tensors)
                if (s0->nb[1] % (s0->type == GGML_TYPE_Q4_K ? 16 : 2) != 0) return false;
            }
            return true;
        }
        case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV:
            return s0 && s1 && s0->type == GGML_TYPE_F32 && s1->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 &&
                   same_shape(s0, op) && can_repeat(s1, s0);
        case GGML_OP_RMS_NORM:
            return s0 && s0->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && s0->nb[0] == 4 && op->nb[0] == 4;
        case GGML_OP_SCALE:
            return s0 && s0->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && is_contiguous(s0) && is_contiguous(op);
        case GGML_OP_UNARY: {
            if (!s0 || s0->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || !is_contiguous(s0) || !is_contiguous(op)) return false;
            switch (op_param_i32(op, 0)) {
                case GGML_UNARY_OP_ABS: case GGML_UNARY_OP_SGN: case GGML_UNARY_OP_NEG: case GGML_UNARY_OP_STEP: case GGML_UNARY_OP_TANH:
                case GGML_UNARY_OP_ELU: case GGML_UNARY_OP_RELU: case GGML_UNARY_OP_SIGMOID: case GGML_UNARY_OP_GELU: case GGML_UNARY_OP_GELU_QUICK:
                case GGML_UNARY_OP_SILU: case GGML_UNARY_OP_HARDSWISH: case GGML_UNARY_OP_HARDSIGMOID: case GGML_UNARY_OP_EXP: case GGML_UNARY_OP_GELU_ERF:
                    return true;
                default: return false;
            }
        }
        case GGML_OP_GLU: {
            if (!s0 || s0->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32) return false;
            if (!is_contiguous_1(s0) || !is_contiguous_1(op) || (s1 && (!is_contiguous_1(s1) || s1->type != GGML_TYPE_F32))) return false;
            switch (op_param_i32(op, 0)) {
                case GGML_GLU_OP_REGLU: case GGML_GLU_OP_GEGLU: case GGML_GLU_OP_SWIGLU: case GGML_GLU_OP_GEGLU_ERF: case GGML_GLU_OP_GEGLU_QUICK: return true;
                default: return false;
            }
        }
        case GGML_OP_ROPE: {
            if (!s0 || !s1 || s0->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || s1->type != GGML_TYPE_I32) return false;
            const int mode = op_param_i32(op, 2);
            if (mode != GGML_ROPE_TYPE_NORMAL && mode != GGML_ROPE_TYPE_NEOX) return false;     // mrope / vision: CPU
            if (op->src[2] && op->src[2]->type != GGML_TYPE_F32) return false;
            return s0->nb[0] == 4 && op->nb[0] == 4 && (op_param_i32(op, 1) % 2) == 0;
        }
        case GGML_OP_SOFT_MAX:
            if (!s0 || s0->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || !is_contiguous(op) || s0->nb[0] != 4) return false;
            if (s1 && s1->type != GGML_TYPE_F16 && s1->type != GGML_TYPE_F32) return false;
            if (s1 && s1->nb[0] != type_size(s1->type)) return false;
            if (op->src[2] && op->src[2]->type != GGML_TYPE_F32) return false;
            return s0->ne[0] * 4 <= 150 * 1024;
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP: {
            if (!s0) return false;
            const int a = s0->type, b = op->type;
            const bool fl = (a == GGML_TYPE_F32 || a == GGML_TYPE_F16) && (b == GGML_TYPE_F32 || b == GGML_TYPE_F16);
            return (fl || (a == GGML_TYPE_I32 && b == GGML_TYPE_I32)) && nelements(s0) == nelements(op);
        }
        case GGML_OP_GET_ROWS: {
            if (!s0 || !s1 || s1->type != GGML_TYPE_I32 || op->type != GGML_TYPE_F32 || op->nb[0] != 4) return false;
            switch (s0->type) {
                case GGML_TYPE_F32: case GGML_TYPE_F16: case GGML_TYPE_Q8_0: case GGML_TYPE_Q4_K: case GGML_TYPE_Q6_K:
                    return s0->nb[0] == type_size(s0->type);
                default: return false;
            }
        }
        case GGML_OP_SET_ROWS:
            return s0 && s1 && s0->type == GGML_TYPE_F32 && (op->type == GGML_TYPE_F16 || op->type == GGML_TYPE_F32) &&
                   (s1->type == GGML_TYPE_I64 || s1->type == GGML_TYPE_I32) && s0->nb[0] == 4 && op->nb[0] == type_size(op->type);
        case GGML_OP_FLASH_ATTN_EXT: {
            const ggml_tensor * q = s0, * k = s1, * v = op->src[2], * m = op->src[3];
            if (!q || !k || !v) return false;
            if (q->type != GGML_TYPE_F32 || k->type != GGML_TYPE_F16 || v->type != GGML_TYPE_F16 || op->type != GGML_TYPE_F32) return false;
            if (q->ne[0] != k->ne[0] || k->ne[0] != v->ne[0]) return false;
            if (q->ne[0] != 64 && q->ne[0] != 128) return false;
            if (q->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2) return false;
            if (k->nb[1] % 16 != 0 || v->nb[1] % 16 != 0 || k->nb[2] % 16 != 0 || v->nb[2] % 16 != 0) return false;
            if (m && (m->type != GGML_TYPE_F16 || m->nb[0] != 2)) return false;
            if (op->src[4] && op->src[4]->type != GGML_TYPE_F32) return false;
            if (k->ne[2] == 0 || q->ne[2] % k->ne[2] != 0 || k->ne[2] != v->ne[2] || q->ne[3] != k->ne[3]) return false;
            return is_contiguous(op);
        }
        default:
            return false;
    }
}

// ------------------------------------------------------------------------------------------------ scratch sizing
static size_t graph_act_scratch_need(const ggml_cgraph * g) {
    size_t need = 0;
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        if (n->op != GGML_OP_MUL_MAT || is_empty(n)) continue;
        const act_kind k = act_kind_for(n->src[0]->type);
        const size_t b = act_image_bytes(k, n->src[1]->ne[0]) * (size_t) (n->src[1]->ne[1] * n->src[1]->ne[2] * n->src[1]->ne[3]);
        if (b > need) need = b;
    }
    return need;
}
static void ensure_scratch(backend_ctx * c, void ** p, size_t * have, size_t need) {
    if (need <= *have) return;
    HIP_CHECK(hipStreamSynchronize(c->stream));             // nothing in flight may still read the old block
    if (*p) HIP_CHECK(hipFree(*p));
    size_t n = need + need / 4; n = (n + ((size_t) 1 << 20) - 1) & ~(((size_t) 1 << 20) - 1);
    HIP_CHECK(hipMalloc(p, n));
    *have = n;
    c->execs.clear();                                       // captured graphs baked the old pointer in
}

// ------------------------------------------------------------------------------------------------ MUL_MAT
static void op_mul_mat(exec_state & s, const ggml_tensor * dst) {
    const ggml_tensor * w = dst->src[0];
    const ggml_tensor * x = dst->src[1];
    const int64_t K = w->ne[0], M = w->ne[1], N = x->ne[1];
    const int64_t ne12 = x->ne[2], ne13 = x->ne[3];
    const int64_t r2 = ne12 / w->ne[2], r3 = ne13 / w->ne[3];
    const act_kind kind = act_kind_for(w->type);
    const size_t img = act_image_bytes(kind, K);

    // ---- activation conversion (skipped when the previous MUL_MAT already converted the very same src1)
    const bool cached = kind != ACT_F32 && s.a_src == x->data && s.a_kind == kind && s.a_K == K && s.a_ne[0] == N && s.a_ne[1] == ne12 &&
                        s.a_ne[2] == ne13 && s.a_nb[0] == x->nb[1] && s.a_nb[1] == x->nb[2] && s.a_nb[2] == x->nb[3];
    if (kind != ACT_F32 && !cached) {
        auto conv = [&](const float * src, size_t xs, void * out, int64_t rows) {
            if      (kind == ACT_Q8K) quantize_q8k_image(src, xs, out, K, rows, s.st);
            else if (kind == ACT_Q80) quantize_q80_image(src, xs, out, K, rows, s.st);
            else                      convert_f32_f16_rows(src, xs, (uint16_t *) out, img, K, rows, s.st);
            ++s.n_kernels;
        };
        const bool flat = (ne12 == 1 || x->nb[2] == (size_t) N * x->nb[1]) && (ne13 == 1 || x->nb[3] == (size_t) ne12 * x->nb[2]);
        if (flat) {
            conv((const float *) x->data, x->nb[1], s.c->act_scratch, N * ne12 * ne13);
        } else {
            for (int64_t i13 = 0; i13 < ne13; ++i13)
                for (int64_t i12 = 0; i12 < ne12; ++i12)
                    conv((const float *) ((const char *) x->data + i12 * x->nb[2] + i13 * x->nb[3]), x->nb[1],
                         (char *) s.c->act_scratch + (size_t) ((i13 * ne12 + i12) * N) * img, N);
        }
        s.a_src = x->data; s.a_kind = kind; s.a_K = K; s.a_ne[0] = N; s.a_ne[1] = ne12; s.a_ne[2] = ne13;
        s.a_nb[0] = x->nb[1]; s.a_nb[1] = x->nb[2]; s.a_nb[2] = x->nb[3];
    }

    // ---- weight streaming
    const double wbytes = (double) M * (double) row_size(w->type, K);
    const char * cls = w->type == GGML_TYPE_Q4_K ? "mmv_q4k" : w->type == GGML_TYPE_Q6_K ? "mmv_q6k" : w->type == GGML_TYPE_Q8_0 ? "mmv_q80" :
                       w->type == GGML_TYPE_F16 ? "mmv_f16" : "mmv_f32";
    for (int64_t i13 = 0; i13 < ne13; ++i13) {
        for (int64_t i12 = 0; i12 < ne12; ++i12) {
            const char * wp = (const char *) w->data + (i12 / r2) * w->nb[2] + (i13 / r3) * w->nb[3];
            char *       dp = (char *) dst->data + i12 * dst->nb[2] + i13 * dst->nb[3];
            for (int64_t c0 = 0; c0 < N; c0 += MI_MMVQ_MAX_COLS) {
                mmv_args a;
                a.W = wp; a.w_rs = w->nb[1]; a.K = K; a.nrows = M;
                a.ncols = (int) (N - c0 < MI_MMVQ_MAX_COLS ? N - c0 : MI_MMVQ_MAX_COLS);
                a.dst = (float *) (dp + c0 * dst->nb[1]); a.dst_cs = dst->nb[1];
                if (kind == ACT_F32) {
                    a.act = (const char *) x->data + i12 * x->nb[2] + i13 * x->nb[3] + c0 * x->nb[1]; a.act_cs = x->nb[1];
                } else {
                    a.act = (const char *) s.c->act_scratch + (size_t) ((i13 * ne12 + i12) * N + c0) * img; a.act_cs = img;
                }
                prof_scope ps(s, cls, wbytes);
                switch (w->type) {
                    case GGML_TYPE_Q4_K: mmv_q4_K(a, s.st); break;
                    case GGML_TYPE_Q6_K: mmv_q6_K(a, s.st); break;
                    case GGML_TYPE_Q8_0: mmv_q8_0(a, s.st); break;
                    case GGML_TYPE_F16:  mmv_f16(a, s.st); break;
                    default:             mmv_f32(a, s.st); break;
                }
                ++s.n_kernels;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ node dispatch
// returns the number of graph nodes consumed (>1 when a fusion fired)
static int compute_node(exec_state & s, ggml_cgraph * g, int i) {
    ggml_tensor * n = g->nodes[i];
    if (is_noop(n)) return 1;
    if (n->op != GGML_OP_MUL_MAT) s.a_src = nullptr;                 // any other writer invalidates the activation cache

    switch (n->op) {
        case GGML_OP_MUL_MAT:
            op_mul_mat(s, n);
            return 1;
        case GGML_OP_RMS_NORM: {
            const float eps = op_param_f32(n, 0);
            // fusion: RMS_NORM -> MUL(norm, w) where the norm output has no other consumer inside this graph
            if (s.c->opt_fusion && i + 1 < g->n_nodes) {
                ggml_tensor * m = g->nodes[i + 1];
                if (m->op == GGML_OP_MUL && m->type == GGML_TYPE_F32 && (m->src[0] == n || m->src[1] == n)) {
                    const ggml_tensor * wt = m->src[0] == n ? m->src[1] : m->src[0];
                    bool only_use = true;
                    for (int j = i + 2; j < g->n_nodes && only_use; ++j)
                        for (int k = 0; k < GGML_MAX_SRC; ++k) if (g->nodes[j]->src[k] == n) { only_use = false; break; }
                    if (only_use && !(n->flags & GGML_TENSOR_FLAG_OUTPUT) && wt->type == GGML_TYPE_F32 && wt->nb[0] == 4 && same_shape(m, n) &&
                        can_repeat(wt, n) && m->nb[0] == 4 && wt != n) {
                        const tdesc wd = td(wt);
                        prof_scope ps(s, "rms_norm_mul", 0);
                        rms_norm(td(n->src[0]), td(m), eps, &wd, s.st); ++s.n_kernels;
                        return 2;
                    }
                }
            }
            prof_scope ps(s, "rms_norm", 0);
            rms_norm(td(n->src[0]), td(n), eps, nullptr, s.st); ++s.n_kernels;
            return 1;
        }
        case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: {
            prof_scope ps(s, "bin", 0);
            bin_bcast_f32(n->op, td(n->src[0]), td(n->src[1]), td(n), s.st); ++s.n_kernels;
            return 1;
        }
        case GGML_OP_SCALE: {
            prof_scope ps(s, "scale", 0);
            scale_f32((const float *) n->src[0]->data, (float *) n->data, nelements(n), op_param_f32(n, 0), op_param_f32(n, 1), s.st); ++s.n_kernels;
            return 1;
        }
        case GGML_OP_UNARY: {
            prof_scope ps(s, "unary", 0);
            unary_f32(op_param_i32(n, 0), (const float *) n->src[0]->data, (float *) n->data, nelements(n), s.st); ++s.n_kernels;
            return 1;
        }
        case GGML_OP_GLU: {
            prof_scope ps(s, "glu", 0);
            tdesc b; if (n->src[1]) b = td(n->src[1]);
            glu_f32(op_param_i32(n, 0), td(n->src[0]), n->src[1] ? &b : nullptr, op_param_i32(n, 1) != 0, td(n), s.st); ++s.n_kernels;
            return 1;
        }
        case GGML_OP_ROPE: {
            rope_params rp;
            rp.n_dims = op_param_i32(n, 1); rp.mode = op_param_i32(n, 2); rp.n_ctx_orig = op_param_i32(n, 4);
            rp.freq_base = op_param_f32(n, 5); rp.freq_scale = op_param_f32(n, 6); rp.ext_factor = op_param_f32(n, 7);
            rp.attn_factor = op_param_f32(n, 8); rp.beta_fast = op_param_f32(n, 9); rp.beta_slow = op_param_f32(n, 10);
            prof_scope ps(s, "rope", 0);
            rope_f32(td(n->src[0]), (const int32_t *) n->src[1]->data, n->src[2] ? (const float *) n->src[2]->data : nullptr, td(n), rp, s.st); ++s.n_kernels;
            return 1;
        }
        case GGML_OP_SOFT_MAX: {
            tdesc m; if (n->src[1]) m = td(n->src[1]);
            prof_scope ps(s, "soft_max", 0);
            soft_max_f32(td(n->src[0]), n->src[1] ? &m : nullptr, n->src[1] ? n->src[1]->type : 0,
                         n->src[2] ? (const float *) n->src[2]->data : nullptr, td(n), op_param_f32(n, 0), op_param_f32(n, 1), s.st); ++s.n_kernels;
            return 1;
        }
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP: {
            prof_scope ps(s, "cpy", 0);
            const ggml_tensor * src = n->src[0];
            // CPY writes into src[1]'s storage, which `n` is a view of; n->data is the destination in all three ops
            cpy_strided(td(src), src->type, td(n), n->type, s.st); ++s.n_kernels;
            return 1;
        }
        case GGML_OP_GET_ROWS: {
            prof_scope ps(s, "get_rows", 0);
            get_rows(td(n->src[0]), n->src[0]->type, td(n->src[1]), td(n), s.st); ++s.n_kernels;
            return 1;
        }
        case GGML_OP_SET_ROWS: {
            prof_scope ps(s, "set_rows", 0);
            set_rows(td(n->src[0]), td(n->src[1]), n->src[1]->type, td(n), n->type, s.st); ++s.n_kernels;
            return 1;
        }
        case GGML_OP_FLASH_ATTN_EXT: {
            fattn_args f;
            f.q = td(n->src[0]); f.k = td(n->src[1]); f.v = td(n->src[2]); f.dst = td(n);
            tdesc m; if (n->src[3]) m = td(n->src[3]);
            f.mask = n->src[3] ? &m : nullptr;
            f.sinks = n->src[4] ? (const float *) n->src[4]->data : nullptr;
            f.scale = op_param_f32(n, 0); f.max_bias = op_param_f32(n, 1); f.logit_softcap = op_param_f32(n, 2);
            f.scratch = nullptr; f.scratch_bytes = 0;
            prof_scope ps(s, "fattn", 0);
            flash_attn_ext_f16(f, s.st); ++s.n_kernels;
            return 1;
        }
        default:
            log_msg(GGML_LOG_LEVEL_ERROR, "[mi355x] graph_compute: op %d (%s) reached the backend but is not implemented -- supports_op bug\n", (int) n->op, n->name);
            abort();
    }
}

static void run_nodes(exec_state & s, ggml_cgraph * g) {
    for (int i = 0; i < g->n_nodes;) i += compute_node(s, g, i);
}

// ------------------------------------------------------------------------------------------------ fingerprint
static inline uint64_t mix(uint64_t h, uint64_t v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); return h; }
static uint64_t fingerprint(const ggml_cgraph * g) {
    uint64_t h = 0xcbf29ce484222325ull;
    h = mix(h, (uint64_t) g->n_nodes);
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        h = mix(h, (uint64_t) n->op); h = mix(h, (uint64_t) n->type); h = mix(h, (uint64_t) (uintptr_t) n->data);
        for (int d = 0; d < 4; ++d) { h = mix(h, (uint64_t) n->ne[d]); h = mix(h, (uint64_t) n->nb[d]); }
        for (int p = 0; p < GGML_MAX_OP_PARAMS / 4; ++p) h = mix(h, (uint64_t) (uint32_t) n->op_params[p]);
        for (int k = 0; k < GGML_MAX_SRC; ++k) {
            const ggml_tensor * sr = n->src[k];
            if (!sr) { h = mix(h, 0x5bd1e995u + k); continue; }
            h = mix(h, (uint64_t) (uintptr_t) sr->data); h = mix(h, (uint64_t) sr->type);
            for (int d = 0; d < 4; ++d) { h = mix(h, (uint64_t) sr->ne[d]); h = mix(h, (uint64_t) sr->nb[d]); }
        }
    }
    return h;
}

// ------------------------------------------------------------------------------------------------ graph_compute
enum ggml_status graph_compute(backend_ctx * c, ggml_cgraph * g) {
    if (g->n_nodes == 0) return GGML_STATUS_SUCCESS;
    ensure_scratch(c, &c->act_scratch, &c->act_scratch_bytes, graph_act_scratch_need(g));

    int n_real = 0;
    for (int i = 0; i < g->n_nodes; ++i) n_real += !is_noop(g->nodes[i]);

    const bool try_graph = c->opt_graphs && !c->opt_profile && n_real >= 8;
    if (try_graph) {
        const uint64_t fp = fingerprint(g);
        graph_exec * ge = nullptr;
        for (auto & e : c->execs) if (e.fingerprint == fp) { ge = &e; break; }
        if (!ge) {
            if (c->execs.size() >= 8) {                      // evict the least recently used
                size_t lru = 0;
                for (size_t k = 1; k < c->execs.size(); ++k) if (c->execs[k].last_use < c->execs[lru].last_use) lru = k;
                if (c->execs[lru].exec)  HIP_CHECK(hipGraphExecDestroy(c->execs[lru].exec));
                if (c->execs[lru].graph) HIP_CHECK(hipGraphDestroy(c->execs[lru].graph));
                c->execs.erase(c->execs.begin() + lru);
            }
            c->execs.push_back(graph_exec());
            ge = &c->execs.back();
            ge->fingerprint = fp;
        }
        ge->last_use = ++c->tick;
        ge->seen++;
        if (ge->exec) {
            HIP_CHECK(hipGraphLaunch(ge->exec, c->stream));
            c->stat_replays++; c->stat_kernels_last = ge->n_kernels;
            return GGML_STATUS_SUCCESS;
        }
        if (ge->seen >= 2) {
            // second submission of an identical graph: capture it (the first, eager run already set every
            // function attribute and sized every scratch buffer, so the capture region only holds launches)
            exec_state s; s.c = c; s.st = c->stream; s.capturing = true;
            HIP_CHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            run_nodes(s, g);
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamEndCapture(c->stream, &graph);
            if (e == hipSuccess && graph) {
                hipGraphExec_t ex = nullptr;
                e = hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0);
                if (e == hipSuccess) {
                    ge->graph = graph; ge->exec = ex; ge->n_kernels = (int) s.n_kernels;
                    HIP_CHECK(hipGraphLaunch(ex, c->stream));
                    c->stat_captures++; c->stat_kernels_last = s.n_kernels;
                    return GGML_STATUS_SUCCESS;
                }
                HIP_CHECK(hipGraphDestroy(graph));
            }
            (void) hipGetLastError();
            log_msg(GGML_LOG_LEVEL_WARN, "[mi355x] hipGraph capture failed (%s); running eagerly\n", hipGetErrorString(e));
            c->opt_graphs = false;
        }
    }

    exec_state s; s.c = c; s.st = c->stream;
    run_nodes(s, g);
    c->stat_eager++; c->stat_kernels_last = s.n_kernels;
    if (c->opt_profile) prof_drain(c);
    return GGML_STATUS_SUCCESS;
}

void backend_ctx_init(backend_ctx * c) {
    const char * e;
    if ((e = getenv("MI355X_GRAPHS")))  c->opt_graphs  = atoi(e) != 0;
    if ((e = getenv("MI355X_FUSION")))  c->opt_fusion  = atoi(e) != 0;
    if ((e = getenv("MI355X_PROFILE"))) c->opt_profile = atoi(e) != 0;
}
void backend_ctx_release(backend_ctx * c) {
    for (auto & e : c->execs) { if (e.exec) (void) hipGraphExecDestroy(e.exec); if (e.graph) (void) hipGraphDestroy(e.graph); }
    c->execs.clear();
    for (auto ev : c->prof_event_pool) (void) hipEventDestroy(ev);
    if (c->act_scratch) (void) hipFree(c->act_scratch);
    if (c->w_scratch) (void) hipFree(c->w_scratch);
    if (c->copy_event) (void) hipEventDestroy(c->copy_event);
    if (c->stream) (void) hipStreamDestroy(c->stream);
}

} // namespace mi

extern "C" {
int mi355x_set_option(struct ggml_backend * backend, const char * key, long value) {
    mi::backend_ctx * c = (mi::backend_ctx *) backend->context;
    if (!strcmp(key, "graphs"))  { c->opt_graphs = value != 0; return 0; }
    if (!strcmp(key, "fusion"))  { c->opt_fusion = value != 0; c->execs.clear(); return 0; }
    if (!strcmp(key, "profile")) { c->opt_profile = value != 0; return 0; }
    if (!strcmp(key, "reset_stats")) { c->prof.clear(); c->stat_replays = c->stat_captures = c->stat_eager = 0; return 0; }
    return -1;
}
double mi355x_get_stat(struct ggml_backend * backend, const char * key) {
    mi::backend_ctx * c = (mi::backend_ctx *) backend->context;
    if (!strcmp(key, "graph_replays"))      return (double) c->stat_replays;
    if (!strcmp(key, "graph_captures"))     return (double) c->stat_captures;
    if (!strcmp(key, "eager_graphs"))       return (double) c->stat_eager;
    if (!strcmp(key, "kernels_last_graph")) return (double) c->stat_kernels_last;
    if (!strncmp(key, "prof_", 5)) {
        std::string k(key + 5);
        const size_t us = k.rfind("_us"), nn = k.rfind("_n"), by = k.rfind("_bytes");
        auto get = [&](const std::string & cls) -> const mi::prof_class * { auto it = c->prof.find(cls); return it == c->prof.end() ? nullptr : &it->second; };
        if (us != std::string::npos && us + 3 == k.size()) { auto p = get(k.substr(0, us)); return p ? p->us : 0.0; }
        if (by != std::string::npos && by + 6 == k.size()) { auto p = get(k.substr(0, by)); return p ? p->bytes : 0.0; }
        if (nn != std::string::npos && nn + 2 == k.size()) { auto p = get(k.substr(0, nn)); return p ? (double) p->n : 0.0; }
    }
    return -1.0;
}
}

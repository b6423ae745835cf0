// graph_plan.cpp -- planner side of the graph executor: supports_op (what the backend admits), which kernel family a MUL_MAT takes by shape and type,
// and the scratch a cgraph needs before its first launch.  (Split out of graph.cpp in round 4; no behaviour change.)
#include "graph_internal.hpp"
#include <algorithm>

namespace mi {

// ------------------------------------------------------------------------------------------------ supports_op
bool supports_op(const ggml_tensor * op) {
    const ggml_tensor * s0 = op->src[0];
    const ggml_tensor * s1 = op->src[1];
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            return true;
        case GGML_OP_MUL_MAT: {
            if (!s0 || !s1) return false;
            if (s0->type == GGML_TYPE_BF16)                          // BF16 weights: the any-shape f32-MFMA GEMM at every column count (gemm_any.hip)
                return s1->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && s0->nb[0] == 2 && s1->nb[0] == 4 && op->nb[0] == 4 && s0->nb[1] >= (size_t) s0->ne[0] * 2 &&
                       s0->ne[2] != 0 && s0->ne[3] != 0 && s1->ne[2] % s0->ne[2] == 0 && s1->ne[3] % s0->ne[3] == 0 && s1->ne[2] * s1->ne[3] <= 65535;
            const act_kind k = act_kind_for(s0->type);
            if (k == ACT_NONE || op->type != GGML_TYPE_F32) return false;
            if (s1->type != GGML_TYPE_F32 && !(s1->type == GGML_TYPE_F16 && k == ACT_F16)) return false;      // F16 x F16: the convolutions' mat-mul
            if (s0->ne[0] % blck_size(s0->type) != 0) return false;
            if (s0->nb[0] != type_size(s0->type) || s1->nb[0] != type_size(s1->type) || op->nb[0] != sizeof(float)) return false;
            if (s0->nb[1] < row_size(s0->type, s0->ne[0])) return false;          // transposed weights: not handled
            if (s0->ne[2] == 0 || s0->ne[3] == 0 || s1->ne[2] % s0->ne[2] != 0 || s1->ne[3] % s0->ne[3] != 0) return false;
            if (k == ACT_Q8K || k == ACT_Q80) {
                // 16-B / 2-B vector paths assume block-aligned rows (always true for ggml-allocated tensors)
                if (s0->nb[1] % ((s0->type == GGML_TYPE_Q4_K || s0->type == GGML_TYPE_Q5_K) ? 16 : 2) != 0) return false;
            }
            // every mat-vec path (up to 8 columns per launch; F32 weights at any width) stages one activation column in LDS: a column
            // beyond 152 KiB has no kernel (e.g. attention without FLASH_ATTN_EXT past ~77k cache rows: K = n_kv) -> leave it to the CPU
            if (!mm_uses_gemm(op) && !mm_uses_mmq(op) && !mm_takes_gemm_any(op)) {       // (gemm_any stages nothing in LDS)
                const size_t col = k == ACT_F32 ? (size_t) s0->ne[0] * 4 : act_image_bytes(is_image_quant(s0->type) ? ACT_F16 : k, s0->ne[0]);
                if (col > (size_t) 152 * 1024) return false;
            }
            return true;
        }
        case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV:
            return s0 && s1 && s0->type == GGML_TYPE_F32 && s1->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 &&
                   same_shape(s0, op) && can_repeat(s1, s0);
        case GGML_OP_RMS_NORM:
        case GGML_OP_NORM:
            return s0 && s0->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && s0->nb[0] == 4 && op->nb[0] == 4;
        case GGML_OP_IM2COL:
            // src0 = kernel (shape only), src1 = f32 image with dense [IH, IW] planes, dst dense f16 / f32
            return s0 && s1 && s1->type == GGML_TYPE_F32 && (op->type == GGML_TYPE_F16 || op->type == GGML_TYPE_F32) && is_contiguous(op) && s1->nb[0] == 4 &&
                   (op_param_i32(op, 6) != 1 || s1->nb[1] == (size_t) s1->ne[0] * 4) && nelements(op) < ((int64_t) 1 << 40);
        case GGML_OP_POOL_2D:
            return s0 && (s0->type == GGML_TYPE_F32 || s0->type == GGML_TYPE_F16) && op->type == GGML_TYPE_F32 && is_contiguous(op) &&
                   s0->nb[0] == (s0->type == GGML_TYPE_F32 ? 4u : 2u) && (s0->ne[3] == 1 || s0->nb[3] == (size_t) s0->ne[2] * s0->nb[2]) &&
                   (op_param_i32(op, 0) == GGML_OP_POOL_AVG || op_param_i32(op, 0) == GGML_OP_POOL_MAX);
        case GGML_OP_POOL_1D:                                 // the reference implements k == s, p == 0 only (ops.cpp:7270-7276)
            return s0 && (s0->type == GGML_TYPE_F32 || s0->type == GGML_TYPE_F16) && op->type == GGML_TYPE_F32 && is_contiguous(op) && is_contiguous(s0) &&
                   op_param_i32(op, 1) == op_param_i32(op, 2) && op_param_i32(op, 3) == 0 &&
                   (op_param_i32(op, 0) == GGML_OP_POOL_AVG || op_param_i32(op, 0) == GGML_OP_POOL_MAX);
        case GGML_OP_SCALE:
            return s0 && s0->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && is_contiguous(s0) && is_contiguous(op);
        // ---- the Token2Wav graphs' extra ops (kernels/t2w_ops.hip)
        case GGML_OP_SQR: case GGML_OP_SQRT: case GGML_OP_LOG: case GGML_OP_SIN: case GGML_OP_COS: case GGML_OP_CLAMP: case GGML_OP_LEAKY_RELU:
            return s0 && s0->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && is_contiguous(s0) && is_contiguous(op);
        case GGML_OP_CONCAT: {
            if (!s0 || !s1 || s0->type != op->type || s1->type != op->type) return false;
            const int t = op->type;
            return t == GGML_TYPE_F32 || t == GGML_TYPE_I32 || t == GGML_TYPE_F16;
        }
        case GGML_OP_REPEAT: {
            if (!s0 || s0->type != op->type || !can_repeat(s0, op)) return false;
            const int t = op->type;
            return t == GGML_TYPE_F32 || t == GGML_TYPE_I32 || t == GGML_TYPE_F16;
        }
        case GGML_OP_PAD:
            return s0 && s0->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && s0->nb[0] == 4 && is_contiguous(op);
        case GGML_OP_PAD_REFLECT_1D:
            return s0 && s0->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && op_param_i32(op, 0) < s0->ne[0] && op_param_i32(op, 1) < s0->ne[0];
        case GGML_OP_ARANGE:
            return op->type == GGML_TYPE_F32 && is_contiguous(op);
        case GGML_OP_TIMESTEP_EMBEDDING:
            return s0 && s0->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && s0->nb[0] == 4 && op->nb[0] == 4;
        case GGML_OP_SUM_ROWS:
            return s0 && s0->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && s0->nb[0] == 4;
        case GGML_OP_CONV_TRANSPOSE_1D:          // (ggml_conv_transpose_1d asserts p0 == 0, d0 == 1 and a 2-D src1)
            return s0 && s1 && (s0->type == GGML_TYPE_F16 || s0->type == GGML_TYPE_F32) && s1->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 &&
                   s0->nb[0] == (s0->type == GGML_TYPE_F16 ? 2u : 4u) && s1->nb[0] == 4 && op->nb[0] == 4 && s1->ne[2] == 1 && s1->ne[3] == 1 && s0->ne[3] == 1;
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
            const bool fi = (a == GGML_TYPE_F32 && b == GGML_TYPE_I32) || (a == GGML_TYPE_I32 && b == GGML_TYPE_F32);      // ggml_cast to / from i32 (Token2Wav masks)
            return (fl || fi || (a == GGML_TYPE_I32 && b == GGML_TYPE_I32)) && nelements(s0) == nelements(op);
        }
        case GGML_OP_GET_ROWS: {
            if (!s0 || !s1 || s1->type != GGML_TYPE_I32 || op->type != GGML_TYPE_F32 || op->nb[0] != 4) return false;
            switch (s0->type) {
                case GGML_TYPE_F32: case GGML_TYPE_F16: case GGML_TYPE_BF16: case GGML_TYPE_Q8_0: case GGML_TYPE_Q4_K: case GGML_TYPE_Q6_K:
                case GGML_TYPE_Q4_0: case GGML_TYPE_Q4_1: case GGML_TYPE_Q5_0: case GGML_TYPE_Q5_1: case GGML_TYPE_Q2_K: case GGML_TYPE_Q3_K: case GGML_TYPE_Q5_K:
                    return s0->nb[0] == type_size(s0->type);
                default: return false;
            }
        }
        case GGML_OP_SET_ROWS:
            return s0 && s1 && s0->type == GGML_TYPE_F32 &&
                   (op->type == GGML_TYPE_F16 || op->type == GGML_TYPE_F32 || op->type == GGML_TYPE_BF16 || ((op->type == GGML_TYPE_Q8_0 || op->type == GGML_TYPE_Q4_0) && s0->ne[0] % 32 == 0)) &&
                   (s1->type == GGML_TYPE_I64 || s1->type == GGML_TYPE_I32) && s0->nb[0] == 4 && op->nb[0] == type_size(op->type);
        case GGML_OP_FLASH_ATTN_EXT: {
            const ggml_tensor * q = s0, * k = s1, * v = op->src[2], * m = op->src[3];
            if (!q || !k || !v) return false;
            if (q->type != GGML_TYPE_F32 || k->type != v->type || op->type != GGML_TYPE_F32) return false;
            if (k->type != GGML_TYPE_F16 && k->type != GGML_TYPE_F32 && k->type != GGML_TYPE_BF16 && k->type != GGML_TYPE_Q8_0 && k->type != GGML_TYPE_Q4_0) return false;
            if (q->ne[0] != k->ne[0]) return false;
            // the MFMA / streaming / one-token kernels (F16 cache, head 64 / 128); anything else -- other head sizes, a quantised / BF16 / F32 cache -- fattn_any.hip
            const bool special = k->type == GGML_TYPE_F16 && (q->ne[0] == 64 || q->ne[0] == 128) && v->ne[0] == q->ne[0];
            if (!special && (q->ne[0] > 576 || v->ne[0] > 576)) return false;
            if ((k->type == GGML_TYPE_Q8_0 || k->type == GGML_TYPE_Q4_0) && (k->ne[0] % 32 != 0 || v->ne[0] % 32 != 0)) return false;
            if (q->nb[0] != 4 || k->nb[0] != type_size(k->type) || v->nb[0] != type_size(v->type)) return false;
            if (special && (k->nb[1] % 16 != 0 || v->nb[1] % 16 != 0 || k->nb[2] % 16 != 0 || v->nb[2] % 16 != 0)) return false;
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
// batches of more than 8 columns go to the MFMA GEMM (activations rounded to f16; quantised weights de-quantised to f16 first)
// ... except K-quant weights against up to MMQ_MAX_COLS columns (several sequences decoded together, drafts, small ubatches): those
// read the quantised blocks themselves on the int8 matrix cores (mmq.hip) -- 0.56 / 0.82 bytes per weight instead of the 2 of an f16 image
int64_t mmq_max_cols() {
    static const int64_t v = getenv("MI355X_MMQ_MAX_COLS") ? atoll(getenv("MI355X_MMQ_MAX_COLS")) : 64;
    return v;
}
static int64_t mmq_min_cols() {       // narrower batches stay on the dot4 mat-vec kernels (one column: 4.3 TB/s; the MFMA tile would be 1/32 full).
    // measured crossover (12288 x 4096 Q4_K: dot4 8.0 / 11.2 / 19.0 us at 2 / 4 / 8 columns, mmq 13.3 us flat): 6 columns
    static const int64_t v = getenv("MI355X_MMQ_MIN_COLS") ? atoll(getenv("MI355X_MMQ_MIN_COLS")) : 6;
    return v;
}
bool mm_uses_mmq(const ggml_tensor * n) {
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    return (w->type == GGML_TYPE_Q4_K || w->type == GGML_TYPE_Q5_K || w->type == GGML_TYPE_Q6_K) && x->type == GGML_TYPE_F32 && x->ne[1] >= mmq_min_cols() && x->ne[1] <= mmq_max_cols() &&
           mmq_ok(w->type, w->ne[0], w->data, w->nb[1]) && (w->ne[2] == 1 || mmq_ok(w->type, w->ne[0], (const char *) w->data + w->nb[2], w->nb[1])) &&
           (w->ne[3] == 1 || mmq_ok(w->type, w->ne[0], (const char *) w->data + w->nb[3], w->nb[1]));
}
// Q4_K weights against a prefill ubatch (more than mmq_max_cols() columns): the tiled int8-MFMA kernel on the blocks themselves and the Q8_K-quantised
// activations -- the oracle's integers (mmq_tile.hip).  A sub-case of mm_uses_gemm(): the grouping / residual / split-K machinery of the GEMM path serves it.
static int g_mmq_tile = -1;                                   // option "mmq_tile": -1 = MI355X_MMQ_TILE decides (default off), 0 off, 1 on
void mmq_tile_set_mode(int m) { g_mmq_tile = m; }
bool mm_uses_mmq_tile(const ggml_tensor * n) {
    static const int env = getenv("MI355X_MMQ_TILE") ? atoi(getenv("MI355X_MMQ_TILE")) : 0;      // (off by default: exact, but slower than the F16-image GEMM -- DESIGN.md section 7, profiles/r05_mmq_tile.txt)
    if (!(g_mmq_tile >= 0 ? g_mmq_tile : env)) return false;
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    if (n->op != GGML_OP_MUL_MAT || !w || !x || w->type != GGML_TYPE_Q4_K || x->type != GGML_TYPE_F32 || n->type != GGML_TYPE_F32) return false;
    if (x->ne[1] <= mmq_max_cols() || x->ne[2] != 1 || x->ne[3] != 1 || w->ne[2] != 1 || w->ne[3] != 1 || x->nb[0] != 4 || n->nb[0] != 4) return false;
    if (x->nb[1] % 16 != 0 || ((uintptr_t) x->data & 15) != 0) return false;
    return mmq_tile_ok(w->type, w->ne[0], w->data, w->nb[1]);
}
// The activations of an MFMA GEMM node: f16-rounded rows -- or, for K-quant weights (the reference CPU backend quantises src1 to Q8_K for every one of them,
// ggml-cpu.c:1272-1306 with vec_dot_type Q8_K), f16 rows of the Q8_K-QUANTISED values, so that the product on the weights' F16 image differs from the reference's
// integer arithmetic by f16 rounding only (NMSE 2e-6 per mat-mul), not by the reference's own 8-bit quantisation noise (5e-5).  Option "prefill_q8k", default OFF: it costs two
// re-quantisation launches per layer (pp512 44.7 k -> 41.5 k tok/s) and end to end both forms sit on the reference's own rounding-flip floor (tests/test_round5_gpu.py).
static int g_prefill_q8k = -1;
void prefill_q8k_set_mode(int m) { g_prefill_q8k = m; }
act_kind gemm_act_kind(const ggml_tensor * n) {
    static const int env = getenv("MI355X_PREFILL_Q8K") ? atoi(getenv("MI355X_PREFILL_Q8K")) : 0;
    if (!(g_prefill_q8k >= 0 ? g_prefill_q8k : env)) return ACT_F16;
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    const int t = w->type;
    const bool kq = t == GGML_TYPE_Q2_K || t == GGML_TYPE_Q3_K || t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K;
    return kq && x->type == GGML_TYPE_F32 && x->ne[0] % 256 == 0 ? ACT_F16Q : ACT_F16;
}
bool mm_uses_gemm(const ggml_tensor * n) {
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    static const bool no_gemm = getenv("MI355X_NO_GEMM") != nullptr;
    if (x->ne[1] < GEMM_MIN_COLS || no_gemm) return false;
    if (mm_uses_mmq(n)) return false;
    if (w->type != GGML_TYPE_F16 && w->type != GGML_TYPE_Q4_K && w->type != GGML_TYPE_Q5_K && w->type != GGML_TYPE_Q6_K && w->type != GGML_TYPE_Q8_0 && !is_image_quant(w->type) && !is_q40_like(w->type)) return false;
    const int64_t K = w->ne[0];
    if (K % 32 != 0) return false;
    if (w->type == GGML_TYPE_F16 && (w->nb[1] % 16 != 0 || w->nb[2] % 16 != 0 || w->nb[3] % 16 != 0 || ((uintptr_t) w->data & 15) != 0)) return false;
    // per-head products (an encoder's V^T . P over a growing K/V cache: K = 50, 100, ... positions): the DMA GEMM batches heads only at K % 64 == 0 and would
    // otherwise go out head by head (Whisper streaming chunk 16, K = 800: 384 extra launches); the any-shape f16 kernel takes every head in one launch
    if (w->type == GGML_TYPE_F16 && x->ne[2] * x->ne[3] > 1 && K % 64 != 0 && x->type == GGML_TYPE_F32 && x->ne[2] * x->ne[3] <= 65535) return false;
    return true;
}
// MUL_MAT that op_mul_mat sends to the any-shape GEMM's f16 kernel (gemm_any.hip k_gemm_any_h): F16 weights the DMA GEMMs do not take (odd K,
// unaligned rows) against more than 8 f32 columns -- it reads a ready-made f16 activation image as well as the f32 rows
bool mm_uses_gemm_any_f16(const ggml_tensor * n) {
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    static const bool off = getenv("MI355X_NO_GEMM_ANY") != nullptr || getenv("MI355X_NO_GEMM_ANY_H") != nullptr;
    return !off && !mm_uses_gemm(n) && w->type == GGML_TYPE_F16 && x->type == GGML_TYPE_F32 && x->ne[1] > MI_MMVQ_MAX_COLS && x->nb[0] == 4 && w->nb[0] == 2 && n->nb[0] == 4 &&
           x->ne[2] * x->ne[3] <= 65535 && w->ne[1] < (1ll << 31) && x->ne[1] < (1ll << 31) && w->ne[0] < (1ll << 31);
}
size_t graph_act_scratch_need(const ggml_cgraph * g) {
    size_t need = 0;
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        if (n->op != GGML_OP_MUL_MAT || is_empty(n)) continue;
        const act_kind k = mm_uses_gemm(n) ? ACT_F16 : act_kind_for(n->src[0]->type);
        size_t b = act_image_bytes(k, n->src[1]->ne[0]) * (size_t) (n->src[1]->ne[1] * n->src[1]->ne[2] * n->src[1]->ne[3]);
        if (mm_uses_mmq_tile(n)) b = mmqt_image_bytes(n->src[1]->ne[0], n->src[1]->ne[1]);
        if (b > need) need = b;
    }
    return need;
}
size_t graph_w_scratch_need(const ggml_cgraph * g) {
    size_t need = 0;
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        if (n->op != GGML_OP_MUL_MAT || is_empty(n) || n->src[0]->type == GGML_TYPE_F16 || !(mm_uses_gemm(n) || is_image_quant(n->src[0]->type)) || mm_uses_mmq_tile(n)) continue;
        const size_t b = (size_t) n->src[0]->ne[0] * (size_t) n->src[0]->ne[1] * 2;
        if (b > need) need = b;
    }
    return need;
}
void fill_fattn_args(const ggml_tensor * n, fattn_args & f, tdesc & m) {
    f.q = td(n->src[0]); f.k = td(n->src[1]); f.v = td(n->src[2]); f.dst = td(n);
    if (n->src[3]) m = td(n->src[3]);
    f.mask = n->src[3] ? &m : nullptr;
    f.sinks = n->src[4] ? (const float *) n->src[4]->data : nullptr;
    f.scale = op_param_f32(n, 0); f.max_bias = op_param_f32(n, 1); f.logit_softcap = op_param_f32(n, 2);
    f.scratch = nullptr; f.scratch_bytes = 0; f.kv_type = n->src[1]->type;
}
size_t graph_fa_scratch_need(const ggml_cgraph * g) {
    size_t need = 0;
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        if (n->op == GGML_OP_SOFT_MAX && !is_empty(n) && n->ne[1] == 1 && n->ne[3] == 1 && n->ne[0] > 256) {      // flash-attention off, one token, deep cache: the slices' partial rows (attn_one_sm)
            const size_t b = (size_t) n->ne[2] * (size_t) ((n->ne[0] + 255) / 256) * (128 + 2) * 4;
            if (b > need) need = b;
            continue;
        }
        if (n->op == GGML_OP_SOFT_MAX && !is_empty(n) && n->ne[1] > 32 && n->src[1] && n->src[1]->ne[2] == 1 && n->src[1]->ne[3] == 1) {      // flash-attention off, a batch of rows: mask tile map + f16 copy of an f32 mask (exec_attn_sm_prefill)
            const size_t b = ((fattn_map_bytes_host(n->ne[1], n->ne[0]) + 255) & ~(size_t) 255) + (size_t) n->src[1]->ne[1] * (size_t) n->ne[0] * 2;
            if (b > need) need = b;
            continue;
        }
        if (n->op != GGML_OP_FLASH_ATTN_EXT || is_empty(n)) continue;
        fattn_args f; tdesc m; fill_fattn_args(n, f, m);
        size_t b = fattn_scratch_bytes(f);
        if (n->src[0]->ne[1] == 1 && n->src[0]->ne[3] == 1 && n->src[0]->ne[0] == 128) b = std::max(b, fattn_gs_parts_bytes((int) n->src[0]->ne[2], 128));     // one token: the slices' partial states (k_fattn_gs)
        if (b > need) need = b;
    }
    return need;
}
size_t graph_rope_scratch_need(const ggml_cgraph * g) {
    size_t need = 0;
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        if (n->op != GGML_OP_ROPE || (n->ne[2] < ROPE_TABLE_MIN_TOKENS && n->ne[2] != 1)) continue;     // (one token: the table of fattn_one.hip)
        const size_t b = (size_t) n->ne[2] * (size_t) n->ne[0] * 4;
        if (b > need) need = b;
    }
    return need;
}
int64_t gemm_group_split_max_cols() {                 // grouped launches (wq / wk / wv) may split K up to this many columns
    static const int64_t v = getenv("MI355X_GROUP_SPLIT_MAX_COLS") ? atoll(getenv("MI355X_GROUP_SPLIT_MAX_COLS")) : 512;
    return v;
}
size_t graph_gemm_partial_need(const ggml_cgraph * g) {
    size_t need = 0;
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        if (n->op != GGML_OP_MUL_MAT || is_empty(n)) continue;
        if (mm_takes_gemm_any(n) && n->src[0]->type == GGML_TYPE_F32 && n->src[1]->type == GGML_TYPE_F32) {      // small f32 x f32 products may split K over workgroups (gemm_any.hip)
            const int nbatch = (int) (n->src[1]->ne[2] * n->src[1]->ne[3]);
            size_t b = gemm_any_split_scratch_bytes(n->src[0]->ne[1], n->src[1]->ne[1], n->src[0]->ne[0], nbatch, true);
            if (n->src[0]->op == GGML_OP_RESHAPE && n->src[0]->src[0] && n->src[0]->src[0]->op == GGML_OP_IM2COL) {      // a streaming causal convolution's per-batch-element product: run with the roles swapped, both batch elements in one launch (exec_causal_conv)
                const size_t b2 = gemm_any_split_scratch_bytes(n->src[1]->ne[1], n->src[0]->ne[1], n->src[0]->ne[0], 2, true);
                if (b2 > b) b = b2;
            }
            if (b > need) need = b;
        }
        if (n->src[1]->ne[2] != 1 || n->src[1]->ne[3] != 1) continue;
        if (!mm_uses_gemm(n)) {                              // the split form of op_mul_mat (F16 weights, K a few columns past a multiple of 64): its MFMA part is a lone, usually under-filled GEMM
            const int64_t K = n->src[0]->ne[0];
            if (n->src[0]->type == GGML_TYPE_F16 && n->src[1]->type == GGML_TYPE_F32 && K % 64 != 0 && K >= 512 && n->src[1]->ne[1] > MI_MMVQ_MAX_COLS) {
                const size_t b = gemm_split_scratch_bytes(n->src[0]->ne[1], n->src[1]->ne[1], K - K % 64);
                if (b > need) need = b;
            }
            continue;
        }
        int64_t m_sum = n->src[0]->ne[1];                    // the mat-muls that share this activation may go out as one launch (exec_gemm_group)
        if (n->src[1]->ne[1] <= gemm_group_split_max_cols()) {
            int grouped = 1;
            for (int j = i + 1; j < g->n_nodes && j < i + 32 && grouped < 3; ++j) {
                const ggml_tensor * c = g->nodes[j];
                if (c->op == GGML_OP_MUL_MAT && !is_empty(c) && c->src[1]->data == n->src[1]->data && c->src[1]->ne[0] == n->src[1]->ne[0] && c->src[1]->ne[1] == n->src[1]->ne[1]) { m_sum += c->src[0]->ne[1]; ++grouped; }
            }
        }
        size_t b = gemm_split_scratch_bytes(m_sum, n->src[1]->ne[1], n->src[0]->ne[0]);
        if (mm_uses_mmq_tile(n)) {                           // (its own split rule: one workgroup per CU)
            const size_t b1 = mmq_tile_split_scratch_bytes(n->src[0]->ne[1], n->src[1]->ne[1], n->src[0]->ne[0]), b2 = mmq_tile_split_scratch_bytes(m_sum, n->src[1]->ne[1], n->src[0]->ne[0]);
            if (b1 > b) b = b1;
            if (b2 > b) b = b2;
        }
        if (b > need) need = b;
    }
    return need;
}
void drop_graph_execs(backend_ctx * c) {                   // captured graphs bake pointers / fusion decisions in: destroy, do not just forget
    for (auto & e : c->execs) { if (e.exec) (void) hipGraphExecDestroy(e.exec); if (e.graph) (void) hipGraphDestroy(e.graph); }
    c->execs.clear();
}
bool ensure_scratch(backend_ctx * c, void ** p, size_t * have, size_t need) {
    if (need <= *have) return true;
    HIP_CHECK(hipStreamSynchronize(c->stream));             // nothing in flight may still read the old block
    if (*p) HIP_CHECK(hipFree(*p));
    *p = nullptr; *have = 0;
    size_t n = need + need / 4; n = (n + ((size_t) 1 << 20) - 1) & ~(((size_t) 1 << 20) - 1);
    if (hipMalloc(p, n) != hipSuccess) {                    // the resident F16 weight images are the first thing to give back
        (void) hipGetLastError();
        shadow_drop_all(c->device, c->shadow_hold);
        if (hipMalloc(p, n) != hipSuccess) { (void) hipGetLastError(); *p = nullptr; drop_graph_execs(c); return false; }
    }
    *have = n;
    drop_graph_execs(c);                                    // captured graphs baked the old pointer in
    return true;
}


} // namespace mi

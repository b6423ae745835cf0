// ggml_util.hpp -- the handful of ggml-base helpers the backend needs, restated so the shared object has
// no hard link-time dependency on libggml-base (reference: ggml/src/ggml.c type-traits table :600-900,
// ggml_nbytes :1180, ggml_is_contiguous_n :1340).  Only the types the backend can name are tabulated;
// anything else reports size 0 and is rejected by supports_op.
#pragma once
#include "ggml_abi.h"
#include <string.h>

namespace mi {

struct type_info { int blck; int size; const char * name; };

static inline type_info type_traits(int t) {
    switch (t) {
        case GGML_TYPE_F32:  return {1, 4, "f32"};
        case GGML_TYPE_F16:  return {1, 2, "f16"};
        case GGML_TYPE_BF16: return {1, 2, "bf16"};
        case GGML_TYPE_Q4_0: return {32, 18, "q4_0"};
        case GGML_TYPE_Q4_1: return {32, 20, "q4_1"};
        case GGML_TYPE_Q5_0: return {32, 22, "q5_0"};
        case GGML_TYPE_Q5_1: return {32, 24, "q5_1"};
        case GGML_TYPE_Q8_0: return {32, 34, "q8_0"};
        case GGML_TYPE_Q8_1: return {32, 36, "q8_1"};
        case GGML_TYPE_Q2_K: return {256, 84, "q2_K"};
        case GGML_TYPE_Q3_K: return {256, 110, "q3_K"};
        case GGML_TYPE_Q4_K: return {256, 144, "q4_K"};
        case GGML_TYPE_Q5_K: return {256, 176, "q5_K"};
        case GGML_TYPE_Q6_K: return {256, 210, "q6_K"};
        case GGML_TYPE_Q8_K: return {256, 292, "q8_K"};
        case GGML_TYPE_I8:   return {1, 1, "i8"};
        case GGML_TYPE_I16:  return {1, 2, "i16"};
        case GGML_TYPE_I32:  return {1, 4, "i32"};
        case GGML_TYPE_I64:  return {1, 8, "i64"};
        case GGML_TYPE_F64:  return {1, 8, "f64"};
        default:             return {0, 0, "?"};
    }
}
static inline size_t  type_size(int t) { return (size_t) type_traits(t).size; }
static inline int64_t blck_size(int t) { return type_traits(t).blck; }
static inline bool    is_quantized(int t) { return type_traits(t).blck > 1; }
static inline size_t  row_size(int t, int64_t ne) { const type_info i = type_traits(t); return i.blck ? (size_t) (ne / i.blck) * i.size : 0; }

static inline int64_t nelements(const ggml_tensor * t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
static inline int64_t nrows(const ggml_tensor * t) { return t->ne[1] * t->ne[2] * t->ne[3]; }
static inline bool    is_empty(const ggml_tensor * t) { return nelements(t) == 0; }

static inline size_t nbytes(const ggml_tensor * t) {
    if (is_empty(t)) return 0;
    const type_info ti = type_traits(t->type);
    size_t n;
    if (ti.blck == 1) {
        n = ti.size;
        for (int i = 0; i < GGML_MAX_DIMS; ++i) n += (t->ne[i] - 1) * t->nb[i];
    } else {
        n = t->ne[0] * t->nb[0] / ti.blck;
        for (int i = 1; i < GGML_MAX_DIMS; ++i) n += (t->ne[i] - 1) * t->nb[i];
    }
    return n;
}

// contiguous from dimension n upward (n = 0: fully contiguous; n = 1: rows may be strided)
static inline bool is_contiguous_n(const ggml_tensor * t, int n) {
    const type_info ti = type_traits(t->type);
    if (!ti.blck) return false;
    size_t next_nb = ti.size;
    if (t->ne[0] != ti.blck && t->nb[0] != next_nb) return false;
    next_nb *= t->ne[0] / ti.blck;
    for (int i = 1; i < GGML_MAX_DIMS; ++i) {
        if (t->ne[i] != 1) {
            if (i > n) { if (t->nb[i] != next_nb) return false; next_nb *= t->ne[i]; }
            else       { next_nb = t->ne[i] * t->nb[i]; }
        }
    }
    return true;
}
static inline bool is_contiguous(const ggml_tensor * t)   { return is_contiguous_n(t, 0); }
static inline bool is_contiguous_1(const ggml_tensor * t) { return is_contiguous_n(t, 1); }
static inline bool same_shape(const ggml_tensor * a, const ggml_tensor * b) {
    return a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3];
}
static inline bool can_repeat(const ggml_tensor * small, const ggml_tensor * big) {   // ggml_can_repeat
    if (is_empty(small)) return is_empty(big);
    for (int i = 0; i < 4; ++i) if (big->ne[i] % small->ne[i] != 0) return false;
    return true;
}
static inline int32_t op_param_i32(const ggml_tensor * t, int i) { return t->op_params[i]; }
static inline float   op_param_f32(const ggml_tensor * t, int i) { float f; memcpy(&f, &t->op_params[i], 4); return f; }

static inline const char * op_name(int op) {
    switch (op) {
        case GGML_OP_NONE: return "NONE"; case GGML_OP_DUP: return "DUP"; case GGML_OP_ADD: return "ADD"; case GGML_OP_SUB: return "SUB";
        case GGML_OP_MUL: return "MUL"; case GGML_OP_DIV: return "DIV"; case GGML_OP_RMS_NORM: return "RMS_NORM"; case GGML_OP_MUL_MAT: return "MUL_MAT";
        case GGML_OP_SCALE: return "SCALE"; case GGML_OP_CPY: return "CPY"; case GGML_OP_CONT: return "CONT"; case GGML_OP_RESHAPE: return "RESHAPE";
        case GGML_OP_VIEW: return "VIEW"; case GGML_OP_PERMUTE: return "PERMUTE"; case GGML_OP_TRANSPOSE: return "TRANSPOSE";
        case GGML_OP_GET_ROWS: return "GET_ROWS"; case GGML_OP_SET_ROWS: return "SET_ROWS"; case GGML_OP_SOFT_MAX: return "SOFT_MAX";
        case GGML_OP_ROPE: return "ROPE"; case GGML_OP_FLASH_ATTN_EXT: return "FLASH_ATTN_EXT"; case GGML_OP_UNARY: return "UNARY"; case GGML_OP_GLU: return "GLU";
        default: return "OP?";
    }
}

} // namespace mi

// graph_exec_internal.hpp -- what the translation units of the executor share (round 6: graph_exec.cpp split by module): graph_exec.cpp = activation images,
// the MUL_MAT executor, deferral bookkeeping, compute_node and run_nodes; graph_exec_llm.cpp = the text-decoder matchers (grouped GEMMs, norm / rope chains and their
// hand-over to the attention launches, split-K folds); graph_exec_t2w.cpp = the encoder / Token2Wav matchers (f32 attention chain, element-wise chains, lazy copies,
// streaming convolutions, modulated norms).  Pure movement: no function body changed.
#pragma once
#include "graph_internal.hpp"
#include <map>

namespace mi {


// ------------------------------------------------------------------------------------------------ MUL_MAT
struct byte_range { const char * lo; const char * hi; };
// out / bias: the ADD of a [M] row vector behind the mat-mul, folded into the any-shape GEMM's epilogue (exec_mul_mat decides; only that path takes them)
// sib / nsib: up to two more F32-weight mat-muls over the same activation (same weight shape and strides) for the launch; *sib_taken tells whether they went along
struct mm_sibling { const ggml_tensor * w; const ggml_tensor * out; const float * bias; };

// RMS_NORM(j) -> MUL(w[D]) -> ROPE [-> SET_ROWS of the rotated rows viewed as [D*H, T] into an f16 table]; shape checks only
struct nr_chain {
    int norm, mul, rope, store;                    // norm / mul = -1: a ROPE-only chain (llama architecture: no q / k norm)
    const ggml_tensor * wt, * pos, * ff;           // wt = null: no norm
    const ggml_tensor * xin; int first;            // the f32 heads the chain starts from, and the chain's first node
    int D, H, T; float eps; rope_params rp;
};

// LayerNorm -> MUL(n, scale) -> ADD(n, .) -> ADD(., shift) with scale / shift one row per dim-2 slice ([C, 1, B] views of the DiT's adaLN product, token2wav-impl.cpp:1121-1164):
// the three element-wise nodes ride in the norm launch's epilogue, rounded as they round.  The norm has exactly these two readers.
struct norm_mod_match { int mi_, a1i, a2i; const ggml_tensor * sv, * tv, * out; };

static inline byte_range range_of(const ggml_tensor * t) { const char * p = (const char *) t->data; return { p, p + nbytes(t) }; }
static inline bool overlap(byte_range a, byte_range b) { return a.lo < b.hi && b.lo < a.hi && a.lo != a.hi && b.lo != b.hi; }

static inline bool is_kquant(int t) { return t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K; }   // the formats with integer-dot kernels on Q8_K activations
// Is t read by somebody this executor does not see?  Graph outputs, and -- when the scheduler cut the graph into splits -- tensors whose
// whole-graph use count (ggml_cgraph::use_counts, shared by the split views: ggml_graph_view) exceeds the uses inside this split: a later
// split (on this or another backend) reads them, so their f32 value must be written and no fusion may swallow them (cf. ggml_can_fuse).
static inline bool is_out(exec_state & s, const ggml_tensor * t) { return (t->flags & GGML_TENSOR_FLAG_OUTPUT) || s.external.count(t) != 0; }

static inline size_t attn_sm_mask16_off(int64_t nq, int64_t nkv) { return (fattn_map_bytes_host(nq, nkv) + 255) & ~(size_t) 255; }

static inline tdesc swapped01(tdesc d) { std::swap(d.ne[0], d.ne[1]); std::swap(d.nb[0], d.nb[1]); return d; }

// deferred copies (graph_exec.cpp): queue `nj` jobs of ONE node (they may write interleaved parts of one tensor: not checked against each other) / launch what is pending
bool copy_queue_on(exec_state & s);
void copy_queue(exec_state & s, const copy_pair * jobs, int nj, const tdesc & Y, const ggml_tensor * node, int dim1 = -1);      // Y: the node's output; dim1 >= 0: job 1's box starts at jobs[0].dst.ne[dim1] along that dimension
void copy_flush(exec_state & s);
void materialise_norm(exec_state & s);
void lazy_net(exec_state & s, int i);
void lazy_materialise(exec_state & s, const ggml_tensor * t, int reader_op = -1);
byte_range range_of(const tdesc & d);
size_t prepare_act(exec_state & s, const ggml_tensor * x, act_kind kind);
const char * mmv_class(int type);
const uint16_t * weight_shadow(exec_state & s, const ggml_tensor * w, const char * wp, int64_t K, int64_t M);
bool mm_takes_gemm_any(const ggml_tensor * n);
void op_mul_mat(exec_state & s, const ggml_tensor * dst, const ggml_tensor * out = nullptr, const float * bias = nullptr, const mm_sibling * sib = nullptr, int nsib = 0, bool * sib_taken = nullptr, int act = 0);
bool plain_kq_matvec(const ggml_tensor * n, int max_cols);
bool kq_mm_ok(const ggml_tensor * n);
bool q80_mv1_node(exec_state & s, const ggml_tensor * n);
bool mv1_node_ok(exec_state & s, const ggml_tensor * n);
bool norm_in_kernel(exec_state & s, const ggml_tensor * x, const ggml_tensor * const * outs, int n_outs, int n_consumers, mmv_norm & nr);
void gs_materialise(exec_state & s);
void mv1_source(exec_state & s, const ggml_tensor * x, const ggml_tensor * const * outs, int n_outs, int n_consumers, mv1_args & v);
bool same_act(const ggml_tensor * a, const ggml_tensor * b);
int n_users(exec_state & s, const ggml_tensor * t);
int sole_user(exec_state & s, const ggml_tensor * t);
int next_real_node(exec_state & s, int i);
bool ready_before(exec_state & s, const ggml_tensor * src, int i, const int * item, int n_item);
bool can_hoist(exec_state & s, int i, int j, const int * item, int n_item);
void note_write(exec_state & s, const ggml_tensor * t);
void materialise_reduce(exec_state & s);
void materialise_group(exec_state & s, int skip_mask = 0);
bool reads_pending_group(exec_state & s, const ggml_tensor * n);
bool kq_in_staging(exec_state & s, const ggml_tensor * w, int64_t N);
bool gemm_operand(exec_state & s, const ggml_tensor * w, const uint16_t ** w16, size_t * rs);
bool gemm_groupable(const ggml_tensor * c);
bool gemm_only_consumers(exec_state & s, const ggml_tensor * t, int64_t K, int64_t N, const ggml_tensor ** x_out);
void seed_act_f16(exec_state & s, const ggml_tensor * x, bool quantised = false);
bool exec_attn_sm_prefill(exec_state & s, int i, bool dry);
bool exec_gemm_group(exec_state & s, int i);
void exec_mul_mat(exec_state & s, int i);
bool match_norm_rope(exec_state & s, int j, nr_chain & c);
norm_rope_job chain_job(exec_state & s, const nr_chain & c);
act_kind consumers_act_kind(exec_state & s, const ggml_tensor * x);
bool match_norm_modulate(exec_state & s, int i, norm_mod_match & M, bool consecutive);
bool exec_norm_modulate(exec_state & s, int i);
bool exec_gate_norm(exec_state & s, int i);
bool exec_norm(exec_state & s, int i);
bool try_defer_qkv_to_attention(exec_state & s, const nr_chain & A, const nr_chain * B, int vj, const int * item, int ni);
bool try_defer_qkv_to_softmax_attention(exec_state & s, const nr_chain & A, const nr_chain * B, int vsj, const int * item, int ni);
bool match_rope_only(exec_state & s, int j, nr_chain & c);
bool exec_rope_chain(exec_state & s, int i);
bool exec_rms_norm(exec_state & s, int i);
bool exec_attn_f32(exec_state & s, int i);
void materialise_vt(exec_state & s);
bool try_alias_vt(exec_state & s, int i);
void compute_node(exec_state & s, int i);
int cont_sink(exec_state & s, int i);
int exec_ew_chain(exec_state & s, int i, int * taken);
bool lazy_try_register(exec_state & s, int i);
bool exec_causal_conv(exec_state & s, int i);
bool exec_concat_tail(exec_state & s, int i);
bool exec_conv1d_tc(exec_state & s, int i);
void run_nodes(exec_state & s, ggml_cgraph * g);

} // namespace mi

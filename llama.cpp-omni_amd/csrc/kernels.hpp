// kernels.hpp -- host-callable launchers of the hand-written gfx950 kernels.
// All pointers are device pointers; every launcher enqueues on `st` and returns immediately.
#pragma once
#include "common.hpp"

namespace mi {

// ---- activation quantisers (reference: ggml-cpu.c:1272-1306 `from_float` stage of mul_mat)
// x: [nrows, K] f32 with row stride xs (bytes). img: nrows images of q8k_image_bytes(K) each.
void quantize_q8k_image(const float * x, size_t xs, void * img, int64_t K, int64_t nrows, hipStream_t st);
void quantize_q80_image(const float * x, size_t xs, void * img, int64_t K, int64_t nrows, hipStream_t st);
// f32 -> f16 (RNE) rows, dst row stride ys bytes
void convert_f32_f16_rows(const float * x, size_t xs, uint16_t * y, size_t ys, int64_t K, int64_t nrows, hipStream_t st);
// f16 rows of the Q8_K-quantised values (d * q, what the reference's integer dot products multiply K-quant weights with): K % 256 == 0; in place for rows a launch left as f16
void convert_f32_f16q_rows(const float * x, size_t xs, uint16_t * y, size_t ys, int64_t K, int64_t nrows, hipStream_t st);
void requant_f16_rows_q8k(uint16_t * y, size_t ys, int64_t K, int64_t nrows, hipStream_t st);
void convert_f32_f16_rows3(const float * x, size_t nb1, size_t nb2, size_t nb3, int64_t n1, int64_t n2, int64_t n3, uint16_t * y, size_t ys, int64_t K, hipStream_t st);

// ---- mat-vec on quantised weights: dst[col*ds + row] = dot(W[row, :], act[col, :]), ncols <= MMVQ_MAX_COLS
#define MI_MMVQ_MAX_COLS 8
struct mmv_args {
    const void * W;        // weight rows
    size_t       w_rs;     // weight row stride in bytes (nb01)
    const void * act;      // activation images (q8k / q80 / f16 rows), one per column
    size_t       act_cs;   // activation column stride in bytes
    float *      dst;      // f32 output
    size_t       dst_cs;   // output column stride in bytes (nb1)
    int64_t      K;        // contraction length
    int64_t      nrows;    // weight rows (= dst ne0)
    int          ncols;    // activation columns (tokens)
    // broadcast batch (mmv_f16 / mmv_f32 only): nbatch = ne12 * ne13 matrices in one launch, batch b = i13 * ne12 + i12 reads
    // W + (i12 / r2) * w_nb2 + (i13 / r3) * w_nb3, act + b * act_bs, writes dst + i12 * dst_nb2 + i13 * dst_nb3
    int          nbatch = 1, ne12 = 1, r2 = 1, r3 = 1;
    size_t       w_nb2 = 0, w_nb3 = 0, act_bs = 0, dst_nb2 = 0, dst_nb3 = 0;
};
void mmv_q4_K(const mmv_args & a, hipStream_t st);
void mmv_q6_K(const mmv_args & a, hipStream_t st);
void mmv_q5_K(const mmv_args & a, hipStream_t st);
void mmv_q8_0(const mmv_args & a, hipStream_t st);
void mmv_q4_0(const mmv_args & a, hipStream_t st);   // Q4_0 / Q5_0 weights x Q8_0 activation images (the reference's vec_dot_q4_0_q8_0 / _q5_0_q8_0 integers)
void mmv_q5_0(const mmv_args & a, hipStream_t st);
void mmv_f16 (const mmv_args & a, hipStream_t st);   // act = f16 rows
void mmv_f32 (const mmv_args & a, hipStream_t st);   // W f32, act = f32 rows

// ---- fused K-quant launches (decode): up to 3 matrices sharing one Q8_K activation image set in one launch
// (wq/wk/wv), Q4_K / Q6_K mixed, optional residual add epilogue (dst = W.x + resid)
struct mmv_mat {
    const void * W; size_t w_rs;
    float * dst;    size_t dst_cs;
    const float * resid; size_t resid_cs;      // may be null
    int64_t nrows; int type;
};
// norm.x != null: the activation is rms_norm(x) * w of these f32 rows (column stride x_cs); every workgroup builds the Q8_K image
// itself (mmvk.hip), `act` is ignored
bool mmv_norm_ok(int64_t K, int ncols);        // shapes the in-kernel norm supports (one column, K <= 4096)
struct mmv_norm { const float * x = nullptr; size_t x_cs = 0; const float * w = nullptr; float eps = 0.0f; };
struct mmv_multi_args {
    mmv_norm norm;
    mmv_mat m[3]; int nmat;
    const void * act; size_t act_cs;           // Q8_K images, one per column
    int64_t K; int ncols;                      // ncols * q8k_image_bytes(K) must fit LDS (<= 152 KiB)
};
void mmv_kquant_multi(const mmv_multi_args & a, hipStream_t st);
// ffn_gate + ffn_up + SWIGLU: dst[col][r] = silu(Wg[r].x) * (Wu[r].x); both matrices `type`, same shape; ncols <= 8
void mmv_kquant_pair_swiglu(int type, const void * Wg, const void * Wu, size_t w_rs, const void * act, size_t act_cs, float * dst, size_t dst_cs,
                            int64_t K, int64_t nrows, int ncols, hipStream_t st, const mmv_norm * norm = nullptr);

// ---- batch-1 decode form (mmv1.hip): ONE activation column, Q4_K / Q6_K, K a multiple of 256 up to 16384 (whole steps of 4096 at the Qwen3-8B widths, TAIL instances otherwise).  The launch takes the f32
// activation row itself -- x, or rms_norm(x) * norm_w (the RMS_NORM + MUL nodes in front of src1) -- and every workgroup builds the Q8_K
// image in its prologue, so no norm / quantise launch precedes it; img != null hands over a ready image (q8k_image_bytes layout) instead.
// W_up != null: m[0] is ffn_gate, W_up ffn_up (same type / shape / stride), dst = silu(gate.x) * (up.x).  dst / resid strides unused.
struct mv1_args {
    mmv_mat m[3]; int nmat = 0; const void * W_up = nullptr;
    const float * x = nullptr; const float * norm_w = nullptr; float eps = 0.0f; const void * img = nullptr;
    int64_t K = 0;
    // the activation row as attention slices' partial states (fattn_gs_parts_bytes: nslice x [K] partial outputs + nslice x [K / 128] x (M, S)), folded in the launch's
    // prologue -- the LDS-DMA engine only (mmv2_ok): one K-quant matrix, K = 4096, no norm
    const float * parts = nullptr; int nslice = 0;
};
bool mmv1_ok(const mv1_args & a);                 // Q4_K / Q6_K (mmv1.hip) or Q8_0 (mmv1q.hip: K % 32 == 0, K <= 4096; img = Q8_0 image)
void mmv1(const mv1_args & a, hipStream_t st);
// the loader / consumer LDS-DMA engine (mmv2.hip): K = 4096 / 12288, Q4_K / Q6_K, 16-byte aligned rows; mmv1() routes to it when it applies
// (MI355X_MV2=0 or mmv2_enable(false) keeps the register-load kernels)
bool mmv2_ok(const mv1_args & a);
void mmv2(const mv1_args & a, hipStream_t st);
void mmv2_enable(bool on);
bool mmv2_enabled();                                // (mmv1() routes to the engine)
bool mmv1q_ok(const mv1_args & a);
void mmv1q(const mv1_args & a, hipStream_t st);

// ---- K-quant weights against 2 .. 32 activation columns on the int8 matrix cores (mmq.hip): up to 3 matrices (Q4_K / Q6_K mixed)
// sharing one set of Q8_K activation images; dst[col*dst_cs + row], optional residual add epilogue
struct mmq_mat { const void * W; size_t w_rs; float * dst; size_t dst_cs; int64_t nrows; int type; const float * resid = nullptr; size_t resid_cs = 0; };   // resid: optional dst = W.x + resid
struct mmq_args { mmq_mat m[3]; int nmat; const void * act; size_t act_cs; int64_t K; int ncols; };
bool mmq_ok(int type, int64_t K, const void * W, size_t w_rs);
void mmq_kquant(const mmq_args & a, hipStream_t st);
// ---- Q4_K weights against a prefill ubatch (> 64 columns) on the int8 matrix cores, tiled (mmq_tile.hip): up to 3 matrices sharing the block-major
// Q8_K image of the activations (quantize_q8k_tile_image); `resid`: dst = W.x + resid (single matrix, un-split); split-K slabs / deferred reductions as gemm_f16_multi
struct mmqt_mat { const void * W; size_t w_rs; float * dst; size_t dst_cs; int64_t M; const float * resid = nullptr; size_t resid_cs = 0; };
struct mmqt_args {
    mmqt_mat m[3]; int nmat = 0; const void * img = nullptr; int64_t N = 0, K = 0;
    float * partial = nullptr; size_t partial_bytes = 0; int * deferred_split = nullptr; bool defer_multi = false;
};
size_t mmqt_image_bytes(int64_t K, int64_t N);
void   quantize_q8k_tile_image(const float * x, size_t xs, void * img, int64_t K, int64_t N, hipStream_t st);
bool   mmq_tile_ok(int type, int64_t K, const void * W, size_t w_rs);
void   mmq_tile(const mmqt_args & a, hipStream_t st);
size_t mmq_tile_split_scratch_bytes(int64_t m_sum, int64_t N, int64_t K);
long   mmq_tile_launches();

// RMS_NORM + MUL(w) + Q8_K image of the result in one launch (one workgroup per row); y may be null when only the
// image is consumed.  Same arithmetic as rms_norm() followed by quantize_q8k_image().
void rms_norm_mul_quant(const float * x, size_t xs, const float * w, float * y, size_t ys, void * img, int64_t n, int64_t nrows, float eps, hipStream_t st);

// ---- full dequantisation (GET_ROWS on quantised tables, dequant->GEMM prefill path, CPY q->f32)
// src row r at src + r*src_rs ; dst row r at dst + r*dst_rs ; K elements per row
void dequant_rows_f32(int type, const void * src, size_t src_rs, float * dst, size_t dst_rs, int64_t K, int64_t nrows, hipStream_t st);
void dequant_rows_f16(int type, const void * src, size_t src_rs, uint16_t * dst, size_t dst_rs, int64_t K, int64_t nrows, hipStream_t st);

// ---- tensor descriptor for the strided element-wise kernels
struct tdesc {
    void *  p;
    int64_t ne[4];
    size_t  nb[4];
};

// RMS_NORM (ops.cpp:3517-3566), optionally fused with the following MUL by `w` (broadcast over rows) and ADD
// y16 != null (2-D, with mul_w): also emit f16-rounded rows (row stride y16_rs) for the prefill GEMM; write_f32 = false skips y
void rms_norm(const tdesc & x, const tdesc & y, float eps, const tdesc * mul_w, hipStream_t st, uint16_t * y16 = nullptr, size_t y16_rs = 0, bool write_f32 = true, bool y16_q8 = false);   // y16_q8: as gemm_reduce_rms_norm (returns false semantics: aborts when the wave-per-row kernel cannot take the shape)
// IM2COL (ops.cpp:6150-6301): x f32 [IW, IH, IC, N] (2-D) or [IW, IC, N] (1-D) -> y f16 / f32 [IC*KH*KW, OW, OH, N]; p = op_params (s0,s1,p0,p1,d0,d1,is_2D)
void im2col_f32(const tdesc & kernel, const tdesc & x, const tdesc & y, int y_type, const int32_t * p, hipStream_t st);
// POOL_2D (ops.cpp:7281-7355) / POOL_1D with k == s, p == 0 (ops.cpp:7212-7260): avg / max windows of f32 / f16 planes -> f32; p = op_params
void pool_f32(const tdesc & x, int x_type, const tdesc & y, const int32_t * p, bool is_2d, hipStream_t st);
// NORM (ops.cpp:3450-3495): y = (x - mean) / sqrt(var + eps) per row
void norm_f32(const tdesc & x, const tdesc & y, float eps, hipStream_t st);
bool norm_rows_ok(const tdesc & x, const tdesc & y);                      // many 16-byte aligned rows of at most 4096 elements: the wave-per-row kernel
struct norm_gate { const float * y, * resid, * gate; size_t g_bs; };      // x = resid + y * gate computed (and written to x) in front of the norm: rows of y / resid laid out like x, gate one row per dim-2 slice
void norm_rows_f32(const tdesc & x, const tdesc & y, float eps, const float * w, const float * b, uint16_t * y16, size_t y16_rs, bool write_f32, hipStream_t st, size_t w_bs = 0, size_t b_bs = 0, bool mod = false,
                   const norm_gate * gate = nullptr);   // + MUL w, ADD b, f16 image
long norm_from_split_launches();
bool norm_rows_from_split_ok(const tdesc & x, const tdesc & y, int nsplit, size_t resid_cs, size_t resid2_cs, const void * resid, const void * resid2, const void * part);
void norm_rows_from_split(const tdesc & x, const tdesc & y, float eps, const float * w, const float * b, uint16_t * y16, size_t y16_rs, bool write_f32,
                          const float * part, int nsplit, size_t split_elems, const float * resid, size_t resid_cs, const float * resid2, size_t resid2_cs, hipStream_t st);   // x = split-K slabs + addends, written, then the LayerNorm
// ROPE f32 (ops.cpp:5534-5720): modes NORMAL / NEOX, optional freq factors, YaRN
struct rope_params {
    int   n_dims, mode, n_ctx_orig;
    float freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
};
void rope_f32(const tdesc & x, const int32_t * pos, const float * freq_factors, const tdesc & y, const rope_params & rp, hipStream_t st);
// SOFT_MAX (ops.cpp:5072-5182): y = softmax(x*scale + slope*mask)
void soft_max_f32(const tdesc & x, const tdesc * mask, int mask_type, const float * sinks, const tdesc & y, float scale, float max_bias, hipStream_t st,
                  uint16_t * y16 = nullptr, size_t y16_rs = 0, bool write_f32 = true);      // y16: also / only the f16 rows (the next MUL_MAT's activation image); needs soft_max_rows_ok
bool soft_max_rows_ok(const tdesc & x, const tdesc * mask, int mask_type, const float * sinks, const tdesc & y);
// GLU (split or single-tensor forms; ops.cpp:2934-2990 for swiglu)
void glu_f32(int glu_op, const tdesc & a, const tdesc * b, bool swapped, const tdesc & y, hipStream_t st, uint16_t * y16 = nullptr, size_t y16_rs = 0, bool write_f32 = true);
// SWIGLU (split form) straight into the Q8_K activation images of its rows (+ optionally the f32 result): the GLU of an FFN whose
// down projection is a K-quant mat-mul at several columns
bool swiglu_q8k_ok(const tdesc & a, const tdesc & b, const tdesc & y);
void swiglu_q8k(const tdesc & a, const tdesc & b, const tdesc & y, bool write_f32, void * img, hipStream_t st);
// unary ops on contiguous f32
void unary_f32(int uop, const float * x, float * y, int64_t n, hipStream_t st, uint16_t * y16 = nullptr, bool write_f32 = true);   // y16: dense f16 copy of the result (n % 4 == 0, aligned)
// ADD / SUB / MUL / DIV with ggml broadcast semantics (src1 repeats over src0)
void bin_bcast_f32(int op, const tdesc & a, const tdesc & b, const tdesc & y, hipStream_t st);
void scale_f32(const float * x, float * y, int64_t n, float s, float b, hipStream_t st);
// ---- ops of the Token2Wav graphs (kernels/t2w_ops.hip; reference tools/omni/token2wav/token2wav-impl.cpp)
// SQR / SQRT / LOG / SIN / COS / CLAMP(p0 = min, p1 = max) / LEAKY_RELU(p0 = slope) on dense f32
void math_f32(int op, const float * x, float * y, int64_t n, float p0, float p1, hipStream_t st);
void conv1d_weight_t(const float * w, float * y, int KK, int Cout, hipStream_t st);               // [Cout][KK] -> [KK][Cout] (exec_conv1d_tc)
void conv1d_tc(const float * x, const float * wt, const float * bias, float * y, int T, int OW, int Cin, int Cout, int KW, int dil, int pad, hipStream_t st);      // y[t + OW co] = bias[co] + sum_{c,k} w[k, c, co] x[t + k dil - pad + T c]
void conv1d_weight_rows(const float * w, float * y, int KW, int C, int Cout, hipStream_t st);     // [KW, C, Cout] -> [Cout][KW][C] (exec_causal_conv)
void concat(const tdesc & a, const tdesc & b, const tdesc & y, int dim, int elem_size, hipStream_t st);            // ops.cpp:1968-2009
void repeat(const tdesc & x, const tdesc & y, int elem_size, hipStream_t st);                                      // ops.cpp:1637-1679
void pad_f32(const tdesc & x, const tdesc & y, const int32_t * p, hipStream_t st);                                 // ops.cpp:7592-7638; p = {lp0, rp0, ..., lp3, rp3}
void pad_reflect_1d_f32(const tdesc & x, const tdesc & y, int p0, int p1, hipStream_t st);                         // ops.cpp:7660-7691
void arange_f32(float * y, int64_t n, float start, float step, hipStream_t st);                                    // ops.cpp:7762-7783
void timestep_embedding_f32(const float * ts, const tdesc & y, int64_t n, int dim, int max_period, hipStream_t st);  // ops.cpp:7800-7831
void sum_rows_f32(const tdesc & x, const tdesc & y, hipStream_t st);                                               // ops.cpp:1399-1430
void conv_transpose_1d_f32(const tdesc & w, int w_type, const tdesc & x, const tdesc & y, int s0, hipStream_t st); // ops.cpp:5952-6122
void cast_f32_i32(const tdesc & src, bool src_is_f32, const tdesc & dst, hipStream_t st);                          // ops.cpp:555, 558-561
// CPY / CONT / DUP between f32 / f16 with arbitrary strides (same element count)
void cpy_strided(const tdesc & src, int src_type, const tdesc & dst, int dst_type, hipStream_t st);
// up to COPY_BATCH_MAX same-type strided copies (element size 2 or 4 bytes, < 2^31 elements each, shapes of equal element count) as one launch; the CALLER guarantees that
// no job reads or writes bytes another job of the batch writes (graph_exec.cpp copy_queue)
#define COPY_BATCH_MAX 32
struct copy_pair { tdesc src, dst; int es; };
bool copy_batch_ok(const tdesc & src, const tdesc & dst, int es);
void copy_batch(const copy_pair * jobs, int n, hipStream_t st);
// n small copies in one launch: entry i copies ents[i].size bytes from base + ents[i].off to ents[i].dst (base / ents: device-visible pinned memory)
struct upload_ent { void * dst; uint32_t off, size; };
void upload_small(const upload_ent * ents, const char * base, int n, hipStream_t st);
// GET_ROWS (f32 / f16 / quantised tables -> f32), SET_ROWS (f32 -> f32 / f16, i64 or i32 indices)
void get_rows(const tdesc & src, int src_type, const tdesc & idx, const tdesc & dst, hipStream_t st);
void set_rows(const tdesc & src, const tdesc & idx, int idx_type, const tdesc & dst, int dst_type, hipStream_t st, int64_t period = 0);   // period: rows r = t * period + j of single-element rows (a work-order hint)

// decode pre-stage of FLASH_ATTN_EXT: the layer's q chain, k chain + store and v store (what norm_rope_store() does in its own launch)
// executed by the attention kernel itself; one token, one sequence (fattn_pre_ok)
struct fattn_pre {
    const float * qraw; int64_t q_hs; const float * kraw; int64_t k_hs; const float * vraw; int64_t v_hs;   // f32 heads, head stride in bytes
    const float * qw; const float * kw; const int32_t * pos; const float * ff; float eps; rope_params rp;
    void * kcache; int64_t kc_rs; void * vcache; int64_t vc_rs; const void * kidx; const void * vidx; int idx_is64;
};
// FLASH_ATTN_EXT (ops.cpp:7912-8148): q f32 [D, nq, nh, ns], k/v f16 [D, nkv, nhkv, ns], mask f16 [nkv, >=nq, 1|nh?, ns]
struct fattn_args {
    tdesc q, k, v, dst;
    int kv_type = 1;         // GGML type of K and V (F16 = 1: every specialised kernel; F32 / BF16 / Q8_0 / Q4_0: fattn_any.hip)
    const tdesc * mask;      // may be null
    const float * sinks;     // may be null
    float scale, max_bias, logit_softcap;
    void * scratch;          // mask tile map of the prefill kernel (fattn_scratch_bytes(); unused by the decode kernel)
    unsigned * counters = nullptr;   // >= n_head zeroed arrival counters: the sliced one-token kernel merges its slices itself (else: k_fattn_merge)
    size_t scratch_bytes;
    bool   map_valid = false; // scratch already holds the tile map of THIS mask (same tensor used by an earlier node of the graph)
    void * img = nullptr;    // optional: also emit Q8_K images of the output rows [nh*D] (one per (seq, query row))
    const fattn_pre * pre = nullptr;   // optional q/k/v pre-stage (decode)
    uint16_t * out16 = nullptr; size_t out16_rs = 0; bool write_f32 = true;   // prefill kernel: also / only emit f16 rows [nh*D] per (seq, query)
    const float * rope_tab = nullptr;  // (cos, sin) pairs [D/2] of the token (rope_table), required by the one-token kernel (fattn_one_ok)
    // v is V^T: tdesc of the [n_kv, D, HK, ns] tensor (cells contiguous, nb[1] = stride between d rows) -- the flash-attention-OFF graphs' MUL_MAT(v, soft_max(..))
    // operand; only the matrix-core prefill kernel takes it (fattn_sm_prefill_ok)
    bool v_transposed = false;
    // one token over <= 256 rows, head size 128, four query heads per KV head (fattn_gs_ok): leave FGS slices' partial (O, M, S) states here instead of dst -- the wo
    // mat-vec launch folds them in its prologue (mv1_args::parts); fattn_gs_merge() materialises dst from them when it cannot
    float * gs_parts = nullptr;
};
// MUL_MAT(k, q) -> SOFT_MAX_EXT -> MUL_MAT(v^T, p) -> PERMUTE -> CONT for a batch of query rows as ONE flash-attention launch (the prefill kernel with V^T staging): the
// [n_kv, n_q, H] score / probability blocks are never written.  Same roundings as the separate nodes up to the order of the soft-max sums (q and p rounded to f16, f32 sums).
bool   fattn_sm_prefill_ok(const fattn_args & f);
size_t fattn_map_bytes_host(int64_t nq, int64_t nkv);           // bytes of the prefill kernel's mask tile map for one shared [n_kv, n_q] mask
// One decode token through the reference's flash-attention-OFF attention (MUL_MAT(k, q) -> SOFT_MAX_EXT -> MUL_MAT(v^T, p) -> PERMUTE -> CONT on
// the TRANSPOSED v cache) with the q / k / v pre-stage, as ONE launch (fattn_one.hip: k_attn_one_sm)
struct attn_sm_args {
    const fattn_pre * pre = nullptr;       // q chain, k chain + row store, v: vcache = base of the transposed cache, vidx = the scatter indices
    const float * rope_tab = nullptr;
    const void * k = nullptr; int64_t knb1 = 0, knb2 = 0;      // f16 K view [D, n_kv, HK]: cell stride, kv-head stride (bytes)
    const void * v = nullptr; int64_t vnb1 = 0, vnb2 = 0;      // f16 V^T view [n_kv, D, HK]: dim stride, kv-head stride (bytes); cells contiguous
    const void * mask = nullptr; int64_t mnb2 = 0, mne2 = 1;   // f32 mask row(s) [n_kv]
    void * dst = nullptr; int64_t dnb1 = 0;                    // f32 [D, H]: head stride (bytes)
    int64_t vidx_n = 0;
    int D = 0, nkv = 0, n_head = 0, n_head_kv = 0; float scale = 1.0f;
    void * part = nullptr; size_t part_bytes = 0; unsigned * counters = nullptr;      // more than 256 cells: partial (O, M, S) rows [n_head][slices][D + 2] and zeroed per-head arrival counters
};
bool   attn_one_sm_ok(const attn_sm_args & a);
void   attn_one_sm(const attn_sm_args & a, hipStream_t st);
void   fattn_set_one(bool on);             // one-token kernel on (default) / off: the round-1 decode kernels take the shape (cross-check)
int    fattn_one_nsplit(const fattn_args & a);   // 256-row KV slices of the one-token kernel (> 1: partial rows + k_fattn_merge, needs the scratch)
bool   fattn_one_ok(const fattn_args & a);       // one token, one sequence, pre-stage, <= 256 cache rows: the latency-optimised kernel (fattn_one.hip) runs
bool   fattn_gs_ok(const fattn_args & a);        // ... and the group-slice form applies (fattn_one.hip k_fattn_gs)
size_t fattn_gs_parts_bytes(int n_head, int D);
int    fattn_gs_nslice();
void   fattn_set_gs(int m);          // option "fattn_gs": the group-slice one-token attention + fold in wo's prologue: -1 default (on), 0 off, 1 on
long   fattn_gs_launches(); long fattn_gs_far_launches();
void   fattn_gs_merge(const float * parts, float * dst, int n_head, int D, hipStream_t st);
// (cos, sin) * mscale of every (token, rotation pair): tab[T][D/2][2], what ggml_rope_cache_init / rope_yarn give for these positions
void   rope_table(const int32_t * pos, const float * ff, const rope_params & rp, int T, int D, float * tab, hipStream_t st);
size_t fattn_scratch_bytes(const fattn_args & a);
bool   fattn_can_emit_image(const fattn_args & a);
bool   fattn_pre_ok(const fattn_args & a);
void   fattn_set_dma(int m);      // the LDS-DMA ring form of the prefill kernel (head 128, >= 512 workgroups): -1 default (on), 0 off, 1 on
long   fattn_dma_launches();
void   fattn_set_gqa(bool on);   // matrix-core decode kernel on (default) / off: the streaming kernel takes every few-token shape (cross-check)
bool   fattn_uses_mma(const fattn_args & a);      // the matrix-core (prefill) kernel will run: out16 is honoured
void   flash_attn_ext_f16(const fattn_args & a, hipStream_t st);

// RMS_NORM -> MUL(w) -> ROPE [-> SET_ROWS into an f16 table] on a [D, H, T] f32 activation, one launch
// up to three such jobs share a launch (q chain, k chain + store, plain f32 -> f16 v store: w == null)
struct norm_rope_job {
    const float * x; int64_t xnb1, xnb2;     // head / token byte strides
    const float * w;                         // norm weight [D]; null = plain store job, or (rope_only) a ROPE chain without norm
    float * y; int64_t ynb1, ynb2;           // rope output (null when only the store is consumed)
    void * kv; int64_t kv_rs;                // f16 table base and row stride (null when there is no store)
    const void * idx; int idx_is64; int64_t idx_nb0;
    int H;
    int rope_only = 0;                       // w == null and the job still rotates (llama-architecture q / k chains)
    // optional f16 copy of the rope output as the activation image of a per-head MUL_MAT (flash-attention off: K . q): row h * T + t, y16_rs bytes apart
    void * y16 = nullptr; int64_t y16_rs = 0;
    int nsplit = 1; int64_t split_bytes = 0; // > 1: x is slab 0 of the split-K slabs of the mat-mul in front (dense [T][M] blocks, split_bytes apart): the rows are their sum (norm_rope_takes_split)
};
struct norm_rope_args {
    norm_rope_job j[3]; int njobs;
    const int32_t * pos; const float * ff;
    int D, T; float eps; rope_params rp;
    // optional scratch for the ubatch's (cos, sin) table, T * D/2 float pairs; rope_tab_valid: it already holds THIS (pos, rope params)
    float * rope_tab = nullptr; bool rope_tab_valid = false;
};
long norm_rope_split_launches();
bool norm_rope_takes_split(const norm_rope_args & a);                   // norm_rope_store will run the kernel that sums split-K slab sources (nsplit > 1)
void norm_rope_store(const norm_rope_args & a, hipStream_t st);
bool rms_norm_mul_quant_ok(int64_t n);

// dense GEMM on the matrix cores: dst[n, m] (f32) = sum_k W[m,k] (f16) * X[n,k] (f16), f32 accumulate
void gemm_f16_mfma(const uint16_t * W, size_t w_rs, const uint16_t * X, size_t x_rs, float * dst, size_t dst_cs,
                   int64_t M, int64_t N, int64_t K, hipStream_t st);

// grouped form: up to three matrices sharing X in one launch, optional residual epilogue (dst = W.x + resid), and -- for a lone
// under-filled matrix when `partial` scratch (gemm_split_scratch_bytes) is supplied -- deterministic split-K
struct gemm_mat { const uint16_t * W; size_t w_rs; float * dst; size_t dst_cs; int64_t M; const float * resid; size_t resid_cs;
                  int qtype = 0;
                  const float * resid2 = nullptr; size_t resid2_cs = 0;       // a second addend behind the first (an encoder's bias, then the residual stream): (acc + resid) + resid2, each an f32
                  // split-K launches of at most 128 columns only (gemm_f16_small_n_ksplit() > 1): a GELU / GELU_QUICK (GGML_UNARY_OP_*, -1 = none) applied to the reduced value in the
                  // reduction's epilogue, its f16 rows written to y16 (the activation image of the next mat-mul), the f32 rows to dst only when y32
                  int unary = -1; uint16_t * y16 = nullptr; size_t y16_rs = 0; bool y32 = true; size_t y16_ms = 2; };   // (y16 without unary: the plain f16 copy of the rows -- a K / V cache store -- element (m, n) at y16 + m * y16_ms + n * y16_rs bytes)
                  //   // rounding like the two ADD nodes       // qtype != 0 (GGML_TYPE_Q4_K / Q6_K): W points at the block rows, de-quantised inside the GEMM's staging (all matrices of a launch alike; forces 128-row tiles)
struct gemm_multi_args {
    gemm_mat m[3]; int nmat; const uint16_t * X; size_t x_rs; int64_t N, K; float * partial; size_t partial_bytes = (size_t) -1;
    // broadcast batch (nmat == 1, K % 64 == 0): nbatch = ne12 * ne13 products in one launch; batch b = i13 * ne12 + i12 reads
    // W + (i12 / r2) * w_nb2 + (i13 / r3) * w_nb3 and X + b * x_bs, writes dst + i12 * dst_nb2 + i13 * dst_nb3
    int nbatch = 1, ne12 = 1, r2 = 1, r3 = 1; size_t w_nb2 = 0, w_nb3 = 0, x_bs = 0, dst_nb2 = 0, dst_nb3 = 0;
    // non-null: when a split-K is chosen the reduction is NOT run; *deferred_split = number of [N][M] slabs left in `partial` (0: none,
    // dst is complete) and the caller owes gemm_reduce() or gemm_reduce_rms_norm()
    int * deferred_split = nullptr;
    int * probe_path = nullptr;               // non-null: NO launch; *probe_path = 1 when these arguments go to the k_gemm_f16_glds<MB> family (whose tile epilogue and reduction both write gemm_mat::y16 rows), else 0
    bool defer_multi = false;                 // the same for a grouped launch (nmat > 1, no addends): slab s = partial + s * (sum of M_i) * N floats, matrix i a dense [N][M_i] block at + (M_0 + .. + M_{i-1}) * N
    // gate / up + SWIGLU (gemm_glu_ok): no f32 outputs; f16 rows of silu(W[glu_gate].x) * (W[1 - glu_gate].x) go to glu_out16
    uint16_t * glu_out16 = nullptr; size_t glu_out16_rs = 0; int glu_gate = 0;
    // every matrix given as Q4_K blocks (qtype) AND the activations as the block-major Q8_K image: the launch goes to mmq_tile() (X / x_rs unused)
    const void * qt_img = nullptr;
};
int    device_cu_count();                                        // CUs of the current device
void   gemm_reduce(const float * partial, int nsplit, const float * resid, size_t resid_cs, float * dst, size_t dst_cs, int64_t M, int64_t N, hipStream_t st);
void   gemm_reduce_group(const float * partial, int nsplit, size_t slab_elems, int nmat, const size_t * off, const int64_t * M, int64_t N, float * const * dst, const size_t * dst_cs, hipStream_t st);   // the reduction a deferred grouped launch owes (no addends)
void   gemm_reduce2(const float * partial, int nsplit, const float * resid, size_t resid_cs, const float * resid2, size_t resid2_cs, float * dst, size_t dst_cs, int64_t M, int64_t N, hipStream_t st);
// the same reduction fused with the RMS_NORM -> MUL(w) of the result: dst = sum + resid (f32); y = rms_norm(dst) * w -> y32 / f16 rows y16
bool   gemm_reduce_rms_norm_ok(int64_t M);
void   gemm_reduce_rms_norm(const float * partial, int nsplit, const float * resid, size_t resid_cs, float * dst, size_t dst_cs, const float * w, float eps,
                            float * y32, size_t y32_cs, uint16_t * y16, size_t y16_rs, int64_t M, int64_t N, hipStream_t st, bool y16_q8 = false);      // y16_q8: the f16 rows carry the Q8_K-quantised values (q8k_requant4; M % 256 == 0)
// A chain of element-wise f32 nodes in one launch (elementwise.hip k_ew_chain): op j reads external inputs (selector 0..5) or the result of an earlier op of the chain
// (selector 8 + index) and only the last result is stored.  Each op is the arithmetic of its stand-alone kernel (the library is built with -ffp-contract=off: nothing fuses
// across ops), so a chain gives the stand-alone launches' values bit for bit.
struct ew_op_desc { int kind; int sub; int a, b; float p0, p1; };     // kind: GGML_OP_ADD / SUB / MUL / DIV / SCALE / UNARY (sub = the unary op) / SQR / SQRT / LOG / SIN / COS / CLAMP / LEAKY_RELU
struct ew_chain_args {
    int n_ops = 0; ew_op_desc op[8];
    int n_in = 0; const float * in[6]; int in_mode[6]; uint32_t in_n04[6];      // mode 0: the chain's shape, element for element; 1: one row of 4 * in_n04 floats repeated; 2: one value;
    uint32_t in_per4[6] = { 0 }, in_bs4[6] = { 0 };                            // 3: one row per slice of 4 * in_per4 chain elements, the rows 4 * in_bs4 floats apart (the DiT's per-batch-element shift / scale / gate vectors)
    float * out = nullptr; int64_t total = 0;                                  // total % 4 == 0, every pointer 16-byte aligned (mode 2: 4-byte)
};
void   ew_chain(const ew_chain_args & a, hipStream_t st);
void   gemm_f16_multi(const gemm_multi_args & a, hipStream_t st);
int    gemm_f16_small_n_ksplit(const gemm_multi_args & a);      // the K split a launch of <= 128 columns will get (> 1: reduction epilogue, two addends per matrix possible)
size_t gemm_split_scratch_bytes(int64_t M, int64_t N, int64_t K);
long   gemm_variant_launches(int v);      // launches so far of the 256 x 256 (0) / 192-row (1) / gate-up-SWIGLU (2) tile kernels (test instrumentation)
bool   gemm_glu_ok(const gemm_multi_args & a);
// any-shape f32-MFMA GEMM (gemm_any.hip): F32 weights, or F16 weights whose K the F16 GEMM does not take; X f32 rows (rounded to f16 when w_f16)
struct gemm_any_args {
    const void * W; size_t w_rs, w_nb2 = 0, w_nb3 = 0; bool w_f16; bool w_bf16 = false;      // (w_bf16: 16-bit weights are BF16, activations rounded to bf16: ggml_vec_dot_bf16)
    const void * X; size_t x_rs, x_nb2 = 0, x_nb3 = 0; bool x_f16 = false;     // f32 rows, or f16 rows (with F16 weights)
    float * dst; size_t dst_cs, dst_nb2 = 0, dst_nb3 = 0;
    int64_t M, N, K; int nbatch = 1, ne12 = 1, r2 = 1, r3 = 1;
    bool accumulate = false;                 // dst += W.X (the K tail behind a gemm_f16 launch over the first K - K % 64 columns)
    const float * bias = nullptr;            // dst[n][m] = W.X + bias[m]: the ADD of a [M] row vector that follows the mat-mul (one more f32 rounding, as the separate op)
    int act = 0;                             // 1: GELU of the finished value (the UNARY node behind the bias ADD: a DiT block's FFN), same arithmetic as the element-wise kernel
    // optional, for small f32 x f32 products with a long K: split K over workgroups too; partial tiles go to `partial`, the last workgroup of a tile to arrive (ticket in
    // `counters`, zero between launches) folds them in split order.  Both belong to the calling backend context (one stream: launches do not overlap).
    float * partial = nullptr; size_t partial_bytes = 0; unsigned * counters = nullptr; int n_counters = 0;
    // up to two more products over the same X with weights of the same shape and strides (a DiT block's q / k / v projections) in the launch: gemm_any_group_ok() first
    int nmat = 1; const void * W_more[2] = { nullptr, nullptr }; float * dst_more[2] = { nullptr, nullptr }; const float * bias_more[2] = { nullptr, nullptr };
};
void   gemm_any(const gemm_any_args & a, hipStream_t st);
bool   gemm_any_group_ok(const gemm_any_args & a);       // would gemm_any take a.nmat > 1 (the 16 x 16-tile f32 kernel's shapes)?
// attention as separate f32 nodes over a few hundred keys, one launch (attn_f32.hip): Q [D, nq, HB], K [D, nkv, HB], V^T [nkv, D, HB] f32 with K-contiguous rows;
// scores x * s1 + b1 (the SCALE node, when has_scale) then * s2 (the soft-max's scale), soft-max, . V; element (d, q, h, s) of the result (head-batch = h + H * s) through the strides
struct attn_f32_args {
    const void * q, * k, * vt; void * dst;
    size_t q_rs, q_bs, k_rs, k_bs, v_rs, v_bs, d_nb_q, d_nb_h, d_nb_s;
    int64_t D, nq, nkv, HB, H;
    float s1 = 1.0f, b1 = 0.0f, s2 = 1.0f; bool has_scale = false;
    size_t q_bs2 = 0, k_bs2 = 0; int64_t q_H = 0, k_H = 0;      // q_H / k_H > 0: that operand's head-batch index is h + H s at h * bs + s * bs2 (a permuted 4-D view read in place)
    size_t v_ks = 0, v_bs2 = 0; int64_t v_H = 0;                // v_ks != 0: `vt` is V itself, element (key k, dim d) at k * v_ks + d * 4; head-batch index h + v_H s at h * v_bs + s * v_bs2
};
bool   attn_f32_ok(const attn_f32_args & a);
void   attn_f32(const attn_f32_args & a, hipStream_t st);
size_t gemm_any_split_scratch_bytes(int64_t M, int64_t N, int64_t K, int nbatch, bool f32_operands);      // what the split above needs for this shape (0: it would not split)

} // namespace mi

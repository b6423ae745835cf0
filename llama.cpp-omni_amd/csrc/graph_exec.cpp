// graph_exec.cpp -- executor side: the node executors, the fusion matchers in front of them (activation images, grouped GEMMs, norm / rope chains, attention
// chains, element-wise chains) and run_nodes.  (Split out of graph.cpp in round 4; no behaviour change.)
#include "graph_internal.hpp"
#include <map>

namespace mi {

// ------------------------------------------------------------------------------------------------ MUL_MAT
struct byte_range { const char * lo; const char * hi; };
static byte_range range_of(const ggml_tensor * t) { const char * p = (const char *) t->data; return { p, p + nbytes(t) }; }
static bool overlap(byte_range a, byte_range b) { return a.lo < b.hi && b.lo < a.hi && a.lo != a.hi && b.lo != b.hi; }

// convert src1 of a MUL_MAT into the activation format of `kind` (or reuse the cached conversion); returns the image stride
static void materialise_norm(exec_state & s);
static void lazy_net(exec_state & s, int i);
static void lazy_materialise(exec_state & s, const ggml_tensor * t, int reader_op = -1);
static byte_range range_of(const tdesc & d);
static size_t prepare_act(exec_state & s, const ggml_tensor * x, act_kind kind) {
    const int64_t K = x->ne[0], N = x->ne[1], ne12 = x->ne[2], ne13 = x->ne[3];
    const size_t img = act_image_bytes(kind, K);
    if (s.pn.m && x == s.pn.m) materialise_norm(s);                      // a consumer outside the in-kernel-norm launches
    if (kind == ACT_F32) return 0;
    if (kind == ACT_Q8KT && (ne12 != 1 || ne13 != 1 || x->type != GGML_TYPE_F32)) { fprintf(stderr, "[mi355x] prepare_act: the block-major Q8_K image takes one 2-D f32 activation\n"); abort(); }
    const bool cached = s.a_src == x->data && s.a_kind == kind && s.a_K == K && s.a_ne[0] == N && s.a_ne[1] == ne12 &&
                        s.a_ne[2] == ne13 && s.a_nb[0] == x->nb[1] && s.a_nb[1] == x->nb[2] && s.a_nb[2] == x->nb[3];
    if (cached) return img;
    auto conv = [&](const float * src, size_t xs, void * out, int64_t rows) {
        if      (kind == ACT_Q8K) quantize_q8k_image(src, xs, out, K, rows, s.st);
        else if (kind == ACT_Q8KT) quantize_q8k_tile_image(src, xs, out, K, rows, s.st);
        else if (kind == ACT_F16Q) convert_f32_f16q_rows(src, xs, (uint16_t *) out, img, K, rows, s.st);
        else if (kind == ACT_Q80) quantize_q80_image(src, xs, out, K, rows, s.st);
        else                      convert_f32_f16_rows(src, xs, (uint16_t *) out, img, K, rows, s.st);
        ++s.n_kernels;
    };
    prof_scope ps(s, "act_convert", 0);
    const bool flat = (ne12 == 1 || x->nb[2] == (size_t) N * x->nb[1]) && (ne13 == 1 || x->nb[3] == (size_t) ne12 * x->nb[2]);
    if (x->type == GGML_TYPE_F16) {
        // F16 x F16 (the MUL_MAT of ggml_conv_1d / ggml_conv_2d: im2col columns against an f16 kernel): the activation rows are already
        // in the GEMM's format; gather them into the dense image (supports_op admits F16 src1 only next to F16 src0)
        tdesc d; d.p = s.c->act_scratch; d.ne[0] = K; d.ne[1] = N; d.ne[2] = ne12; d.ne[3] = ne13;
        d.nb[0] = 2; d.nb[1] = img; d.nb[2] = img * (size_t) N; d.nb[3] = img * (size_t) (N * ne12);
        cpy_strided(td(x), GGML_TYPE_F16, d, GGML_TYPE_F16, s.st);
        ++s.n_kernels;
    } else if (flat) {
        conv((const float *) x->data, x->nb[1], s.c->act_scratch, N * ne12 * ne13);
    } else if (kind == ACT_F16 && N * ne12 * ne13 <= 65535) {      // (ACT_F16Q never gets here: K-quant weights take 2-D activations in every graph the planner sends to the GEMM)            // permuted rows (q seen per head): one strided launch
        convert_f32_f16_rows3((const float *) x->data, x->nb[1], x->nb[2], x->nb[3], N, ne12, ne13, (uint16_t *) s.c->act_scratch, img, K, s.st);
        ++s.n_kernels;
    } else {
        for (int64_t i13 = 0; i13 < ne13; ++i13)
            for (int64_t i12 = 0; i12 < ne12; ++i12)
                conv((const float *) ((const char *) x->data + i12 * x->nb[2] + i13 * x->nb[3]), x->nb[1],
                     (char *) s.c->act_scratch + (size_t) ((i13 * ne12 + i12) * N) * img, N);
    }
    s.a_src = x->data; s.a_kind = kind; s.a_K = K; s.a_ne[0] = N; s.a_ne[1] = ne12; s.a_ne[2] = ne13;
    s.a_nb[0] = x->nb[1]; s.a_nb[1] = x->nb[2]; s.a_nb[2] = x->nb[3];
    s.a_range_lo = (const char *) x->data; s.a_range_hi = (const char *) x->data + nbytes(x);
    return img;
}

static const char * mmv_class(int type) {
    return type == GGML_TYPE_Q4_K ? "mmv_q4k" : type == GGML_TYPE_Q6_K ? "mmv_q6k" : type == GGML_TYPE_Q8_0 ? "mmv_q80" : type == GGML_TYPE_F16 ? "mmv_f16" : "mmv_f32";
}

// resident F16 image of a quantised weight matrix (shadow.hpp): built on first use outside of graph capture, only for tensors
// that live in a buffer marked GGML_BACKEND_BUFFER_USAGE_WEIGHTS
static const uint16_t * weight_shadow(exec_state & s, const ggml_tensor * w, const char * wp, int64_t K, int64_t M) {
    const ggml_tensor * root = w;
    while (root->view_src) root = root->view_src;
    if (root->op != GGML_OP_NONE || !root->buffer || root->buffer->usage != GGML_BACKEND_BUFFER_USAGE_WEIGHTS) return nullptr;
    bool created = false;
    uint16_t * p = shadow_get_or_create(s.c->device, wp, (size_t) (M - 1) * w->nb[1] + row_size(w->type, K), w->type, K, M, w->nb[1], s.st, s.capturing, &created);
    if (!p || !created) return p;
    {
        prof_scope ps(s, "dequant_f16", (double) M * (double) row_size(w->type, K));
        dequant_rows_f16(w->type, wp, w->nb[1], p, (size_t) K * 2, K, M, s.st); ++s.n_kernels;
    }
    shadow_mark_ready(p, s.st);
    return p;
}

// does op_mul_mat send this MUL_MAT to the any-shape GEMM (gemm_any.hip)?  (after the MFMA GEMM and BF16 branches)
bool mm_takes_gemm_any(const ggml_tensor * n) {
    static const bool no_gemm_any = getenv("MI355X_NO_GEMM_ANY") != nullptr;
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    const int64_t K = w->ne[0], M = w->ne[1], N = x->ne[1];
    if (no_gemm_any || mm_uses_gemm(n) || w->type == GGML_TYPE_BF16) return false;
    return (w->type == GGML_TYPE_F32 || w->type == GGML_TYPE_F16) && N > MI_MMVQ_MAX_COLS &&
           ((x->type == GGML_TYPE_F32 && x->nb[0] == 4) || (x->type == GGML_TYPE_F16 && x->nb[0] == 2 && w->type == GGML_TYPE_F16)) && w->nb[0] == (w->type == GGML_TYPE_F16 ? 2u : 4u) && n->nb[0] == 4 &&
           x->ne[2] * x->ne[3] <= 65535 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31);
}
// out / bias: the ADD of a [M] row vector behind the mat-mul, folded into the any-shape GEMM's epilogue (exec_mul_mat decides; only that path takes them)
// sib / nsib: up to two more F32-weight mat-muls over the same activation (same weight shape and strides) for the launch; *sib_taken tells whether they went along
struct mm_sibling { const ggml_tensor * w; const ggml_tensor * out; const float * bias; };
static void op_mul_mat(exec_state & s, const ggml_tensor * dst, const ggml_tensor * out = nullptr, const float * bias = nullptr, const mm_sibling * sib = nullptr, int nsib = 0, bool * sib_taken = nullptr) {
    const ggml_tensor * w = dst->src[0];
    const ggml_tensor * x = dst->src[1];
    if (!out) out = dst;
    const int64_t K = w->ne[0], M = w->ne[1], N = x->ne[1];
    const int64_t ne12 = x->ne[2], ne13 = x->ne[3];
    const int64_t r2 = ne12 / w->ne[2], r3 = ne13 / w->ne[3];

    if (mm_uses_mmq_tile(dst)) {                                // Q4_K x a prefill ubatch: the tiled int8-MFMA kernel on the Q8_K image (mmq_tile.hip)
        prepare_act(s, x, ACT_Q8KT);
        mmqt_args q;
        q.nmat = 1; q.m[0] = { w->data, w->nb[1], (float *) dst->data, dst->nb[1], M }; q.img = s.c->act_scratch; q.N = N; q.K = K;
        if (dst->nb[1] % 16 == 0 && gemm_split_scratch_bytes(M, N, K) <= s.c->gemm_partial_bytes) { q.partial = (float *) s.c->gemm_partial; q.partial_bytes = s.c->gemm_partial_bytes; }
        prof_scope ps(s, "mmq_tile", 2.0 * (double) M * (double) N * (double) K);
        mmq_tile(q, s.st);
        ++s.n_kernels;
        return;
    }
    if (mm_uses_gemm(dst)) {
        // ---- prefill: MFMA GEMM.  X -> f16 rows (what the reference does for F16 weights, ggml-cpu.c:1245-1268); quantised W -> f16
        const size_t ximg = prepare_act(s, x, x->ne[2] * x->ne[3] == 1 ? gemm_act_kind(dst) : ACT_F16);
        // attention without FLASH_ATTN_EXT at prefill: every head's K.Q^T (or V^T.P) product in one launch
        if (w->type == GGML_TYPE_F16 && ne12 * ne13 > 1 && ne12 * ne13 <= 65535 && K % 64 == 0 && w->nb[1] % 16 == 0 && w->nb[2] % 16 == 0 && w->nb[3] % 16 == 0 &&
            ((uintptr_t) w->data & 15) == 0 && dst->nb[0] == 4) {
            gemm_multi_args a;
            a.nmat = 1; a.m[0] = { (const uint16_t *) w->data, w->nb[1], (float *) dst->data, dst->nb[1], M, nullptr, 0 };
            a.X = (const uint16_t *) s.c->act_scratch; a.x_rs = ximg; a.N = N; a.K = K; a.partial = nullptr;
            a.nbatch = (int) (ne12 * ne13); a.ne12 = (int) ne12; a.r2 = (int) r2; a.r3 = (int) r3;
            a.w_nb2 = w->nb[2]; a.w_nb3 = w->nb[3]; a.x_bs = (size_t) N * ximg; a.dst_nb2 = dst->nb[2]; a.dst_nb3 = dst->nb[3];
            prof_scope ps(s, "gemm_f16", 2.0 * (double) M * (double) N * (double) K * (double) (ne12 * ne13));
            gemm_f16_multi(a, s.st);
            ++s.n_kernels;
            return;
        }
        const void * last_w = nullptr;
        for (int64_t i13 = 0; i13 < ne13; ++i13) {
            for (int64_t i12 = 0; i12 < ne12; ++i12) {
                const char * wp = (const char *) w->data + (i12 / r2) * w->nb[2] + (i13 / r3) * w->nb[3];
                const uint16_t * w16 = (const uint16_t *) wp; size_t w16_rs = w->nb[1];
                const uint16_t * sh = w->type != GGML_TYPE_F16 ? weight_shadow(s, w, wp, K, M) : nullptr;
                if (sh) { w16 = sh; w16_rs = (size_t) K * 2; }
                else if (w->type != GGML_TYPE_F16) {
                    if (wp != last_w) {
                        prof_scope ps(s, "dequant_f16", (double) M * (double) row_size(w->type, K));
                        dequant_rows_f16(w->type, wp, w->nb[1], (uint16_t *) s.c->w_scratch, (size_t) K * 2, K, M, s.st); ++s.n_kernels;
                        last_w = wp;
                    }
                    w16 = (const uint16_t *) s.c->w_scratch; w16_rs = (size_t) K * 2;
                }
                const char * xp = (const char *) s.c->act_scratch + (size_t) ((i13 * ne12 + i12) * N) * ximg;
                prof_scope ps(s, "gemm_f16", 2.0 * (double) M * (double) N * (double) K);
                gemm_f16_mfma(w16, w16_rs, (const uint16_t *) xp, ximg, (float *) ((char *) dst->data + i12 * dst->nb[2] + i13 * dst->nb[3]), dst->nb[1], M, N, K, s.st);
                ++s.n_kernels;
            }
        }
        return;
    }

    if (w->type == GGML_TYPE_BF16) {
        if (s.pn.m && x == s.pn.m) materialise_norm(s);
        gemm_any_args a;
        a.W = w->data; a.w_rs = w->nb[1]; a.w_nb2 = w->nb[2]; a.w_nb3 = w->nb[3]; a.w_f16 = false; a.w_bf16 = true;
        a.X = x->data; a.x_rs = x->nb[1]; a.x_nb2 = x->nb[2]; a.x_nb3 = x->nb[3];
        a.dst = (float *) dst->data; a.dst_cs = dst->nb[1]; a.dst_nb2 = dst->nb[2]; a.dst_nb3 = dst->nb[3];
        a.M = M; a.N = N; a.K = K; a.nbatch = (int) (ne12 * ne13); a.ne12 = (int) ne12; a.r2 = (int) r2; a.r3 = (int) r3;
        prof_scope ps(s, "gemm_any_bf16", 2.0 * (double) M * (double) N * (double) K * (double) (ne12 * ne13));
        gemm_any(a, s.st);
        ++s.n_kernels;
        return;
    }
    const act_kind kind = act_kind_for(w->type);
    // more than 8 columns against F32 weights, or F16 weights with a contraction length the F16 GEMM does not take (the omni encoders, Token2Wav):
    // one f32-MFMA launch over every (head, batch) instead of a mat-vec launch per 8 columns per head
    if (mm_takes_gemm_any(dst)) {
        if (s.pn.m && x == s.pn.m) materialise_norm(s);
        gemm_any_args a;
        int64_t k_done = 0;
        // the producer (SOFT_MAX of an encoder's / a flash-attention-off prefill's scores, or an earlier mat-mul on the same x) left the f16 image of x in
        // the scratch -- and possibly did not write the f32 block at all
        bool x_img = w->type == GGML_TYPE_F16 && x->type == GGML_TYPE_F32 && s.a_src == x->data && s.a_kind == ACT_F16 && s.a_K == K && s.a_ne[0] == N && s.a_ne[1] == ne12 &&
                     s.a_ne[2] == ne13 && s.a_nb[0] == x->nb[1] && s.a_nb[1] == x->nb[2] && s.a_nb[2] == x->nb[3];
        // F16 weights, K a few columns past a multiple of 64 (SigLip2's n_ff 4304): the F16 MFMA GEMM takes the first K - K % 64 columns, this kernel adds the tail
        if (w->type == GGML_TYPE_F16 && x->type == GGML_TYPE_F32 && ne12 * ne13 == 1 && K % 64 != 0 && K >= 512 && w->nb[1] % 16 == 0 && ((uintptr_t) w->data & 15) == 0 &&
            out->nb[1] % 16 == 0 && act_image_bytes(ACT_F16, K) * (size_t) N <= s.c->act_scratch_bytes) {
            const size_t ximg = prepare_act(s, x, ACT_F16);        // (nothing to do when the image is there already)
            x_img = true;
            k_done = K - K % 64;
            prof_scope ps(s, "gemm_f16", 2.0 * (double) M * (double) N * (double) k_done);
            gemm_multi_args ga;                                  // (split along K when the tiles do not fill the chip: SigLip2's fc2, 1152 x 1024 outputs, went from 79 to 24 us)
            ga.nmat = 1; ga.m[0] = { (const uint16_t *) w->data, w->nb[1], (float *) out->data, out->nb[1], M, nullptr, 0 };
            ga.X = (const uint16_t *) s.c->act_scratch; ga.x_rs = ximg; ga.N = N; ga.K = k_done;
            ga.partial = gemm_split_scratch_bytes(M, N, k_done) <= s.c->gemm_partial_bytes ? (float *) s.c->gemm_partial : nullptr; ga.partial_bytes = s.c->gemm_partial_bytes;
            gemm_f16_multi(ga, s.st);
            ++s.n_kernels;
        }
        a.W = (const char *) w->data + k_done * (w->type == GGML_TYPE_F16 ? 2 : 4); a.w_rs = w->nb[1]; a.w_nb2 = w->nb[2]; a.w_nb3 = w->nb[3]; a.w_f16 = w->type == GGML_TYPE_F16;
        if (x_img) {                                             // rows of the image: [ne13][ne12][N] x act_image_bytes
            const size_t img = act_image_bytes(ACT_F16, K);
            a.X = (const char *) s.c->act_scratch + k_done * 2; a.x_rs = img; a.x_nb2 = img * (size_t) N; a.x_nb3 = img * (size_t) (N * ne12); a.x_f16 = true;
        } else {
            a.X = (const char *) x->data + k_done * (x->type == GGML_TYPE_F16 ? 2 : 4); a.x_rs = x->nb[1]; a.x_nb2 = x->nb[2]; a.x_nb3 = x->nb[3]; a.x_f16 = x->type == GGML_TYPE_F16;
        }
        a.dst = (float *) out->data; a.dst_cs = out->nb[1]; a.dst_nb2 = out->nb[2]; a.dst_nb3 = out->nb[3]; a.accumulate = k_done > 0; a.bias = bias;
        a.M = M; a.N = N; a.K = K - k_done; a.nbatch = (int) (ne12 * ne13); a.ne12 = (int) ne12; a.r2 = (int) r2; a.r3 = (int) r3;
        if (s.c->gemm_partial && !s.c->fa_counters && !s.capturing) {      // (first use is an eager submission: captures come from the second on)
            if (hipMalloc((void **) &s.c->fa_counters, 1024 * sizeof(unsigned)) == hipSuccess) HIP_CHECK(hipMemsetAsync(s.c->fa_counters, 0, 1024 * sizeof(unsigned), s.st));
            else { (void) hipGetLastError(); s.c->fa_counters = nullptr; }
        }
        if (s.c->gemm_partial && s.c->fa_counters) { a.partial = (float *) s.c->gemm_partial; a.partial_bytes = s.c->gemm_partial_bytes; a.counters = s.c->fa_counters; a.n_counters = 1024; }
        if (nsib > 0 && sib_taken) {
            *sib_taken = false;
            if (k_done == 0 && !x_img && w->type == GGML_TYPE_F32 && x->type == GGML_TYPE_F32) {
                a.nmat = 1 + nsib;
                for (int q = 0; q < nsib; ++q) { a.W_more[q] = sib[q].w->data; a.dst_more[q] = (float *) sib[q].out->data; a.bias_more[q] = sib[q].bias; }
                if (gemm_any_group_ok(a)) *sib_taken = true; else a.nmat = 1;
            }
        }
        prof_scope ps(s, w->type == GGML_TYPE_F16 ? "gemm_any_f16" : "gemm_any_f32", 2.0 * (double) M * (double) N * (double) (K - k_done) * (double) (ne12 * ne13) * (double) a.nmat);
        gemm_any(a, s.st);
        ++s.n_kernels;
        return;
    }
    const size_t img = prepare_act(s, x, kind);

    const double wbytes = (double) M * (double) row_size(w->type, K);
    // attention without FLASH_ATTN_EXT: K / V^T per KV head against one activation per query head -- every head in ONE launch
    if ((w->type == GGML_TYPE_F16 || w->type == GGML_TYPE_F32) && ne12 * ne13 > 1 && N <= MI_MMVQ_MAX_COLS && ne12 * ne13 <= 65535) {
        mmv_args a;
        a.W = w->data; a.w_rs = w->nb[1]; a.K = K; a.nrows = M; a.ncols = (int) N;
        a.dst = (float *) dst->data; a.dst_cs = dst->nb[1];
        a.nbatch = (int) (ne12 * ne13); a.ne12 = (int) ne12; a.r2 = (int) r2; a.r3 = (int) r3;
        a.w_nb2 = w->nb[2]; a.w_nb3 = w->nb[3]; a.dst_nb2 = dst->nb[2]; a.dst_nb3 = dst->nb[3];
        bool ok = true;
        if (kind == ACT_F32) {
            a.act = x->data; a.act_cs = x->nb[1]; a.act_bs = x->nb[2];
            ok = ne13 == 1 || x->nb[3] == (size_t) ne12 * x->nb[2];
        } else { a.act = s.c->act_scratch; a.act_cs = img; a.act_bs = (size_t) N * img; }
        if (ok) {
            prof_scope ps(s, mmv_class(w->type), wbytes * (double) (ne12 * ne13) / (double) (r2 * r3));
            if (w->type == GGML_TYPE_F16) mmv_f16(a, s.st); else mmv_f32(a, s.st);
            ++s.n_kernels;
            return;
        }
    }
    for (int64_t i13 = 0; i13 < ne13; ++i13) {
        for (int64_t i12 = 0; i12 < ne12; ++i12) {
            const char * wp = (const char *) w->data + (i12 / r2) * w->nb[2] + (i13 / r3) * w->nb[3];
            char *       dp = (char *) dst->data + i12 * dst->nb[2] + i13 * dst->nb[3];
            if (mm_uses_mmq(dst)) {                                   // 9 .. 64 columns of a K-quant matrix: int8 MFMA, 32 columns per launch
                for (int64_t c0 = 0; c0 < N; c0 += 32) {
                    mmq_args q;
                    q.nmat = 1; q.m[0] = { wp, w->nb[1], (float *) (dp + c0 * dst->nb[1]), dst->nb[1], M, (int) w->type };
                    q.act = (const char *) s.c->act_scratch + (size_t) ((i13 * ne12 + i12) * N + c0) * img; q.act_cs = img;
                    q.K = K; q.ncols = (int) (N - c0 < 32 ? N - c0 : 32);
                    prof_scope ps(s, w->type == GGML_TYPE_Q4_K ? "mmq_q4k" : "mmq_q6k", wbytes);
                    mmq_kquant(q, s.st); ++s.n_kernels;
                }
                continue;
            }
            const void * wv = wp; size_t wv_rs = w->nb[1]; int wv_type = w->type;
            if (is_image_quant(w->type)) {                            // mat-vec on the F16 image of the block format
                const uint16_t * sh = weight_shadow(s, w, wp, K, M);
                if (!sh) {
                    prof_scope ps(s, "dequant_f16", wbytes);
                    dequant_rows_f16(w->type, wp, w->nb[1], (uint16_t *) s.c->w_scratch, (size_t) K * 2, K, M, s.st); ++s.n_kernels;
                    sh = (const uint16_t *) s.c->w_scratch;
                }
                wv = sh; wv_rs = (size_t) K * 2; wv_type = GGML_TYPE_F16;
            }
            for (int64_t c0 = 0; c0 < N; c0 += MI_MMVQ_MAX_COLS) {
                mmv_args a;
                a.W = wv; a.w_rs = wv_rs; a.K = K; a.nrows = M;
                a.ncols = (int) (N - c0 < MI_MMVQ_MAX_COLS ? N - c0 : MI_MMVQ_MAX_COLS);
                a.dst = (float *) (dp + c0 * dst->nb[1]); a.dst_cs = dst->nb[1];
                if (kind == ACT_F32) {
                    a.act = (const char *) x->data + i12 * x->nb[2] + i13 * x->nb[3] + c0 * x->nb[1]; a.act_cs = x->nb[1];
                } else {
                    a.act = (const char *) s.c->act_scratch + (size_t) ((i13 * ne12 + i12) * N + c0) * img; a.act_cs = img;
                }
                prof_scope ps(s, mmv_class(wv_type), wbytes);
                switch (wv_type) {
                    case GGML_TYPE_Q4_K: mmv_q4_K(a, s.st); break;
                    case GGML_TYPE_Q5_K: mmv_q5_K(a, s.st); break;
                    case GGML_TYPE_Q6_K: mmv_q6_K(a, s.st); break;
                    case GGML_TYPE_Q8_0: mmv_q8_0(a, s.st); break;
                    case GGML_TYPE_Q4_0: mmv_q4_0(a, s.st); break;
                    case GGML_TYPE_Q5_0: mmv_q5_0(a, s.st); break;
                    case GGML_TYPE_F16:  mmv_f16(a, s.st); break;
                    default:             mmv_f32(a, s.st); break;
                }
                ++s.n_kernels;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ fusion planner
// Fusions are found on the DATA FLOW, not on adjacency: libllama's node order interleaves the q/k/v chains
// (ggml_build_forward_expand visits each chain depth-first), so wk's MUL_MAT sits five nodes after wq's.  A node j is
// executed early, together with node i < j, only when that cannot change any byte another node observes:
//   * every source of j is a leaf, was computed before i, or is produced inside the fused item, and
//   * no node strictly between i and j (and outside the item) reads or writes memory overlapping j's output,
//     nor writes memory overlapping j's inputs (ggml-alloc re-uses the storage of dead tensors).
static bool is_kquant(int t) { return t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K; }   // the formats with integer-dot kernels on Q8_K activations

static bool plain_kq_matvec(const ggml_tensor * n, int max_cols) {      // MUL_MAT(K-quant W [K,M], f32 x [K,N<=max]) with no broadcast
    if (n->op != GGML_OP_MUL_MAT || is_empty(n)) return false;
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    return is_kquant(w->type) && w->ne[2] == 1 && w->ne[3] == 1 && x->ne[2] == 1 && x->ne[3] == 1 && x->ne[1] <= max_cols &&
           q8k_image_bytes(w->ne[0]) * (size_t) x->ne[1] <= 152 * 1024 && n->nb[0] == 4;
}
// ... or against 9 .. 64 columns on the int8 matrix cores (mmq.hip): the same fusions (sibling batching, residual epilogue, norm image)
static bool kq_mm_ok(const ggml_tensor * n) {
    if (n->op != GGML_OP_MUL_MAT || is_empty(n)) return false;
    if (!mm_uses_mmq(n)) return plain_kq_matvec(n, MI_MMVQ_MAX_COLS);
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    return w->ne[2] == 1 && w->ne[3] == 1 && x->ne[2] == 1 && x->ne[3] == 1 && n->nb[0] == 4 && x->nb[0] == 4;
}
// the Q8_0 twin (mmv1q.hip): MUL_MAT(Q8_0 W [K, M], f32 x [K, 1]), no broadcast -- the TTS / Token2Wav modules' decode mat-vecs
static bool q80_mv1_node(exec_state & s, const ggml_tensor * n) {
    if (!s.c->opt_mv1 || n->op != GGML_OP_MUL_MAT || is_empty(n)) return false;
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    if ((w->type != GGML_TYPE_Q8_0 && w->type != GGML_TYPE_F16) || x->type != GGML_TYPE_F32 || w->ne[2] != 1 || w->ne[3] != 1 || x->ne[1] != 1 || x->ne[2] != 1 || x->ne[3] != 1 || n->nb[0] != 4 || x->nb[0] != 4) return false;
    mv1_args v; v.nmat = 1; v.K = w->ne[0];
    v.m[0] = { w->data, w->nb[1], (float *) n->data, 0, nullptr, 0, w->ne[1], (int) w->type };
    v.img = (const void *) 16;
    return mmv1_ok(v);
}
// batch-1 decode form (mmv1.hip): one column, Q4_K / Q6_K, K a multiple of 256 up to 16384, aligned rows; or the Q8_0 twin
static bool mv1_node_ok(exec_state & s, const ggml_tensor * n) {
    if (q80_mv1_node(s, n)) return true;
    if (!s.c->opt_mv1 || !plain_kq_matvec(n, 1)) return false;
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    if (x->ne[1] != 1 || x->type != GGML_TYPE_F32 || (w->type != GGML_TYPE_Q4_K && w->type != GGML_TYPE_Q6_K)) return false;
    mv1_args v; v.nmat = 1; v.K = w->ne[0];
    v.m[0] = { w->data, w->nb[1], (float *) n->data, 0, nullptr, 0, w->ne[1], (int) w->type };
    v.img = (const void *) 16;                                             // (source checked separately)
    return mmv1_ok(v);
}
// activation source of an mmv1 launch on x: the pending norm (computed inside the launch), the cached Q8_K image, the f32 row itself
// (quantised inside the launch), or -- when an output would overwrite x while the launch reads it -- a quantise launch first
static size_t prepare_act(exec_state & s, const ggml_tensor * x, act_kind kind);
static bool norm_in_kernel(exec_state & s, const ggml_tensor * x, const ggml_tensor * const * outs, int n_outs, int n_consumers, mmv_norm & nr);
// the attention rows this mat-vec reads still lie as slices' partial states: fold them in the launch's prologue if the LDS-DMA engine takes the launch, else write the rows first
static void gs_materialise(exec_state & s) {
    if (!s.gs.n) return;
    prof_scope ps(s, "fattn", 0);
    fattn_gs_merge((const float *) s.c->fa_scratch, (float *) s.gs.n->data, s.gs.nh, s.gs.D, s.st); ++s.n_kernels;
    s.gs.n = nullptr; s.gs.consumer = -1;
}
static void mv1_source(exec_state & s, const ggml_tensor * x, const ggml_tensor * const * outs, int n_outs, int n_consumers, mv1_args & v) {
    if (s.gs.n && x->data == s.gs.n->data) {
        mv1_args t = v; t.x = nullptr; t.norm_w = nullptr; t.img = nullptr; t.parts = (const float *) s.c->fa_scratch; t.nslice = fattn_gs_nslice();
        if (!(s.pn.m && x == s.pn.m) && x->ne[0] == (int64_t) s.gs.nh * s.gs.D && mmv2_enabled() && mmv2_ok(t)) {
            v.parts = t.parts; v.nslice = t.nslice; v.x = nullptr; v.norm_w = nullptr; v.img = nullptr;
            s.gs.n = nullptr; s.gs.consumer = -1; ++s.n_fused;
            return;
        }
        gs_materialise(s);
    }
    mmv_norm nr;
    if (s.pn.m && x == s.pn.m && ((uintptr_t) s.pn.x->data & 15) == 0 && ((uintptr_t) s.pn.wt->data & 15) == 0 && norm_in_kernel(s, x, outs, n_outs, n_consumers, nr)) {
        v.x = nr.x; v.norm_w = nr.w; v.eps = nr.eps;
        return;
    }
    const int64_t K = x->ne[0];
    const act_kind kind = v.m[0].type == GGML_TYPE_Q8_0 ? ACT_Q80 : (v.m[0].type == GGML_TYPE_F16 ? ACT_F16 : ACT_Q8K);       // (v.m[] is filled before the source is chosen)
    const bool cached = s.a_src == x->data && s.a_kind == kind && s.a_K == K && s.a_ne[0] == 1 && s.a_ne[1] == x->ne[2] && s.a_ne[2] == x->ne[3];
    bool plain = !cached && !(s.pn.m && x == s.pn.m) && ((uintptr_t) x->data & 15) == 0;
    if (plain) {
        const byte_range rx = range_of(x);
        for (int i = 0; i < n_outs; ++i) if (outs[i] && overlap(range_of(outs[i]), rx)) plain = false;
    }
    if (plain) { v.x = (const float *) x->data; v.norm_w = nullptr; return; }
    prepare_act(s, x, kind);
    v.img = s.c->act_scratch;
}
static bool same_act(const ggml_tensor * a, const ggml_tensor * b) {
    return a->data == b->data && a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3] &&
           a->nb[1] == b->nb[1] && a->nb[2] == b->nb[2] && a->nb[3] == b->nb[3];
}
// Is t read by somebody this executor does not see?  Graph outputs, and -- when the scheduler cut the graph into splits -- tensors whose
// whole-graph use count (ggml_cgraph::use_counts, shared by the split views: ggml_graph_view) exceeds the uses inside this split: a later
// split (on this or another backend) reads them, so their f32 value must be written and no fusion may swallow them (cf. ggml_can_fuse).
static bool is_out(exec_state & s, const ggml_tensor * t) { return (t->flags & GGML_TENSOR_FLAG_OUTPUT) || s.external.count(t) != 0; }
static int n_users(exec_state & s, const ggml_tensor * t) {
    auto it = s.users.find(t);
    return (it == s.users.end() ? 0 : (int) it->second.size()) + (s.external.count(t) ? 1 : 0);
}
static int sole_user(exec_state & s, const ggml_tensor * t) {           // index of the only consumer node, or -1
    auto it = s.users.find(t);
    if (it == s.users.end() || it->second.size() != 1 || is_out(s, t)) return -1;
    return it->second[0];
}
static int next_real_node(exec_state & s, int i) {                      // the next node after i that will launch something (-1: none)
    for (int j = i + 1; j < s.g->n_nodes; ++j) if (!s.done[j] && !is_noop(s.g->nodes[j])) return j;
    return -1;
}
static bool ready_before(exec_state & s, const ggml_tensor * src, int i, const int * item, int n_item) {
    if (!src) return true;
    const ggml_tensor * t = src;
    while (t) {                                                          // walk through view chains down to the producing node
        auto it = s.index.find(t);
        if (it != s.index.end()) {
            const int k = it->second;
            if (!is_noop(s.g->nodes[k])) {
                if (k < i || s.done[k]) return true;                     // computed already (in order, or hoisted earlier)
                for (int q = 0; q < n_item; ++q) if (item[q] == k) return true;
                return false;
            }
            if (k >= i) {                                                // a view node created after i: its base must still be ready
                bool ok = true;
                for (int q = 0; q < GGML_MAX_SRC && ok; ++q) if (s.g->nodes[k]->src[q]) ok = ready_before(s, s.g->nodes[k]->src[q], i, item, n_item);
                return ok;
            }
        }
        t = t->view_src;
    }
    return true;                                                         // leaf (weight / graph input)
}
static bool can_hoist(exec_state & s, int i, int j, const int * item, int n_item) {
    const ggml_tensor * nj = s.g->nodes[j];
    for (int k = 0; k < GGML_MAX_SRC; ++k) if (nj->src[k] != nj && !ready_before(s, nj->src[k], i, item, n_item)) return false;      // (ggml_cast names its result as its own src[1])
    const byte_range dj = range_of(nj);
    // a copy that was left un-run (s.lazy) keeps READING its source until its last reader has run: ggml-alloc considers that source dead behind the CONT and may have placed
    // nj's result on it -- writing it early would feed the lazy readers clobbered data.  The copies' own buffers count as written by whoever materialises them.
    for (const auto & kv : s.lazy) if (overlap(dj, range_of(kv.second.src)) || overlap(dj, range_of(kv.first))) return false;
    for (int m = i + 1; m < j; ++m) {
        const ggml_tensor * nm = s.g->nodes[m];
        bool in_item = false;
        for (int q = 0; q < n_item; ++q) in_item |= item[q] == m;
        if (in_item || s.done[m] || is_noop(nm)) continue;
        const byte_range dm = range_of(nm);
        if (overlap(dj, dm)) return false;
        for (int k = 0; k < GGML_MAX_SRC; ++k) {
            if (nm->src[k] && overlap(dj, range_of(nm->src[k]))) return false;
            if (nj->src[k] && overlap(dm, range_of(nj->src[k]))) return false;
        }
    }
    return true;
}
static void note_write(exec_state & s, const ggml_tensor * t) {          // a kernel wrote t: drop the activation cache if it aliased
    if (s.fa_mask) { const char * p = (const char *) t->data; if ((const char *) s.fa_mask >= p && (const char *) s.fa_mask < p + nbytes(t)) s.fa_mask = nullptr; }
    if (!s.a_src) return;
    const byte_range r = range_of(t);
    if (r.lo < s.a_range_hi && s.a_range_lo < r.hi) s.a_src = nullptr;
}

// ---- deferred norm (see exec_state::pn)
static void materialise_norm(exec_state & s) {                           // run the stand-alone kernel now: f32 result + Q8_K image, seeds the cache
    const ggml_tensor * m = s.pn.m, * x = s.pn.x, * wt = s.pn.wt;
    s.pn.m = nullptr;
    {
        prof_scope ps(s, "rms_norm_mul_quant", 0);
        rms_norm_mul_quant((const float *) x->data, x->nb[1], (const float *) wt->data, (float *) m->data, m->nb[1], s.c->act_scratch, m->ne[0], m->ne[1], s.pn.eps, s.st);
    }
    ++s.n_kernels;
    s.a_src = m->data; s.a_kind = ACT_Q8K; s.a_K = m->ne[0]; s.a_ne[0] = m->ne[1]; s.a_ne[1] = 1; s.a_ne[2] = 1;
    s.a_nb[0] = m->nb[1]; s.a_nb[1] = m->nb[2]; s.a_nb[2] = m->nb[3];
    s.a_range_lo = (const char *) m->data; s.a_range_hi = (const char *) m->data + nbytes(m);
}
// may the launch that writes `outs` take its activation from the pending norm of x?  (it reads the norm's INPUT while it runs)
static bool norm_in_kernel(exec_state & s, const ggml_tensor * x, const ggml_tensor * const * outs, int n_outs, int n_consumers, mmv_norm & nr) {
    if (!s.pn.m || x != s.pn.m) return false;                             // identity of the tensor, not of its address (ggml-alloc re-uses memory)
    const byte_range rx = range_of(s.pn.x);
    for (int i = 0; i < n_outs; ++i) if (outs[i] && overlap(range_of(outs[i]), rx)) { materialise_norm(s); return false; }
    nr.x = (const float *) s.pn.x->data; nr.x_cs = s.pn.x->nb[1]; nr.w = (const float *) s.pn.wt->data; nr.eps = s.pn.eps;
    s.pn.left -= n_consumers;
    if (s.pn.left <= 0) s.pn.m = nullptr;                                 // every consumer served: m's memory is nobody's business any more
    return true;
}

// prefill: MUL_MAT at node i goes to the MFMA GEMM together with the other MUL_MATs that consume the same activation (wq / wk / wv,
// ffn_gate / ffn_up: one launch fills the chip where wk alone is 32 tiles), with the residual ADD folded into the epilogue; a lone
// under-filled matrix (wo, ffn_down at ubatch 512) is split along K instead.  Returns false when the plain path must run.
static void materialise_reduce(exec_state & s) {
    const ggml_tensor * A = s.pr.A;
    s.pr.A = nullptr;
    prof_scope ps(s, "gemm_reduce", 0);
    gemm_reduce2((const float *) s.c->gemm_partial, s.pr.nsplit, s.pr.resid, s.pr.resid_cs, s.pr.resid2, s.pr.resid2_cs, (float *) A->data, A->nb[1], A->ne[0], A->ne[1], s.st);
    ++s.n_kernels;
}
static void materialise_group(exec_state & s, int skip_mask = 0) {       // skip_mask: results somebody has taken as slabs
    float * dst[3] = { nullptr, nullptr, nullptr }; size_t cs[3] = { 0, 0, 0 }, off[3] = { 0, 0, 0 }; int64_t M[3] = { 0, 0, 0 }; int n = 0;
    for (int q = 0; q < s.prm.n; ++q) if (!(skip_mask & (1 << q))) { dst[n] = (float *) s.prm.A[q]->data; cs[n] = s.prm.A[q]->nb[1]; off[n] = s.prm.off[q]; M[n] = s.prm.M[q]; ++n; }
    if (n > 0) {
        prof_scope ps(s, "gemm_reduce", 0);
        gemm_reduce_group((const float *) s.c->gemm_partial, s.prm.nsplit, s.prm.slab, n, off, M, s.prm.N, dst, cs, s.st);
        ++s.n_kernels;
    }
    s.prm.n = 0;
}
static bool reads_pending_group(exec_state & s, const ggml_tensor * n) {          // an RMS_NORM on (a view of) one of the pending grouped results
    if (n->op != GGML_OP_RMS_NORM || !n->src[0]) return false;
    for (int q = 0; q < s.prm.n; ++q) if (n->src[0]->data == s.prm.A[q]->data) return true;
    return false;
}
// Q4_K / Q6_K weights with NO resident F16 image (MI355X_NO_F16_SHADOW, the image budget spent, out of memory): the GEMM de-quantises the blocks
// inside its LDS staging (k_gemm_kq_glds) instead of running a de-quantise-to-scratch launch in front of every mat-mul.  With the image resident
// the F16 kernel is faster at every column count (the in-staging form spends ~900 VALU cycles per wave and K-step on nibbles, scales and f16
// rounding against 512 MFMA cycles: measured pp100 9.9 vs 7.6 ms, pp256 13.5 vs 9.4 ms), so otherwise it is only taken on request: MI355X_KQ_STAGING=1 / set_option("kq_staging") (<= MAX_COLS columns).
static bool kq_in_staging(exec_state & s, const ggml_tensor * w, int64_t N) {
    static const bool off = getenv("MI355X_NO_KQ_STAGING") != nullptr;
    static const int64_t max_n = getenv("MI355X_KQ_STAGING_MAX_COLS") ? atoll(getenv("MI355X_KQ_STAGING_MAX_COLS")) : 256;
    if (off || !(w->type == GGML_TYPE_Q4_K || w->type == GGML_TYPE_Q6_K) || w->ne[0] % 256 != 0 || w->ne[2] != 1 || w->ne[3] != 1 ||
        w->nb[1] % (w->type == GGML_TYPE_Q4_K ? 16 : 2) != 0 || ((uintptr_t) w->data & 15) != 0) return false;
    if (s.c->opt_kq_staging) return N <= max_n;
    return weight_shadow(s, w, (const char *) w->data, w->ne[0], w->ne[1]) == nullptr;
}
static bool gemm_operand(exec_state & s, const ggml_tensor * w, const uint16_t ** w16, size_t * rs) {
    if (w->ne[2] != 1 || w->ne[3] != 1) return false;
    if (w->type == GGML_TYPE_F16) { *w16 = (const uint16_t *) w->data; *rs = w->nb[1]; return true; }
    const uint16_t * sh = weight_shadow(s, w, (const char *) w->data, w->ne[0], w->ne[1]);
    if (!sh) return false;
    *w16 = sh; *rs = (size_t) w->ne[0] * 2;
    return true;
}
static bool gemm_groupable(const ggml_tensor * c) {
    if (c->op != GGML_OP_MUL_MAT || is_empty(c) || !mm_uses_gemm(c)) return false;
    const ggml_tensor * x = c->src[1];
    return x->ne[2] == 1 && x->ne[3] == 1 && c->src[0]->ne[0] % 64 == 0 && c->nb[0] == 4 && c->type == GGML_TYPE_F32;
}
static bool gemm_only_consumers(exec_state & s, const ggml_tensor * t, int64_t K, int64_t N, const ggml_tensor ** x_out);
static void seed_act_f16(exec_state & s, const ggml_tensor * x, bool quantised = false);
static bool exec_attn_sm_prefill(exec_state & s, int i, bool dry);
static bool exec_gemm_group(exec_state & s, int i) {
    ggml_cgraph * g = s.g;
    ggml_tensor * n = g->nodes[i];
    if (!gemm_groupable(n)) return false;
    const ggml_tensor * x = n->src[1];
    const int64_t K = x->ne[0], N = x->ne[1];
    gemm_multi_args a;
    a.nmat = 0; a.N = N; a.K = K; a.partial = nullptr;
    int mm_idx[3] = { i, -1, -1 };
    const bool qt = mm_uses_mmq_tile(n);                       // Q4_K blocks x the block-major Q8_K image on the int8 matrix cores (mmq_tile.hip): raw blocks like kq
    const bool kq = qt || kq_in_staging(s, n->src[0], N);      // then every matrix of the launch must be K-quant blocks too
    const act_kind xkind = qt ? ACT_Q8KT : gemm_act_kind(n);    // (siblings join the launch only when they take the same image)
    {
        const uint16_t * w16; size_t rs;
        if (kq) { w16 = (const uint16_t *) n->src[0]->data; rs = n->src[0]->nb[1]; }
        else if (!gemm_operand(s, n->src[0], &w16, &rs)) return false;
        a.m[a.nmat++] = { w16, rs, (float *) n->data, n->nb[1], n->src[0]->ne[1], nullptr, 0, kq ? (int) n->src[0]->type : 0 };
    }
    for (int j = i + 1; j < g->n_nodes && j < i + 32 && a.nmat < 3; ++j) {
        ggml_tensor * c = g->nodes[j];
        if (s.done[j] || !gemm_groupable(c) || !same_act(c->src[1], x)) continue;
        if (!can_hoist(s, i, j, mm_idx, a.nmat)) continue;
        if (qt != mm_uses_mmq_tile(c)) continue;
        if (!qt && gemm_act_kind(c) != xkind) continue;
        if (!qt && kq != kq_in_staging(s, c->src[0], N)) continue;
        const uint16_t * w16; size_t rs;
        if (kq) { w16 = (const uint16_t *) c->src[0]->data; rs = c->src[0]->nb[1]; }
        else if (!gemm_operand(s, c->src[0], &w16, &rs)) continue;
        mm_idx[a.nmat] = j;
        a.m[a.nmat++] = { w16, rs, (float *) c->data, c->nb[1], c->src[0]->ne[1], nullptr, 0, kq ? (int) c->src[0]->type : 0 };
    }
    // residual: the only consumer is ADD(c, r) / ADD(r, c) with r of the same shape, available now
    int add_idx[3] = { -1, -1, -1 };
    for (int q = 0; q < a.nmat; ++q) {
        ggml_tensor * c = g->nodes[mm_idx[q]];
        const int ai = sole_user(s, c);
        if (ai > mm_idx[q] && g->nodes[ai]->op == GGML_OP_ADD && !s.done[ai]) {
            ggml_tensor * A = g->nodes[ai];
            const ggml_tensor * r = A->src[0] == c ? A->src[1] : A->src[0];
            // ... or a bias: r one row of ne0 elements broadcast over the columns (the encoders' linear layers) = a residual with column stride 0
            const bool bias = r && A->src[0] == c && r->ne[0] == c->ne[0] && r->ne[1] * r->ne[2] * r->ne[3] == 1 && c->ne[1] > 1 && ((uintptr_t) r->data & 15) == 0;
            if (((A->src[0] == c) != (A->src[1] == c)) && r && r != c && r->type == GGML_TYPE_F32 && (same_shape(r, c) || bias) && same_shape(A, c) && r->nb[0] == 4 && A->nb[0] == 4 &&
                A->type == GGML_TYPE_F32 && (bias || r->nb[1] % 16 == 0) && A->nb[1] % 16 == 0) {
                int item[7]; int ni = 0;
                for (int t = 0; t < a.nmat; ++t) item[ni++] = mm_idx[t];
                for (int t = 0; t < q; ++t) if (add_idx[t] >= 0) item[ni++] = add_idx[t];
                item[ni++] = ai;
                if (can_hoist(s, i, ai, item, ni)) {
                    a.m[q].resid = (const float *) r->data; a.m[q].resid_cs = bias ? 0 : r->nb[1];
                    a.m[q].dst = (float *) A->data; a.m[q].dst_cs = A->nb[1];
                    add_idx[q] = ai;
                }
            }
        }
    }
    // ffn_gate / ffn_up whose only reader is one GLU(SWIGLU, split) that only feeds GEMMs: SWIGLU runs in the epilogue and the launch writes
    // the f16 activation image of ffn_down (into the alternate scratch: this launch still reads its own input image from act_scratch)
    int glu_idx = -1; const ggml_tensor * glu_x = nullptr;
    if (a.nmat == 2 && !kq && add_idx[0] < 0 && add_idx[1] < 0 && s.c->act_scratch_alt) {
        const int g0 = sole_user(s, g->nodes[mm_idx[0]]), g1 = sole_user(s, g->nodes[mm_idx[1]]);
        if (g0 >= 0 && g0 == g1 && g0 > mm_idx[1] && !s.done[g0]) {
            const ggml_tensor * G = g->nodes[g0];
            const ggml_tensor * m0 = g->nodes[mm_idx[0]], * m1 = g->nodes[mm_idx[1]];
            int item[3] = { mm_idx[0], mm_idx[1], g0 };
            if (G->op == GGML_OP_GLU && op_param_i32(G, 0) == GGML_GLU_OP_SWIGLU && op_param_i32(G, 1) == 0 && G->src[0] && G->src[1] &&
                ((G->src[0] == m0 && G->src[1] == m1) || (G->src[0] == m1 && G->src[1] == m0)) && G->type == GGML_TYPE_F32 && G->ne[2] == 1 && G->ne[3] == 1 &&
                G->ne[0] == m0->ne[0] && G->ne[1] == N && G->nb[1] == (size_t) G->ne[0] * 4 && !is_out(s, m0) && !is_out(s, m1) &&
                act_image_bytes(ACT_F16, G->ne[0]) * (size_t) N <= s.c->act_scratch_alt_bytes && gemm_only_consumers(s, G, G->ne[0], G->ne[1], &glu_x) && gemm_glu_ok(a) &&
                can_hoist(s, i, g0, item, 3)) {
                glu_idx = g0;
                a.glu_out16 = (uint16_t *) s.c->act_scratch_alt; a.glu_out16_rs = act_image_bytes(ACT_F16, G->ne[0]); a.glu_gate = G->src[0] == m0 ? 0 : 1;
            }
        }
    }
    const size_t ximg = prepare_act(s, x, xkind);
    a.X = (const uint16_t *) s.c->act_scratch; a.x_rs = ximg;
    if (qt) a.qt_img = s.c->act_scratch;
    if (glu_idx >= 0) {
        double flops = 2.0 * 2.0 * (double) a.m[0].M * (double) N * (double) K;
        {
            prof_scope ps(s, "gemm_f16", flops);
            gemm_f16_multi(a, s.st);
        }
        ++s.n_kernels; s.n_fused += 2;
        s.done[mm_idx[1]] = 1; s.done[glu_idx] = 1;
        std::swap(s.c->act_scratch, s.c->act_scratch_alt); std::swap(s.c->act_scratch_bytes, s.c->act_scratch_alt_bytes);
        seed_act_f16(s, glu_x);
        return true;
    }
    if (a.nmat == 1 && a.m[0].dst_cs % 16 == 0 && gemm_split_scratch_bytes(a.m[0].M, N, K) <= s.c->gemm_partial_bytes) a.partial = (float *) s.c->gemm_partial;
    else if (a.nmat > 1 && N <= gemm_group_split_max_cols() && s.c->gemm_partial_bytes > 0) a.partial = (float *) s.c->gemm_partial;      // short prompts: split-K for the grouped launches too
    a.partial_bytes = s.c->gemm_partial_bytes;
    // a streaming encoder chunk (<= 128 columns: split K, the result goes through the reduction launch): linear -> + bias -> + residual stream.  The second ADD
    // (only reader of the first, same shape, its other operand ready) rides in the reduction's epilogue too: (acc + bias) + residual, the two roundings of the two nodes.
    // ... and so does the GELU behind the bias of fc1 (linear -> + bias -> GELU -> fc2, the only reader chain): the reduction applies it and writes the f16 image fc2 reads; the f32
    // rows only when somebody else reads them
    int un_idx = -1; const ggml_tensor * un_x = nullptr;
    static const bool no_act = getenv("MI355X_NO_GEMM_ACT") != nullptr;
    if (!no_act && a.nmat == 1 && add_idx[0] >= 0 && a.m[0].resid_cs == 0 && N <= 128 && gemm_f16_small_n_ksplit(a) > 1) {
        const ggml_tensor * A = g->nodes[add_idx[0]];
        const int u = sole_user(s, A);
        if (u > add_idx[0] && !s.done[u] && next_real_node(s, add_idx[0]) == u && g->nodes[u]->op == GGML_OP_UNARY && !is_out(s, A)) {
            const ggml_tensor * U = g->nodes[u];
            const int uop = op_param_i32(U, 0);
            const ggml_tensor * xg = nullptr;
            if ((uop == GGML_UNARY_OP_GELU || uop == GGML_UNARY_OP_GELU_QUICK) && U->src[0] == A && U->type == GGML_TYPE_F32 && same_shape(U, A) && U->ne[2] == 1 && U->ne[3] == 1 && U->ne[0] % 8 == 0 &&
                U->nb[0] == 4 && U->nb[1] == (size_t) U->ne[0] * 4 && ((uintptr_t) U->data & 15) == 0 && gemm_only_consumers(s, U, U->ne[0], U->ne[1], &xg)) {
                const int u1 = sole_user(s, U);
                a.m[0].unary = uop; a.m[0].y16 = (uint16_t *) s.c->act_scratch; a.m[0].y16_rs = act_image_bytes(ACT_F16, U->ne[0]);
                a.m[0].y32 = !(u1 > u && next_real_node(s, u) == u1);
                a.m[0].dst = (float *) U->data; a.m[0].dst_cs = U->nb[1];
                un_idx = u; un_x = xg;
            }
        }
    }
    int add2_idx[3] = { -1, -1, -1 };
    static const bool no_add2 = getenv("MI355X_NO_GEMM_ADD2") != nullptr;
    if (!no_add2 && un_idx < 0 && N <= 128 && gemm_f16_small_n_ksplit(a) > 1)
        for (int q = 0; q < a.nmat; ++q) {
            if (add_idx[q] < 0 || a.m[q].resid_cs != 0) continue;                       // (first addend: a bias row)
            ggml_tensor * A = g->nodes[add_idx[q]];
            const int a2 = sole_user(s, A);
            if (a2 <= add_idx[q] || g->nodes[a2]->op != GGML_OP_ADD || s.done[a2]) continue;
            ggml_tensor * A2 = g->nodes[a2];
            const ggml_tensor * r2 = A2->src[0] == A ? A2->src[1] : A2->src[0];
            if (((A2->src[0] == A) == (A2->src[1] == A)) || !r2 || r2 == A || r2->type != GGML_TYPE_F32 || A2->type != GGML_TYPE_F32 || !same_shape(r2, A) || !same_shape(A2, A) ||
                r2->nb[0] != 4 || A2->nb[0] != 4 || r2->nb[1] % 16 != 0 || A2->nb[1] % 16 != 0 || A->ne[2] * A->ne[3] != 1) continue;
            int item[10]; int ni = 0;
            for (int t = 0; t < a.nmat; ++t) { item[ni++] = mm_idx[t]; if (add_idx[t] >= 0) item[ni++] = add_idx[t]; }
            for (int t = 0; t < q; ++t) if (add2_idx[t] >= 0) item[ni++] = add2_idx[t];
            item[ni++] = a2;
            if (!can_hoist(s, i, a2, item, ni)) continue;
            a.m[q].resid2 = (const float *) r2->data; a.m[q].resid2_cs = r2->nb[1];
            a.m[q].dst = (float *) A2->data; a.m[q].dst_cs = A2->nb[1];
            add2_idx[q] = a2;
        }
    // ... and the CPY of a streaming encoder's new K / V rows into its f16 cache (audition.cpp:519-556: Kcur -> a contiguous run of the K cache; Vcur (+ bias) -> TRANSPOSE -> a
    // [n_tokens, n_state] view of the transposed V cache, rows a cache pitch apart): the only reader of the f32 rows, so the reduction writes the f16 cells itself and the f32
    // rows never exist
    int cpy_idx[3] = { -1, -1, -1 };
    static const bool no_cpy16 = getenv("MI355X_NO_GEMM_CPY16") != nullptr;
    bool y16_path = N <= 128 && gemm_f16_small_n_ksplit(a) > 1;          // the reduction launch writes the f16 rows ...
    if (!no_cpy16 && un_idx < 0 && !y16_path && !kq) {                    // ... and so does the tile epilogue of the k_gemm_f16_glds<MB> family (an encoder's K / V CAST at full length)
        gemm_multi_args pa = a; int path = 0; pa.probe_path = &path;
        gemm_f16_multi(pa, s.st);
        y16_path = path == 1;
    }
    if (!no_cpy16 && un_idx < 0 && y16_path)
        for (int q = 0; q < a.nmat; ++q) {
            if (add2_idx[q] >= 0 || a.m[q].M % 4 != 0) continue;
            const int ri = add_idx[q] >= 0 ? add_idx[q] : mm_idx[q];
            const ggml_tensor * R = g->nodes[ri];
            if (is_out(s, R) || R->ne[2] != 1 || R->ne[3] != 1 || R->nb[0] != 4 || R->nb[1] != (size_t) R->ne[0] * 4) continue;
            const ggml_tensor * t = R; int cj = -1;
            for (int hop = 0; hop < 5; ++hop) {
                const int u = sole_user(s, t);
                if (u < 0) break;
                if (g->nodes[u]->op == GGML_OP_CPY) { cj = u; break; }
                if (!is_noop(g->nodes[u])) break;
                t = g->nodes[u];
            }
            if (cj <= ri || s.done[cj]) continue;
            const ggml_tensor * Cp = g->nodes[cj], * S = Cp->src[0];
            if (Cp->type != GGML_TYPE_F16 || !S || S->type != GGML_TYPE_F32 || S->data != R->data || nelements(S) != nelements(R) || nelements(Cp) != nelements(R) || is_out(s, Cp)) continue;
            { const ggml_tensor * w = S; while (w && w != R) w = w->view_src; if (!w) continue; }
            const int64_t M = R->ne[0];
            size_t ms = 0, rs = 0;
            if (is_contiguous(S) && is_contiguous(Cp)) { ms = 2; rs = (size_t) M * 2; }                                                  // same linear order: K rows
            else if (S->ne[0] == N && S->ne[1] == M && S->ne[2] == 1 && S->ne[3] == 1 && S->nb[0] == R->nb[1] && S->nb[1] == 4 &&
                     Cp->ne[0] == N && Cp->ne[1] == M && Cp->ne[2] == 1 && Cp->ne[3] == 1 && Cp->nb[0] == 2 && Cp->nb[1] % 2 == 0) { ms = Cp->nb[1]; rs = 2; }   // the transposed view: V rows
            else if (S->ne[0] == N && S->ne[1] > 0 && S->ne[1] * S->ne[2] == M && S->ne[3] == 1 && S->nb[0] == R->nb[1] && S->nb[1] == 4 && S->nb[2] == (size_t) S->ne[1] * 4 &&
                     is_contiguous(Cp) && Cp->ne[0] == N && Cp->ne[1] == S->ne[1] && Cp->ne[2] == S->ne[2] && Cp->ne[3] == 1) { ms = (size_t) N * 2; rs = 2; }       // [n_tokens, D, H] of PERMUTE(1, 2, 0, 3): V^T per head (an encoder's V CAST)
            else continue;
            if (ms == 2 && (((uintptr_t) Cp->data & 7) != 0 || rs % 8 != 0)) continue;
            int item[8]; int ni = 0;
            for (int k = 0; k < a.nmat; ++k) { item[ni++] = mm_idx[k]; if (add_idx[k] >= 0) item[ni++] = add_idx[k]; }
            for (int k = 0; k < q; ++k) if (cpy_idx[k] >= 0 && ni < 8) item[ni++] = cpy_idx[k];
            if (!can_hoist(s, i, cj, item, ni)) continue;
            a.m[q].y16 = (uint16_t *) Cp->data; a.m[q].y16_ms = ms; a.m[q].y16_rs = rs; a.m[q].y32 = false;
            if (add_idx[q] < 0) { a.m[q].dst = (float *) R->data; a.m[q].dst_cs = R->nb[1]; }
            cpy_idx[q] = cj;
        }
    double flops = 0;
    for (int q = 0; q < a.nmat; ++q) flops += 2.0 * (double) a.m[q].M * (double) N * (double) K;
    // a split-K result whose next reader is RMS_NORM (wo / ffn_down + residual -> the next norm): leave the slabs, the norm reduces them
    int nsplit = 0;
    const ggml_tensor * Aout = a.nmat == 1 ? g->nodes[add2_idx[0] >= 0 ? add2_idx[0] : (add_idx[0] >= 0 ? add_idx[0] : i)] : nullptr;
    static const bool no_defer_reduce = getenv("MI355X_NO_REDUCE_IN_NORM") != nullptr;
    if (!no_defer_reduce && un_idx < 0 && a.partial && Aout && Aout->ne[2] == 1 && Aout->ne[3] == 1 && gemm_reduce_rms_norm_ok(Aout->ne[0]) && Aout->nb[1] % 16 == 0 &&
        (!a.m[0].resid || a.m[0].resid_cs % 16 == 0)) {
        int nx = (add2_idx[0] >= 0 ? add2_idx[0] : (add_idx[0] >= 0 ? add_idx[0] : i)) + 1;
        while (nx < g->n_nodes && (s.done[nx] || is_noop(g->nodes[nx]) || nx == add_idx[0] || nx == add2_idx[0])) ++nx;
        if (nx < g->n_nodes && g->nodes[nx]->op == GGML_OP_RMS_NORM && g->nodes[nx]->src[0] == Aout && add2_idx[0] < 0) a.deferred_split = &nsplit;
        // ... or a LayerNorm (the encoders' wo / fc2 + bias + residual -> ln): k_norm_rows sums the slabs and both addends itself (exec_norm decides; it falls back to the
        // reduction launch when it cannot take the row)
        static const bool no_defer_ln = getenv("MI355X_NO_REDUCE_IN_LAYER_NORM") != nullptr;
        if (!no_defer_ln && nx < g->n_nodes && g->nodes[nx]->op == GGML_OP_NORM && g->nodes[nx]->src[0] == Aout && Aout->ne[0] <= 4096 && Aout->ne[1] >= 2 &&
            (!a.m[0].resid2 || a.m[0].resid2_cs % 16 == 0)) a.deferred_split = &nsplit;
    }
    // a grouped launch of a prefill ubatch (wq / wk / wv, two K halves at 512 tokens) whose results go straight into the q / k norm + rope + store launch: leave the slabs, that
    // launch sums them (k_norm_rope_v4 with slab sources) -- judged here only by the next launching node being an RMS_NORM on one of the results; exec_rms_norm takes the slabs when
    // every chain of its launch maps onto them and runs the reduction launch itself otherwise
    bool group_deferred = false;
    static const bool no_defer_group = getenv("MI355X_NO_REDUCE_IN_NORM_ROPE") != nullptr;
    if (!no_defer_group && !no_defer_reduce && a.nmat >= 2 && un_idx < 0 && a.partial && (!kq || qt) && N > MI_MMVQ_MAX_COLS && !s.prm.n) {
        bool ok = true;
        for (int q = 0; q < a.nmat && ok; ++q) {
            const ggml_tensor * R = g->nodes[mm_idx[q]];
            ok = add_idx[q] < 0 && add2_idx[q] < 0 && cpy_idx[q] < 0 && !a.m[q].resid && R->ne[2] == 1 && R->ne[3] == 1 && R->nb[1] == (size_t) R->ne[0] * 4 && a.m[q].M % 4 == 0 && !is_out(s, R) &&
                 n_users(s, R) == 1;      // (the norm chain / V store must be R's ONLY reader -- a second one would read rows materialise_group() skipped)
        }
        int nx = i + 1;                                                   // (the group's other mat-muls were hoisted up to node i: skip them)
        auto mine = [&](int k) { for (int q = 0; q < a.nmat; ++q) if (mm_idx[q] == k) return true; return false; };
        while (nx < g->n_nodes && (s.done[nx] || is_noop(g->nodes[nx]) || mine(nx))) ++nx;
        bool hit = false;
        if (ok && nx < g->n_nodes && g->nodes[nx]->op == GGML_OP_RMS_NORM && g->nodes[nx]->src[0])
            for (int q = 0; q < a.nmat; ++q) hit = hit || g->nodes[nx]->src[0]->data == g->nodes[mm_idx[q]]->data;
        if (ok && hit) { a.deferred_split = &nsplit; a.defer_multi = true; group_deferred = true; }
    }
    {
        prof_scope ps(s, qt ? "mmq_tile" : "gemm_f16", flops);
        gemm_f16_multi(a, s.st);
    }
    ++s.n_kernels;
    if (group_deferred && nsplit > 1) {
        s.prm.n = a.nmat; s.prm.nsplit = nsplit; s.prm.N = N;
        size_t off = 0;
        for (int q = 0; q < a.nmat; ++q) { s.prm.A[q] = g->nodes[mm_idx[q]]; s.prm.off[q] = off; s.prm.M[q] = a.m[q].M; off += (size_t) a.m[q].M * (size_t) N; }
        s.prm.slab = off;
        nsplit = 0;
    }
    if (nsplit > 1) { s.pr.A = Aout; s.pr.nsplit = nsplit; s.pr.resid = a.m[0].resid; s.pr.resid_cs = a.m[0].resid_cs; s.pr.resid2 = a.m[0].resid2; s.pr.resid2_cs = a.m[0].resid2_cs; }
    if (un_idx >= 0) {                                          // (the bias ADD's rows are never written: its one reader ran in the reduction)
        s.done[add_idx[0]] = 1; s.done[un_idx] = 1; s.n_fused += 2;
        note_write(s, g->nodes[un_idx]);
        seed_act_f16(s, un_x);
        return true;
    }
    for (int q = 0; q < a.nmat; ++q) if (cpy_idx[q] >= 0) { s.done[cpy_idx[q]] = 1; ++s.n_fused; note_write(s, g->nodes[cpy_idx[q]]); }
    for (int q = 0; q < a.nmat; ++q) {
        if (q > 0) { s.done[mm_idx[q]] = 1; ++s.n_fused; }
        if (add2_idx[q] >= 0) { s.done[add_idx[q]] = 1; s.done[add2_idx[q]] = 1; s.n_fused += 2; note_write(s, g->nodes[add2_idx[q]]); }
        else if (add_idx[q] >= 0) { s.done[add_idx[q]] = 1; ++s.n_fused; note_write(s, g->nodes[add_idx[q]]); }
        else note_write(s, g->nodes[mm_idx[q]]);
    }
    return true;
}

// MUL_MAT at node i: try gate/up/SWIGLU, then q/k/v batching, then residual-add epilogue; falls back to the plain path
static void exec_mul_mat(exec_state & s, int i) {
    ggml_cgraph * g = s.g;
    ggml_tensor * n = g->nodes[i];
    if (s.c->opt_fusion && exec_gemm_group(s, i)) return;
    const bool q80 = s.c->opt_fusion && q80_mv1_node(s, n);                  // Q8_0, one column: the same fusions on mmv1q.hip
    if (s.c->opt_fusion && mm_takes_gemm_any(n) && !is_out(s, n)) {
        // the bias ADD behind an F32-weight / odd-K linear layer (Token2Wav's DiT and HiFT blocks): a [M] row vector, the only reader, the next launch -> the GEMM's epilogue
        static const bool off = getenv("MI355X_NO_GEMM_ANY_BIAS") != nullptr;
        const int ai = off ? -1 : sole_user(s, n);
        if (ai > i && next_real_node(s, i) == ai && g->nodes[ai]->op == GGML_OP_ADD) {
            const ggml_tensor * A = g->nodes[ai];
            const ggml_tensor * r = A->src[0] == n ? A->src[1] : (A->src[1] == n ? A->src[0] : nullptr);
            bool ok = r && r != n && r->type == GGML_TYPE_F32 && A->type == GGML_TYPE_F32 && r->ne[0] == n->ne[0] && r->ne[1] * r->ne[2] * r->ne[3] == 1 && r->nb[0] == 4 && n->ne[1] > 1;
            for (int d = 0; ok && d < 4; ++d) ok = A->ne[d] == n->ne[d] && A->nb[d] == n->nb[d];
            // (ggml-alloc may have given the ADD's result the memory of the mat-mul's dead operands: the launch reads them while it writes the result)
            ok = ok && !overlap(range_of(A), range_of(n->src[0])) && !overlap(range_of(A), range_of(n->src[1])) && !overlap(range_of(A), range_of(r));
            if (ok) {
                // the block's other projections of the same activation (q / k / v of a DiT block: F32 weights of one shape, each with its bias ADD behind it) join the launch
                static const bool no_group = getenv("MI355X_NO_GEMM_ANY_GROUP") != nullptr;
                mm_sibling sib[2]; int sib_mm[2], sib_add[2], nsib = 0;
                int item[6] = { i, ai, -1, -1, -1, -1 }; int ni = 2;
                const ggml_tensor * w0 = n->src[0], * x0 = n->src[1];
                for (int j = ai + 1; !no_group && j < g->n_nodes && j < i + 40 && nsib < 2; ++j) {
                    const ggml_tensor * c = g->nodes[j];
                    if (s.done[j] || c->op != GGML_OP_MUL_MAT || c->src[1] != x0 || !mm_takes_gemm_any(c) || is_out(s, c)) continue;
                    const ggml_tensor * wc = c->src[0];
                    if (wc->type != GGML_TYPE_F32 || w0->type != GGML_TYPE_F32 || !wc->data || wc == w0) continue;
                    bool same = true;
                    for (int d = 0; d < 4; ++d) same = same && wc->ne[d] == w0->ne[d] && wc->nb[d] == w0->nb[d] && c->ne[d] == n->ne[d] && c->nb[d] == n->nb[d];
                    if (!same) continue;
                    const int aj = sole_user(s, c);
                    if (aj <= j || next_real_node(s, j) != aj || g->nodes[aj]->op != GGML_OP_ADD) continue;
                    const ggml_tensor * Aj = g->nodes[aj];
                    const ggml_tensor * rj = Aj->src[0] == c ? Aj->src[1] : (Aj->src[1] == c ? Aj->src[0] : nullptr);
                    bool okj = rj && rj != c && rj->type == GGML_TYPE_F32 && Aj->type == GGML_TYPE_F32 && rj->ne[0] == c->ne[0] && rj->ne[1] * rj->ne[2] * rj->ne[3] == 1 && rj->nb[0] == 4 && rj->data;
                    for (int d = 0; okj && d < 4; ++d) okj = Aj->ne[d] == c->ne[d] && Aj->nb[d] == c->nb[d];
                    okj = okj && !overlap(range_of(Aj), range_of(wc)) && !overlap(range_of(Aj), range_of(x0)) && !overlap(range_of(Aj), range_of(rj)) && !overlap(range_of(Aj), range_of(A)) &&
                          !overlap(range_of(Aj), range_of(w0)) && !overlap(range_of(Aj), range_of(r));
                    for (int q = 0; okj && q < nsib; ++q) okj = !overlap(range_of(Aj), range_of(sib[q].out)) && !overlap(range_of(Aj), range_of(sib[q].w));
                    if (!okj) continue;
                    int it2[6]; for (int q = 0; q < ni; ++q) it2[q] = item[q];
                    it2[ni] = j; it2[ni + 1] = aj;
                    if (!can_hoist(s, i, j, it2, ni + 2) || !can_hoist(s, i, aj, it2, ni + 2)) continue;
                    item[ni++] = j; item[ni++] = aj;
                    sib[nsib] = { wc, Aj, (const float *) rj->data }; sib_mm[nsib] = j; sib_add[nsib] = aj; ++nsib;
                }
                bool taken = false;
                op_mul_mat(s, n, A, (const float *) r->data, sib, nsib, &taken);
                s.done[ai] = 1; ++s.n_fused;
                note_write(s, A);
                if (taken) for (int q = 0; q < nsib; ++q) { s.done[sib_mm[q]] = 1; s.done[sib_add[q]] = 1; s.n_fused += 2; note_write(s, sib[q].out); }
                return;
            }
        }
    }
    if (!s.c->opt_fusion || (!kq_mm_ok(n) && !q80)) { op_mul_mat(s, n); note_write(s, n); return; }
    const ggml_tensor * x = n->src[1];
    const int64_t K = x->ne[0]; const int N = (int) x->ne[1];

    // ---- (a) ffn_up / ffn_gate + GLU(SWIGLU, split): one launch, intermediates never written
    const bool use_mmq = mm_uses_mmq(n);
    if (!use_mmq) {
        const int gi = sole_user(s, n);
        if (gi > i && g->nodes[gi]->op == GGML_OP_GLU && op_param_i32(g->nodes[gi], 0) == GGML_GLU_OP_SWIGLU && op_param_i32(g->nodes[gi], 1) == 0 &&
            g->nodes[gi]->src[0] && g->nodes[gi]->src[1]) {
            ggml_tensor * G = g->nodes[gi];
            ggml_tensor * other = G->src[0] == n ? G->src[1] : (G->src[1] == n ? G->src[0] : nullptr);
            auto oit = other ? s.index.find(other) : s.index.end();
            if (other && oit != s.index.end() && oit->second > i && !s.done[oit->second] && (q80 ? q80_mv1_node(s, other) : plain_kq_matvec(other, MI_MMVQ_MAX_COLS)) && sole_user(s, other) == gi &&
                same_act(other->src[1], x) && other->src[0]->type == n->src[0]->type && other->src[0]->ne[1] == n->src[0]->ne[1] &&
                other->src[0]->nb[1] == n->src[0]->nb[1] && G->nb[0] == 4 && G->ne[0] == n->ne[0] && is_contiguous_1(G)) {
                const int oi = oit->second;
                const int item[3] = { i, oi, gi };
                if (can_hoist(s, i, oi, item, 3) && can_hoist(s, i, gi, item, 3)) {
                    mmv_norm nrm;
                    const ggml_tensor * outs[1] = { G };
                    const ggml_tensor * gate_n = G->src[0], * up_n = G->src[1];
                    if (N == 1 && mv1_node_ok(s, gate_n) && mv1_node_ok(s, up_n) && ((uintptr_t) G->data & 3) == 0) {
                        mv1_args v; v.nmat = 1; v.K = K;
                        v.m[0] = { gate_n->src[0]->data, gate_n->src[0]->nb[1], (float *) G->data, 0, nullptr, 0, gate_n->src[0]->ne[1], (int) gate_n->src[0]->type };
                        v.W_up = up_n->src[0]->data;
                        mv1_source(s, x, outs, 1, 2, v);
                        prof_scope ps(s, q80 ? (n->src[0]->type == GGML_TYPE_F16 ? "mmv_f16" : "mmv_q80") : (n->src[0]->type == GGML_TYPE_Q4_K ? "mmv_q4k" : "mmv_q6k"), 2.0 * (double) n->src[0]->ne[1] * (double) row_size(n->src[0]->type, K));
                        mmv1(v, s.st);
                        ++s.n_kernels; s.n_fused += 2;
                        s.done[oi] = s.done[gi] = 1;
                        note_write(s, G);
                        return;
                    }
                    if (q80) { fprintf(stderr, "[mi355x] exec_mul_mat: a Q8_0 gate / up pair that mmv1q refuses\n"); abort(); }   // (q80_mv1_node accepted both halves)
                    const size_t img = norm_in_kernel(s, x, outs, 1, 2, nrm) ? q8k_image_bytes(K) : prepare_act(s, x, ACT_Q8K);
                    const ggml_tensor * gate = G->src[0], * up = G->src[1];
                    prof_scope ps(s, n->src[0]->type == GGML_TYPE_Q4_K ? "mmv_q4k" : "mmv_q6k", 2.0 * (double) n->src[0]->ne[1] * (double) row_size(n->src[0]->type, K));
                    mmv_kquant_pair_swiglu(n->src[0]->type, gate->src[0]->data, up->src[0]->data, n->src[0]->nb[1], s.c->act_scratch, img,
                                           (float *) G->data, G->nb[1], K, n->src[0]->ne[1], N, s.st, &nrm);
                    ++s.n_kernels; s.n_fused += 2;
                    s.done[oi] = s.done[gi] = 1;
                    note_write(s, G);
                    return;
                }
            }
        }
    }

    // ---- (b) batch MUL_MATs that consume the same activation (wq / wk / wv), each with an optional residual ADD
    int   mm_idx[3] = { i, -1, -1 }; int nm = 1;
    for (int j = i + 1; j < g->n_nodes && j < i + 32 && nm < 3; ++j) {
        ggml_tensor * c = g->nodes[j];
        if (s.done[j] || !(q80 ? (q80_mv1_node(s, c) && c->src[0]->type == n->src[0]->type) : kq_mm_ok(c)) || !same_act(c->src[1], x)) continue;
        // do not steal one half of a gate/up pair (that fusion is worth more; it exists for the mat-vec widths only)
        const int cu = sole_user(s, c);
        if (!use_mmq && cu > 0 && g->nodes[cu]->op == GGML_OP_GLU) continue;
        if (!can_hoist(s, i, j, mm_idx, nm)) continue;
        mm_idx[nm++] = j;
    }
    mmv_multi_args a;
    a.nmat = nm; a.K = K; a.ncols = N;
    int add_idx[3] = { -1, -1, -1 };
    double bytes_q4 = 0, bytes_q6 = 0;
    for (int q = 0; q < nm; ++q) {
        ggml_tensor * c = g->nodes[mm_idx[q]];
        const ggml_tensor * w = c->src[0];
        a.m[q] = { w->data, w->nb[1], (float *) c->data, c->nb[1], nullptr, 0, w->ne[1], (int) w->type };
        (w->type == GGML_TYPE_Q4_K ? bytes_q4 : bytes_q6) += (double) w->ne[1] * (double) row_size(w->type, K);
        // residual: the only consumer is ADD(c, r) / ADD(r, c) with r of the same shape, available now
        const int ai = sole_user(s, c);
        if (ai > mm_idx[q] && g->nodes[ai]->op == GGML_OP_ADD && !s.done[ai]) {
            ggml_tensor * A = g->nodes[ai];
            const ggml_tensor * r = A->src[0] == c ? A->src[1] : A->src[0];
            if (((A->src[0] == c) != (A->src[1] == c)) && r && r != c && r->type == GGML_TYPE_F32 && same_shape(r, c) && same_shape(A, c) && r->nb[0] == 4 && A->nb[0] == 4 && A->type == GGML_TYPE_F32) {
                int item[7]; int ni = 0;
                for (int t = 0; t < nm; ++t) item[ni++] = mm_idx[t];
                for (int t = 0; t < q; ++t) if (add_idx[t] >= 0) item[ni++] = add_idx[t];
                item[ni++] = ai;
                if (can_hoist(s, i, ai, item, ni)) {
                    a.m[q].resid = (const float *) r->data; a.m[q].resid_cs = r->nb[1];
                    a.m[q].dst = (float *) A->data; a.m[q].dst_cs = A->nb[1];
                    add_idx[q] = ai;
                }
            }
        }
    }
    if (N == 1 && !use_mmq) {
        // the launch as a whole (the node checks above looked at every matrix alone, without its residual): e.g. a residual on a matrix of more rows than
        // the engine's residual staging holds -- drop the epilogue fusion rather than the batch-1 kernel
        mv1_args t; t.nmat = nm; t.K = K; t.img = (const void *) 16;
        for (int q = 0; q < nm; ++q) t.m[q] = a.m[q];
        if (!mmv1_ok(t)) {
            for (int q = 0; q < nm; ++q) if (add_idx[q] >= 0) {
                ggml_tensor * c = g->nodes[mm_idx[q]];
                a.m[q].resid = nullptr; a.m[q].resid_cs = 0; a.m[q].dst = (float *) c->data; a.m[q].dst_cs = c->nb[1];
                add_idx[q] = -1;
            }
        }
    }
    const ggml_tensor * outs[3] = { nullptr, nullptr, nullptr };
    for (int q = 0; q < nm; ++q) outs[q] = add_idx[q] >= 0 ? g->nodes[add_idx[q]] : g->nodes[mm_idx[q]];
    bool all_mv1 = N == 1 && !use_mmq;
    for (int q = 0; q < nm && all_mv1; ++q) all_mv1 = mv1_node_ok(s, g->nodes[mm_idx[q]]) && ((uintptr_t) a.m[q].dst & 3) == 0 && ((uintptr_t) a.m[q].resid & 3) == 0;
    if (all_mv1) {
        mv1_args v; v.nmat = nm; v.K = K;
        for (int q = 0; q < nm; ++q) v.m[q] = a.m[q];
        mv1_source(s, x, outs, nm, nm, v);
        {
            prof_scope ps(s, q80 ? (n->src[0]->type == GGML_TYPE_F16 ? "mmv_f16" : "mmv_q80") : (bytes_q4 >= bytes_q6 ? "mmv_q4k" : "mmv_q6k"), bytes_q4 + bytes_q6);
            mmv1(v, s.st);
        }
        ++s.n_kernels;
        for (int q = 0; q < nm; ++q) {
            if (q > 0) { s.done[mm_idx[q]] = 1; ++s.n_fused; }
            if (add_idx[q] >= 0) { s.done[add_idx[q]] = 1; ++s.n_fused; note_write(s, g->nodes[add_idx[q]]); }
            else note_write(s, g->nodes[mm_idx[q]]);
        }
        return;
    }
    if (q80) { fprintf(stderr, "[mi355x] exec_mul_mat: a Q8_0 batch that mmv1q refuses\n"); abort(); }                 // (every member passed q80_mv1_node)
    const size_t img = norm_in_kernel(s, x, outs, nm, nm, a.norm) ? q8k_image_bytes(K) : prepare_act(s, x, ACT_Q8K);
    a.act = s.c->act_scratch; a.act_cs = img;
    if (use_mmq) {                                                    // int8 matrix cores, 32 columns per launch
        for (int c0 = 0; c0 < N; c0 += 32) {
            mmq_args q;
            q.nmat = nm; q.act = (const char *) s.c->act_scratch + (size_t) c0 * img; q.act_cs = img; q.K = K; q.ncols = N - c0 < 32 ? N - c0 : 32;
            for (int t = 0; t < nm; ++t) {
                const mmv_mat & m = a.m[t];
                q.m[t].W = m.W; q.m[t].w_rs = m.w_rs; q.m[t].dst = (float *) ((char *) m.dst + (size_t) c0 * m.dst_cs); q.m[t].dst_cs = m.dst_cs;
                q.m[t].nrows = m.nrows; q.m[t].type = m.type;
                q.m[t].resid = m.resid ? (const float *) ((const char *) m.resid + (size_t) c0 * m.resid_cs) : nullptr; q.m[t].resid_cs = m.resid_cs;
            }
            prof_scope ps(s, bytes_q4 >= bytes_q6 ? "mmq_q4k" : "mmq_q6k", bytes_q4 + bytes_q6);
            mmq_kquant(q, s.st); ++s.n_kernels;
        }
    } else {
        // profile class: the launch is attributed to the type that carries most of its bytes
        prof_scope ps(s, bytes_q4 >= bytes_q6 ? "mmv_q4k" : "mmv_q6k", bytes_q4 + bytes_q6);
        mmv_kquant_multi(a, s.st);
        ++s.n_kernels;
    }
    for (int q = 0; q < nm; ++q) {
        if (q > 0) { s.done[mm_idx[q]] = 1; ++s.n_fused; }
        if (add_idx[q] >= 0) { s.done[add_idx[q]] = 1; ++s.n_fused; note_write(s, g->nodes[add_idx[q]]); }
        else note_write(s, g->nodes[mm_idx[q]]);
    }
}

// RMS_NORM(j) -> MUL(w[D]) -> ROPE [-> SET_ROWS of the rotated rows viewed as [D*H, T] into an f16 table]; shape checks only
struct nr_chain {
    int norm, mul, rope, store;                    // norm / mul = -1: a ROPE-only chain (llama architecture: no q / k norm)
    const ggml_tensor * wt, * pos, * ff;           // wt = null: no norm
    const ggml_tensor * xin; int first;            // the f32 heads the chain starts from, and the chain's first node
    int D, H, T; float eps; rope_params rp;
};
static bool match_norm_rope(exec_state & s, int j, nr_chain & c) {
    ggml_cgraph * g = s.g;
    ggml_tensor * n = g->nodes[j];
    const int mi_ = sole_user(s, n);
    if (mi_ <= j || g->nodes[mi_]->op != GGML_OP_MUL || s.done[mi_]) return false;
    ggml_tensor * m = g->nodes[mi_];
    if ((m->src[0] == n) == (m->src[1] == n)) return false;
    const ggml_tensor * wt = m->src[0] == n ? m->src[1] : m->src[0];
    const int64_t D = n->ne[0];
    if (!wt || m->type != GGML_TYPE_F32 || wt->type != GGML_TYPE_F32 || wt->nb[0] != 4 || !same_shape(m, n) || m->nb[0] != 4 ||
        wt->ne[0] != D || wt->ne[1] * wt->ne[2] * wt->ne[3] != 1 || n->src[0]->nb[0] != 4 || n->src[0]->type != GGML_TYPE_F32) return false;
    const int ri = sole_user(s, m);
    if (!(ri > mi_ && g->nodes[ri]->op == GGML_OP_ROPE && g->nodes[ri]->src[0] == m && !s.done[ri] && D % 2 == 0 && D <= 256 && n->ne[3] == 1)) return false;
    ggml_tensor * r = g->nodes[ri];
    const int mode = op_param_i32(r, 2);
    const ggml_tensor * pos = r->src[1], * ff = r->src[2];
    if (!((mode == GGML_ROPE_TYPE_NORMAL || mode == GGML_ROPE_TYPE_NEOX) && op_param_i32(r, 1) == D && r->nb[0] == 4 && pos && pos->type == GGML_TYPE_I32 &&
          pos->nb[0] == 4 && (!ff || (ff->type == GGML_TYPE_F32 && ff->nb[0] == 4)))) return false;
    c.norm = j; c.mul = mi_; c.rope = ri; c.store = -1; c.wt = wt; c.pos = pos; c.ff = ff; c.xin = n->src[0]; c.first = j;
    c.D = (int) D; c.H = (int) n->ne[1]; c.T = (int) n->ne[2]; c.eps = op_param_f32(n, 0);
    memset(&c.rp, 0, sizeof(c.rp));
    c.rp.n_dims = op_param_i32(r, 1); c.rp.mode = mode; c.rp.n_ctx_orig = op_param_i32(r, 4);
    c.rp.freq_base = op_param_f32(r, 5); c.rp.freq_scale = op_param_f32(r, 6); c.rp.ext_factor = op_param_f32(r, 7);
    c.rp.attn_factor = op_param_f32(r, 8); c.rp.beta_fast = op_param_f32(r, 9); c.rp.beta_slow = op_param_f32(r, 10);
    // optional store of the rotated rows (llama_kv_cache::cpy_k): the rope output's only consumer
    const int si = sole_user(s, r);
    if (si > ri && g->nodes[si]->op == GGML_OP_SET_ROWS && !s.done[si]) {
        const ggml_tensor * S = g->nodes[si], * V = S->src[0], * idx = S->src[1];
        if (V && idx && V->data == r->data && V->ne[0] == D * n->ne[1] && V->ne[1] == n->ne[2] && V->ne[2] == 1 && V->ne[3] == 1 &&
            V->nb[1] == r->nb[2] && r->nb[1] == (size_t) D * 4 && S->type == GGML_TYPE_F16 && S->nb[0] == 2 &&
            (idx->type == GGML_TYPE_I64 || idx->type == GGML_TYPE_I32) && idx->ne[0] == n->ne[2] && idx->ne[1] == 1 && idx->ne[2] == 1) c.store = si;
    }
    return true;
}
static norm_rope_job chain_job(exec_state & s, const nr_chain & c) {
    ggml_cgraph * g = s.g;
    const ggml_tensor * x = c.xin; ggml_tensor * r = g->nodes[c.rope];
    norm_rope_job j;
    j.x = (const float *) x->data; j.xnb1 = x->nb[1]; j.xnb2 = x->nb[2]; j.w = c.wt ? (const float *) c.wt->data : nullptr; j.rope_only = c.wt ? 0 : 1;
    j.y = (float *) r->data; j.ynb1 = r->nb[1]; j.ynb2 = r->nb[2];
    j.kv = nullptr; j.kv_rs = 0; j.idx = nullptr; j.idx_is64 = 0; j.idx_nb0 = 0; j.H = c.H;
    if (c.store >= 0) {
        const ggml_tensor * S = g->nodes[c.store], * idx = S->src[1];
        j.kv = S->data; j.kv_rs = S->nb[1]; j.idx = idx->data; j.idx_is64 = idx->type == GGML_TYPE_I64; j.idx_nb0 = idx->nb[0];
        j.y = nullptr;                                                    // the only consumer was the store
    }
    return j;
}

// Prefill: does every consumer of t read all of it as the [K, N] activation of a MUL_MAT that goes to the MFMA GEMM (directly or through
// a reshape of the same bytes)?  Then the producer can emit the f16 rows the GEMM wants and the separate conversion launch disappears.
static bool gemm_only_consumers(exec_state & s, const ggml_tensor * t, int64_t K, int64_t N, const ggml_tensor ** x_out) {
    static const bool off = getenv("MI355X_NO_F16_EMIT") != nullptr;
    if (off || !s.c->opt_fusion || is_out(s, t) || N <= MI_MMVQ_MAX_COLS) return false;
    auto it = s.users.find(t);
    if (it == s.users.end() || it->second.empty()) return false;
    if (act_image_bytes(ACT_F16, K) * (size_t) N > s.c->act_scratch_bytes) return false;
    const ggml_tensor * x0 = nullptr; act_kind k0 = ACT_F16;
    for (int u : it->second) {
        const ggml_tensor * c = s.g->nodes[u];
        if (c->op != GGML_OP_MUL_MAT || is_empty(c) || !mm_uses_gemm(c) || mm_uses_mmq_tile(c)) return false;      // (mmq_tile.hip reads the Q8_K image it builds from the f32 rows)
        const ggml_tensor * x = c->src[1];
        if (x->type != GGML_TYPE_F32 || x->data != t->data || x->ne[0] != K || x->ne[1] != N || x->ne[2] != 1 || x->ne[3] != 1 || x->nb[1] != (size_t) K * 4 ||
            c->src[0]->data == t->data) return false;
        if (x0 && (!same_act(x0, x) || gemm_act_kind(c) != k0)) return false;
        if (!x0) k0 = gemm_act_kind(c);
        x0 = x;
    }
    *x_out = x0;
    return true;
}
// which image do the GEMMs that read x want?  (gemm_only_consumers made sure they agree)
static act_kind consumers_act_kind(exec_state & s, const ggml_tensor * x) {
    auto it = s.users.find(x);
    if (it == s.users.end()) return ACT_F16;
    for (int u : it->second) { const ggml_tensor * c = s.g->nodes[u]; if (c->op == GGML_OP_MUL_MAT && c->src[1] && c->src[1]->data == x->data) return gemm_act_kind(c); }
    return ACT_F16;
}
static void seed_act_f16(exec_state & s, const ggml_tensor * x, bool quantised) {   // the f16 image of x now sits in act_scratch (quantised: the emitter wrote the Q8_K-quantised values already)
    const act_kind want = consumers_act_kind(s, x);
    if (want == ACT_F16Q && !quantised) {                                // K-quant consumers: re-quantise the rows in place (from f16: the emitting launch -- attention, SwiGLU -- has no f32 copy)
        prof_scope ps(s, "act_convert", 0);
        requant_f16_rows_q8k((uint16_t *) s.c->act_scratch, act_image_bytes(ACT_F16, x->ne[0]), x->ne[0], x->ne[1] * x->ne[2] * x->ne[3], s.st);
        ++s.n_kernels;
    }
    s.a_src = x->data; s.a_kind = want; s.a_K = x->ne[0]; s.a_ne[0] = x->ne[1]; s.a_ne[1] = 1; s.a_ne[2] = 1;
    s.a_nb[0] = x->nb[1]; s.a_nb[1] = x->nb[2]; s.a_nb[2] = x->nb[3];
    s.a_range_lo = (const char *) x->data; s.a_range_hi = (const char *) x->data + nbytes(x);
}

// The encoders' LayerNorm: NORM -> MUL by the [n] weight -> ADD of the [n] bias (audition.cpp / vision.cpp build_norm), each the next launching node
// and the only reader of the one before, on many rows: one launch of the wave-per-row kernel, which also emits the f16 image when only MFMA GEMMs
// read the result (wq / wk / wv, fc1).  Same three f32 roundings as the separate ops.
// LayerNorm -> MUL(n, scale) -> ADD(n, .) -> ADD(., shift) with scale / shift one row per dim-2 slice ([C, 1, B] views of the DiT's adaLN product, token2wav-impl.cpp:1121-1164):
// the three element-wise nodes ride in the norm launch's epilogue, rounded as they round.  The norm has exactly these two readers.
struct norm_mod_match { int mi_, a1i, a2i; const ggml_tensor * sv, * tv, * out; };
// `consecutive`: the norm's readers must be the launches right behind it (false: the caller checks with can_hoist that they may run at its own position)
static bool match_norm_modulate(exec_state & s, int i, norm_mod_match & M, bool consecutive) {
    static const bool off = getenv("MI355X_NO_NORM_FUSE") != nullptr;
    ggml_cgraph * g = s.g;
    const ggml_tensor * n = g->nodes[i];
    if (off || !s.c->opt_fusion || n->op != GGML_OP_NORM || is_out(s, n) || n->src[0]->type != GGML_TYPE_F32 || n->type != GGML_TYPE_F32 || !is_contiguous(n) || n->ne[3] != 1) return false;
    auto it = s.users.find(n);
    if (it == s.users.end() || it->second.size() != 2) return false;
    const int mi_ = it->second[0], a1i = it->second[1];
    if (mi_ <= i || a1i <= mi_ || s.done[mi_] || s.done[a1i]) return false;
    if (consecutive && (next_real_node(s, i) != mi_ || next_real_node(s, mi_) != a1i)) return false;
    const ggml_tensor * m = g->nodes[mi_], * a1 = g->nodes[a1i];
    auto row_vec = [&](const ggml_tensor * v) {              // one row of C floats per dim-2 slice (or one row altogether)
        return v && v->type == GGML_TYPE_F32 && v->data && v->ne[0] == n->ne[0] && v->ne[1] == 1 && (v->ne[2] == n->ne[2] || v->ne[2] == 1) && v->ne[3] == 1 && v->nb[0] == 4 &&
               v->nb[2] % 16 == 0 && ((uintptr_t) v->data & 15) == 0;
    };
    if (m->op != GGML_OP_MUL || m->src[0] != n || !row_vec(m->src[1]) || !same_shape(m, n) || !is_contiguous(m) || is_out(s, m) || sole_user(s, m) != a1i) return false;
    if (a1->op != GGML_OP_ADD || a1->src[0] != n || a1->src[1] != m || !same_shape(a1, n) || !is_contiguous(a1) || is_out(s, a1)) return false;
    const int a2i = sole_user(s, a1);
    if (a2i <= a1i || s.done[a2i] || (consecutive && next_real_node(s, a1i) != a2i)) return false;
    const ggml_tensor * a2 = g->nodes[a2i];
    if (a2->op != GGML_OP_ADD || a2->src[0] != a1 || !row_vec(a2->src[1]) || !same_shape(a2, n) || !is_contiguous(a2) || a2->type != GGML_TYPE_F32) return false;
    const ggml_tensor * sv = m->src[1], * tv = a2->src[1];
    if (!norm_rows_ok(td(n->src[0]), td(a2))) return false;
    if (overlap(range_of(a2), range_of(sv)) || overlap(range_of(a2), range_of(tv)) || (overlap(range_of(a2), range_of(n->src[0])) && a2->data != n->src[0]->data)) return false;
    M = { mi_, a1i, a2i, sv, tv, a2 };
    return true;
}
static bool exec_norm_modulate(exec_state & s, int i) {
    norm_mod_match M;
    if (!match_norm_modulate(s, i, M, true)) return false;
    const ggml_tensor * n = s.g->nodes[i];
    if (s.pr.A) materialise_reduce(s);
    if (s.prm.n) materialise_group(s);
    {
        prof_scope ps(s, "norm", 0);
        norm_rows_f32(td(n->src[0]), td(M.out), op_param_f32(n, 0), (const float *) M.sv->data, (const float *) M.tv->data, nullptr, 0, true, s.st,
                      M.sv->ne[2] > 1 ? M.sv->nb[2] / 4 : 0, M.tv->ne[2] > 1 ? M.tv->nb[2] / 4 : 0, true);
    }
    ++s.n_kernels;
    for (int k : { M.mi_, M.a1i, M.a2i }) { s.done[k] = 1; ++s.n_fused; }
    note_write(s, M.out);
    return true;
}
// The DiT's gated residual in front of that: MUL(y, gate) -> ADD(resid, .) = x, whose LayerNorm + modulation follows (possibly behind a few unrelated small copies --
// the convolution caches' -- which the chain is hoisted over when can_hoist allows): x is computed and written in the norm launch.  `i` is the MUL.
static bool exec_gate_norm(exec_state & s, int i) {
    static const bool off = getenv("MI355X_NO_GATE_NORM") != nullptr;
    ggml_cgraph * g = s.g;
    const ggml_tensor * m = g->nodes[i];
    if (off || !s.c->opt_fusion || m->op != GGML_OP_MUL || m->type != GGML_TYPE_F32 || !is_contiguous(m) || m->ne[3] != 1 || is_out(s, m) || m->view_src) return false;
    const ggml_tensor * y = m->src[0], * gv = m->src[1];
    if (!y || !gv || y->type != GGML_TYPE_F32 || !is_contiguous(y) || !same_shape(y, m) || !y->data) return false;
    if (gv->type != GGML_TYPE_F32 || !gv->data || gv->ne[0] != m->ne[0] || gv->ne[1] != 1 || (gv->ne[2] != m->ne[2] && gv->ne[2] != 1) || gv->ne[3] != 1 || gv->nb[0] != 4 || gv->nb[2] % 16 != 0) return false;
    const int ai = sole_user(s, m);
    if (ai <= i || next_real_node(s, i) != ai) return false;
    const ggml_tensor * a = g->nodes[ai];
    if (a->op != GGML_OP_ADD || a->src[1] != m || a->type != GGML_TYPE_F32 || !is_contiguous(a) || !same_shape(a, m) || a->view_src) return false;
    const ggml_tensor * r = a->src[0];
    if (!r || r->type != GGML_TYPE_F32 || !is_contiguous(r) || !same_shape(r, a) || !r->data) return false;
    // the LayerNorm of x among its readers, the first launching reader
    auto it = s.users.find(a);
    if (it == s.users.end()) return false;
    int ni = -1;
    for (int u : it->second) if (u > ai && g->nodes[u]->op == GGML_OP_NORM && g->nodes[u]->src[0] == a) { ni = u; break; }
    if (ni < 0 || s.done[ni] || ni > ai + 24) return false;
    for (int u : it->second) if (u < ni && u != ai) return false;               // somebody reads x before its norm: it must exist by then (keep the separate launches)
    norm_mod_match M;
    if (!match_norm_modulate(s, ni, M, false)) return false;
    const int item[6] = { i, ai, ni, M.mi_, M.a1i, M.a2i };
    if (next_real_node(s, ai) != ni || next_real_node(s, ni) != M.mi_ || next_real_node(s, M.mi_) != M.a1i || next_real_node(s, M.a1i) != M.a2i) {
        for (int k : { ni, M.mi_, M.a1i, M.a2i }) if (!can_hoist(s, ai, k, item, 6)) return false;
    }
    if (((uintptr_t) y->data | (uintptr_t) r->data | (uintptr_t) gv->data | (uintptr_t) a->data) & 15) return false;
    // x is written row by row while other rows of y / resid are still being read: it may sit exactly on one of them (same rows), not across
    if ((overlap(range_of(a), range_of(y)) && a->data != y->data) || (overlap(range_of(a), range_of(r)) && a->data != r->data) || overlap(range_of(a), range_of(gv))) return false;
    if (overlap(range_of(M.out), range_of(y)) || overlap(range_of(M.out), range_of(r)) || overlap(range_of(M.out), range_of(gv))) return false;
    const ggml_tensor * n = g->nodes[ni];
    if (s.pr.A) materialise_reduce(s);
    if (s.prm.n) materialise_group(s);
    if (s.pn.m && (s.pn.m == y || s.pn.m == r)) materialise_norm(s);
    {
        prof_scope ps(s, "norm", 0);
        const norm_gate ng = { (const float *) y->data, (const float *) r->data, (const float *) gv->data, gv->ne[2] > 1 ? gv->nb[2] / 4 : 0 };
        norm_rows_f32(td(a), td(M.out), op_param_f32(n, 0), (const float *) M.sv->data, (const float *) M.tv->data, nullptr, 0, true, s.st,
                      M.sv->ne[2] > 1 ? M.sv->nb[2] / 4 : 0, M.tv->ne[2] > 1 ? M.tv->nb[2] / 4 : 0, true, &ng);
    }
    ++s.n_kernels;
    for (int k : { ai, ni, M.mi_, M.a1i, M.a2i }) { s.done[k] = 1; ++s.n_fused; }
    note_write(s, a); note_write(s, M.out);
    return true;
}

static bool exec_norm(exec_state & s, int i) {
    static const bool off = getenv("MI355X_NO_NORM_FUSE") != nullptr;
    ggml_cgraph * g = s.g;
    const ggml_tensor * n = g->nodes[i];
    if (off || !s.c->opt_fusion || is_out(s, n) || n->src[0]->type != GGML_TYPE_F32) return false;
    auto vec_of = [&](const ggml_tensor * op, const ggml_tensor * in) -> const ggml_tensor * {
        const ggml_tensor * v = op->src[0] == in ? op->src[1] : (op->src[1] == in ? op->src[0] : nullptr);
        if (!v || v == in || v->type != GGML_TYPE_F32 || v->ne[0] != in->ne[0] || v->ne[1] * v->ne[2] * v->ne[3] != 1 || v->nb[0] != 4 || ((uintptr_t) v->data & 15) != 0) return nullptr;
        for (int d = 0; d < 4; ++d) if (op->ne[d] != in->ne[d] || op->nb[d] != in->nb[d]) return nullptr;
        return op->type == GGML_TYPE_F32 ? v : nullptr;
    };
    const int mi_ = sole_user(s, n);
    if (mi_ <= i || next_real_node(s, i) != mi_ || g->nodes[mi_]->op != GGML_OP_MUL) return false;
    const ggml_tensor * m = g->nodes[mi_];
    const ggml_tensor * wt = vec_of(m, n);
    if (!wt) return false;
    const ggml_tensor * out = m, * bt = nullptr;
    int ai = -1;
    if (!is_out(s, m)) {
        const int u = sole_user(s, m);
        if (u > mi_ && next_real_node(s, mi_) == u && g->nodes[u]->op == GGML_OP_ADD) {
            bt = vec_of(g->nodes[u], m);
            if (bt) { ai = u; out = g->nodes[u]; }
        }
    }
    if (!norm_rows_ok(td(n->src[0]), td(out))) return false;
    const int last = ai >= 0 ? ai : mi_;
    const ggml_tensor * xg = nullptr;
    const bool emit16 = out->ne[2] == 1 && out->ne[3] == 1 && out->nb[1] == (size_t) out->ne[0] * 4 && gemm_only_consumers(s, out, out->ne[0], out->ne[1], &xg);
    // the f32 rows may be skipped only when the single reader is the very next launch (the image is still in the scratch then)
    const int u1 = emit16 ? sole_user(s, out) : -1;
    const bool w32 = !(emit16 && u1 > last && next_real_node(s, last) == u1);
    // the rows still lie as split-K slabs of the mat-mul in front (+ bias / residual): summed, written and normalised in this launch
    const bool from_split = s.pr.A && s.pr.A == n->src[0];
    if (from_split && !norm_rows_from_split_ok(td(n->src[0]), td(out), s.pr.nsplit, s.pr.resid_cs, s.pr.resid2_cs, s.pr.resid, s.pr.resid2, s.c->gemm_partial)) materialise_reduce(s);
    {
        prof_scope ps(s, "norm", 0);
        if (s.pr.A && s.pr.A == n->src[0]) {
            norm_rows_from_split(td(n->src[0]), td(out), op_param_f32(n, 0), (const float *) wt->data, bt ? (const float *) bt->data : nullptr,
                                 emit16 ? (uint16_t *) s.c->act_scratch : nullptr, emit16 ? act_image_bytes(ACT_F16, out->ne[0]) : 0, w32,
                                 (const float *) s.c->gemm_partial, s.pr.nsplit, (size_t) n->src[0]->ne[0] * (size_t) n->src[0]->ne[1], s.pr.resid, s.pr.resid_cs, s.pr.resid2, s.pr.resid2_cs, s.st);
            s.pr.A = nullptr; ++s.n_fused;
        } else
        norm_rows_f32(td(n->src[0]), td(out), op_param_f32(n, 0), (const float *) wt->data, bt ? (const float *) bt->data : nullptr,
                      emit16 ? (uint16_t *) s.c->act_scratch : nullptr, emit16 ? act_image_bytes(ACT_F16, out->ne[0]) : 0, w32, s.st);
    }
    ++s.n_kernels;
    s.done[mi_] = 1; ++s.n_fused;
    if (ai >= 0) { s.done[ai] = 1; ++s.n_fused; }
    note_write(s, out);
    if (emit16) { seed_act_f16(s, xg); ++s.n_fused; }
    return true;
}

// Decode (one token, one sequence): can the layer's q chain, k chain + store and v store run INSIDE the attention kernel?  Needs the
// rope(q) output to be consumed by exactly one FLASH_ATTN_EXT node (through views), that node to read the very cache rows the two
// stores write, and nothing but views between the chains and the attention node.  On success the chains are not launched; the
// attention node picks the work up (compute_node).
static bool try_defer_qkv_to_attention(exec_state & s, const nr_chain & A, const nr_chain * B, int vj, const int * item, int ni) {
    static const bool off = getenv("MI355X_NO_QKV_IN_ATTN") != nullptr;
    ggml_cgraph * g = s.g;
    if (off || A.T != 1 || !B || B->store < 0 || vj < 0 || A.store >= 0 || (A.D != 64 && A.D != 128)) return false;
    const ggml_tensor * rq = g->nodes[A.rope];
    // follow the single-consumer view chain from rope(q) to the attention node
    const ggml_tensor * t = rq; int fi = -1;
    for (int hop = 0; hop < 4; ++hop) {
        const int u = sole_user(s, t);
        if (u < 0) return false;
        const ggml_tensor * c = g->nodes[u];
        if (c->op == GGML_OP_FLASH_ATTN_EXT) {                          // (the consumer map attributes users of a view to its root too)
            const ggml_tensor * w = c->src[0];
            while (w && w != t) w = w->view_src;
            if (!w) return false;
            fi = u; break;
        }
        if (!is_noop(c)) return false;
        t = c;
    }
    if (fi < 0 || s.done[fi]) return false;
    const ggml_tensor * f = g->nodes[fi];
    const ggml_tensor * fq = f->src[0], * fk = f->src[1], * fv = f->src[2];
    const ggml_tensor * Sk = g->nodes[B->store], * Sv = g->nodes[vj];
    const int64_t D = A.D;
    if (fq->data != rq->data || fq->ne[0] != D || fq->ne[1] != 1 || fq->ne[2] != A.H || fq->ne[3] != 1 || fq->nb[2] != rq->nb[1] || rq->nb[0] != 4) return false;
    if (fk->type != GGML_TYPE_F16 || fv->type != GGML_TYPE_F16 || fk->data != Sk->data || fv->data != Sv->data || fk->nb[1] != Sk->nb[1] || fv->nb[1] != Sv->nb[1] ||
        fk->nb[2] != (size_t) D * 2 || fv->nb[2] != (size_t) D * 2 || fk->ne[2] != B->H || fv->ne[2] != B->H || fk->ne[3] != 1 || fv->ne[0] != D) return false;
    int last = 0;
    for (int q = 0; q < ni; ++q) if (item[q] > last) last = item[q];
    for (int k = A.first + 1; k < fi; ++k) {
        bool mine = false;
        for (int q = 0; q < ni; ++q) mine |= item[q] == k;
        if (!mine && !s.done[k] && !is_noop(g->nodes[k])) return false;             // something else runs in between: keep the separate launch
    }
    fattn_args fa; tdesc m; fill_fattn_args(f, fa, m);
    if (!fattn_pre_ok(fa) || (A.wt == nullptr) != (B->wt == nullptr)) return false;
    const ggml_tensor * xq = A.xin, * xk = B->xin, * xv = Sv->src[0], * kidx = Sk->src[1], * vidx = Sv->src[1];
    if (kidx->type != vidx->type) return false;
    fattn_pre & p = s.pq.pre;
    p.qraw = (const float *) xq->data; p.q_hs = xq->nb[1]; p.kraw = (const float *) xk->data; p.k_hs = xk->nb[1];
    p.vraw = (const float *) xv->data; p.v_hs = (int64_t) D * 4;
    p.qw = A.wt ? (const float *) A.wt->data : nullptr; p.kw = B->wt ? (const float *) B->wt->data : nullptr; p.pos = (const int32_t *) A.pos->data; p.ff = A.ff ? (const float *) A.ff->data : nullptr;
    p.eps = A.eps; p.rp = A.rp;
    p.kcache = Sk->data; p.kc_rs = Sk->nb[1]; p.vcache = Sv->data; p.vc_rs = Sv->nb[1]; p.kidx = kidx->data; p.vidx = vidx->data; p.idx_is64 = kidx->type == GGML_TYPE_I64;
    s.pq.fa = fi; s.pq.kst = B->store; s.pq.vst = vj;
    return true;
}

// The same for the flash-attention-OFF graph (src/llama-graph.cpp:1362-1420): rope(q) feeds MUL_MAT(k, q) -> SOFT_MAX_EXT(mask f32, scale) ->
// MUL_MAT(v^T, p) -> PERMUTE -> CONT, the k chain stores a cache row, the v store is the single-element scatter into the TRANSPOSED cache
// (llama-kv-cache.cpp:1091-1109).  On success the first MUL_MAT node runs the whole step as one launch (attn_one_sm, fattn_one.hip).
static bool try_defer_qkv_to_softmax_attention(exec_state & s, const nr_chain & A, const nr_chain * B, int vsj, const int * item, int ni) {
    static const bool off = getenv("MI355X_NO_QKV_IN_ATTN") != nullptr || getenv("MI355X_NO_ATTN_SM") != nullptr;
    ggml_cgraph * g = s.g;
    if (off || A.T != 1 || !B || B->store < 0 || vsj < 0 || A.store >= 0 || (A.D != 64 && A.D != 128)) return false;
    const ggml_tensor * rq = g->nodes[A.rope];
    auto views_back_to = [](const ggml_tensor * w, const ggml_tensor * t) { while (w && w != t) w = w->view_src; return w != nullptr; };
    // rope(q) -> [views] -> MUL_MAT(k, q)
    const ggml_tensor * t = rq; int m1 = -1;
    for (int hop = 0; hop < 4; ++hop) {
        const int u = sole_user(s, t);
        if (u < 0) return false;
        const ggml_tensor * c = g->nodes[u];
        if (c->op == GGML_OP_MUL_MAT) { if (!views_back_to(c->src[1], t)) return false; m1 = u; break; }
        if (!is_noop(c)) return false;
        t = c;
    }
    if (m1 < 0 || s.done[m1]) return false;
    const ggml_tensor * M1 = g->nodes[m1], * fk = M1->src[0], * fq = M1->src[1];
    const ggml_tensor * Sk = g->nodes[B->store], * Sv = g->nodes[vsj];
    const int64_t D = A.D, H = A.H, HK = B->H;
    if (fq->data != rq->data || fq->type != GGML_TYPE_F32 || fq->ne[0] != D || fq->ne[1] != 1 || fq->ne[2] != H || fq->ne[3] != 1 || fq->nb[2] != rq->nb[1] || rq->nb[0] != 4) return false;
    if (fk->type != GGML_TYPE_F16 || fk->data != Sk->data || fk->ne[0] != D || fk->ne[2] != HK || fk->ne[3] != 1 || fk->nb[0] != 2 || fk->nb[1] != Sk->nb[1] || fk->nb[2] != (size_t) D * 2) return false;
    const int64_t nkv = fk->ne[1];
    if (M1->type != GGML_TYPE_F32 || M1->ne[0] != nkv || M1->ne[1] != 1 || M1->ne[2] != H || M1->ne[3] != 1) return false;
    // -> SOFT_MAX_EXT
    const int smi = sole_user(s, M1);
    if (smi < 0 || s.done[smi]) return false;
    const ggml_tensor * SM = g->nodes[smi];
    if (SM->op != GGML_OP_SOFT_MAX || SM->src[0] != M1 || SM->src[2] || op_param_f32(SM, 1) != 0.0f || !same_shape(SM, M1)) return false;
    const ggml_tensor * mk = SM->src[1];
    if (mk && (mk->type != GGML_TYPE_F32 || mk->ne[0] != nkv || mk->nb[0] != 4 || mk->ne[2] != 1 || mk->ne[3] != 1)) return false;
    // -> MUL_MAT(v^T, p)
    const int m2 = sole_user(s, SM);
    if (m2 < 0 || s.done[m2]) return false;
    const ggml_tensor * M2 = g->nodes[m2];
    if (M2->op != GGML_OP_MUL_MAT || M2->src[1] != SM) return false;
    const ggml_tensor * fv = M2->src[0];
    if (fv->type != GGML_TYPE_F16 || fv->data != Sv->data || fv->ne[0] != nkv || fv->ne[1] != D || fv->ne[2] != HK || fv->ne[3] != 1 || fv->nb[0] != 2 ||
        fv->nb[2] != (size_t) D * fv->nb[1]) return false;
    if (M2->type != GGML_TYPE_F32 || M2->ne[0] != D || M2->ne[1] != 1 || M2->ne[2] != H || M2->ne[3] != 1 || M2->nb[0] != 4) return false;
    // -> PERMUTE -> CONT [D * H]
    t = M2; int ci = -1;
    for (int hop = 0; hop < 4; ++hop) {
        const int u = sole_user(s, t);
        if (u < 0) return false;
        const ggml_tensor * c = g->nodes[u];
        if (c->op == GGML_OP_CONT) { if (!views_back_to(c->src[0], t)) return false; ci = u; break; }
        if (!is_noop(c)) return false;
        t = c;
    }
    if (ci < 0 || s.done[ci]) return false;
    const ggml_tensor * C = g->nodes[ci], * cs = C->src[0];
    if (C->type != GGML_TYPE_F32 || !is_contiguous(C) || nelements(C) != D * H || cs->data != M2->data || cs->ne[0] != D || cs->ne[1] != H || cs->ne[2] != 1 || cs->ne[3] != 1 ||
        cs->nb[0] != 4 || cs->nb[1] != M2->nb[2]) return false;
    // the v scatter: one f16 element per index into the same transposed cache
    const ggml_tensor * xv = Sv->src[0], * vidx = Sv->src[1], * kidx = Sk->src[1];
    if (Sv->type != GGML_TYPE_F16 || Sv->ne[0] != 1 || Sv->nb[1] != 2 || xv->type != GGML_TYPE_F32 || xv->ne[0] != 1 || xv->ne[1] != D * HK || xv->nb[1] != 4 ||
        nelements(xv) != D * HK || vidx->ne[0] != D * HK || kidx->type != vidx->type || (vidx->type != GGML_TYPE_I64 && vidx->type != GGML_TYPE_I32) ||
        vidx->nb[0] != (vidx->type == GGML_TYPE_I64 ? 8u : 4u)) return false;
    if ((A.wt == nullptr) != (B->wt == nullptr)) return false;
    for (int k = A.first + 1; k < ci; ++k) {
        bool mine = k == m1 || k == smi || k == m2;
        for (int q = 0; q < ni; ++q) mine |= item[q] == k;
        if (!mine && !s.done[k] && !is_noop(g->nodes[k])) return false;             // something else runs in between: keep the separate launches
    }
    const ggml_tensor * xq = A.xin, * xk = B->xin;
    fattn_pre & p = s.pq.pre;
    p.qraw = (const float *) xq->data; p.q_hs = xq->nb[1]; p.kraw = (const float *) xk->data; p.k_hs = xk->nb[1];
    p.vraw = (const float *) xv->data; p.v_hs = (int64_t) D * 4;
    p.qw = A.wt ? (const float *) A.wt->data : nullptr; p.kw = B->wt ? (const float *) B->wt->data : nullptr; p.pos = (const int32_t *) A.pos->data; p.ff = A.ff ? (const float *) A.ff->data : nullptr;
    p.eps = A.eps; p.rp = A.rp;
    p.kcache = Sk->data; p.kc_rs = Sk->nb[1]; p.vcache = Sv->data; p.vc_rs = 2; p.kidx = kidx->data; p.vidx = vidx->data; p.idx_is64 = kidx->type == GGML_TYPE_I64;
    attn_sm_args & a = s.pq.sma;
    a = attn_sm_args();
    a.pre = &s.pq.pre; a.k = fk->data; a.knb1 = fk->nb[1]; a.knb2 = fk->nb[2]; a.v = fv->data; a.vnb1 = fv->nb[1]; a.vnb2 = fv->nb[2];
    a.mask = mk ? mk->data : nullptr; a.mnb2 = 0; a.mne2 = 1; a.dst = C->data; a.dnb1 = (int64_t) D * 4; a.vidx_n = vidx->ne[0];
    a.D = (int) D; a.nkv = (int) nkv; a.n_head = (int) H; a.n_head_kv = (int) HK; a.scale = op_param_f32(SM, 0);
    a.rope_tab = (const float *) s.c->rope_scratch;                                  // filled when the launch happens
    if (nkv > 256) {                                                                 // slices: partial rows in the attention scratch, arrival counters
        if (!s.c->fa_counters && !s.capturing) {
            if (hipMalloc((void **) &s.c->fa_counters, 1024 * sizeof(unsigned)) == hipSuccess) HIP_CHECK(hipMemsetAsync(s.c->fa_counters, 0, 1024 * sizeof(unsigned), s.st));
            else { (void) hipGetLastError(); s.c->fa_counters = nullptr; }
        }
        a.part = s.c->fa_scratch; a.part_bytes = s.c->fa_scratch_bytes; a.counters = s.c->fa_counters;
        s.fa_mask = nullptr;                                                         // (the scratch no longer holds a mask tile map)
    }
    if (s.c->rope_scratch_bytes < (size_t) D * 4 || !attn_one_sm_ok(a)) return false;
    s.pq.fa = m1; s.pq.sm = true; s.pq.kst = B->store; s.pq.vst = vsj; s.pq.sm_soft = smi; s.pq.sm_mm2 = m2; s.pq.sm_cont = ci;
    return true;
}

// ROPE at node i without a norm in front (llama architecture: the omni TTS decoder, src/llama-model.cpp llm_build_llama): the q chain is
// ROPE alone, the k chain ROPE -> SET_ROWS, v a plain (or, flash-attention off, scattered) store.  Same three outcomes as the Qwen3 chains:
// everything inside the one-token attention launch, or one norm_rope launch for both chains + the v store, or (no match) the plain op.
static bool match_rope_only(exec_state & s, int j, nr_chain & c) {
    ggml_cgraph * g = s.g;
    ggml_tensor * r = g->nodes[j];
    if (r->op != GGML_OP_ROPE || s.done[j]) return false;
    const ggml_tensor * x = r->src[0], * pos = r->src[1], * ff = r->src[2];
    const int64_t D = r->ne[0];
    const int mode = op_param_i32(r, 2);
    if (!x || x->type != GGML_TYPE_F32 || r->type != GGML_TYPE_F32 || x->nb[0] != 4 || r->nb[0] != 4 || D % 2 != 0 || D > 256 || r->ne[3] != 1 || !same_shape(x, r)) return false;
    if (!((mode == GGML_ROPE_TYPE_NORMAL || mode == GGML_ROPE_TYPE_NEOX) && op_param_i32(r, 1) == D && pos && pos->type == GGML_TYPE_I32 && pos->nb[0] == 4 &&
          (!ff || (ff->type == GGML_TYPE_F32 && ff->nb[0] == 4)))) return false;
    c.norm = -1; c.mul = -1; c.rope = j; c.store = -1; c.wt = nullptr; c.pos = pos; c.ff = ff; c.xin = x; c.first = j;
    c.D = (int) D; c.H = (int) r->ne[1]; c.T = (int) r->ne[2]; c.eps = 0.0f;
    memset(&c.rp, 0, sizeof(c.rp));
    c.rp.n_dims = op_param_i32(r, 1); c.rp.mode = mode; c.rp.n_ctx_orig = op_param_i32(r, 4);
    c.rp.freq_base = op_param_f32(r, 5); c.rp.freq_scale = op_param_f32(r, 6); c.rp.ext_factor = op_param_f32(r, 7);
    c.rp.attn_factor = op_param_f32(r, 8); c.rp.beta_fast = op_param_f32(r, 9); c.rp.beta_slow = op_param_f32(r, 10);
    const int si = sole_user(s, r);
    if (si > j && g->nodes[si]->op == GGML_OP_SET_ROWS && !s.done[si]) {
        const ggml_tensor * S = g->nodes[si], * V = S->src[0], * idx = S->src[1];
        if (V && idx && V->data == r->data && V->ne[0] == D * r->ne[1] && V->ne[1] == r->ne[2] && V->ne[2] == 1 && V->ne[3] == 1 &&
            V->nb[1] == r->nb[2] && r->nb[1] == (size_t) D * 4 && S->type == GGML_TYPE_F16 && S->nb[0] == 2 &&
            (idx->type == GGML_TYPE_I64 || idx->type == GGML_TYPE_I32) && idx->ne[0] == r->ne[2] && idx->ne[1] == 1 && idx->ne[2] == 1) c.store = si;
    }
    return true;
}
static bool try_defer_qkv_to_attention(exec_state & s, const nr_chain & A, const nr_chain * B, int vj, const int * item, int ni);
static bool try_defer_qkv_to_softmax_attention(exec_state & s, const nr_chain & A, const nr_chain * B, int vsj, const int * item, int ni);
static bool exec_rope_chain(exec_state & s, int i) {
    static const bool off = getenv("MI355X_NO_ROPE_CHAIN") != nullptr;
    ggml_cgraph * g = s.g;
    nr_chain A;
    if (off || !match_rope_only(s, i, A) || A.store >= 0) return false;             // (starts at the q chain: the first ROPE of a layer in llm_build_llama)
    int item[8]; int ni = 0;
    item[ni++] = A.rope;
    nr_chain B; int bj = -1;
    for (int j = i + 1; j < g->n_nodes && j < i + 24; ++j) {
        if (s.done[j] || g->nodes[j]->op != GGML_OP_ROPE || !match_rope_only(s, j, B)) continue;
        if (B.D != A.D || B.T != A.T || B.pos != A.pos || B.ff != A.ff || memcmp(&B.rp, &A.rp, sizeof(rope_params)) != 0 || B.store < 0) continue;
        int it2[8]; int n2 = ni;
        memcpy(it2, item, sizeof(int) * ni);
        it2[n2++] = B.rope;
        if (!can_hoist(s, i, B.rope, it2, n2)) break;
        it2[n2++] = B.store;
        if (!can_hoist(s, i, B.store, it2, n2)) break;
        bj = j; memcpy(item, it2, sizeof(int) * n2); ni = n2;
        break;
    }
    if (bj < 0) return false;
    // v store: plain rows, or (flash-attention off, one token) the single-element scatter
    int vj = -1; norm_rope_job vjob;
    for (int j = i + 1; j < g->n_nodes && j < i + 32; ++j) {
        ggml_tensor * S = g->nodes[j];
        if (s.done[j] || S->op != GGML_OP_SET_ROWS) continue;
        bool mine = false;
        for (int q = 0; q < ni; ++q) mine |= item[q] == j;
        if (mine) continue;
        const ggml_tensor * V = S->src[0], * idx = S->src[1];
        if (V && V->type == GGML_TYPE_F32 && S->type == GGML_TYPE_F16 && V->ne[0] == 1 && S->ne[0] == 1 && A.T == 1 && V->ne[1] == (int64_t) A.D * B.H) {
            item[ni++] = j;
            if (can_hoist(s, i, j, item, ni) && try_defer_qkv_to_softmax_attention(s, A, &B, j, item, ni)) {
                for (int q = 1; q < ni; ++q) { s.done[item[q]] = 1; ++s.n_fused; }
                ++s.n_fused;
                return true;
            }
            --ni;
            break;
        }
        if (!(V && idx && V->type == GGML_TYPE_F32 && S->type == GGML_TYPE_F16 && S->nb[0] == 2 && V->nb[0] == 4 && V->ne[0] % A.D == 0 &&
              V->ne[1] == A.T && V->ne[2] == 1 && V->ne[3] == 1 && (idx->type == GGML_TYPE_I64 || idx->type == GGML_TYPE_I32) &&
              idx->ne[0] == A.T && idx->ne[1] == 1 && idx->ne[2] == 1)) continue;
        item[ni++] = j;
        if (can_hoist(s, i, j, item, ni)) {
            vj = j;
            vjob = { (const float *) V->data, (int64_t) A.D * 4, (int64_t) V->nb[1], nullptr, nullptr, 0, 0,
                     S->data, (int64_t) S->nb[1], idx->data, idx->type == GGML_TYPE_I64, (int64_t) idx->nb[0], (int) (V->ne[0] / A.D) };
        } else --ni;
        break;
    }
    if (try_defer_qkv_to_attention(s, A, &B, vj, item, ni)) {
        for (int q = 1; q < ni; ++q) { s.done[item[q]] = 1; ++s.n_fused; }
        ++s.n_fused;
        return true;
    }
    norm_rope_args a;
    a.njobs = 0; a.pos = (const int32_t *) A.pos->data; a.ff = A.ff ? (const float *) A.ff->data : nullptr;
    a.D = A.D; a.T = A.T; a.eps = 0.0f; a.rp = A.rp;
    a.j[a.njobs++] = chain_job(s, A);
    a.j[a.njobs++] = chain_job(s, B);
    if (vj >= 0) a.j[a.njobs++] = vjob;
    {
        prof_scope ps(s, "norm_rope", 0);
        norm_rope_store(a, s.st);
    }
    ++s.n_kernels;
    for (int q = 1; q < ni; ++q) { s.done[item[q]] = 1; ++s.n_fused; }
    note_write(s, g->nodes[A.rope]);
    note_write(s, g->nodes[B.store]);
    if (vj >= 0) note_write(s, g->nodes[vj]);
    return true;
}

// RMS_NORM at node i: fold the following MUL(w) in, and -- when every consumer is a K-quant MUL_MAT -- also emit the Q8_K image
static bool exec_rms_norm(exec_state & s, int i) {
    ggml_cgraph * g = s.g;
    ggml_tensor * n = g->nodes[i];
    const float eps = op_param_f32(n, 0);
    // (a pending split-K result is folded in only by the plain 2-D norm + mul path at the end; every other path reads it from memory)
    if (s.pr.A && s.pr.A == n->src[0] && !(n->ne[2] == 1 && n->ne[3] == 1 && n->ne[1] > MI_MMVQ_MAX_COLS && n->ne[0] > 256)) materialise_reduce(s);
    if (s.prm.n) {                                                       // pending slabs of wq / wk / wv: only the prefill norm + rope launch below can take them
        nr_chain A0;
        if (!s.c->opt_fusion || !match_norm_rope(s, i, A0) || A0.T <= MI_MMVQ_MAX_COLS) materialise_group(s);
    }
    if (!s.c->opt_fusion) return false;
    const int mi_ = sole_user(s, n);
    if (mi_ != i + 1 || g->nodes[mi_]->op != GGML_OP_MUL) return false;
    ggml_tensor * m = g->nodes[mi_];
    const ggml_tensor * wt = m->src[0] == n ? m->src[1] : m->src[0];
    if ((m->src[0] == n) == (m->src[1] == n)) return false;
    if (!wt || wt == n || m->type != GGML_TYPE_F32 || wt->type != GGML_TYPE_F32 || wt->nb[0] != 4 || !same_shape(m, n) || !can_repeat(wt, n) || m->nb[0] != 4) return false;
    // chain variant: RMS_NORM -> MUL(w[D]) -> ROPE [-> SET_ROWS(view as [D*H, T]) into an f16 table]: the q / k chains of a decoder
    // layer; a second chain with the same rope parameters and one plain f32 -> f16 SET_ROWS (the v store) join the launch
    {
        nr_chain A;
        if (match_norm_rope(s, i, A)) {
            nr_chain B; int bj = -1, vj = -1; norm_rope_job vjob;
            int item[12]; int ni = 0;
            item[ni++] = A.norm; item[ni++] = A.mul; item[ni++] = A.rope;
            bool okA = can_hoist(s, i, A.rope, item, ni);
            if (okA && A.store >= 0) {
                item[ni++] = A.store;
                if (!can_hoist(s, i, A.store, item, ni)) { --ni; A.store = -1; }
            }
            if (okA) {
                // second chain
                for (int j = i + 1; j < g->n_nodes && j < i + 24; ++j) {
                    if (s.done[j] || g->nodes[j]->op != GGML_OP_RMS_NORM) continue;
                    if (!match_norm_rope(s, j, B) || B.D != A.D || B.T != A.T || B.eps != A.eps || B.pos != A.pos || B.ff != A.ff ||
                        memcmp(&B.rp, &A.rp, sizeof(rope_params)) != 0) break;
                    int it2[12]; int n2 = ni;
                    memcpy(it2, item, sizeof(int) * ni);
                    it2[n2++] = B.norm; it2[n2++] = B.mul; it2[n2++] = B.rope;
                    bool ok = can_hoist(s, i, B.norm, it2, n2) && can_hoist(s, i, B.mul, it2, n2) && can_hoist(s, i, B.rope, it2, n2);
                    if (ok && B.store >= 0) {
                        it2[n2++] = B.store;
                        if (!can_hoist(s, i, B.store, it2, n2)) { --n2; B.store = -1; }
                    }
                    if (ok) { bj = j; memcpy(item, it2, sizeof(int) * n2); ni = n2; }
                    break;
                }
                // plain store of rows of D-element groups (v_cur -> v cache)
                for (int j = i + 1; j < g->n_nodes && j < i + 32; ++j) {
                    ggml_tensor * S = g->nodes[j];
                    if (s.done[j] || S->op != GGML_OP_SET_ROWS) continue;
                    bool mine = false;
                    for (int q = 0; q < ni; ++q) mine |= item[q] == j;
                    if (mine) continue;
                    const ggml_tensor * V = S->src[0], * idx = S->src[1];
                    if (!(V && idx && V->type == GGML_TYPE_F32 && S->type == GGML_TYPE_F16 && S->nb[0] == 2 && V->nb[0] == 4 && V->ne[0] % A.D == 0 &&
                          V->ne[1] == A.T && V->ne[2] == 1 && V->ne[3] == 1 && (idx->type == GGML_TYPE_I64 || idx->type == GGML_TYPE_I32) &&
                          idx->ne[0] == A.T && idx->ne[1] == 1 && idx->ne[2] == 1)) continue;
                    item[ni++] = j;
                    if (can_hoist(s, i, j, item, ni)) {
                        vj = j;
                        vjob = { (const float *) V->data, (int64_t) A.D * 4, (int64_t) V->nb[1], nullptr, nullptr, 0, 0,
                                 S->data, (int64_t) S->nb[1], idx->data, idx->type == GGML_TYPE_I64, (int64_t) idx->nb[0], (int) (V->ne[0] / A.D) };
                    } else --ni;
                    break;
                }
                // flash-attention off, one token: the v store is a scatter of single elements into the transposed cache
                if (vj < 0 && A.T == 1 && bj >= 0) {
                    for (int j = i + 1; j < g->n_nodes && j < i + 32; ++j) {
                        ggml_tensor * S = g->nodes[j];
                        if (s.done[j] || S->op != GGML_OP_SET_ROWS) continue;
                        bool mine = false;
                        for (int q = 0; q < ni; ++q) mine |= item[q] == j;
                        if (mine) continue;
                        const ggml_tensor * V = S->src[0];
                        if (!(V && V->type == GGML_TYPE_F32 && S->type == GGML_TYPE_F16 && V->ne[0] == 1 && S->ne[0] == 1 && V->ne[1] == (int64_t) A.D * B.H)) continue;
                        item[ni++] = j;
                        if (can_hoist(s, i, j, item, ni) && try_defer_qkv_to_softmax_attention(s, A, &B, j, item, ni)) {
                            for (int q = 1; q < ni; ++q) { s.done[item[q]] = 1; ++s.n_fused; }
                            ++s.n_fused;
                            return true;
                        }
                        --ni;
                        break;
                    }
                }
                if (try_defer_qkv_to_attention(s, A, bj >= 0 ? &B : nullptr, vj, item, ni)) {
                    for (int q = 1; q < ni; ++q) { s.done[item[q]] = 1; ++s.n_fused; }
                    ++s.n_fused;
                    return true;
                }
                norm_rope_args a;
                a.njobs = 0; a.pos = (const int32_t *) A.pos->data; a.ff = A.ff ? (const float *) A.ff->data : nullptr;
                a.D = A.D; a.T = A.T; a.eps = A.eps; a.rp = A.rp;
                a.j[a.njobs++] = chain_job(s, A);
                if (bj >= 0) a.j[a.njobs++] = chain_job(s, B);
                if (vj >= 0) a.j[a.njobs++] = vjob;
                // flash-attention off, prefill: rope(q) is read (through views) by exactly one per-head MUL_MAT on the MFMA GEMM (K . q): write
                // its f16 activation image here instead of the f32 rows + a conversion launch (not when that MUL_MAT starts a soft-max attention chain that runs as
                // one flash-attention launch: that kernel reads the f32 rows and rounds them itself)
                const ggml_tensor * q16 = nullptr;
                if (A.store < 0 && A.T > MI_MMVQ_MAX_COLS && !getenv("MI355X_NO_F16_EMIT")) {
                    const ggml_tensor * rq = g->nodes[A.rope];
                    const int u = sole_user(s, rq);                      // (consumers are counted through view chains)
                    const ggml_tensor * c = u >= 0 ? g->nodes[u] : nullptr;
                    const ggml_tensor * t = c && c->op == GGML_OP_MUL_MAT ? c->src[1] : nullptr;
                    const ggml_tensor * base = t;
                    while (base && base != rq && (base->op == GGML_OP_RESHAPE || base->op == GGML_OP_VIEW || base->op == GGML_OP_PERMUTE || base->op == GGML_OP_TRANSPOSE)) base = base->src[0];
                    if (t && base == rq && c->src[0] != t && mm_uses_gemm(c) && !exec_attn_sm_prefill(s, u, true) && t->data == rq->data && t->type == GGML_TYPE_F32 && !is_out(s, t) &&
                        t->ne[0] == A.D && t->ne[1] == A.T && t->ne[2] == A.H && t->ne[3] == 1 && t->nb[0] == 4 && t->nb[1] == (size_t) rq->nb[2] &&
                        t->nb[2] == (size_t) rq->nb[1] && act_image_bytes(ACT_F16, A.D) * (size_t) (A.T * A.H) <= s.c->act_scratch_bytes) q16 = t;
                }
                if (q16) { a.j[0].y = nullptr; a.j[0].y16 = s.c->act_scratch; a.j[0].y16_rs = (int64_t) act_image_bytes(ACT_F16, A.D); }
                if (A.T >= ROPE_TABLE_MIN_TOKENS && (size_t) A.T * A.D * 4 <= s.c->rope_scratch_bytes) {
                    // prefill: the angles depend on (position, pair) only -- one table per graph instead of sincos per head, layer and chain
                    a.rope_tab = (float *) s.c->rope_scratch;
                    a.rope_tab_valid = s.rt.pos == A.pos->data && s.rt.ff == (A.ff ? A.ff->data : nullptr) && s.rt.T == A.T && s.rt.D == A.D &&
                                       memcmp(&s.rt.rp, &A.rp, sizeof(rope_params)) == 0;
                    if (!a.rope_tab_valid) { s.rt.pos = A.pos->data; s.rt.ff = A.ff ? A.ff->data : nullptr; s.rt.T = A.T; s.rt.D = A.D; s.rt.rp = A.rp; ++s.n_kernels; }
                }
                if (s.prm.n) {
                    // every job of this launch reads one of the pending results whole, each result once: point the jobs at the slabs; anything else gets the reduction launch
                    // (a result no job reads -- the V rows of a flash-attention-off graph, whose store is a scatter launch of its own -- gets the reduction launch alone)
                    norm_rope_args b = a;
                    bool ok = a.njobs <= s.prm.n; int used = 0;
                    for (int jb = 0; jb < a.njobs && ok; ++jb) {
                        int q = -1;
                        for (int k = 0; k < s.prm.n; ++k) if ((const void *) a.j[jb].x == s.prm.A[k]->data && !(used & (1 << k))) q = k;
                        ok = q >= 0 && a.j[jb].xnb1 == (int64_t) a.D * 4 && a.j[jb].xnb2 == s.prm.M[q] * 4 && (int64_t) a.j[jb].H * a.D == s.prm.M[q] && a.T == s.prm.N;
                        if (ok) { used |= 1 << q; b.j[jb].x = (const float *) s.c->gemm_partial + s.prm.off[q]; b.j[jb].nsplit = s.prm.nsplit; b.j[jb].split_bytes = (int64_t) s.prm.slab * 4; }
                    }
                    if (ok && norm_rope_takes_split(b)) { a = b; materialise_group(s, used); ++s.n_fused; }
                    else materialise_group(s);
                }
                {
                    prof_scope ps(s, "norm_rope", 0);
                    norm_rope_store(a, s.st);
                }
                ++s.n_kernels;
                for (int q = 1; q < ni; ++q) { s.done[item[q]] = 1; ++s.n_fused; }
                note_write(s, g->nodes[A.store >= 0 ? A.store : A.rope]);
                if (bj >= 0) note_write(s, g->nodes[B.store >= 0 ? B.store : B.rope]);
                if (vj >= 0) note_write(s, g->nodes[vj]);
                if (q16) {                                              // the image of the permuted view [D, T, H] now sits in act_scratch (rows h * T + t)
                    s.a_src = q16->data; s.a_kind = ACT_F16; s.a_K = q16->ne[0]; s.a_ne[0] = q16->ne[1]; s.a_ne[1] = q16->ne[2]; s.a_ne[2] = q16->ne[3];
                    s.a_nb[0] = q16->nb[1]; s.a_nb[1] = q16->nb[2]; s.a_nb[2] = q16->nb[3];
                    s.a_range_lo = (const char *) q16->data; s.a_range_hi = (const char *) q16->data + nbytes(q16);
                    ++s.n_fused;
                }
                return true;
            }
        }
    }
    // every consumer a Q8_0 batch-1 mat-vec (mmv1q.hip: the TTS / Token2Wav decoders): the norm is computed inside their launches
    if (s.c->opt_mv1 && n->ne[1] == 1 && n->ne[2] == 1 && n->ne[3] == 1 && rms_norm_mul_quant_ok(n->ne[0]) && wt->ne[0] == n->ne[0] && wt->ne[1] * wt->ne[2] * wt->ne[3] == 1 &&
        n->src[0]->nb[0] == 4 && n_users(s, m) > 0 && !is_out(s, m) && ((uintptr_t) n->src[0]->data & 15) == 0 && ((uintptr_t) wt->data & 15) == 0) {
        bool all_q80 = true; int last_user = mi_;
        for (int u : s.users[m]) { const ggml_tensor * c = g->nodes[u]; all_q80 = all_q80 && c->src[1] == m && q80_mv1_node(s, c); if (u > last_user) last_user = u; }
        if (all_q80) {
            const ggml_tensor * xs = n->src[0];
            const byte_range rx = range_of(xs);
            for (int k = mi_ + 1; k < last_user && all_q80; ++k) {           // nothing that runs before the last consumer may write over the norm's input
                const ggml_tensor * nk = g->nodes[k];
                if (is_noop(nk) || s.done[k]) continue;
                bool is_user = false;
                for (int u : s.users[m]) is_user |= u == k;
                if (!is_user && overlap(range_of(nk), rx)) all_q80 = false;
            }
            if (all_q80) {
                if (s.pn.m) materialise_norm(s);
                s.done[mi_] = 1; s.n_fused += 2;
                s.pn.m = m; s.pn.x = xs; s.pn.wt = wt; s.pn.eps = eps; s.pn.left = n_users(s, m);
                if (s.a_src == m->data) s.a_src = nullptr;
                return true;
            }
        }
    }
    if (s.prm.n) materialise_group(s);                                  // (the norm + rope launch did not happen: the paths below read the rows from memory)
    // image variant: row-contiguous 2-D activation, weight a plain [ne0] vector, every consumer a K-quant mat-vec on it
    bool want_img = rms_norm_mul_quant_ok(n->ne[0]) && n->ne[2] == 1 && n->ne[3] == 1 && n->ne[1] <= mmq_max_cols() && wt->ne[0] == n->ne[0] &&
                    wt->ne[1] * wt->ne[2] * wt->ne[3] == 1 && n->src[0]->nb[0] == 4 && n_users(s, m) > 0;
    if (want_img) {
        for (int u : s.users[m]) {
            const ggml_tensor * c = g->nodes[u];
            if (!(c->op == GGML_OP_MUL_MAT && c->src[1] == m && is_kquant(c->src[0]->type) && c->src[0]->ne[2] == 1 && c->src[0]->ne[3] == 1 &&
                  (n->ne[1] <= MI_MMVQ_MAX_COLS || mm_uses_mmq(c)))) { want_img = false; break; }
        }
    }
    if (want_img) {
        if (s.pn.m) materialise_norm(s);                              // (an earlier deferred norm that was never consumed in-kernel)
        // defer: the consumers build the image themselves.  Needs: every consumer a fused K-quant mat-vec, 16-byte aligned rows,
        // and nothing that runs before the last consumer may write over the norm's input
        const ggml_tensor * xs = n->src[0];
        // (measured on MI355X, decode of Qwen3-8B: the in-kernel norm removes 73 launches per token and costs the consumers exactly
        //  what it saves -- 378 tok/s either way, DESIGN.md section 7 -- so it is opt-in: option "norm_in_kernel" / MI355X_NORM_IN_KERNEL=1)
        // the batch-1 decode launches (mmv1.hip) always take the norm in: their prologue builds the image from x and the norm weights
        bool all_mv1 = n->ne[1] == 1;
        for (int u : s.users[m]) all_mv1 = all_mv1 && mv1_node_ok(s, g->nodes[u]);
        bool defer = (all_mv1 || (s.c->opt_norm_in_kernel && mmv_norm_ok(n->ne[0], (int) n->ne[1]))) && !is_out(s, m) && ((uintptr_t) xs->data & 15) == 0 && xs->nb[1] % 16 == 0 && ((uintptr_t) wt->data & 15) == 0;
        int last_user = mi_;
        for (int u : s.users[m]) { defer = defer && plain_kq_matvec(g->nodes[u], MI_MMVQ_MAX_COLS); if (u > last_user) last_user = u; }
        if (defer) {
            const byte_range rx = range_of(xs);
            for (int k = mi_ + 1; k < last_user && defer; ++k) {
                const ggml_tensor * nk = g->nodes[k];
                if (is_noop(nk) || s.done[k]) continue;
                bool is_user = false;
                for (int u : s.users[m]) is_user |= u == k;
                if (!is_user && overlap(range_of(nk), rx)) defer = false;
            }
        }
        s.done[mi_] = 1; s.n_fused += 1;
        s.pn.m = m; s.pn.x = xs; s.pn.wt = wt; s.pn.eps = eps; s.pn.left = n_users(s, m);
        if (!defer) { note_write(s, m); materialise_norm(s); }
        else { ++s.n_fused; if (s.a_src == m->data) s.a_src = nullptr; }
        return true;
    }
    const tdesc wd = td(wt);
    const ggml_tensor * xg = nullptr;
    const bool emit16 = n->ne[2] == 1 && n->ne[3] == 1 && m->nb[1] == (size_t) m->ne[0] * 4 && gemm_only_consumers(s, m, m->ne[0], m->ne[1], &xg);
    const bool q8 = emit16 && consumers_act_kind(s, xg) == ACT_F16Q;       // K-quant GEMMs read the rows: the image carries the Q8_K-quantised values (quantised from the f32 values, inside the norm launch)
    const bool from_split = s.pr.A && s.pr.A == n->src[0];
    if (from_split && !(n->ne[2] == 1 && n->ne[3] == 1 && wt->ne[0] == n->ne[0] && wt->ne[1] * wt->ne[2] * wt->ne[3] == 1 && m->nb[1] % 16 == 0 && ((uintptr_t) wt->data & 15) == 0))
        materialise_reduce(s);
    if (s.pr.A && s.pr.A == n->src[0]) {
        // the norm's input still lies as split-K slabs: reduce, add the residual, write it, and normalise in one pass
        const ggml_tensor * A = s.pr.A;
        const bool w32 = !emit16 || n_users(s, m) > 1;
        prof_scope ps(s, "rms_norm_mul", 0);
        gemm_reduce_rms_norm((const float *) s.c->gemm_partial, s.pr.nsplit, s.pr.resid, s.pr.resid_cs, (float *) A->data, A->nb[1], (const float *) wt->data, eps,
                             w32 ? (float *) m->data : nullptr, m->nb[1], emit16 ? (uint16_t *) s.c->act_scratch : nullptr, act_image_bytes(ACT_F16, m->ne[0]),
                             A->ne[0], A->ne[1], s.st, q8);
        s.pr.A = nullptr; ++s.n_fused;
    } else {
        prof_scope ps(s, "rms_norm_mul", 0);
        if (emit16) rms_norm(td(n->src[0]), td(m), eps, &wd, s.st, (uint16_t *) s.c->act_scratch, act_image_bytes(ACT_F16, m->ne[0]), n_users(s, m) > 1, q8);
        else        rms_norm(td(n->src[0]), td(m), eps, &wd, s.st);
    }
    ++s.n_kernels; s.n_fused += 1; s.done[mi_] = 1;
    note_write(s, m);
    if (emit16) { seed_act_f16(s, xg, q8); ++s.n_fused; }
    return true;
}

// Flash-attention OFF, a batch of query rows (llama-bench's default prefill, the Whisper / SigLip encoders): MUL_MAT(k, q) -> SOFT_MAX_EXT(mask, scale) -> MUL_MAT(v^T, p) ->
// PERMUTE -> CONT is one flash-attention launch reading V^T as it lies (reference: ggml_compute_forward_soft_max_f32, ops.cpp:5072-5182, between two ggml_compute_forward_mul_mat;
// the [n_kv, n_q, H] blocks -- 146 MB written and read back per Whisper layer -- are never materialised).  An f32 mask is cast to f16 once per graph run (what the
// reference's own flash-attention graphs do, llama-graph.cpp build_attn_inp_kv: ggml_cast(kq_mask, F16); 0 and -inf are exact) behind the mask tile map in the attention scratch.
static size_t attn_sm_mask16_off(int64_t nq, int64_t nkv) { return (fattn_map_bytes_host(nq, nkv) + 255) & ~(size_t) 255; }
// K.Q -> [SCALE] -> SOFT_MAX (no mask) -> V^T.P -> [views -> CONT of the [D, H, nq, ns] permutation], everything f32 and nothing else in between: one attn_f32 launch
// (the reference's Token2Wav DiT attention, token2wav-impl.cpp:406-439).  `i` is the K.Q MUL_MAT.
static bool exec_attn_f32(exec_state & s, int i) {
    ggml_cgraph * g = s.g;
    static const bool dbg = getenv("MI355X_SINK_DEBUG") != nullptr;          // which line turned an f32 x f32 batched MUL_MAT down, tallied (stderr at process exit)
    static std::map<int, long> why;
    struct dump { ~dump() { if (dbg) { fprintf(stderr, "[mi355x] attn_f32: taken %ld, refusals by source line:", why[0]); for (auto & kv : why) if (kv.first) fprintf(stderr, " %d:%ld", kv.first, kv.second); fprintf(stderr, "\n"); } } };
    static dump at_exit;
    auto no = [&](int line) { if (dbg) ++why[line]; return false; };
    const ggml_tensor * M1 = g->nodes[i];
    if (!s.c->opt_fusion || M1->op != GGML_OP_MUL_MAT || is_out(s, M1)) return false;
    const ggml_tensor * fk = M1->src[0], * fq = M1->src[1];
    if (fk->type != GGML_TYPE_F32 || fq->type != GGML_TYPE_F32 || M1->type != GGML_TYPE_F32 || fk->nb[0] != 4 || fq->nb[0] != 4 || fk->ne[3] != 1 || fq->ne[3] != 1 || !fk->data || !fq->data) return false;
    const int64_t D = fk->ne[0], nkv = fk->ne[1], HB = fk->ne[2], nq = fq->ne[1];
    if (fq->ne[0] != D || fq->ne[2] != HB || nq <= MI_MMVQ_MAX_COLS || !is_contiguous(M1)) return no(__LINE__);
    int u = sole_user(s, M1);
    if (u <= i || s.done[u]) return no(__LINE__);
    const ggml_tensor * SC = nullptr, * prev = M1; int sci = -1;
    if (g->nodes[u]->op == GGML_OP_SCALE) {
        SC = g->nodes[u]; sci = u;
        if (SC->src[0] != M1 || !same_shape(SC, M1) || SC->type != GGML_TYPE_F32 || is_out(s, SC)) return no(__LINE__);
        prev = SC; u = sole_user(s, SC);
        if (u <= sci || s.done[u]) return no(__LINE__);
    }
    const int smi = u;
    const ggml_tensor * SM = g->nodes[smi];
    if (SM->op != GGML_OP_SOFT_MAX || SM->src[0] != prev || SM->src[1] || SM->src[2] || op_param_f32(SM, 1) != 0.0f || !same_shape(SM, M1) || SM->type != GGML_TYPE_F32 || is_out(s, SM)) return no(__LINE__);
    const int m2 = sole_user(s, SM);
    if (m2 <= smi || s.done[m2]) return no(__LINE__);
    const ggml_tensor * M2 = g->nodes[m2];
    if (M2->op != GGML_OP_MUL_MAT || M2->src[1] != SM || M2->type != GGML_TYPE_F32 || !is_contiguous(M2)) return no(__LINE__);
    const ggml_tensor * fv = M2->src[0];
    if (fv->type != GGML_TYPE_F32 || fv->ne[0] != nkv || fv->ne[1] != D || fv->ne[2] != HB || fv->ne[3] != 1 || fv->nb[0] != 4 || !fv->data) return no(__LINE__);
    if (M2->ne[0] != D || M2->ne[1] != nq || M2->ne[2] != HB || M2->ne[3] != 1) return no(__LINE__);
    attn_f32_args a;
    a.q = fq->data; a.q_rs = fq->nb[1]; a.q_bs = fq->nb[2]; a.k = fk->data; a.k_rs = fk->nb[1]; a.k_bs = fk->nb[2]; a.vt = fv->data; a.v_rs = fv->nb[1]; a.v_bs = fv->nb[2];
    a.D = D; a.nq = nq; a.nkv = nkv; a.HB = HB;
    // Q / K whose flattening copy CONT(PERMUTE([D, H, n, B])) was left un-run (lazy_try_register, case C): read through the permuted view's strides
    auto lazy_root = [&](const ggml_tensor * t) -> const ggml_tensor * { while (t && t->op == GGML_OP_RESHAPE) t = t->src[0]; return t && s.lazy.count(t) ? t : nullptr; };
    const ggml_tensor * lq = lazy_root(fq), * lk = lazy_root(fk);
    byte_range rq = range_of(fq), rk = range_of(fk);
    if (lq) { const tdesc & d = s.lazy[lq].src; if (d.ne[0] != D || d.ne[1] != nq || d.ne[2] * d.ne[3] != HB || d.nb[0] != 4) return no(__LINE__);
              a.q = d.p; a.q_rs = d.nb[1]; a.q_bs = d.nb[2]; a.q_bs2 = d.nb[3]; a.q_H = d.ne[2]; rq = range_of(d); }
    if (lk) { const tdesc & d = s.lazy[lk].src; if (d.ne[0] != D || d.ne[1] != nkv || d.ne[2] * d.ne[3] != HB || d.nb[0] != 4) return no(__LINE__);
              a.k = d.p; a.k_rs = d.nb[1]; a.k_bs = d.nb[2]; a.k_bs2 = d.nb[3]; a.k_H = d.ne[2]; rk = range_of(d); }
    a.has_scale = SC != nullptr; if (SC) { a.s1 = op_param_f32(SC, 0); a.b1 = op_param_f32(SC, 1); } a.s2 = op_param_f32(SM, 0);
    // the result as it is, or through views into the CONT of its [D, H, nq, ns] permutation
    const ggml_tensor * out = M2; int ci = -1;
    a.dst = M2->data; a.d_nb_q = M2->nb[1]; a.d_nb_h = M2->nb[2]; a.d_nb_s = 0; a.H = HB;
    if (!is_out(s, M2)) {
        auto views_back_to = [](const ggml_tensor * w, const ggml_tensor * t) { while (w && w != t) w = (w->op == GGML_OP_RESHAPE || w->op == GGML_OP_VIEW || w->op == GGML_OP_PERMUTE || w->op == GGML_OP_TRANSPOSE) ? w->src[0] : nullptr; return w != nullptr; };
        const int cu = sole_user(s, M2);
        if (cu > m2 && !s.done[cu] && g->nodes[cu]->op == GGML_OP_CONT && next_real_node(s, m2) == cu) {
            const ggml_tensor * C = g->nodes[cu], * cs = C->src[0];
            const int64_t H = cs->ne[1], ns = cs->ne[3];
            if (views_back_to(cs, M2) && C->type == GGML_TYPE_F32 && is_contiguous(C) && !C->view_src && C->data && cs->data == M2->data && cs->ne[0] == D && cs->ne[2] == nq && H * ns == HB &&
                cs->nb[0] == 4 && cs->nb[1] == M2->nb[2] && cs->nb[2] == M2->nb[1] && (ns == 1 || cs->nb[3] == (size_t) H * M2->nb[2])) {
                bool inner_ok = true;
                for (const ggml_tensor * w = cs; w != M2; w = w->src[0]) if (is_out(s, w)) inner_ok = false;
                if (inner_ok && nelements(C) == D * HB * nq) {           // (the CONT may carry any shape of the same elements -- ggml_cont_2d in the encoders: strides of the dense [D, H, nq, ns] order)
                    out = C; ci = cu; a.dst = C->data; a.d_nb_h = (size_t) D * 4; a.d_nb_q = (size_t) D * (size_t) H * 4; a.d_nb_s = (size_t) D * (size_t) H * (size_t) nq * 4; a.H = H;
                }
            }
        }
    }
    const int last = ci >= 0 ? ci : m2;
    for (int k = i + 1; k < last; ++k)
        if (k != sci && k != smi && k != m2 && !s.done[k] && !is_noop(g->nodes[k])) return no(__LINE__);       // something else runs in between: keep the separate launches
    if (!attn_f32_ok(a)) return no(__LINE__);
    // the result is written while other workgroups still read the operands: its buffer (placed by ggml-alloc for a later point of the graph) must not sit on them
    // (a lazy operand's source is dead for ggml-alloc behind its copy's node, so the result may have been placed on it: then the copy is made after all and read instead)
    if (lq && overlap(range_of(out), rq)) { lazy_materialise(s, lq, (int) GGML_OP_MUL_MAT); lq = nullptr; a.q = fq->data; a.q_rs = fq->nb[1]; a.q_bs = fq->nb[2]; a.q_bs2 = 0; a.q_H = 0; rq = range_of(fq); }
    if (lk && overlap(range_of(out), rk)) { lazy_materialise(s, lk, (int) GGML_OP_MUL_MAT); lk = nullptr; a.k = fk->data; a.k_rs = fk->nb[1]; a.k_bs = fk->nb[2]; a.k_bs2 = 0; a.k_H = 0; rk = range_of(fk); }
    if (overlap(range_of(out), rq) || overlap(range_of(out), rk)) return no(__LINE__);
    // V^T = CONT(PERMUTE(V)) left un-run (case C'): the second product reads V itself, [nkv, D, H, B] with the keys a row apart
    const ggml_tensor * lv = lazy_root(fv);
    byte_range rv = range_of(fv);
    if (lv) {
        const tdesc & d = s.lazy[lv].src;
        if (d.ne[0] == nkv && d.ne[1] == D && d.ne[2] * d.ne[3] == HB && d.nb[1] == 4 && (d.nb[0] & 3) == 0 && !overlap(range_of(out), range_of(d))) {
            a.vt = d.p; a.v_ks = d.nb[0]; a.v_bs = d.nb[2]; a.v_bs2 = d.nb[3]; a.v_H = d.ne[2]; a.v_rs = 0; rv = range_of(d);
        } else { lazy_materialise(s, lv, (int) GGML_OP_MUL_MAT); lv = nullptr; }
    }
    if (overlap(range_of(out), rv)) return no(__LINE__);
    if (s.pr.A) materialise_reduce(s);
    if (s.prm.n) materialise_group(s);
    if (s.pn.m && (s.pn.m == fq || s.pn.m == fk || s.pn.m == fv)) materialise_norm(s);
    {
        prof_scope ps(s, "attn_f32", 4.0 * (double) D * (double) nq * (double) nkv * (double) HB);
        attn_f32(a, s.st); ++s.n_kernels;
    }
    for (int k : { sci, smi, m2, ci }) if (k >= 0) { s.done[k] = 1; ++s.n_fused; }
    if (lq) s.lazy.erase(lq);                                             // their one reader has run: the copies are never made
    if (lk) s.lazy.erase(lk);
    if (lv) s.lazy.erase(lv);
    note_write(s, out);
    if (dbg) ++why[0];
    return true;
}

static bool exec_attn_sm_prefill(exec_state & s, int i, bool dry) {       // dry: would this MUL_MAT be taken?  (no launches, no state)
    static const bool off = getenv("MI355X_NO_ATTN_SM_PREFILL") != nullptr;
    ggml_cgraph * g = s.g;
    const ggml_tensor * M1 = g->nodes[i];
    if (off || !s.c->opt_fusion || M1->op != GGML_OP_MUL_MAT || is_out(s, M1)) return false;
    const ggml_tensor * fk = M1->src[0], * fq = M1->src[1];
    if (fk->type != GGML_TYPE_F16 || fq->type != GGML_TYPE_F32 || M1->type != GGML_TYPE_F32 || fk->nb[0] != 2 || fq->nb[0] != 4) return false;
    const int64_t D = fk->ne[0], nkv = fk->ne[1], HK = fk->ne[2], ns = fk->ne[3], nq = fq->ne[1], H = fq->ne[2];
    if ((D != 64 && D != 128) || fq->ne[0] != D || nq <= 32 || HK <= 0 || H % HK != 0 || fq->ne[3] != ns || nkv <= 0) return false;
    const int smi = sole_user(s, M1);
    if (smi <= i || s.done[smi]) return false;
    const ggml_tensor * SM = g->nodes[smi];
    if (SM->op != GGML_OP_SOFT_MAX || SM->src[0] != M1 || SM->src[2] || op_param_f32(SM, 1) != 0.0f || !same_shape(SM, M1) || is_out(s, SM)) return false;
    const ggml_tensor * mk = SM->src[1];
    if (mk && ((mk->type != GGML_TYPE_F32 && mk->type != GGML_TYPE_F16) || mk->ne[0] != nkv || mk->ne[1] < nq || mk->ne[2] != 1 || mk->ne[3] != 1 ||
               mk->nb[0] != (mk->type == GGML_TYPE_F32 ? 4u : 2u))) return false;
    const int m2 = sole_user(s, SM);
    if (m2 <= smi || s.done[m2]) return false;
    const ggml_tensor * M2 = g->nodes[m2];
    if (M2->op != GGML_OP_MUL_MAT || M2->src[1] != SM || M2->type != GGML_TYPE_F32 || is_out(s, M2)) return false;
    const ggml_tensor * fv = M2->src[0];
    if (fv->type != GGML_TYPE_F16 || fv->ne[0] != nkv || fv->ne[1] != D || fv->ne[2] != HK || fv->ne[3] != ns || fv->nb[0] != 2) return false;
    if (M2->ne[0] != D || M2->ne[1] != nq || M2->ne[2] != H || M2->ne[3] != ns || M2->nb[0] != 4) return false;
    // -> views -> CONT of the [D, H, nq, ns] permutation
    auto views_back_to = [](const ggml_tensor * w, const ggml_tensor * t) { while (w && w != t) w = w->view_src; return w != nullptr; };
    const ggml_tensor * t = M2; int ci = -1;
    for (int hop = 0; hop < 4; ++hop) {
        const int u = sole_user(s, t);
        if (u < 0) return false;
        const ggml_tensor * c = g->nodes[u];
        if (c->op == GGML_OP_CONT) { if (!views_back_to(c->src[0], t)) return false; ci = u; break; }
        if (!is_noop(c)) return false;
        t = c;
    }
    if (ci <= m2 || s.done[ci]) return false;
    const ggml_tensor * C = g->nodes[ci], * cs = C->src[0];
    if (C->type != GGML_TYPE_F32 || !is_contiguous(C) || nelements(C) != D * H * nq * ns || cs->data != M2->data || cs->ne[0] != D || cs->ne[1] != H || cs->ne[2] != nq || cs->ne[3] != ns ||
        cs->nb[0] != 4 || cs->nb[1] != M2->nb[2] || cs->nb[2] != M2->nb[1] || (ns > 1 && cs->nb[3] != M2->nb[3])) return false;
    for (int k = i + 1; k < ci; ++k)
        if (k != smi && k != m2 && !s.done[k] && !is_noop(g->nodes[k])) return false;       // something else runs in between: keep the separate launches
    fattn_args f; tdesc m;
    f.q = td(fq); f.k = td(fk); f.v = s.va.cast == fv ? s.va.v : td(fv); f.v_transposed = true; f.kv_type = GGML_TYPE_F16;
    f.dst = td(C);
    f.dst.ne[0] = D; f.dst.ne[1] = H; f.dst.ne[2] = nq; f.dst.ne[3] = ns;
    f.dst.nb[0] = 4; f.dst.nb[1] = (size_t) D * 4; f.dst.nb[2] = (size_t) D * H * 4; f.dst.nb[3] = (size_t) D * H * nq * 4;
    f.mask = nullptr; f.sinks = nullptr; f.scale = op_param_f32(SM, 0); f.max_bias = 0.0f; f.logit_softcap = 0.0f;
    f.scratch = nullptr; f.scratch_bytes = 0;
    if (!fattn_sm_prefill_ok(f)) return false;
    if (mk) {
        const size_t map_b0 = attn_sm_mask16_off(nq, nkv), m16_b0 = mk->type == GGML_TYPE_F32 ? (size_t) mk->ne[1] * (size_t) nkv * 2 : 0;
        if (!s.c->fa_scratch || s.c->fa_scratch_bytes < map_b0 + m16_b0) return false;
    }
    if (dry) return true;
    if (mk) {
        m = td(mk);
        const size_t map_b = attn_sm_mask16_off(nq, nkv), m16_b = mk->type == GGML_TYPE_F32 ? (size_t) mk->ne[1] * (size_t) nkv * 2 : 0;
        if (!s.c->fa_scratch || s.c->fa_scratch_bytes < map_b + m16_b) return false;
        const bool valid = s.fa_mask == mk->data && s.fa_dims[0] == mk->ne[0] && s.fa_dims[1] == nq && s.fa_dims[2] == mk->ne[2] && s.fa_dims[3] == mk->ne[3] && s.fa_mnb1 == mk->nb[1];
        if (mk->type == GGML_TYPE_F32) {
            tdesc m16 = m;
            m16.p = (char *) s.c->fa_scratch + map_b; m16.nb[0] = 2; m16.nb[1] = (size_t) nkv * 2; m16.nb[2] = m16.nb[1] * (size_t) mk->ne[1]; m16.nb[3] = m16.nb[2];
            if (!valid) { prof_scope ps(s, "cpy", 0); cpy_strided(m, GGML_TYPE_F32, m16, GGML_TYPE_F16, s.st); ++s.n_kernels; }
            m = m16;
        }
        f.mask = &m; f.scratch = s.c->fa_scratch; f.scratch_bytes = map_b; f.map_valid = valid;
        if (!valid) { s.fa_mask = mk->data; s.fa_dims[0] = mk->ne[0]; s.fa_dims[1] = nq; s.fa_dims[2] = mk->ne[2]; s.fa_dims[3] = mk->ne[3]; s.fa_mnb1 = mk->nb[1]; ++s.n_kernels; }
    }
    // the CONT's rows [D * H, nq * ns] read only by GEMMs (wo): emit them in f16 from the kernel
    const ggml_tensor * xg16 = nullptr;
    if (gemm_only_consumers(s, C, D * H, nq * ns, &xg16)) {
        f.out16 = (uint16_t *) s.c->act_scratch; f.out16_rs = act_image_bytes(ACT_F16, D * H); f.write_f32 = n_users(s, C) > 1;
    }
    {
        prof_scope ps(s, "fattn", 0);
        flash_attn_ext_f16(f, s.st); ++s.n_kernels;
    }
    s.done[smi] = 1; s.done[m2] = 1; s.done[ci] = 1; s.n_fused += 3;
    if (s.va.cast == fv) s.va.cast = nullptr;
    note_write(s, C);
    if (xg16) { seed_act_f16(s, xg16); ++s.n_fused; }
    return true;
}

// The streaming Whisper graph (audition.cpp:519-607) stores V TRANSPOSED in its cache -- row (h, d) of V^T holds the cells contiguously, kv_size apart -- and then, every
// chunk, copies the whole window back twice: V_2d_t = CONT(TRANSPOSE(view of the cache)) and V = CAST(PERMUTE(RESHAPE(V_2d_t)), F16), a contiguous [n_kv, D, H] block, which is
// what the second mat-mul of the attention reads.  Element (kv, d, h) of that block is element (h D + d, kv) of the cache view: exactly the V^T rows the fused soft-max
// attention stages as they lie (fa_dev::vt).  At the CONT node: if its only reader chain is that CAST and the CAST's only reader is the second mat-mul of a chain
// exec_attn_sm_prefill accepts with the aliased V, neither copy runs -- 2 launches and 2 x the window per layer and chunk.
// (the attention was not fused after all -- cannot happen while the dry run and the real one see the same graph state, but a reader of the CAST's block must never find it
// unwritten: run the two copies now)
static void materialise_vt(exec_state & s) {
    ggml_cgraph * g = s.g;
    const ggml_tensor * C = g->nodes[s.va.cont_i], * K = g->nodes[s.va.cast_i];
    s.va.cast = nullptr;
    prof_scope ps(s, "cpy", 0);
    cpy_strided(td(C->src[0]), C->src[0]->type, td(C), C->type, s.st);
    cpy_strided(td(K->src[0]), K->src[0]->type, td(K), K->type, s.st);
    s.n_kernels += 2;
}
static bool try_alias_vt(exec_state & s, int i) {
    static const bool off = getenv("MI355X_NO_ATTN_VT_ALIAS") != nullptr;
    ggml_cgraph * g = s.g;
    const ggml_tensor * C = g->nodes[i];
    if (off || !s.c->opt_fusion || s.va.cast || C->op != GGML_OP_CONT || C->type != GGML_TYPE_F16 || is_out(s, C) || !is_contiguous(C) || C->ne[2] != 1 || C->ne[3] != 1) return false;
    const ggml_tensor * T = C->src[0];                                  // [n_state, n_kv] with the cells contiguous: nb[1] == 2, nb[0] = the cache's row pitch
    if (!T || T->type != GGML_TYPE_F16 || T->ne[0] != C->ne[0] || T->ne[1] != C->ne[1] || T->ne[2] != 1 || T->ne[3] != 1 || T->nb[1] != 2 || T->nb[0] < (size_t) T->ne[1] * 2 || T->nb[0] % 2 != 0) return false;
    const int64_t n_state = C->ne[0], nkv = C->ne[1];
    int ci = -1;                                                        // the CAST: a CPY whose source is a [n_kv, D, H] view of C, behind the RESHAPE / PERMUTE view nodes
    { const ggml_tensor * t = C;
      for (int hop = 0; hop < 4; ++hop) {
          const int u = sole_user(s, t);
          if (u < 0) return false;
          if (g->nodes[u]->op == GGML_OP_CPY) { ci = u; break; }
          if (!is_noop(g->nodes[u])) return false;
          t = g->nodes[u];
      } }
    if (ci <= i || s.done[ci]) return false;
    const ggml_tensor * K = g->nodes[ci];
    const ggml_tensor * P = K->src[0];
    if (K->op != GGML_OP_CPY || K->type != GGML_TYPE_F16 || is_out(s, K) || !is_contiguous(K) || !P || P->type != GGML_TYPE_F16 || P->data != C->data) return false;
    { const ggml_tensor * w = P; while (w && w != C) w = w->view_src; if (!w) return false; }
    const int64_t D = P->ne[1], H = P->ne[2];
    if (P->ne[0] != nkv || D <= 0 || H <= 0 || D * H != n_state || P->ne[3] != 1 || P->nb[0] != (size_t) n_state * 2 || P->nb[1] != 2 || P->nb[2] != (size_t) D * 2) return false;
    if (K->ne[0] != nkv || K->ne[1] != D || K->ne[2] != H || K->ne[3] != 1) return false;
    for (int k = i + 1; k < ci; ++k) if (!s.done[k] && !is_noop(g->nodes[k])) return false;
    int m2 = -1;                                                        // (ggml_cast names its result as its own src[1]: the CAST is among its own users)
    { auto uit = s.users.find(K);
      if (uit == s.users.end() || is_out(s, K)) return false;
      for (int u : uit->second) { if (u == ci) continue; if (m2 >= 0 && u != m2) return false; m2 = u; } }
    if (m2 <= ci || s.done[m2]) return false;
    const ggml_tensor * M2 = g->nodes[m2];
    if (M2->op != GGML_OP_MUL_MAT || M2->src[0] != K || !M2->src[1] || M2->src[1]->op != GGML_OP_SOFT_MAX) return false;
    const ggml_tensor * M1 = M2->src[1]->src[0];
    auto it = M1 ? s.index.find(M1) : s.index.end();
    if (it == s.index.end() || it->second <= ci || s.done[it->second]) return false;
    // nothing between the CAST and the attention may write the cache rows (it is read at the attention launch, not here)
    const byte_range rv = { (const char *) T->data, (const char *) T->data + (size_t) (n_state - 1) * T->nb[0] + (size_t) nkv * 2 };
    for (int k = ci + 1; k < it->second; ++k) if (!s.done[k] && !is_noop(g->nodes[k]) && overlap(range_of(g->nodes[k]), rv)) return false;
    s.va.cast = K; s.va.cont_i = i; s.va.cast_i = ci;
    s.va.v.p = (char *) T->data;
    s.va.v.ne[0] = nkv; s.va.v.ne[1] = D; s.va.v.ne[2] = H; s.va.v.ne[3] = 1;
    s.va.v.nb[0] = 2; s.va.v.nb[1] = T->nb[0]; s.va.v.nb[2] = (size_t) D * T->nb[0]; s.va.v.nb[3] = (size_t) n_state * T->nb[0];
    if (!exec_attn_sm_prefill(s, it->second, true)) { s.va.cast = nullptr; return false; }
    s.done[ci] = 1; s.n_fused += 2;                                      // (this CONT and the CAST: never launched, their blocks never written)
    return true;
}

// ------------------------------------------------------------------------------------------------ node dispatch
static void compute_node(exec_state & s, int i) {
    ggml_cgraph * g = s.g;
    ggml_tensor * n = g->nodes[i];
    if (is_noop(n)) return;
    if (s.pr.A && !((n->op == GGML_OP_RMS_NORM || n->op == GGML_OP_NORM) && n->src[0] == s.pr.A)) materialise_reduce(s);      // somebody else reads the split-K result first
    if (s.prm.n && !reads_pending_group(s, n)) materialise_group(s);

    switch (n->op) {
        case GGML_OP_MUL_MAT:
            if (!s.lazy.empty()) {                                        // (run_nodes leaves the net to this point for f32 x f32 products: the attention chain reads lazy operands in place)
                if (n->src[0]->type == GGML_TYPE_F32 && n->src[1]->type == GGML_TYPE_F32 && exec_attn_f32(s, i)) return;
                lazy_net(s, i);
            }
            if (s.pq.sm && s.pq.fa == i) {                                // flash-attention off, one token: K.q, soft-max, V^T.p, permute + cont and the q / k / v pre-stage in one launch
                const fattn_pre & P = s.pq.pre;
                attn_sm_args & a = s.pq.sma;
                const bool valid = s.rt.pos == (const void *) P.pos && s.rt.ff == (const void *) P.ff && s.rt.T == 1 && s.rt.D == a.D && memcmp(&s.rt.rp, &P.rp, sizeof(rope_params)) == 0;
                if (!valid) {
                    prof_scope ps(s, "rope", 0);
                    rope_table(P.pos, P.ff, P.rp, 1, a.D, (float *) s.c->rope_scratch, s.st); ++s.n_kernels;
                    s.rt.pos = P.pos; s.rt.ff = P.ff; s.rt.T = 1; s.rt.D = a.D; s.rt.rp = P.rp;
                }
                a.rope_tab = (const float *) s.c->rope_scratch;
                {
                    prof_scope ps(s, "fattn", 0);
                    attn_one_sm(a, s.st); ++s.n_kernels;
                }
                s.done[s.pq.sm_soft] = 1; s.done[s.pq.sm_mm2] = 1; s.done[s.pq.sm_cont] = 1; s.n_fused += 3;
                note_write(s, g->nodes[s.pq.sm_cont]); note_write(s, g->nodes[s.pq.kst]); note_write(s, g->nodes[s.pq.vst]);
                s.pq.fa = -1; s.pq.sm = false;
                return;
            }
            if (exec_attn_sm_prefill(s, i, false)) return;
            if (exec_attn_f32(s, i)) return;
            if (s.va.cast && (n->src[0] == s.va.cast || g->nodes[i]->src[1]->op == GGML_OP_SOFT_MAX)) materialise_vt(s);
            exec_mul_mat(s, i);
            return;
        case GGML_OP_IM2COL: {
            prof_scope ps(s, "im2col", 0);
            im2col_f32(td(n->src[0]), td(n->src[1]), td(n), n->type, n->op_params, s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_POOL_1D: case GGML_OP_POOL_2D: {
            prof_scope ps(s, "pool", 0);
            pool_f32(td(n->src[0]), n->src[0]->type, td(n), n->op_params, n->op == GGML_OP_POOL_2D, s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_NORM: {
            if (exec_norm_modulate(s, i)) return;
            if (exec_norm(s, i)) return;
            if (s.pr.A) materialise_reduce(s);
            prof_scope ps(s, "norm", 0);
            norm_f32(td(n->src[0]), td(n), op_param_f32(n, 0), s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_RMS_NORM: {
            if (exec_rms_norm(s, i)) return;
            if (s.pr.A) materialise_reduce(s);
            if (s.prm.n) materialise_group(s);
            prof_scope ps(s, "rms_norm", 0);
            rms_norm(td(n->src[0]), td(n), op_param_f32(n, 0), nullptr, s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: {
            prof_scope ps(s, "bin", 0);
            bin_bcast_f32(n->op, td(n->src[0]), td(n->src[1]), td(n), s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_SCALE: {
            prof_scope ps(s, "scale", 0);
            scale_f32((const float *) n->src[0]->data, (float *) n->data, nelements(n), op_param_f32(n, 0), op_param_f32(n, 1), s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_SQR: case GGML_OP_SQRT: case GGML_OP_LOG: case GGML_OP_SIN: case GGML_OP_COS: case GGML_OP_CLAMP: case GGML_OP_LEAKY_RELU: {
            prof_scope ps(s, "math", 0);
            math_f32(n->op, (const float *) n->src[0]->data, (float *) n->data, nelements(n), op_param_f32(n, 0), op_param_f32(n, 1), s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_CONCAT: {
            // an operand whose copy was left un-run (lazy_try_register) is read where its source lies; anything lazy further up a view chain is made real first
            tdesc sd[2] = { td(n->src[0]), td(n->src[1]) };
            const ggml_tensor * lz[2] = { nullptr, nullptr };
            if (!s.lazy.empty()) {
                for (int k = 0; k < 2; ++k) {
                    auto it = s.lazy.find(n->src[k]);
                    if (it != s.lazy.end() && it->second.deadline > i && n->type == GGML_TYPE_F32 && !overlap(range_of(n), range_of(it->second.src))) { sd[k] = it->second.src; lz[k] = n->src[k]; }
                }
                // taken in place.  The entry goes only when this node was the copy's LAST reader: a cached-frames copy (lazy_try_register case A) is also read by the
                // transposing CONT behind it -- dropped here, that reader would find no entry and copy from a buffer nobody wrote
                for (int k = 0; k < 2; ++k) if (lz[k]) {
                    bool other = false;
                    auto us = s.users.find(lz[k]);
                    if (us != s.users.end()) for (int u : us->second) if (u != i && u > i && !s.done[u]) other = true;
                    if (!other) s.lazy.erase(lz[k]);
                }
                lazy_net(s, i);                                            // everything else this node reads, and the deadlines
            }
            prof_scope ps(s, "concat", 0);
            concat(sd[0], sd[1], td(n), op_param_i32(n, 0), n->type == GGML_TYPE_F16 ? 2 : 4, s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_REPEAT: {
            prof_scope ps(s, "repeat", 0);
            repeat(td(n->src[0]), td(n), n->type == GGML_TYPE_F16 ? 2 : 4, s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_PAD: {
            prof_scope ps(s, "pad", 0);
            pad_f32(td(n->src[0]), td(n), n->op_params, s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_PAD_REFLECT_1D: {
            prof_scope ps(s, "pad_reflect", 0);
            pad_reflect_1d_f32(td(n->src[0]), td(n), op_param_i32(n, 0), op_param_i32(n, 1), s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_ARANGE: {
            prof_scope ps(s, "arange", 0);
            arange_f32((float *) n->data, nelements(n), op_param_f32(n, 0), op_param_f32(n, 2), s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_TIMESTEP_EMBEDDING: {
            prof_scope ps(s, "timestep_embedding", 0);
            timestep_embedding_f32((const float *) n->src[0]->data, td(n), n->src[0]->ne[0], op_param_i32(n, 0), op_param_i32(n, 1), s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_SUM_ROWS: {
            prof_scope ps(s, "sum_rows", 0);
            sum_rows_f32(td(n->src[0]), td(n), s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_CONV_TRANSPOSE_1D: {
            prof_scope ps(s, "conv_transpose_1d", 0);
            conv_transpose_1d_f32(td(n->src[0]), n->src[0]->type, td(n->src[1]), td(n), op_param_i32(n, 0), s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_UNARY: {
            // the activation between two mat-muls of a prefill-sized block (the encoders' GELU between fc1 and fc2): only MFMA GEMMs read it -> its f16 image is
            // written here (dense when the row length is a multiple of 8); the f32 block only when the reader is not the very next launch
            const ggml_tensor * xg = nullptr;
            const bool emit16 = n->ne[2] == 1 && n->ne[3] == 1 && n->ne[0] % 8 == 0 && n->nb[1] == (size_t) n->ne[0] * 4 && (((uintptr_t) n->data | (uintptr_t) n->src[0]->data) & 15) == 0 &&
                                gemm_only_consumers(s, n, n->ne[0], n->ne[1], &xg);
            const int u1 = emit16 ? sole_user(s, n) : -1;
            const bool w32 = !(emit16 && u1 > i && next_real_node(s, i) == u1);
            {
                prof_scope ps(s, "unary", 0);
                unary_f32(op_param_i32(n, 0), (const float *) n->src[0]->data, (float *) n->data, nelements(n), s.st, emit16 ? (uint16_t *) s.c->act_scratch : nullptr, w32);
            }
            ++s.n_kernels;
            if (emit16) { note_write(s, n); seed_act_f16(s, xg); ++s.n_fused; return; }
            break;
        }
        case GGML_OP_GLU: {
            tdesc b; if (n->src[1]) b = td(n->src[1]);
            const ggml_tensor * xg = nullptr;
            const bool emit16 = n->ne[2] == 1 && n->ne[3] == 1 && n->nb[1] == (size_t) n->ne[0] * 4 && gemm_only_consumers(s, n, n->ne[0], n->ne[1], &xg);
            // every consumer a K-quant mat-mul on the whole result (ffn_down at several columns): emit the Q8_K images here
            const ggml_tensor * xq = nullptr;
            if (s.c->opt_fusion && !emit16 && n->src[1] && op_param_i32(n, 0) == GGML_GLU_OP_SWIGLU && op_param_i32(n, 1) == 0 && n_users(s, n) > 0 &&
                !is_out(s, n) && n->nb[1] == (size_t) n->ne[0] * 4 && swiglu_q8k_ok(td(n->src[0]), b, td(n))) {
                bool ok = true;
                for (int u : s.users[n]) {
                    const ggml_tensor * c = g->nodes[u];
                    const ggml_tensor * x = c->op == GGML_OP_MUL_MAT ? c->src[1] : nullptr;
                    if (!x || !kq_mm_ok(c) || x->data != n->data || x->ne[0] != n->ne[0] || x->ne[1] != n->ne[1] || x->nb[1] != n->nb[1] || (xq && !same_act(xq, x))) { ok = false; break; }
                    xq = x;
                }
                if (!ok) xq = nullptr;
            }
            if (xq) {
                {
                    prof_scope ps(s, "glu", 0);
                    swiglu_q8k(td(n->src[0]), b, td(n), true, s.c->act_scratch, s.st);      // (f32 too: the image cache may be dropped before the consumer runs)
                }
                ++s.n_kernels; ++s.n_fused;
                note_write(s, n);
                s.a_src = xq->data; s.a_kind = ACT_Q8K; s.a_K = xq->ne[0]; s.a_ne[0] = xq->ne[1]; s.a_ne[1] = 1; s.a_ne[2] = 1;
                s.a_nb[0] = xq->nb[1]; s.a_nb[1] = xq->nb[2]; s.a_nb[2] = xq->nb[3];
                s.a_range_lo = (const char *) xq->data; s.a_range_hi = (const char *) xq->data + nbytes(xq);
                return;
            }
            {
                prof_scope ps(s, "glu", 0);
                if (emit16) glu_f32(op_param_i32(n, 0), td(n->src[0]), n->src[1] ? &b : nullptr, op_param_i32(n, 1) != 0, td(n), s.st,
                                    (uint16_t *) s.c->act_scratch, act_image_bytes(ACT_F16, n->ne[0]), n_users(s, n) > 1);
                else        glu_f32(op_param_i32(n, 0), td(n->src[0]), n->src[1] ? &b : nullptr, op_param_i32(n, 1) != 0, td(n), s.st);
            }
            ++s.n_kernels;
            note_write(s, n);
            if (emit16) { seed_act_f16(s, xg); ++s.n_fused; }
            return;
        }
        case GGML_OP_ROPE: {
            if (s.c->opt_fusion && exec_rope_chain(s, i)) return;
            rope_params rp;
            rp.n_dims = op_param_i32(n, 1); rp.mode = op_param_i32(n, 2); rp.n_ctx_orig = op_param_i32(n, 4);
            rp.freq_base = op_param_f32(n, 5); rp.freq_scale = op_param_f32(n, 6); rp.ext_factor = op_param_f32(n, 7);
            rp.attn_factor = op_param_f32(n, 8); rp.beta_fast = op_param_f32(n, 9); rp.beta_slow = op_param_f32(n, 10);
            prof_scope ps(s, "rope", 0);
            rope_f32(td(n->src[0]), (const int32_t *) n->src[1]->data, n->src[2] ? (const float *) n->src[2]->data : nullptr, td(n), rp, s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_SOFT_MAX: {
            tdesc m; if (n->src[1]) m = td(n->src[1]);
            // the probabilities of a prefill ubatch without FLASH_ATTN_EXT feed exactly one MUL_MAT (V^T . P, one product per head) on the MFMA GEMM:
            // emit its f16 activation image here -- the f32 block is neither written nor converted
            const ggml_tensor * xg = nullptr;
            {
                const int u = s.c->opt_fusion && !is_out(s, n) ? sole_user(s, n) : -1;
                const ggml_tensor * c = u > i ? g->nodes[u] : nullptr;
                static const bool off = getenv("MI355X_NO_F16_EMIT") != nullptr;
                if (!off && c && c->op == GGML_OP_MUL_MAT && c->src[1] == n && c->src[0] != n && (mm_uses_gemm(c) || mm_uses_gemm_any_f16(c)) && next_real_node(s, i) == u && is_contiguous(n) && n->ne[1] > MI_MMVQ_MAX_COLS &&
                    act_image_bytes(ACT_F16, n->ne[0]) * (size_t) (n->ne[1] * n->ne[2] * n->ne[3]) <= s.c->act_scratch_bytes &&
                    soft_max_rows_ok(td(n->src[0]), n->src[1] ? &m : nullptr, n->src[1] ? n->src[1]->type : 0, n->src[2] ? (const float *) n->src[2]->data : nullptr, td(n)))
                    xg = n;
            }
            prof_scope ps(s, "soft_max", 0);
            soft_max_f32(td(n->src[0]), n->src[1] ? &m : nullptr, n->src[1] ? n->src[1]->type : 0,
                         n->src[2] ? (const float *) n->src[2]->data : nullptr, td(n), op_param_f32(n, 0), op_param_f32(n, 1), s.st,
                         xg ? (uint16_t *) s.c->act_scratch : nullptr, xg ? act_image_bytes(ACT_F16, n->ne[0]) : 0, xg == nullptr);
            ++s.n_kernels;
            if (xg) {
                s.a_src = xg->data; s.a_kind = ACT_F16; s.a_K = xg->ne[0]; s.a_ne[0] = xg->ne[1]; s.a_ne[1] = xg->ne[2]; s.a_ne[2] = xg->ne[3];
                s.a_nb[0] = xg->nb[1]; s.a_nb[1] = xg->nb[2]; s.a_nb[2] = xg->nb[3];
                s.a_range_lo = (const char *) xg->data; s.a_range_hi = (const char *) xg->data + nbytes(xg);
                ++s.n_fused;
                return;                                             // (no note_write: the f32 block was not written)
            }
            break;
        }
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP: {
            if (n->op == GGML_OP_CONT && try_alias_vt(s, i)) return;
            prof_scope ps(s, "cpy", 0);
            const ggml_tensor * src = n->src[0];
            static const bool dbg_cpy = getenv("MI355X_DEBUG_CPY") != nullptr;
            if (dbg_cpy) fprintf(stderr, "[mi355x] cpy node %s: %s [%lld, %lld, %lld, %lld] type %d nb [%zu, %zu, %zu] -> type %d\n", n->name, src->name, (long long) src->ne[0], (long long) src->ne[1],
                                 (long long) src->ne[2], (long long) src->ne[3], (int) src->type, src->nb[1], src->nb[2], src->nb[3], (int) n->type);
            // CPY writes into src[1]'s storage, which `n` is a view of; n->data is the destination in all three ops
            if ((src->type == GGML_TYPE_F32) != (n->type == GGML_TYPE_F32) && (src->type == GGML_TYPE_I32 || n->type == GGML_TYPE_I32)) cast_f32_i32(td(src), src->type == GGML_TYPE_F32, td(n), s.st);
            else {
                // CONT(PERMUTE(kqv)) of a prefill ubatch whose only readers are MFMA GEMMs (wo): gather straight into the f16 activation image
                const ggml_tensor * xg = nullptr;
                // The f32 tensor itself is then never written, so the path is taken only when the one reader is the NEXT launching node (the rule of
                // SOFT_MAX / UNARY / exec_norm): anything in between that prepares another activation image would evict this one, and the reader
                // would convert from n->data.
                const int cu = s.c->opt_fusion && !is_out(s, n) ? sole_user(s, n) : -1;
                if (n->op == GGML_OP_CONT && src->type == GGML_TYPE_F32 && n->type == GGML_TYPE_F32 && n->ne[2] == 1 && n->ne[3] == 1 && n->nb[1] == (size_t) n->ne[0] * 4 &&
                    cu > i && next_real_node(s, i) == cu && n_users(s, n) == 1 && gemm_only_consumers(s, n, n->ne[0], n->ne[1], &xg)) {
                    tdesc d; d.p = s.c->act_scratch; d.ne[0] = n->ne[0]; d.ne[1] = n->ne[1]; d.ne[2] = 1; d.ne[3] = 1;
                    const size_t img = act_image_bytes(ACT_F16, n->ne[0]);
                    d.nb[0] = 2; d.nb[1] = img; d.nb[2] = img * (size_t) n->ne[1]; d.nb[3] = d.nb[2];
                    cpy_strided(td(src), GGML_TYPE_F32, d, GGML_TYPE_F16, s.st);
                    ++s.n_kernels; ++s.n_fused;
                    seed_act_f16(s, xg);
                    return;                                         // (the f32 copy was not written)
                }
                cpy_strided(td(src), src->type, td(n), n->type, s.st);
            }
            ++s.n_kernels;
            break;
        }
        case GGML_OP_GET_ROWS: {
            prof_scope ps(s, "get_rows", 0);
            get_rows(td(n->src[0]), n->src[0]->type, td(n->src[1]), td(n), s.st); ++s.n_kernels;
            break;
        }
        case GGML_OP_SET_ROWS: {
            prof_scope ps(s, "set_rows", 0);
            int64_t period = 0;                                    // single-element rows: the row length of the block the reshape chain started from
            if (n->src[0]->ne[0] == 1)
                for (const ggml_tensor * v = n->src[0]->view_src; v; v = v->view_src) if (v->ne[0] > 1) { period = v->ne[0]; break; }
            set_rows(td(n->src[0]), td(n->src[1]), n->src[1]->type, td(n), n->type, s.st, period); ++s.n_kernels;
            break;
        }
        case GGML_OP_FLASH_ATTN_EXT: {
            fattn_args f; tdesc m;
            fill_fattn_args(n, f, m);
            if ((n->src[0]->ne[0] != 64 && n->src[0]->ne[0] != 128) || n->src[2]->ne[0] != n->src[0]->ne[0] || n->src[1]->type != GGML_TYPE_F16) {      // other head sizes / cache types: the generic kernel, no fused stage
                prof_scope ps(s, "fattn", 0);
                flash_attn_ext_f16(f, s.st); ++s.n_kernels;
                break;
            }
            const bool with_pre = s.pq.fa == i;
            if (with_pre) f.pre = &s.pq.pre;
            // one token over a shallow cache: the latency-optimised kernel (fattn_one.hip) takes the token's (cos, sin) from a table that is
            // computed once per graph, and leaves the Q8_K image to wo's own prologue
            bool one = false;
            if (with_pre && fattn_one_ok(f) && s.c->rope_scratch_bytes >= (size_t) n->src[0]->ne[0] * 4) {
                const fattn_pre & P = s.pq.pre;
                const int D = (int) n->src[0]->ne[0];
                const bool valid = s.rt.pos == (const void *) P.pos && s.rt.ff == (const void *) P.ff && s.rt.T == 1 && s.rt.D == D && memcmp(&s.rt.rp, &P.rp, sizeof(rope_params)) == 0;
                if (!valid) {
                    prof_scope ps(s, "rope", 0);
                    rope_table(P.pos, P.ff, P.rp, 1, D, (float *) s.c->rope_scratch, s.st); ++s.n_kernels;
                    s.rt.pos = P.pos; s.rt.ff = P.ff; s.rt.T = 1; s.rt.D = D; s.rt.rp = P.rp;
                }
                f.rope_tab = (const float *) s.c->rope_scratch;
                one = true;
                // ... as one workgroup per (KV head, 64-row slice) when the ONE reader of the rows is the next launching node, a batch-1 K-quant mat-vec the LDS-DMA engine
                // takes (wo): the slices' partial states stay in fa_scratch and that launch folds them in its prologue (mv1_source) -- the f32 rows are never written
                const int u = s.c->opt_fusion && !is_out(s, n) ? sole_user(s, n) : -1;        // (option "fattn_gs" / MI355X_FA_NO_GS: inside fattn_gs_ok)
                if (u > i && next_real_node(s, i) == u && fattn_gs_ok(f) && s.c->fa_scratch && s.c->fa_scratch_bytes >= fattn_gs_parts_bytes((int) n->ne[1], D) && mmv2_enabled()) {
                    const ggml_tensor * c = g->nodes[u];
                    const ggml_tensor * x = c->op == GGML_OP_MUL_MAT ? c->src[1] : nullptr;
                    if (x && x->data == n->data && x->ne[0] == n->ne[0] * n->ne[1] && x->ne[1] == 1 && x->ne[2] == 1 && x->ne[3] == 1 && x != s.pn.m && mv1_node_ok(s, c) && !q80_mv1_node(s, c)) {
                        mv1_args t; t.nmat = 1; t.K = x->ne[0];
                        t.m[0] = { c->src[0]->data, c->src[0]->nb[1], (float *) c->data, 0, nullptr, 0, c->src[0]->ne[1], (int) c->src[0]->type };
                        t.parts = (const float *) s.c->fa_scratch; t.nslice = fattn_gs_nslice();
                        if (mmv2_ok(t)) { f.gs_parts = (float *) s.c->fa_scratch; s.gs.n = n; s.gs.consumer = u; s.gs.nh = (int) n->ne[1]; s.gs.D = D; s.fa_mask = nullptr; }
                    }
                }
            }
            // epilogue fusion: when the attention output only feeds K-quant mat-vecs (wo), emit its Q8_K image here
            const ggml_tensor * xuse = nullptr;
            if (!one && s.c->opt_fusion && n->ne[3] == 1 && n->ne[2] <= 32 && n_users(s, n) > 0 && !is_out(s, n) &&
                rms_norm_mul_quant_ok(n->ne[0] * n->ne[1]) && fattn_can_emit_image(f)) {
                bool ok = true;
                for (int u : s.users[n]) {
                    const ggml_tensor * c = g->nodes[u];
                    const ggml_tensor * x = c->op == GGML_OP_MUL_MAT ? c->src[1] : nullptr;
                    if (!x || !kq_mm_ok(c) || x->data != n->data || x->ne[0] != n->ne[0] * n->ne[1] || x->ne[1] != n->ne[2] ||
                        x->nb[1] != (size_t) x->ne[0] * 4 || (xuse && !same_act(xuse, x))) { ok = false; break; }
                    xuse = x;
                }
                if (!ok) xuse = nullptr;
            }
            if (xuse) f.img = s.c->act_scratch;
            // prefill: the attention output [D, H, nq, ns] read as [H*D, nq*ns] rows by wo's GEMM -> emit those rows in f16 from the kernel
            const ggml_tensor * xg16 = nullptr;
            if (!xuse && fattn_uses_mma(f) && n->nb[1] == (size_t) n->ne[0] * 4 && n->nb[2] == (size_t) n->ne[0] * n->ne[1] * 4 &&
                n->nb[3] == n->nb[2] * (size_t) n->ne[2] && gemm_only_consumers(s, n, n->ne[0] * n->ne[1], n->ne[2] * n->ne[3], &xg16)) {
                f.out16 = (uint16_t *) s.c->act_scratch; f.out16_rs = act_image_bytes(ACT_F16, n->ne[0] * n->ne[1]); f.write_f32 = n_users(s, n) > 1;
            }
            if (fattn_scratch_bytes(f) > 0 && !fattn_uses_mma(f)) {       // decode kernel at long context: workspace of its KV split
                f.scratch = s.c->fa_scratch; f.scratch_bytes = s.c->fa_scratch_bytes;
                if (!s.c->fa_counters && !s.capturing) {                  // (first use is always an eager submission: captures come from the second on)
                    if (hipMalloc((void **) &s.c->fa_counters, 1024 * sizeof(unsigned)) == hipSuccess) HIP_CHECK(hipMemsetAsync(s.c->fa_counters, 0, 1024 * sizeof(unsigned), s.st));
                    else { (void) hipGetLastError(); s.c->fa_counters = nullptr; }
                }
                f.counters = s.c->fa_counters;
                s.fa_mask = nullptr;                                      // (the scratch no longer holds a mask tile map)
                ++s.n_kernels;
            } else if (fattn_scratch_bytes(f) > 0) {
                // the mask tile map is computed once per mask tensor and graph run (every layer shares the mask)
                const ggml_tensor * mk = n->src[3];
                f.scratch = s.c->fa_scratch; f.scratch_bytes = s.c->fa_scratch_bytes;
                f.map_valid = s.fa_mask == mk->data && s.fa_dims[0] == mk->ne[0] && s.fa_dims[1] == n->src[0]->ne[1] && s.fa_dims[2] == mk->ne[2] &&
                              s.fa_dims[3] == mk->ne[3] && s.fa_mnb1 == mk->nb[1];
                if (!f.map_valid) {
                    s.fa_mask = mk->data; s.fa_dims[0] = mk->ne[0]; s.fa_dims[1] = n->src[0]->ne[1]; s.fa_dims[2] = mk->ne[2]; s.fa_dims[3] = mk->ne[3];
                    s.fa_mnb1 = mk->nb[1]; ++s.n_kernels;
                }
            }
            {
                prof_scope ps(s, "fattn", 0);
                flash_attn_ext_f16(f, s.st); ++s.n_kernels;
            }
            note_write(s, n);
            if (with_pre) { note_write(s, g->nodes[s.pq.kst]); note_write(s, g->nodes[s.pq.vst]); s.pq.fa = -1; }
            if (xg16) { seed_act_f16(s, xg16); ++s.n_fused; }
            if (xuse) {
                s.a_src = xuse->data; s.a_kind = ACT_Q8K; s.a_K = xuse->ne[0]; s.a_ne[0] = xuse->ne[1]; s.a_ne[1] = 1; s.a_ne[2] = 1;
                s.a_nb[0] = xuse->nb[1]; s.a_nb[1] = xuse->nb[2]; s.a_nb[2] = xuse->nb[3];
                s.a_range_lo = (const char *) xuse->data; s.a_range_hi = (const char *) xuse->data + nbytes(xuse);
                ++s.n_fused;
            }
            return;
        }
        default:
            log_msg(GGML_LOG_LEVEL_ERROR, "[mi355x] graph_compute: op %d (%s) reached the backend but is not implemented -- supports_op bug\n", (int) n->op, n->name);
            abort();
    }
    note_write(s, n);
}

// The reference's Token2Wav builders put a ggml_cont behind most ops -- on tensors that are contiguous already (a third of a window's 15 000 launches are such
// copies).  When the producer is a plain element-wise / gather op, the copy is the very next launching node and the producer's only reader (directly or through
// RESHAPEs), the producer writes straight into the copy's buffer and the copy is not launched.  ggml-alloc may have placed the copy's buffer over memory that
// became free when the producer ran -- the producer's own sources -- so that overlap is checked.  Returns the CONT's node index or -1.
static int cont_sink(exec_state & s, int i) {
    static const bool off = getenv("MI355X_NO_CONT_SINK") != nullptr;
    static const bool dbg = getenv("MI355X_SINK_DEBUG") != nullptr;      // why a producer -> CONT pair was NOT folded, tallied per reason (stderr at process exit)
    static long why[8] = { 0 };
    struct dump { ~dump() { if (dbg) fprintf(stderr, "[mi355x] cont_sink: folded %ld | producer not a sink kind %ld | producer not plain %ld | next node no plain CONT %ld | path not RESHAPEs %ld | other readers %ld | CONT over the producer's sources %ld\n", why[0], why[1], why[2], why[3], why[4], why[5], why[6]); } };
    static dump at_exit;
    auto no = [&](int r) { if (dbg) ++why[r]; return -1; };
    if (off || !s.c->opt_fusion) return -1;
    ggml_cgraph * g = s.g;
    const ggml_tensor * p = g->nodes[i];
    if (is_noop(p)) return -1;
    switch (p->op) {
        case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: case GGML_OP_SCALE: case GGML_OP_SQR: case GGML_OP_SQRT: case GGML_OP_LOG: case GGML_OP_SIN: case GGML_OP_COS:
        case GGML_OP_CLAMP: case GGML_OP_LEAKY_RELU: case GGML_OP_CONCAT: case GGML_OP_REPEAT: case GGML_OP_PAD: case GGML_OP_PAD_REFLECT_1D: case GGML_OP_CONT: case GGML_OP_CONV_TRANSPOSE_1D:
            break;
        case GGML_OP_UNARY: break;
        default: { const int j0 = next_real_node(s, i); if (j0 >= 0 && g->nodes[j0]->op == GGML_OP_CONT) return no(1); return -1; }
    }
    if (!p->data || !is_contiguous(p) || is_out(s, p) || p->view_src) return no(2);
    const int j = next_real_node(s, i);
    if (j < 0) return -1;
    const ggml_tensor * c = g->nodes[j];
    if (c->op != GGML_OP_CONT) return -1;
    if (c->type != p->type || !c->data || c->view_src || !is_contiguous(c) || nbytes(c) != nbytes(p) || c->data == p->data) return no(3);
    for (const ggml_tensor * t = c->src[0]; t != p; t = t->src[0]) {                  // directly, or through RESHAPEs of the contiguous result
        if (!t || t->op != GGML_OP_RESHAPE || !is_contiguous(t) || is_out(s, t)) return no(4);
        auto it = s.users.find(t);
        if (it == s.users.end() || it->second.size() != 1 || it->second[0] != j) return no(5);
    }
    if (sole_user(s, p) != j) return no(5);
    const char * lo = (const char *) c->data, * hi = lo + nbytes(c);
    // (an element-wise producer may write over an operand of its own shape that sits at exactly the copy's address: every thread reads its element before it writes it)
    const bool ew = p->op == GGML_OP_ADD || p->op == GGML_OP_SUB || p->op == GGML_OP_MUL || p->op == GGML_OP_DIV || p->op == GGML_OP_SCALE || p->op == GGML_OP_SQR || p->op == GGML_OP_SQRT ||
                    p->op == GGML_OP_LOG || p->op == GGML_OP_SIN || p->op == GGML_OP_COS || p->op == GGML_OP_CLAMP || p->op == GGML_OP_LEAKY_RELU || p->op == GGML_OP_UNARY;
    for (int k = 0; k < GGML_MAX_SRC && p->src[k]; ++k) {
        const char * a = (const char *) p->src[k]->data, * b = a + nbytes(p->src[k]);
        if (a < hi && lo < b) {
            if (ew && a == lo && p->src[k]->type == p->type && same_shape(p->src[k], p) && is_contiguous(p->src[k])) continue;
            return no(6);
        }
    }
    if (dbg) ++why[0];
    return j;
}

// A run of element-wise f32 nodes, each the next launching node and the only reader of the one before (directly or through RESHAPEs), all over the same number of
// contiguous elements: one k_ew_chain launch writes the last node's result (kernels.hpp ew_chain_args).  Other operands are "external": the chain's shape element for
// element, one row of ne0 floats repeated (bias / gain / modulation vectors), or one value.  Returns the number of nodes taken (0: none; the caller marks them done).
static int exec_ew_chain(exec_state & s, int i, int * taken) {
    static const bool off = getenv("MI355X_NO_EW_CHAIN") != nullptr;
    if (off || !s.c->opt_fusion) return 0;
    ggml_cgraph * g = s.g;
    auto ew_kind = [](const ggml_tensor * n) -> bool {
        switch (n->op) {
            case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: case GGML_OP_SCALE: case GGML_OP_UNARY: case GGML_OP_SQR: case GGML_OP_SQRT: case GGML_OP_LOG:
            case GGML_OP_SIN: case GGML_OP_COS: case GGML_OP_CLAMP: case GGML_OP_LEAKY_RELU: return true;
            default: return false;
        }
    };
    auto plain = [](const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32 && t->data && is_contiguous(t) && ((uintptr_t) t->data & 15) == 0; };
    const ggml_tensor * first = g->nodes[i];
    if (!ew_kind(first) || !plain(first) || nelements(first) % 4 != 0 || nelements(first) < 4) return 0;
    const int64_t total = nelements(first);
    ew_chain_args a;
    a.total = total;
    const ggml_tensor * ext[6]; int n_ext = 0;
    const ggml_tensor * res[8]; int idx[8]; int n = 0;
    auto through_reshapes = [&](const ggml_tensor * t, const ggml_tensor * target, int consumer) -> bool {       // t is `target` seen through RESHAPEs read only by `consumer`
        for (; t != target; t = t->src[0]) {
            if (!t || t->op != GGML_OP_RESHAPE || is_out(s, t)) return false;
            auto it = s.users.find(t);
            if (it == s.users.end() || it->second.size() != 1 || it->second[0] != consumer) return false;
        }
        return true;
    };
    int j = i;
    while (n < 8) {
        const ggml_tensor * nd = g->nodes[j];
        if (!ew_kind(nd) || !plain(nd) || nelements(nd) != total) break;
        const bool binary = nd->op == GGML_OP_ADD || nd->op == GGML_OP_SUB || nd->op == GGML_OP_MUL || nd->op == GGML_OP_DIV;
        int sel[2] = { -1, -1 };
        const int n_ext0 = n_ext;
        bool ok = true, uses_prev = n == 0;
        for (int k = 0; k < (binary ? 2 : 1) && ok; ++k) {
            const ggml_tensor * o = nd->src[k];
            if (!o) { ok = false; break; }
            if (n > 0 && through_reshapes(o, res[n - 1], j)) { sel[k] = 8 + (n - 1); uses_prev = true; continue; }
            // an external operand
            if (o->type != GGML_TYPE_F32 || !o->data) { ok = false; break; }
            int mode;
            // one row per dim-2 slice (the DiT's shift / scale / gate: [C, 1, B] views of the adaLN product, repeated over the frames of batch element b)
            const bool row_per_slice = k == 1 && nd->ne[3] == 1 && o->ne[0] == nd->ne[0] && o->ne[1] == 1 && nd->ne[1] > 1 && o->ne[2] == nd->ne[2] && o->ne[2] > 1 && o->ne[3] == 1 &&
                                       o->nb[0] == 4 && o->nb[2] % 16 == 0 && o->ne[0] % 4 == 0 && o->nb[2] / 16 < (1ull << 32);
            if (row_per_slice) mode = 3;
            else if (!is_contiguous(o)) { ok = false; break; }
            else if (nelements(o) == total && (k == 0 || same_shape(o, nd))) mode = 0;
            else if (k == 1 && nelements(o) == 1) mode = 2;
            else if (k == 1 && o->ne[0] == nd->ne[0] && o->ne[1] * o->ne[2] * o->ne[3] == 1 && o->ne[0] % 4 == 0) mode = 1;
            else { ok = false; break; }
            if (mode != 2 && ((uintptr_t) o->data & 15) != 0) { ok = false; break; }
            int e = -1;
            for (int q = 0; q < n_ext; ++q) if (ext[q]->data == o->data && a.in_mode[q] == mode && ((mode != 1 && mode != 3) || a.in_n04[q] == (uint32_t) (o->ne[0] / 4)) && (mode != 3 || a.in_bs4[q] == (uint32_t) (o->nb[2] / 16))) e = q;
            if (e < 0) {
                if (n_ext >= 6) { ok = false; break; }
                e = n_ext++; ext[e] = o; a.in[e] = (const float *) o->data; a.in_mode[e] = mode; a.in_n04[e] = (mode == 1 || mode == 3) ? (uint32_t) (o->ne[0] / 4) : 1;
                a.in_per4[e] = mode == 3 ? (uint32_t) (nd->ne[0] * nd->ne[1] / 4) : 1; a.in_bs4[e] = mode == 3 ? (uint32_t) (o->nb[2] / 16) : 0;
            }
            sel[k] = e;
        }
        if (ok && binary && nd->src[0] && nelements(nd->src[0]) != total) ok = false;      // (ggml: the result has src0's shape)
        if (!ok || !uses_prev) { n_ext = n_ext0; break; }
        ew_op_desc & d = a.op[n];
        d.kind = (int) nd->op; d.sub = nd->op == GGML_OP_UNARY ? op_param_i32(nd, 0) : 0; d.a = sel[0]; d.b = binary ? sel[1] : sel[0];
        d.p0 = op_param_f32(nd, 0); d.p1 = op_param_f32(nd, 1);
        res[n] = nd; idx[n] = j; ++n;
        // may the chain go on?  the result must have exactly one reader, the next launching node -- and it must not be an in-place / view result: an intermediate of the
        // chain is never written, and a view's memory (ggml_add_inplace on a tensor somebody reads later, persistent state) has to change as the eager run changes it
        if (is_out(s, nd) || nd->view_src) break;
        const int u = sole_user(s, nd);
        const int nx = next_real_node(s, j);
        if (u < 0 || u != nx) break;
        j = nx;
    }
    if (n < 2) return 0;
    // trim: the last node's readers are free, but a chain must not end where a fused consumer expects to see the node itself (f16-emitting UNARY in front of a GEMM)
    const ggml_tensor * last = res[n - 1];
    const ggml_tensor * xg = nullptr;
    if (last->ne[2] == 1 && last->ne[3] == 1 && gemm_only_consumers(s, last, last->ne[0], last->ne[1], &xg)) return 0;
    // the result's buffer may sit on memory of the chain's dead inputs: identical position (mode 0) is fine, anything else is not
    const byte_range out = range_of(last);
    for (int q = 0; q < n_ext; ++q) {
        const byte_range r = range_of(ext[q]);
        if (overlap(out, r) && !(a.in_mode[q] == 0 && ext[q]->data == last->data)) return 0;
    }
    a.n_ops = n; a.n_in = n_ext; a.out = (float *) last->data;
    if (n_ext == 0) return 0;
    {
        prof_scope ps(s, "ew_chain", 0);
        ew_chain(a, s.st);
    }
    ++s.n_kernels; s.n_fused += n - 1;
    for (int k = 0; k < n; ++k) taken[k] = idx[k];
    note_write(s, last);
    return n;
}

// ------------------------------------------------------------------------------------------------ lazy copies of cache views
// fmCausalConv1d::build_forward_chunk_graph (token2wav-impl.cpp:952-957) makes two copies of the cached frames before it uses them: cache_in = CONT(view of the packed cache)
// and cache_tcb = CONT(PERMUTE(cache_in)) -- two ~2.5 us launches over 8 KB, 640 of them per window -- and exec_causal_conv then reads the C-fastest frames through a tensor
// descriptor anyway.  Both CONTs are therefore NOT run when they are met: the executor remembers what they would copy (a view of a tensor from outside the graph, intact until
// `deadline`), exec_causal_conv reads the view itself, exec_concat_tail never reads the frames, and ANY other reader -- or the deadline -- materialises the copy first (lazy_net).
static tdesc swapped01(tdesc d) { std::swap(d.ne[0], d.ne[1]); std::swap(d.nb[0], d.nb[1]); return d; }
static byte_range range_of(const tdesc & d) {
    size_t ext = 4;
    for (int k = 0; k < 4; ++k) ext += (size_t) (d.ne[k] > 0 ? d.ne[k] - 1 : 0) * d.nb[k];
    return { (const char *) d.p, (const char *) d.p + ext };
}
static void lazy_materialise(exec_state & s, const ggml_tensor * t, int reader_op) {
    auto it = s.lazy.find(t);
    if (it == s.lazy.end()) return;
    static const bool dbg = getenv("MI355X_SINK_DEBUG") != nullptr;          // who made a lazy copy real after all: reader op (-1: the deadline), tallied (stderr at process exit)
    static std::map<int, long> who;
    struct dump { ~dump() { if (dbg) { fprintf(stderr, "[mi355x] lazy_cont materialised by reader op:"); for (auto & kv : who) fprintf(stderr, " %d:%ld", kv.first, kv.second); fprintf(stderr, "\n"); } } };
    static dump at_exit;
    if (dbg) ++who[reader_op];
    ++s.c->stat_lazy_materialised;
    {
        prof_scope ps(s, "cpy", 0);
        cpy_strided(it->second.src, GGML_TYPE_F32, td(t), GGML_TYPE_F32, s.st); ++s.n_kernels;
    }
    s.lazy.erase(it);
    note_write(s, t);
}
// before node i runs outside the lazy-aware matchers: whatever it reads (through view chains) must exist, and nothing lazy may outlive its source
static void lazy_net(exec_state & s, int i) {
    if (s.lazy.empty()) return;
    const ggml_tensor * n = s.g->nodes[i];
    for (int k = 0; k < GGML_MAX_SRC; ++k)
        for (const ggml_tensor * t = n->src[k]; t; t = (t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE) ? t->src[0] : nullptr)
            if (s.lazy.count(t)) lazy_materialise(s, t, (int) n->op);
    for (auto it = s.lazy.begin(); it != s.lazy.end(); ) {
        if (it->second.deadline > i) { ++it; continue; }
        const ggml_tensor * t = it->first; ++it;
        bool needed = false;                                               // a copy whose readers have all run (or were folded away) is simply never made
        auto us = s.users.find(t);
        if (us != s.users.end()) for (int u : us->second) if (u >= i && !s.done[u]) needed = true;
        if (needed) lazy_materialise(s, t); else s.lazy.erase(t);
    }
}
static bool lazy_try_register(exec_state & s, int i) {
    static const bool off = getenv("MI355X_NO_LAZY_CACHE_CONT") != nullptr;
    if (off || !s.c->opt_fusion) return false;
    static const bool dbg = getenv("MI355X_SINK_DEBUG") != nullptr;          // which line turned a candidate down, tallied (stderr at process exit)
    static std::map<int, long> why;
    struct dump { ~dump() { if (dbg) { fprintf(stderr, "[mi355x] lazy_cont: taken %ld, refusals by source line:", why[0]); for (auto & kv : why) if (kv.first) fprintf(stderr, " %d:%ld", kv.first, kv.second); fprintf(stderr, "\n"); } } };
    static dump at_exit;
    auto no = [&](int line) { if (dbg) ++why[line]; return false; };
    ggml_cgraph * g = s.g;
    const ggml_tensor * n = g->nodes[i];
    if (n->op != GGML_OP_CONT || n->type != GGML_TYPE_F32 || !n->data || !is_contiguous(n) || n->view_src || is_out(s, n)) return false;
    const ggml_tensor * src = n->src[0];
    if (!src || src->type != GGML_TYPE_F32 || !src->data || !same_shape(src, n)) return false;
    auto conv_concat_user = [&](const ggml_tensor * t) -> bool {          // t's one reader is CONCAT(t, CONT(PERMUTE(x)), dim 0): the pattern exec_causal_conv takes
        const int u = sole_user(s, t);
        if (u <= i) return false;
        const ggml_tensor * c = g->nodes[u];
        return c->op == GGML_OP_CONCAT && op_param_i32(c, 0) == 0 && c->src[0] == t && c->src[1] && c->src[1]->op == GGML_OP_CONT && c->src[1]->src[0] && c->src[1]->src[0]->op == GGML_OP_PERMUTE;
    };
    exec_state::lazy_ent e;
    if (src->op == GGML_OP_VIEW) {                                         // cache_in = CONT(view of the packed cache)
        const ggml_tensor * base = src->view_src;
        if (n->ne[3] != 1 || !base || base->op != GGML_OP_NONE || !base->data || src->nb[0] != 4 || (n->flags & GGML_TENSOR_FLAG_OUTPUT)) return no(__LINE__);
        auto us = s.users.find(n);
        if (us == s.users.end()) return no(__LINE__);
        bool has_t = false;
        for (int u : us->second) {
            const ggml_tensor * c = g->nodes[u];
            if (c->op == GGML_OP_CONCAT) continue;                         // (the new-cache chain: exec_concat_tail, which does not read the frames, or the net)
            if (c->op != GGML_OP_CONT || !c->src[0] || c->src[0]->op != GGML_OP_PERMUTE || c->src[0]->src[0] != n || !conv_concat_user(c)) return no(__LINE__);
            has_t = true;
        }
        if (!has_t) return no(__LINE__);
        auto bd = s.lazy_base_deadline.find(base);
        if (bd == s.lazy_base_deadline.end()) {                            // first node that writes over the base's bytes (a CPY into the cache at the end of the graph; a re-used address)
            int dl = g->n_nodes;
            const byte_range rb = range_of(base);
            for (int k = i + 1; k < g->n_nodes; ++k) if (!is_noop(g->nodes[k]) && g->nodes[k]->data && overlap(range_of(g->nodes[k]), rb)) { dl = k; break; }
            bd = s.lazy_base_deadline.emplace(base, dl).first;
        }
        if (bd->second <= i + 1) return no(__LINE__);
        e.src = td(src); e.deadline = bd->second;
    } else if (src->op == GGML_OP_PERMUTE && src->src[0]) {
        static const bool off_c = getenv("MI355X_NO_LAZY_ATTN_CONT") != nullptr;
        // is `M` (node index u) a batched f32 x f32 MUL_MAT reading `w` through reshapes only?
        auto f32_product_of = [&](int u, const ggml_tensor * w, bool second_is_softmax) -> bool {
            const ggml_tensor * M = g->nodes[u];
            if (M->op != GGML_OP_MUL_MAT || M->type != GGML_TYPE_F32 || !M->src[0] || !M->src[1] || M->src[0]->type != GGML_TYPE_F32 || M->src[1]->type != GGML_TYPE_F32 || M->src[0]->ne[3] != 1 || M->src[1]->ne[3] != 1) return false;
            if (second_is_softmax && M->src[1]->op != GGML_OP_SOFT_MAX) return false;
            for (int k = 0; k < (second_is_softmax ? 1 : 2); ++k) { const ggml_tensor * r = M->src[k]; while (r && r->op == GGML_OP_RESHAPE) r = r->src[0]; if (r == w) return true; }
            return false;
        };
        // V^T = CONT(PERMUTE(RESHAPE(c))) with c a CONT [D, n, H, B] and the PERMUTE swapping the first two dims of its [D, n, H B] reshape: returns c
        auto transposed_flat = [&](const ggml_tensor * v) -> const ggml_tensor * {
            if (v->op != GGML_OP_CONT || !v->src[0] || v->src[0]->op != GGML_OP_PERMUTE) return nullptr;
            const ggml_tensor * pm = v->src[0], * r = pm->src[0], * c = r;
            while (c && c->op == GGML_OP_RESHAPE) c = c->src[0];
            if (!r || !c || c->op != GGML_OP_CONT || !is_contiguous(r) || r->data != c->data || pm->data != r->data) return nullptr;
            if (r->ne[0] != c->ne[0] || r->ne[1] != c->ne[1] || r->ne[2] != c->ne[2] * c->ne[3] || r->ne[3] != 1) return nullptr;
            if (pm->ne[0] != r->ne[1] || pm->ne[1] != r->ne[0] || pm->ne[2] != r->ne[2] || pm->ne[3] != 1 || pm->nb[0] != r->nb[1] || pm->nb[1] != r->nb[0] || pm->nb[2] != r->nb[2]) return nullptr;
            return c;
        };
        const ggml_tensor * t = src->src[0];
        const ggml_tensor * cflat = transposed_flat(n);
        auto lc = cflat ? s.lazy.find(cflat) : s.lazy.end();
        auto lt = s.lazy.find(t);
        const int u_n = sole_user(s, n);
        if (lc != s.lazy.end() && !off_c && u_n > i && f32_product_of(u_n, n, true)) {
            // case C', second half: V^T of a flattened V that is itself lazy -- the f32 attention chain's second product reads V where it lies, keys a row apart
            const int u = u_n;
            if (u - i > 96) return no(__LINE__);
            const byte_range rt = range_of(lc->second.src);
            for (int k = i + 1; k <= u; ++k) if (!is_noop(g->nodes[k]) && g->nodes[k]->data && overlap(range_of(g->nodes[k]), rt)) return no(__LINE__);
            e.src = swapped01(lc->second.src); e.deadline = u + 1;
        } else if (lt != s.lazy.end()) {                                   // cache_tcb = CONT(PERMUTE(cache_in)), cache_in still lazy
            const ggml_tensor * q = t;
            if (src->data != q->data || src->ne[0] != q->ne[1] || src->ne[1] != q->ne[0] || src->ne[2] != q->ne[2] || src->nb[0] != q->nb[1] || src->nb[1] != q->nb[0] || src->nb[2] != q->nb[2]) return no(__LINE__);
            if (!conv_concat_user(n)) return no(__LINE__);
            e.src = swapped01(lt->second.src); e.deadline = lt->second.deadline;
        } else {
            // case C: the heads of Q / K / V flattened for the f32 attention chain, CONT(PERMUTE([D, H, n, B] -> [D, n, H, B])), read by one batched MUL_MAT (through reshapes)
            // -- attn_f32 takes the permuted view itself -- or, for V, by the transposing copy above.  The source is a tensor of this graph: lazy only while nothing up to
            // that reader writes over it.
            if (off_c || n->ne[3] < 1 || src->nb[0] != 4 || !t->data || src->data != t->data || !is_contiguous(t) || src->ne[0] != t->ne[0] || src->ne[1] != t->ne[2] || src->ne[2] != t->ne[1] || src->ne[3] != t->ne[3]) return no(__LINE__);
            const int u = sole_user(s, n);
            if (u <= i || u - i > 64) return no(__LINE__);
            if (!f32_product_of(u, n, false)) {
                const ggml_tensor * U = g->nodes[u];
                // ... or by a CONCAT that takes it directly (the new K / V cache rows: CONCAT(CONT(PERMUTE(k)), CONT(PERMUTE(v)), 0), token2wav-impl.cpp:340-347): the generic
                // concat kernel reads both operands through their strides (compute_node, CONCAT)
                static const bool off_cc = getenv("MI355X_NO_LAZY_CONCAT_SRC") != nullptr;
                const bool concat_reader = !off_cc && U->op == GGML_OP_CONCAT && U->type == GGML_TYPE_F32 && (U->src[0] == n || U->src[1] == n) && U->src[0]->type == GGML_TYPE_F32 && U->src[1]->type == GGML_TYPE_F32;
                if (!concat_reader) {
                    const int u2 = transposed_flat(U) == n ? sole_user(s, U) : -1;
                    if (u2 <= u || u2 - u > 96 || !f32_product_of(u2, U, true)) return no(__LINE__);
                }
            }
            const byte_range rt = range_of(t);
            for (int k = i + 1; k <= u; ++k) if (!is_noop(g->nodes[k]) && g->nodes[k]->data && overlap(range_of(g->nodes[k]), rt)) return no(__LINE__);
            e.src = td(src); e.deadline = u + 1;
        }
    } else return no(__LINE__);
    if (dbg) ++why[0];
    ++s.c->stat_lazy_taken;
    s.lazy[n] = e;
    s.done[i] = 1; ++s.n_fused;
    return true;
}

// Token2Wav's streaming causal 1-D convolution the way the reference's builder spells it (token2wav-impl.cpp: the cached P = KW - 1 frames ++ x on the time axis of the
// transposed [T, C, B] copies, then per batch element VIEW -> IM2COL -> MUL_MAT against the [KW*C, Cout] kernel, CONCAT of the batch elements, PERMUTE + CONT back to
// [Cout, T, B], ADD of the bias): 11 launches, five of them transposes or copies.  With x and the cache in their C-fastest layouts the im2col column of frame t is the
// KW*C consecutive floats from frame t of (cache ++ x) -- so: ONE dense concat into the pattern's own [T+P, C, B] buffer (as [C, T+P, B]) and ONE any-shape GEMM whose
// activation rows overlap (row stride C floats, row length KW*C) against the kernel re-laid once to [Cout][KW][C] (a resident image next to the F16 weight images),
// the bias in its epilogue, both batch elements in the launch.  The sums are the reference's with the KW*C products in (k, c) instead of (c, k) order.
// `i` is the CONT of the transposed x.  Returns true when the pattern was taken (its nodes are marked done).
static bool exec_causal_conv(exec_state & s, int i) {
    static const bool off = getenv("MI355X_NO_CONV_FUSE") != nullptr;
    if (off || !s.c->opt_fusion) return false;
    ggml_cgraph * g = s.g;
    static const bool dbg = getenv("MI355X_SINK_DEBUG") != nullptr;          // which line turned a CONT(PERMUTE(x)) candidate down, tallied (stderr at process exit)
    static std::map<int, long> why;
    struct dump { ~dump() { if (dbg) { fprintf(stderr, "[mi355x] causal_conv: refusals by source line:"); for (auto & kv : why) fprintf(stderr, " %d:%ld", kv.first, kv.second); fprintf(stderr, "\n"); } } };
    static dump at_exit;
    auto no = [&](int line) { if (dbg) ++why[line]; return false; };
    auto plain = [](const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32 && t->data && is_contiguous(t) && t->ne[3] == 1; };
    // t = CONT(PERMUTE(q)) with q a plain [C, n, B] tensor and t its [n, C, B] transpose: returns q
    auto untransposed = [&](const ggml_tensor * t) -> const ggml_tensor * {
        if (!plain(t) || t->op != GGML_OP_CONT || t->view_src) return nullptr;
        const ggml_tensor * p = t->src[0];
        if (!p || p->op != GGML_OP_PERMUTE) return nullptr;
        const ggml_tensor * q = p->src[0];
        if (!plain(q) || p->ne[0] != q->ne[1] || p->ne[1] != q->ne[0] || p->ne[2] != q->ne[2] || p->nb[0] != q->nb[1] || p->nb[1] != q->nb[0] || p->nb[2] != q->nb[2] || p->data != q->data) return nullptr;
        return q;
    };
    auto through_reshapes = [&](const ggml_tensor * t, const ggml_tensor * target) -> bool {
        for (; t != target; t = t->src[0]) if (!t || t->op != GGML_OP_RESHAPE || is_out(s, t)) return false;
        return true;
    };
    const ggml_tensor * n1 = g->nodes[i];
    if (n1->op != GGML_OP_CONT) return false;
    const ggml_tensor * x = untransposed(n1);
    if (!x) return false;
    const int64_t C = x->ne[0], T = x->ne[1], B = x->ne[2];
    if (B < 1 || B > 2 || T < 1 || C % 4 != 0) return no(__LINE__);
    const int j2 = sole_user(s, n1);
    if (j2 <= i) return no(__LINE__);
    const ggml_tensor * n2 = g->nodes[j2];
    if (n2->op != GGML_OP_CONCAT || op_param_i32(n2, 0) != 0 || n2->src[1] != n1 || !plain(n2)) return no(__LINE__);
    const ggml_tensor * cacheT = n2->src[0];
    const ggml_tensor * cc = untransposed(cacheT);
    if (!cc || cc->ne[0] != C || cc->ne[2] != B) return no(__LINE__);
    const int64_t P = cc->ne[1], KW = P + 1;
    // the two copies of the cached frames may not have been run (lazy_try_register): then the frames are read where they lie, in the cache
    const auto lzT = s.lazy.find(cacheT);
    const bool frames_lazy = lzT != s.lazy.end();
    const tdesc frames_src = frames_lazy ? swapped01(lzT->second.src) : tdesc();
    // the cache frames are computed before x's copy; their C-fastest original is dead for ggml-alloc once the transposed copy exists, so it is only read when nothing
    // between that copy and here wrote over it -- otherwise the transposed copy is read through swapped strides
    bool cc_intact = !s.lazy.count(cc);                                    // (a cache_in that was never written, behind a cache_tcb that was: read the latter)
    if (frames_lazy) cc_intact = true;
    else {
        auto it = s.index.find(cacheT);
        if (it == s.index.end() || it->second >= i) return no(__LINE__);
        if (i - it->second > 64) cc_intact = false;
        for (int k = it->second + 1; k < i && cc_intact; ++k) if (!is_noop(g->nodes[k]) && g->nodes[k]->data && overlap(range_of(g->nodes[k]), range_of(cc))) cc_intact = false;
    }
    const int j3 = sole_user(s, n2);
    if (j3 <= j2) return no(__LINE__);
    const ggml_tensor * n3 = g->nodes[j3];
    if (n3->op != GGML_OP_CONT || n3->src[0] != n2 || !plain(n3) || n3->view_src || n3->ne[0] != T + P || n3->ne[1] != C || n3->ne[2] != B || is_out(s, n3)) return no(__LINE__);
    auto u3 = s.users.find(n3);
    if (u3 == s.users.end() || (int64_t) u3->second.size() != B) return no(__LINE__);
    int im[2] = { -1, -1 }, mm[2] = { -1, -1 };
    const ggml_tensor * Wk = nullptr;
    for (int q = 0; q < (int) B; ++q) {
        const int ji = u3->second[q];
        const ggml_tensor * ic = g->nodes[ji];
        if (ic->op != GGML_OP_IM2COL || !plain(ic) || ic->ne[0] != KW * C || ic->ne[1] != T || ic->ne[2] != 1) return no(__LINE__);
        const int32_t * ip = ic->op_params;
        if (ip[0] != 1 || ip[2] != 0 || ip[4] != 1 || ip[6] != 0) return no(__LINE__);                          // stride 1, no padding, dilation 1, 1-D
        const ggml_tensor * v = ic->src[1];
        if (!v || v->view_src != n3 || v->type != GGML_TYPE_F32 || v->ne[0] != T + P || v->ne[1] != C || v->ne[2] != 1 || v->ne[3] != 1 || v->nb[1] != n3->nb[1]) return no(__LINE__);
        const size_t off_b = (size_t) ((const char *) v->data - (const char *) n3->data);
        if (off_b % n3->nb[2] != 0) return no(__LINE__);
        const int b = (int) (off_b / n3->nb[2]);
        if (b < 0 || b >= B || im[b] >= 0) return no(__LINE__);
        const ggml_tensor * k = ic->src[0];
        if (!plain(k) || k->ne[0] != KW || k->ne[1] != C || k->op != GGML_OP_NONE || k->view_src || (Wk && k != Wk)) return no(__LINE__);
        Wk = k; im[b] = ji;
        const int jm = sole_user(s, ic);
        if (jm <= ji) return no(__LINE__);
        const ggml_tensor * m = g->nodes[jm];
        if (m->op != GGML_OP_MUL_MAT || !plain(m) || m->ne[0] != T || m->ne[1] != Wk->ne[2] || m->ne[2] != 1 || !through_reshapes(m->src[0], ic)) return no(__LINE__);
        const ggml_tensor * kr = m->src[1];
        if (!kr || kr->ne[0] != KW * C || kr->ne[1] != Wk->ne[2] || kr->ne[2] != 1 || kr->data != Wk->data || !is_contiguous(kr) || kr->type != GGML_TYPE_F32) return no(__LINE__);
        mm[b] = jm;
    }
    const int64_t Cout = Wk->ne[2];
    if (Wk->ne[3] != 1 || !Wk->buffer || Wk->buffer->usage != GGML_BACKEND_BUFFER_USAGE_WEIGHTS) return no(__LINE__);
    int j4 = -1;
    const ggml_tensor * n4 = g->nodes[mm[0]];                                                             // [T, Cout, B], T fastest
    if (B == 2) {
        j4 = sole_user(s, g->nodes[mm[0]]);
        if (j4 < 0 || j4 != sole_user(s, g->nodes[mm[1]]) || j4 <= mm[0] || j4 <= mm[1]) return no(__LINE__);
        n4 = g->nodes[j4];
        if (n4->op != GGML_OP_CONCAT || op_param_i32(n4, 0) != 2 || !plain(n4) || n4->ne[0] != T || n4->ne[1] != Cout || n4->ne[2] != 2 ||
            !through_reshapes(n4->src[0], g->nodes[mm[0]]) || !through_reshapes(n4->src[1], g->nodes[mm[1]])) return no(__LINE__);
    }
    const int j6 = sole_user(s, n4);
    if (j6 < 0 || j6 <= (B == 2 ? j4 : mm[0])) return no(__LINE__);
    const ggml_tensor * n6 = g->nodes[j6];
    if (n6->op != GGML_OP_CONT || !plain(n6) || n6->view_src || n6->ne[0] != Cout || n6->ne[1] != T || n6->ne[2] != B) return no(__LINE__);
    {
        const ggml_tensor * p = n6->src[0];
        if (!p || p->op != GGML_OP_PERMUTE || p->ne[0] != Cout || p->ne[1] != T || p->ne[2] != B || p->nb[0] != (size_t) T * 4 || p->nb[1] != 4 ||
            (B == 2 && p->nb[2] != (size_t) T * (size_t) Cout * 4) || p->data != n4->data || !through_reshapes(p->src[0], n4)) return no(__LINE__);
    }
    const ggml_tensor * out = n6; const float * bias = nullptr; int j7 = -1;
    if (!is_out(s, n6)) {
        const int ja = sole_user(s, n6);
        if (ja > j6 && next_real_node(s, j6) == ja) {
            const ggml_tensor * ad = g->nodes[ja];
            const ggml_tensor * bv = ad->src[1];
            if (ad->op == GGML_OP_ADD && ad->src[0] == n6 && plain(ad) && same_shape(ad, n6) && bv && bv->type == GGML_TYPE_F32 && bv->data && is_contiguous(bv) && bv->ne[0] == Cout && nelements(bv) == Cout) {
                out = ad; bias = (const float *) bv->data; j7 = ja;
            }
        }
    }
    const int last = j7 >= 0 ? j7 : j6;
    auto mine = [&](int k) { return k == i || k == j2 || k == j3 || k == im[0] || k == im[1] || k == mm[0] || k == mm[1] || k == j4 || k == j6 || k == j7; };
    for (int k = i + 1; k < last; ++k) if (!mine(k) && !s.done[k] && !is_noop(g->nodes[k])) return no(__LINE__);   // nothing else runs inside the pattern
    for (int k : { j2, j3, im[0], im[1], mm[0], mm[1], j4, j6 }) if (k >= 0 && k != last && is_out(s, g->nodes[k])) return no(__LINE__);
    if (is_out(s, n1)) return no(__LINE__);
    if (((uintptr_t) x->data & 15) || ((uintptr_t) (frames_lazy ? frames_src.p : cc->data) & 15) || ((uintptr_t) out->data & 15)) return no(__LINE__);
    // ggml-alloc may have put the pattern's buffers over memory that is free by the time their own node runs; here they are written at x's copy
    // (the concatenated frames go to the CONT's buffer, or to the CONCAT's -- same size, both dead outside the pattern -- when the first sits on an input or under the result)
    const ggml_tensor * xbuf = nullptr;
    for (const ggml_tensor * cand : { n3, n2 })
        if (!xbuf && !((uintptr_t) cand->data & 15) && !overlap(range_of(cand), range_of(x)) && !overlap(range_of(cand), frames_lazy ? range_of(frames_src) : range_of(cc_intact ? cc : cacheT)) && !overlap(range_of(out), range_of(cand))) xbuf = cand;
    if (!xbuf) return no(__LINE__);
    // the kernel rows [Cout][KW][C]: built on first use outside capture, kept with the weight images (dropped with them when the source bytes are written)
    bool created = false;
    float * wrows = (float *) shadow_get_or_create(s.c->device, Wk->data, nbytes(Wk), /*type: conv rows*/ 1000 + (int) KW, 2 * KW * C, Cout, (size_t) KW * 4, s.st, s.capturing, &created);
    if (!wrows) return no(__LINE__);
    if (s.pr.A) materialise_reduce(s);
    if (s.prm.n) materialise_group(s);
    if (s.pn.m && (s.pn.m == x || s.pn.m == cc)) materialise_norm(s);
    if (created) {
        prof_scope ps(s, "conv_weight_rows", 0);
        conv1d_weight_rows((const float *) Wk->data, wrows, (int) KW, (int) C, (int) Cout, s.st); ++s.n_kernels;
        shadow_mark_ready((uint16_t *) wrows, s.st);
    }
    {
        prof_scope ps(s, "concat", 0);
        tdesc y; y.p = xbuf->data; y.ne[0] = C; y.ne[1] = T + P; y.ne[2] = B; y.ne[3] = 1; y.nb[0] = 4; y.nb[1] = (size_t) C * 4; y.nb[2] = (size_t) C * (size_t) (T + P) * 4; y.nb[3] = y.nb[2] * (size_t) B;
        tdesc ca = frames_lazy ? frames_src : td(cc);
        if (!frames_lazy && !cc_intact) { ca.p = cacheT->data; ca.nb[0] = cacheT->nb[1]; ca.nb[1] = cacheT->nb[0]; ca.nb[2] = cacheT->nb[2]; ca.nb[3] = cacheT->nb[3]; }
        concat(ca, td(x), y, 1, 4, s.st); ++s.n_kernels;
    }
    note_write(s, xbuf);
    {
        gemm_any_args a;
        a.W = wrows; a.w_rs = (size_t) KW * C * 4; a.w_f16 = false;
        a.X = xbuf->data; a.x_rs = (size_t) C * 4; a.x_nb2 = (size_t) C * (size_t) (T + P) * 4;
        a.dst = (float *) out->data; a.dst_cs = out->nb[1]; a.dst_nb2 = out->nb[2]; a.bias = bias;
        a.M = Cout; a.N = T; a.K = KW * C; a.nbatch = (int) B; a.ne12 = (int) B; a.r2 = (int) B; a.r3 = 1;
        if (s.c->gemm_partial && s.c->fa_counters) { a.partial = (float *) s.c->gemm_partial; a.partial_bytes = s.c->gemm_partial_bytes; a.counters = s.c->fa_counters; a.n_counters = 1024; }
        prof_scope ps(s, "gemm_any_f32", 2.0 * (double) Cout * (double) T * (double) (KW * C) * (double) B);
        gemm_any(a, s.st); ++s.n_kernels;
    }
    note_write(s, out);
    for (int k : { j2, j3, im[0], im[1], mm[0], mm[1], j4, j6, j7 }) if (k >= 0) { s.done[k] = 1; ++s.n_fused; }
    if (frames_lazy) s.lazy.erase(cacheT);                                 // its one reader is done: the copy is never made
    if (dbg) ++why[0];
    return true;
}

// The new cache of a streaming causal convolution (fmCausalConv1d::build_forward_chunk_graph, token2wav-impl.cpp:977-994): CONT(x) -> CONCAT(cache, x) on the frame
// axis -> CONT -> CONT(VIEW of the last K - 1 frames) -- four launches over [C, dt + K - 1, B] to keep K - 1 frames, 320 times per window.  When the kept frames all
// come from x (dt >= K - 1) they are copied from x and the three other nodes are not run.  `i` is the CONT of x, or the CONCAT when x goes in as it is.
static bool exec_concat_tail(exec_state & s, int i) {
    static const bool off = getenv("MI355X_NO_CONCAT_TAIL") != nullptr;
    if (off || !s.c->opt_fusion) return false;
    ggml_cgraph * g = s.g;
    auto plain = [](const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32 && t->data && is_contiguous(t) && t->ne[3] == 1; };
    const ggml_tensor * n = g->nodes[i];
    const ggml_tensor * n0 = nullptr, * x = nullptr; int j1 = i;
    if (n->op == GGML_OP_CONT) {
        n0 = n; x = n->src[0];
        if (!plain(n0) || n0->view_src || !plain(x) || !same_shape(x, n0) || is_out(s, n0)) return false;
        j1 = sole_user(s, n0);
        if (j1 <= i || next_real_node(s, i) != j1) return false;
    } else if (n->op != GGML_OP_CONCAT) return false;
    const ggml_tensor * n1 = g->nodes[j1];
    if (n1->op != GGML_OP_CONCAT || op_param_i32(n1, 0) != 1 || !plain(n1) || is_out(s, n1)) return false;
    if (n0) { if (n1->src[1] != n0) return false; } else { x = n1->src[1]; if (!plain(x)) return false; }
    const ggml_tensor * cache = n1->src[0];
    if (!cache || cache->ne[0] != x->ne[0] || cache->ne[2] != x->ne[2] || cache->ne[3] != 1) return false;
    const int64_t P = cache->ne[1], dt = x->ne[1];
    const int j2 = sole_user(s, n1);
    if (j2 <= j1 || next_real_node(s, j1) != j2) return false;
    const ggml_tensor * n2 = g->nodes[j2];
    if (n2->op != GGML_OP_CONT || n2->src[0] != n1 || !plain(n2) || n2->view_src || !same_shape(n2, n1) || is_out(s, n2)) return false;
    const int j3 = sole_user(s, n2);
    if (j3 <= j2 || next_real_node(s, j2) != j3) return false;
    const ggml_tensor * n3 = g->nodes[j3];
    const ggml_tensor * v = n3->src[0];
    if (n3->op != GGML_OP_CONT || !plain(n3) || n3->view_src || !v || v->op != GGML_OP_VIEW || v->view_src != n2 || v->type != GGML_TYPE_F32 || is_out(s, v)) return false;
    if (v->ne[0] != n2->ne[0] || v->ne[2] != n2->ne[2] || v->ne[3] != 1 || v->nb[0] != 4 || v->nb[1] != n2->nb[1] || v->nb[2] != n2->nb[2] || !same_shape(n3, v)) return false;
    const size_t off_b = (size_t) ((const char *) v->data - (const char *) n2->data);
    if (off_b % n2->nb[1] != 0) return false;
    const int64_t f0 = (int64_t) (off_b / n2->nb[1]), keep = v->ne[1];
    if (f0 < P || f0 + keep > P + dt) return false;                                 // (kept frames that reach into the old cache: the nodes run as they are)
    if (overlap(range_of(n3), range_of(x))) return false;                           // the copy's buffer was placed for a point of the graph where x may be dead
    // x itself (or what it is a view of) may be a copy that was left un-run (lazy_try_register accepts a CONCAT reader in either operand position): this matcher reads
    // x->data directly and runs BEFORE lazy_net -- make such a copy real first
    for (const ggml_tensor * t = x; t; t = (t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE) ? t->src[0] : nullptr)
        if (s.lazy.count(t)) lazy_materialise(s, t, (int) GGML_OP_CONCAT);
    if (s.pr.A) materialise_reduce(s);
    if (s.prm.n) materialise_group(s);
    if (s.pn.m && s.pn.m == x) materialise_norm(s);
    {
        prof_scope ps(s, "cpy", 0);
        tdesc src = td(x);
        src.p = (char *) x->data + (size_t) (f0 - P) * x->nb[1]; src.ne[1] = keep;
        cpy_strided(src, GGML_TYPE_F32, td(n3), GGML_TYPE_F32, s.st); ++s.n_kernels;
    }
    note_write(s, n3);
    for (int k : { n0 ? j1 : -1, j2, j3 }) if (k >= 0) { s.done[k] = 1; ++s.n_fused; }
    return true;
}

// The HiFT vocoder's 1-D convolutions over a T-fastest signal (token2wav-impl.cpp:5136-5235): IM2COL(F32) -> [CONT] -> MUL_MAT against the reshaped kernel ->
// REPEAT(bias) -> ADD, five launches around a [KW*Cin, T] matrix of tens of megabytes.  One conv1d_tc launch (t2w_ops.hip) reads x itself; the kernel transposed once to
// [KW*Cin][Cout] is a resident image.  `i` is the IM2COL; the nodes must be the launches right behind one another.
static bool exec_conv1d_tc(exec_state & s, int i) {
    static const bool off = getenv("MI355X_NO_CONV1D_TC") != nullptr;
    if (off || !s.c->opt_fusion) return false;
    ggml_cgraph * g = s.g;
    const ggml_tensor * n = g->nodes[i];
    auto plain = [](const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32 && t->data && is_contiguous(t); };
    if (n->op != GGML_OP_IM2COL || !plain(n) || is_out(s, n) || n->ne[2] != 1 || n->ne[3] != 1) return false;
    const ggml_tensor * Wk = n->src[0], * x = n->src[1];
    const int32_t * ip = n->op_params;
    if (!plain(Wk) || !plain(x) || ip[0] != 1 || ip[6] != 0 || ip[4] < 1 || ip[2] < 0) return false;
    const int64_t KW = Wk->ne[0], Cin = Wk->ne[1], Cout = Wk->ne[2], T = x->ne[0], OW = n->ne[1];
    if (Wk->ne[3] != 1 || x->ne[1] != Cin || x->ne[2] != 1 || x->ne[3] != 1 || n->ne[0] != KW * Cin || Wk->op != GGML_OP_NONE || Wk->view_src || !Wk->buffer || Wk->buffer->usage != GGML_BACKEND_BUFFER_USAGE_WEIGHTS) return false;
    if (T * Cin >= (1ll << 31) || KW * Cin * Cout >= (1ll << 31) || OW * Cout >= (1ll << 31)) return false;
    // the launches behind the IM2COL, in order; the reference puts a CONT behind nearly every reshape (of the columns, of the KERNEL, of the product, of the bias)
    auto root_of = [](const ggml_tensor * t) { while (t && t->op == GGML_OP_RESHAPE) t = t->src[0]; return t; };
    int taken[8]; int nt = 0;
    int j = next_real_node(s, i);
    const ggml_tensor * col = n, * wsrc = Wk;
    if (j > i && g->nodes[j]->op == GGML_OP_CONT && root_of(g->nodes[j]->src[0]) == n && sole_user(s, n) == j) {
        const ggml_tensor * c1 = g->nodes[j];
        if (!plain(c1) || nelements(c1) != nelements(n) || is_out(s, c1) || c1->view_src) return false;
        taken[nt++] = j; col = c1; j = next_real_node(s, j);
    }
    if (j > i && g->nodes[j]->op == GGML_OP_CONT && root_of(g->nodes[j]->src[0]) == Wk) {                 // a copy of the (reshaped) kernel: not needed
        const ggml_tensor * c2 = g->nodes[j];
        if (!plain(c2) || nelements(c2) != nelements(Wk) || is_out(s, c2) || c2->view_src || !is_contiguous(c2->src[0])) return false;
        taken[nt++] = j; wsrc = c2; j = next_real_node(s, j);
    }
    if (j <= i) return false;
    const int mi_ = j;
    const ggml_tensor * m = g->nodes[mi_];
    if (m->op != GGML_OP_MUL_MAT || !plain(m) || m->ne[0] != OW || m->ne[1] != Cout || m->ne[2] != 1 || m->ne[3] != 1 || root_of(m->src[0]) != col || sole_user(s, col) != mi_) return false;
    const ggml_tensor * kr = m->src[1];
    if (!kr || kr->type != GGML_TYPE_F32 || root_of(kr) != wsrc || kr->ne[0] != KW * Cin || kr->ne[1] != Cout || kr->ne[2] != 1 || !is_contiguous(kr) || (wsrc != Wk && sole_user(s, wsrc) != mi_)) return false;
    taken[nt++] = mi_;
    // how far the launch reaches: up to the ADD of the bias (preferred), the product's CONT, or the product itself -- the first of them whose buffer does not sit on x
    // (ggml-alloc placed those buffers for later points of the graph, where x may be dead; it is not dead here)
    const int nt_m = nt;                                              // taken[0 .. nt_m): up to and including the MUL_MAT
    const ggml_tensor * out = m; const float * bias = nullptr;
    const ggml_tensor * y = m; int n_y = 0, yq = -1;
    const ggml_tensor * out_add = nullptr; const float * bias_add = nullptr; int add_taken[3], n_add = 0;
    if (!is_out(s, m)) {
        int q = next_real_node(s, mi_);
        if (q > mi_ && g->nodes[q]->op == GGML_OP_CONT && root_of(g->nodes[q]->src[0]) == m && sole_user(s, m) == q && plain(g->nodes[q]) && nelements(g->nodes[q]) == nelements(m) && !g->nodes[q]->view_src) {
            y = g->nodes[q]; yq = q; n_y = 1; q = next_real_node(s, q);
        }
        // the bias: [CONT of] a [1, Cout] reshape of a vector -> REPEAT to the product's shape -> ADD
        const ggml_tensor * bvec = nullptr, * bcont = nullptr; int bq = -1;
        if (q > mi_ && g->nodes[q]->op == GGML_OP_CONT && !is_out(s, g->nodes[q]) && plain(g->nodes[q]) && nelements(g->nodes[q]) == Cout) {
            const ggml_tensor * r0 = root_of(g->nodes[q]->src[0]);
            if (r0 && r0->type == GGML_TYPE_F32 && r0->data && is_contiguous(r0) && nelements(r0) == Cout && is_contiguous(g->nodes[q]->src[0])) { bcont = g->nodes[q]; bvec = r0; bq = q; q = next_real_node(s, q); }
        }
        if (q > mi_ && g->nodes[q]->op == GGML_OP_REPEAT && !is_out(s, y)) {
            const ggml_tensor * r = g->nodes[q], * rs = root_of(r->src[0]);
            if (!bcont && rs && rs->type == GGML_TYPE_F32 && rs->data && is_contiguous(rs) && nelements(rs) == Cout && is_contiguous(r->src[0])) bvec = rs;
            const bool src_ok = bcont ? (rs == bcont && sole_user(s, bcont) == q) : (bvec != nullptr);
            const int a0 = sole_user(s, r);
            if (src_ok && bvec && a0 > q && next_real_node(s, q) == a0 && sole_user(s, y) == a0 && plain(r) && !is_out(s, r) && r->ne[0] == OW && r->ne[1] == Cout && nelements(r) == OW * Cout) {
                const ggml_tensor * ad = g->nodes[a0];
                if (ad->op == GGML_OP_ADD && ad->src[1] == r && plain(ad) && nelements(ad) == OW * Cout && ad->ne[0] == OW && root_of(ad->src[0]) == y) {
                    out_add = ad; bias_add = (const float *) bvec->data;
                    if (bq >= 0) add_taken[n_add++] = bq;
                    add_taken[n_add++] = q; add_taken[n_add++] = a0;
                }
            }
        }
    }
    if (out_add && !overlap(range_of(out_add), range_of(x))) {
        out = out_add; bias = bias_add;
        if (n_y) taken[nt++] = yq;
        for (int t = 0; t < n_add; ++t) taken[nt++] = add_taken[t];
    } else if (n_y && !overlap(range_of(y), range_of(x))) { out = y; taken[nt++] = yq; }
    else if (!overlap(range_of(m), range_of(x))) { out = m; nt = nt_m; }
    else return false;
    bool created = false;
    float * wt = (float *) shadow_get_or_create(s.c->device, Wk->data, nbytes(Wk), /*type: transposed conv kernel*/ 2000, 2 * KW * Cin, Cout, (size_t) KW * 4 + 1, s.st, s.capturing, &created);
    if (!wt) return false;
    if (s.pr.A) materialise_reduce(s);
    if (s.prm.n) materialise_group(s);
    if (s.pn.m && s.pn.m == x) materialise_norm(s);
    if (created) {
        prof_scope ps(s, "conv_weight_rows", 0);
        conv1d_weight_t((const float *) Wk->data, wt, (int) (KW * Cin), (int) Cout, s.st); ++s.n_kernels;
        shadow_mark_ready((uint16_t *) wt, s.st);
    }
    {
        prof_scope ps(s, "conv1d_tc", 2.0 * (double) KW * (double) Cin * (double) Cout * (double) OW);
        conv1d_tc((const float *) x->data, wt, bias, (float *) out->data, (int) T, (int) OW, (int) Cin, (int) Cout, (int) KW, ip[4], ip[2], s.st); ++s.n_kernels;
    }
    for (int t = 0; t < nt; ++t) { s.done[taken[t]] = 1; ++s.n_fused; }
    note_write(s, out);
    return true;
}

void run_nodes(exec_state & s, ggml_cgraph * g) {
    static FILE * const launch_log = getenv("MI355X_LAUNCH_LOG") ? fopen(getenv("MI355X_LAUNCH_LOG"), "w") : nullptr;      // one line per node that launched: what a graph's launches are made of (tools/launch_ngrams.py)
    s.g = g;
    s.done.assign(g->n_nodes, 0);
    s.index.clear(); s.users.clear(); s.lazy.clear(); s.lazy_base_deadline.clear(); s.gs = {};
    if (s.c->opt_fusion) {
        s.index.reserve(g->n_nodes * 2); s.users.reserve(g->n_nodes * 2);
        for (int i = 0; i < g->n_nodes; ++i) {
            s.index[g->nodes[i]] = i;
            if (is_noop(g->nodes[i])) continue;
            for (int k = 0; k < GGML_MAX_SRC; ++k) {
                // a consumer of a view counts as a consumer of every tensor on the view chain it reads through
                const ggml_tensor * t = g->nodes[i]->src[k];
                while (t) {
                    auto & v = s.users[t];
                    if (v.empty() || v.back() != i) v.push_back(i);
                    t = (t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE) ? t->src[0] : nullptr;
                }
            }
        }
    }
    s.external.clear();
    if (s.c->opt_fusion && g->use_counts && g->visited_hash_set.size > 0 && g->visited_hash_set.keys && g->visited_hash_set.used) {
        // direct uses inside this cgraph, counted like ggml_build_forward counts them (every src of every node, view nodes included)
        std::unordered_map<const ggml_tensor *, int> direct;
        direct.reserve(g->n_nodes * 2);
        for (int i = 0; i < g->n_nodes; ++i)
            for (int k = 0; k < GGML_MAX_SRC; ++k) if (g->nodes[i]->src[k]) ++direct[g->nodes[i]->src[k]];
        const ggml_hash_set & hs = g->visited_hash_set;
        auto whole = [&](const ggml_tensor * t) -> int {                 // ggml_hash_find (ggml-impl.h:257-270): pointer >> 4, linear probing
            const size_t h = ((size_t) (uintptr_t) t >> 4) % hs.size;
            size_t i = h;
            while ((hs.used[i >> 5] >> (i & 31)) & 1u) {
                if (hs.keys[i] == t) return g->use_counts[i];
                i = (i + 1) % hs.size;
                if (i == h) break;
            }
            return -1;
        };
        for (int i = 0; i < g->n_nodes; ++i) {
            const ggml_tensor * t = g->nodes[i];
            const int w = whole(t);
            auto it = direct.find(t);
            if (w > (it == direct.end() ? 0 : it->second))
                for (const ggml_tensor * r = t; r; r = r->view_src) s.external.insert(r);     // a view read elsewhere keeps its base's bytes alive too
        }
    }
    if (s.c->opt_profile && !s.capturing) {
        // calibration sample: an event pair with nothing in between measures the bracket's own cost, which consumers subtract
        for (int k = 0; k < 4; ++k) { prof_scope ps(s, "empty", 0); }
    }
    for (int i = s.node_lo; i < (s.node_hi < 0 || s.node_hi > g->n_nodes ? g->n_nodes : s.node_hi); ++i) {
        if (s.done[i]) continue;
        static const bool host_prof = getenv("MI355X_HOST_PROF") != nullptr;          // host time of the node walk by op (stderr, per graph): where an eager graph's enqueue time goes
        static double hp_ns[GGML_OP_COUNT]; static long hp_n[GGML_OP_COUNT];
        const auto hp_t0 = host_prof ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        struct hp_guard { bool on; int op; std::chrono::steady_clock::time_point t0; ~hp_guard() { if (on) { hp_ns[op] += (double) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); ++hp_n[op]; } } } hp_g{ host_prof, (int) g->nodes[i]->op, hp_t0 };
        if (host_prof && i == g->n_nodes - 1) {
            fprintf(stderr, "[mi355x] host walk by op (cumulative):");
            for (int o = 0; o < GGML_OP_COUNT; ++o) if (hp_n[o]) fprintf(stderr, " op%d n=%ld %.2fus/node", o, hp_n[o], hp_ns[o] / hp_n[o] * 1e-3);
            fprintf(stderr, "\n");
        }
        struct ll_guard { exec_state & s; ggml_cgraph * g; int i; long k0; long f0; ~ll_guard() {
            if (!launch_log || s.capturing || s.n_kernels == k0) return;
            const ggml_tensor * n = g->nodes[i];
            fprintf(launch_log, "%d %d %ld %ld [%lld,%lld,%lld,%lld]", i, (int) n->op, s.n_kernels - k0, s.n_fused - f0, (long long) n->ne[0], (long long) n->ne[1], (long long) n->ne[2], (long long) n->ne[3]);
            for (int k = 0; k < 3 && n->src[k]; ++k) fprintf(launch_log, " s%d:op%d%s[%lld,%lld,%lld,%lld]", k, (int) n->src[k]->op, is_contiguous(n->src[k]) ? "c" : "n", (long long) n->src[k]->ne[0], (long long) n->src[k]->ne[1], (long long) n->src[k]->ne[2], (long long) n->src[k]->ne[3]);
            fprintf(launch_log, "\n");
        } } ll_g{ s, g, i, s.n_kernels, s.n_fused };
        if (s.gs.n && i != s.gs.consumer && !is_noop(g->nodes[i]) && g->nodes[i] != s.gs.n) gs_materialise(s);       // (somebody else runs before wo folds the attention slices)
        if (g->nodes[i]->op == GGML_OP_IM2COL && exec_conv1d_tc(s, i)) continue;
        if (g->nodes[i]->op == GGML_OP_CONT && exec_causal_conv(s, i)) continue;
        if ((g->nodes[i]->op == GGML_OP_CONT || g->nodes[i]->op == GGML_OP_CONCAT) && exec_concat_tail(s, i)) continue;
        if (g->nodes[i]->op == GGML_OP_CONT && lazy_try_register(s, i)) continue;
        if (!is_noop(g->nodes[i]) && g->nodes[i]->op != GGML_OP_CONCAT && !(g->nodes[i]->op == GGML_OP_MUL_MAT && g->nodes[i]->src[0]->type == GGML_TYPE_F32 && g->nodes[i]->src[1]->type == GGML_TYPE_F32)) lazy_net(s, i);
        if (g->nodes[i]->op == GGML_OP_MUL && exec_gate_norm(s, i)) continue;
        {
            int taken[8];
            const int nt = is_noop(g->nodes[i]) ? 0 : exec_ew_chain(s, i, taken);
            if (nt > 0) { for (int k = 0; k < nt; ++k) s.done[taken[k]] = 1; continue; }
        }
        const int sink = cont_sink(s, i);
        if (sink >= 0) {
            void * own = g->nodes[i]->data;
            g->nodes[i]->data = g->nodes[sink]->data;            // (the launches take the pointer now; the node gets its own back right after)
            compute_node(s, i);
            g->nodes[i]->data = own;
            s.done[sink] = 1; ++s.n_fused;
            note_write(s, g->nodes[sink]);
        } else
            compute_node(s, i);
        if (s.gs.n && i == s.gs.consumer) { fprintf(stderr, "[mi355x] graph_compute: node %d (%s) did not take the attention slices it was chosen for\n", i, g->nodes[i]->name); abort(); }
        if (!s.capturing) {                                  // a launch with an invalid configuration fails silently otherwise (and poisons a later capture)
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) {
                log_msg(GGML_LOG_LEVEL_ERROR, "[mi355x] graph_compute: node %d (%s, op %d, ne = [%lld, %lld, %lld, %lld]) failed to launch: %s\n", i, g->nodes[i]->name, (int) g->nodes[i]->op,
                        (long long) g->nodes[i]->ne[0], (long long) g->nodes[i]->ne[1], (long long) g->nodes[i]->ne[2], (long long) g->nodes[i]->ne[3], hipGetErrorString(e));
                abort();
            }
        }
    }
    gs_materialise(s);                                       // (the attention node was the graph's last launching node)
    if (launch_log && !s.capturing) { fprintf(launch_log, "== end of a graph of %d nodes\n", g->n_nodes); fflush(launch_log); }
}


} // namespace mi

// graph_exec.cpp -- executor side, core: activation images, the MUL_MAT executor, deferral bookkeeping (pending norm / reductions / attention slices), the hoisting
// rules, compute_node and run_nodes.  The fusion matchers live beside it since round 6: graph_exec_llm.cpp (text decoder), graph_exec_t2w.cpp (encoders / Token2Wav);
// shared declarations in graph_exec_internal.hpp.  (Split out of graph.cpp in round 4, by module in round 6; no behaviour change.)
#include "graph_exec_internal.hpp"
#include <map>

namespace mi {

// convert src1 of a MUL_MAT into the activation format of `kind` (or reuse the cached conversion); returns the image stride
size_t prepare_act(exec_state & s, const ggml_tensor * x, act_kind kind) {
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

const char * mmv_class(int type) {
    return type == GGML_TYPE_Q4_K ? "mmv_q4k" : type == GGML_TYPE_Q6_K ? "mmv_q6k" : type == GGML_TYPE_Q8_0 ? "mmv_q80" : type == GGML_TYPE_F16 ? "mmv_f16" : "mmv_f32";
}

// resident F16 image of a quantised weight matrix (shadow.hpp): built on first use outside of graph capture, only for tensors
// that live in a buffer marked GGML_BACKEND_BUFFER_USAGE_WEIGHTS
const uint16_t * weight_shadow(exec_state & s, const ggml_tensor * w, const char * wp, int64_t K, int64_t M) {
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
void op_mul_mat(exec_state & s, const ggml_tensor * dst, const ggml_tensor * out, const float * bias, const mm_sibling * sib, int nsib, bool * sib_taken, int act) {
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
        a.dst = (float *) out->data; a.dst_cs = out->nb[1]; a.dst_nb2 = out->nb[2]; a.dst_nb3 = out->nb[3]; a.accumulate = k_done > 0; a.bias = bias; a.act = act;
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
bool plain_kq_matvec(const ggml_tensor * n, int max_cols) {      // MUL_MAT(K-quant W [K,M], f32 x [K,N<=max]) with no broadcast
    if (n->op != GGML_OP_MUL_MAT || is_empty(n)) return false;
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    return is_kquant(w->type) && w->ne[2] == 1 && w->ne[3] == 1 && x->ne[2] == 1 && x->ne[3] == 1 && x->ne[1] <= max_cols &&
           q8k_image_bytes(w->ne[0]) * (size_t) x->ne[1] <= 152 * 1024 && n->nb[0] == 4;
}
// ... or against 9 .. 64 columns on the int8 matrix cores (mmq.hip): the same fusions (sibling batching, residual epilogue, norm image)
bool kq_mm_ok(const ggml_tensor * n) {
    if (n->op != GGML_OP_MUL_MAT || is_empty(n)) return false;
    if (!mm_uses_mmq(n)) return plain_kq_matvec(n, MI_MMVQ_MAX_COLS);
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    return w->ne[2] == 1 && w->ne[3] == 1 && x->ne[2] == 1 && x->ne[3] == 1 && n->nb[0] == 4 && x->nb[0] == 4;
}
// the Q8_0 twin (mmv1q.hip): MUL_MAT(Q8_0 W [K, M], f32 x [K, 1]), no broadcast -- the TTS / Token2Wav modules' decode mat-vecs
bool q80_mv1_node(exec_state & s, const ggml_tensor * n) {
    if (!s.c->opt_mv1 || n->op != GGML_OP_MUL_MAT || is_empty(n)) return false;
    const ggml_tensor * w = n->src[0], * x = n->src[1];
    if ((w->type != GGML_TYPE_Q8_0 && w->type != GGML_TYPE_F16) || x->type != GGML_TYPE_F32 || w->ne[2] != 1 || w->ne[3] != 1 || x->ne[1] != 1 || x->ne[2] != 1 || x->ne[3] != 1 || n->nb[0] != 4 || x->nb[0] != 4) return false;
    mv1_args v; v.nmat = 1; v.K = w->ne[0];
    v.m[0] = { w->data, w->nb[1], (float *) n->data, 0, nullptr, 0, w->ne[1], (int) w->type };
    v.img = (const void *) 16;
    return mmv1_ok(v);
}
// batch-1 decode form (mmv1.hip): one column, Q4_K / Q6_K, K a multiple of 256 up to 16384, aligned rows; or the Q8_0 twin
bool mv1_node_ok(exec_state & s, const ggml_tensor * n) {
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
// the attention rows this mat-vec reads still lie as slices' partial states: fold them in the launch's prologue if the LDS-DMA engine takes the launch, else write the rows first
void gs_materialise(exec_state & s) {
    if (!s.gs.n) return;
    prof_scope ps(s, "fattn", 0);
    fattn_gs_merge((const float *) s.c->fa_scratch, (float *) s.gs.n->data, s.gs.nh, s.gs.D, s.st); ++s.n_kernels;
    s.gs.n = nullptr; s.gs.consumer = -1;
}
void mv1_source(exec_state & s, const ggml_tensor * x, const ggml_tensor * const * outs, int n_outs, int n_consumers, mv1_args & v) {
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
bool same_act(const ggml_tensor * a, const ggml_tensor * b) {
    return a->data == b->data && a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3] &&
           a->nb[1] == b->nb[1] && a->nb[2] == b->nb[2] && a->nb[3] == b->nb[3];
}
int n_users(exec_state & s, const ggml_tensor * t) {
    auto it = s.users.find(t);
    return (it == s.users.end() ? 0 : (int) it->second.size()) + (s.external.count(t) ? 1 : 0);
}
int sole_user(exec_state & s, const ggml_tensor * t) {           // index of the only consumer node, or -1
    auto it = s.users.find(t);
    if (it == s.users.end() || it->second.size() != 1 || is_out(s, t)) return -1;
    return it->second[0];
}
int next_real_node(exec_state & s, int i) {                      // the next node after i that will launch something (-1: none)
    for (int j = i + 1; j < s.g->n_nodes; ++j) if (!s.done[j] && !is_noop(s.g->nodes[j])) return j;
    return -1;
}
bool ready_before(exec_state & s, const ggml_tensor * src, int i, const int * item, int n_item) {
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
bool can_hoist(exec_state & s, int i, int j, const int * item, int n_item) {
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
void note_write(exec_state & s, const ggml_tensor * t) {          // a kernel wrote t: drop the activation cache if it aliased
    if (s.fa_mask) { const char * p = (const char *) t->data; if ((const char *) s.fa_mask >= p && (const char *) s.fa_mask < p + nbytes(t)) s.fa_mask = nullptr; }
    if (!s.a_src) return;
    const byte_range r = range_of(t);
    if (r.lo < s.a_range_hi && s.a_range_lo < r.hi) s.a_src = nullptr;
}

// ---- deferred norm (see exec_state::pn)
void materialise_norm(exec_state & s) {                           // run the stand-alone kernel now: f32 result + Q8_K image, seeds the cache
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
bool norm_in_kernel(exec_state & s, const ggml_tensor * x, const ggml_tensor * const * outs, int n_outs, int n_consumers, mmv_norm & nr) {
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
void materialise_reduce(exec_state & s) {
    const ggml_tensor * A = s.pr.A;
    s.pr.A = nullptr;
    prof_scope ps(s, "gemm_reduce", 0);
    gemm_reduce2((const float *) s.c->gemm_partial, s.pr.nsplit, s.pr.resid, s.pr.resid_cs, s.pr.resid2, s.pr.resid2_cs, (float *) A->data, A->nb[1], A->ne[0], A->ne[1], s.st);
    ++s.n_kernels;
}
void materialise_group(exec_state & s, int skip_mask) {       // skip_mask: results somebody has taken as slabs
    float * dst[3] = { nullptr, nullptr, nullptr }; size_t cs[3] = { 0, 0, 0 }, off[3] = { 0, 0, 0 }; int64_t M[3] = { 0, 0, 0 }; int n = 0;
    for (int q = 0; q < s.prm.n; ++q) if (!(skip_mask & (1 << q))) { dst[n] = (float *) s.prm.A[q]->data; cs[n] = s.prm.A[q]->nb[1]; off[n] = s.prm.off[q]; M[n] = s.prm.M[q]; ++n; }
    if (n > 0) {
        prof_scope ps(s, "gemm_reduce", 0);
        gemm_reduce_group((const float *) s.c->gemm_partial, s.prm.nsplit, s.prm.slab, n, off, M, s.prm.N, dst, cs, s.st);
        ++s.n_kernels;
    }
    s.prm.n = 0;
}
bool reads_pending_group(exec_state & s, const ggml_tensor * n) {          // an RMS_NORM on (a view of) one of the pending grouped results
    if (n->op != GGML_OP_RMS_NORM || !n->src[0]) return false;
    for (int q = 0; q < s.prm.n; ++q) if (n->src[0]->data == s.prm.A[q]->data) return true;
    return false;
}

// ------------------------------------------------------------------------------------------------ node dispatch
void compute_node(exec_state & s, int i) {
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
            {
                const int dim = op_param_i32(n, 0), es = n->type == GGML_TYPE_F16 ? 2 : 4;
                const tdesc y = td(n);
                copy_pair cp[2];
                for (int k = 0; k < 2; ++k) {                               // operand k lands in its slab of y: the same strides, its own extent, b behind a along `dim`
                    cp[k].src = sd[k]; cp[k].es = es; cp[k].dst = y;
                    for (int d = 0; d < 4; ++d) cp[k].dst.ne[d] = sd[k].ne[d];
                    if (k == 1) cp[k].dst.p = (char *) y.p + (size_t) sd[0].ne[dim] * y.nb[dim];
                }
                const bool same = (n->type == GGML_TYPE_F32 || n->type == GGML_TYPE_F16 || n->type == GGML_TYPE_I32) && n->src[0]->type == n->type && n->src[1]->type == n->type;
                if (copy_queue_on(s) && same && dim >= 0 && dim < 4 && copy_batch_ok(cp[0].src, cp[0].dst, es) && copy_batch_ok(cp[1].src, cp[1].dst, es)) copy_queue(s, cp, 2, y, n, dim);
                else { copy_flush(s); concat(sd[0], sd[1], y, dim, es, s.st); ++s.n_kernels; }
            }
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
            if ((src->type == GGML_TYPE_F32) != (n->type == GGML_TYPE_F32) && (src->type == GGML_TYPE_I32 || n->type == GGML_TYPE_I32)) { copy_flush(s); cast_f32_i32(td(src), src->type == GGML_TYPE_F32, td(n), s.st); }
            else {
                // CONT(PERMUTE(kqv)) of a prefill ubatch whose only readers are MFMA GEMMs (wo): gather straight into the f16 activation image
                const ggml_tensor * xg = nullptr;
                // The f32 tensor itself is then never written, so the path is taken only when the one reader is the NEXT launching node (the rule of
                // SOFT_MAX / UNARY / exec_norm): anything in between that prepares another activation image would evict this one, and the reader
                // would convert from n->data.
                const int cu = s.c->opt_fusion && !is_out(s, n) ? sole_user(s, n) : -1;
                if (n->op == GGML_OP_CONT && src->type == GGML_TYPE_F32 && n->type == GGML_TYPE_F32 && n->ne[2] == 1 && n->ne[3] == 1 && n->nb[1] == (size_t) n->ne[0] * 4 &&
                    cu > i && next_real_node(s, i) == cu && n_users(s, n) == 1 && gemm_only_consumers(s, n, n->ne[0], n->ne[1], &xg)) {
                    copy_flush(s);
                    tdesc d; d.p = s.c->act_scratch; d.ne[0] = n->ne[0]; d.ne[1] = n->ne[1]; d.ne[2] = 1; d.ne[3] = 1;
                    const size_t img = act_image_bytes(ACT_F16, n->ne[0]);
                    d.nb[0] = 2; d.nb[1] = img; d.nb[2] = img * (size_t) n->ne[1]; d.nb[3] = d.nb[2];
                    cpy_strided(td(src), GGML_TYPE_F32, d, GGML_TYPE_F16, s.st);
                    ++s.n_kernels; ++s.n_fused;
                    seed_act_f16(s, xg);
                    return;                                         // (the f32 copy was not written)
                }
                {
                    const int es = src->type == GGML_TYPE_F16 ? 2 : 4;
                    const copy_pair cp = { td(src), td(n), es };
                    if (copy_queue_on(s) && src->type == n->type && (src->type == GGML_TYPE_F32 || src->type == GGML_TYPE_F16 || src->type == GGML_TYPE_I32) && copy_batch_ok(cp.src, cp.dst, es)) {
                        copy_queue(s, &cp, 1, cp.dst, n);
                        note_write(s, n);
                        return;                                         // (counted when the batch leaves)
                    }
                }
                copy_flush(s);
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
                    if (x && x->data == n->data && x->ne[0] == n->ne[0] * n->ne[1] && x->ne[1] == 1 && x->ne[2] == 1 && x->ne[3] == 1 && x != s.pn.m && mv1_node_ok(s, c)) {                 // (K-quant or Q8_0 wo: mmv2_ok decides)
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
byte_range range_of(const tdesc & d) {
    size_t ext = 4;
    for (int k = 0; k < 4; ++k) ext += (size_t) (d.ne[k] > 0 ? d.ne[k] - 1 : 0) * d.nb[k];
    return { (const char *) d.p, (const char *) d.p + ext };
}

// ------------------------------------------------------------------------------------------------ deferred layout copies
// Token2Wav's window graph spends a third of its launches on CONT / CONCAT / CPY nodes that move a few KB each (tools/launch_ngrams.py: runs of 4 and 7 of them inside every
// DiT block, a 350-launch cache-packing suffix), most of them independent of their neighbours.  Such a node does not launch: its copy is queued, and the queue leaves as ONE
// k_copy_batch launch (elementwise.hip) when (a) a node that is not a plain copy is about to run, (b) a new copy reads bytes a pending copy writes, or writes bytes one reads or
// writes (conservative byte ranges of the strided views; ggml-alloc re-uses addresses inside such runs), or (c) COPY_BATCH_MAX jobs are pending.  Within a batch no job
// depends on another, so the result is what the in-order launches gave.  Off with fusion off, in profile mode (per-class event timing) and by MI355X_NO_COPY_BATCH=1.
bool copy_queue_on(exec_state & s) {
    static const bool off = getenv("MI355X_NO_COPY_BATCH") != nullptr;
    return !off && s.c->opt_copy_batch != 0 && s.c->opt_fusion && !s.c->opt_profile;
}
static long g_copy_pruned = 0;
// pending groups nobody will read: dropped (see copy_pending::node).  `keep`: a group that must stay (the one a forwarding in progress reads from)
static void copy_prune(exec_state & s, int keep = -1) {
    static const bool off = getenv("MI355X_NO_COPY_PRUNE") != nullptr;
    if (off || s.cq.empty()) return;
    int dead[COPY_BATCH_MAX]; int nd = 0;
    int last = -1;
    for (const exec_state::copy_pending & P : s.cq) {
        if (P.group == last || P.group == keep) continue;
        last = P.group;
        const ggml_tensor * t = P.node;
        if (!t || t->view_src || t->op == GGML_OP_CPY || is_out(s, t)) continue;      // (a view / a CPY writes somebody else's storage: a persistent cache)
        auto us = s.users.find(t);
        bool live = us == s.users.end() || us->second.empty();                 // (no reader inside the graph at all: a result somebody fetches afterwards)
        if (us != s.users.end()) for (int u : us->second) if (u >= s.cur_node && !s.done[u]) { live = true; break; }      // (the node being processed may still read it un-forwarded)
        if (!live) { const byte_range ry = range_of(P.Y); for (auto & kv : s.lazy) if (overlap(range_of(kv.second.src), ry)) { live = true; break; } }
        if (!live && s.va.cast) { if (overlap(range_of(s.va.v), range_of(P.Y))) live = true; }
        if (!live && nd < COPY_BATCH_MAX) {
            dead[nd++] = P.group;
            static const bool verify = getenv("MI355X_COPY_PRUNE_VERIFY") != nullptr;
            if (verify) {
                const byte_range ry = range_of(P.Y); s.cq_dead.push_back({ ry.lo, ry.hi, t, s.cur_node });
                if (getenv("MI355X_COPY_PRUNE_VERBOSE")) {
                    fprintf(stderr, "[mi355x] PRUNE: dropping group of node %d (%s) at node %d; users:", s.index.count(t) ? s.index[t] : -1, t->name, s.cur_node);
                    if (us != s.users.end()) for (int u : us->second) fprintf(stderr, " %d(op%d%s%s)", u, (int) s.g->nodes[u]->op, s.done[u] ? ",done" : "", s.lazy.count(s.g->nodes[u]) ? ",lazy" : "");
                    fprintf(stderr, "\n");
                }
            }
        }
    }
    if (!nd) return;
    size_t w = 0;
    for (size_t r = 0; r < s.cq.size(); ++r) {
        bool d = false;
        for (int k = 0; k < nd; ++k) if (s.cq[r].group == dead[k]) d = true;
        if (!d) { if (w != r) s.cq[w] = s.cq[r]; ++w; } else { ++s.c->stat_copies_dropped; if (!s.capturing) ++g_copy_pruned; }
    }
    s.cq.resize(w);
}
static long g_copy_hist[COPY_BATCH_MAX + 1], g_copy_why[4];              // MI355X_SINK_DEBUG: batch sizes; flushes by hazard kind (RAW, WAR, WAW, full)
void copy_flush(exec_state & s) {
    if (s.cq.empty()) return;
    static const bool dbg = getenv("MI355X_SINK_DEBUG") != nullptr;
    struct dump { ~dump() { if (dbg) { fprintf(stderr, "[mi355x] copy batches by size:"); for (int k = 1; k <= COPY_BATCH_MAX; ++k) if (g_copy_hist[k]) fprintf(stderr, " %d:%ld", k, g_copy_hist[k]);
                                       fprintf(stderr, "; hazard checks that hit RAW %ld WAR %ld WAW %ld, nodes forwarded %ld, jobs dropped %ld\n", g_copy_why[0], g_copy_why[1], g_copy_why[2], g_copy_why[3], g_copy_pruned); } } };
    static dump at_exit;
    copy_prune(s);
    if (s.cq.empty()) return;
    if (dbg && !s.capturing) ++g_copy_hist[s.cq.size()];
    copy_pair jobs[COPY_BATCH_MAX];
    const int n = (int) s.cq.size();
    for (int k = 0; k < n; ++k) jobs[k] = s.cq[k].job;
    if (n == 1) {                                                           // (one copy: the kernels that know dense rows and 64-bit sizes)
        const int ty = jobs[0].es == 4 ? GGML_TYPE_F32 : GGML_TYPE_F16;
        cpy_strided(jobs[0].src, ty, jobs[0].dst, ty, s.st);
    } else {
        copy_batch(jobs, n, s.st);
        s.n_copies_batched += n; s.c->stat_copies_batched += n;
    }
    ++s.n_kernels;
    s.cq.clear(); s.cq_dead.clear(); s.vplain.t = nullptr;
}
static bool same_desc(const tdesc & a, const tdesc & b) {
    if (a.p != b.p) return false;
    for (int d = 0; d < 4; ++d) if (a.ne[d] != b.ne[d] || (a.ne[d] > 1 && a.nb[d] != b.nb[d])) return false;
    return true;
}
void copy_queue(exec_state & s, const copy_pair * jobs, int nj, const tdesc & Y, const ggml_tensor * node, int dim1) {
    if (nj < 1 || nj > 2) { fprintf(stderr, "[mi355x] copy_queue: %d jobs\n", nj); abort(); }
    typedef exec_state::copy_pending pend;
    auto mk = [&](const copy_pair & j, const int64_t (&org)[4], int group) {
        pend e; e.job = j;
        const byte_range r = range_of(j.src), w = range_of(j.dst);
        e.rlo = r.lo; e.rhi = r.hi; e.wlo = w.lo; e.whi = w.hi; e.group = group; e.Y = Y; e.node = s.cq_owner ? s.cq_owner : node;
        e.same = true;
        for (int d = 0; d < 4; ++d) { e.org[d] = org[d]; if (j.src.ne[d] != j.dst.ne[d]) e.same = false; }
        return e;
    };
    auto hit = [](const char * alo, const char * ahi, const char * blo, const char * bhi) { return alo < bhi && blo < ahi; };
    auto hazard = [&](const std::vector<pend> & v) {
        for (const pend & P : s.cq)
            for (const pend & e : v) {
                const int why = hit(e.rlo, e.rhi, P.wlo, P.whi) ? 0 : hit(e.wlo, e.whi, P.rlo, P.rhi) ? 1 : hit(e.wlo, e.whi, P.wlo, P.whi) ? 2 : -1;
                if (why >= 0) { if (!s.capturing) ++g_copy_why[why]; return true; }
            }
        return false;
    };
    auto undead = [&](const std::vector<pend> & v) {                      // (verify mode) bytes these jobs write are defined again
        for (const pend & e : v) for (size_t q = 0; q < s.cq_dead.size(); ) { if (e.wlo < s.cq_dead[q].hi && s.cq_dead[q].lo < e.whi) s.cq_dead.erase(s.cq_dead.begin() + q); else ++q; }
    };
    const int group = ++s.cq_group;
    std::vector<pend> orig, fwd;
    bool any_fwd = false;
    static const bool no_fwd = getenv("MI355X_NO_COPY_FORWARD") != nullptr;
    for (int k = 0; k < nj; ++k) {
        int64_t org[4] = { 0, 0, 0, 0 };
        if (k == 1 && dim1 >= 0) org[dim1] = jobs[0].dst.ne[dim1];
        const pend e = mk(jobs[k], org, group);
        orig.push_back(e);
        // forwarding: this job reads EXACTLY the output of a pending node whose boxes are same-shape copies -> it reads that node's sources, box by box, instead
        int g = -1;
        if (!no_fwd && e.same) for (const pend & P : s.cq) if (same_desc(P.Y, e.job.src)) { g = P.group; break; }
        bool ok = g >= 0;
        if (ok) for (const pend & P : s.cq) if (P.group == g && (!P.same || P.job.es != e.job.es)) ok = false;
        if (!ok) { fwd.push_back(e); continue; }
        for (const pend & P : s.cq) {
            if (P.group != g) continue;
            copy_pair c; c.es = e.job.es; c.src = P.job.src; c.dst = e.job.dst;
            int64_t o2[4];
            char * dp = (char *) e.job.dst.p;
            for (int d = 0; d < 4; ++d) { c.dst.ne[d] = P.job.dst.ne[d]; dp += (size_t) P.org[d] * e.job.dst.nb[d]; o2[d] = org[d] + P.org[d]; }
            c.dst.p = dp;
            fwd.push_back(mk(c, o2, group));
        }
        any_fwd = true;
    }
    if (any_fwd && (s.cq.size() + fwd.size() > (size_t) COPY_BATCH_MAX || hazard(fwd))) copy_prune(s);          // (dead packs in the way: WAW / WAR against bytes nobody reads)
    if (any_fwd && s.cq.size() + fwd.size() <= (size_t) COPY_BATCH_MAX && !hazard(fwd)) {
        if (!s.capturing) ++g_copy_why[3];                                  // (counted as "forwarded")
        ++s.c->stat_copies_forwarded;
        for (const pend & e : fwd) s.cq.push_back(e);
        undead(fwd);
        return;
    }
    if (s.cq.size() + orig.size() > (size_t) COPY_BATCH_MAX || hazard(orig)) { copy_prune(s); if (s.cq.size() + orig.size() > (size_t) COPY_BATCH_MAX || hazard(orig)) copy_flush(s); }
    for (const pend & e : orig) s.cq.push_back(e);
    undead(orig);
}

void run_nodes(exec_state & s, ggml_cgraph * g) {
    static FILE * const launch_log = getenv("MI355X_LAUNCH_LOG") ? fopen(getenv("MI355X_LAUNCH_LOG"), "w") : nullptr;      // one line per node that launched: what a graph's launches are made of (tools/launch_ngrams.py)
    s.g = g;
    s.done.assign(g->n_nodes, 0);
    s.index.clear(); s.users.clear(); s.lazy.clear(); s.lazy_base_deadline.clear(); s.gs = {};
    s.cq.clear();
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
        s.cur_node = i;
        if (!s.cq_dead.empty() && !is_noop(g->nodes[i])) {                   // (verify mode) does this node read bytes of a dropped copy that nobody has rewritten since?
            const ggml_tensor * n_ = g->nodes[i];
            for (int k = 0; k < GGML_MAX_SRC; ++k) {
                if (!n_->src[k] || !n_->src[k]->data) continue;
                { bool lz = false; for (const ggml_tensor * t_ = n_->src[k]; t_; t_ = t_->view_src ? t_->view_src : ((t_->op == GGML_OP_RESHAPE || t_->op == GGML_OP_VIEW || t_->op == GGML_OP_PERMUTE || t_->op == GGML_OP_TRANSPOSE) ? t_->src[0] : nullptr)) if (s.lazy.count(t_)) lz = true; if (lz) continue; }
                const byte_range r = range_of(n_->src[k]);
                for (const auto & d : s.cq_dead)
                    if (r.lo < d.hi && d.lo < r.hi)
                        fprintf(stderr, "[mi355x] PRUNE VERIFY: node %d (%s, op %d) src%d %s [%lld,%lld,%lld,%lld] reads the dropped output of %s (op %d, dropped at node %d); src is node %d, dropped is node %d, src range %p+%zu dead range %p+%zu\n", i, n_->name, (int) n_->op, k, n_->src[k]->name,
                                (long long) n_->src[k]->ne[0], (long long) n_->src[k]->ne[1], (long long) n_->src[k]->ne[2], (long long) n_->src[k]->ne[3], d.node->name, (int) d.node->op, d.at, s.index.count(n_->src[k]) ? s.index[n_->src[k]] : -1, s.index.count(d.node) ? s.index[d.node] : -1, (const void *) r.lo, (size_t) (r.hi - r.lo), (const void *) d.lo, (size_t) (d.hi - d.lo));
            }
        }
        struct dead_guard { exec_state & s; const ggml_tensor * n; ~dead_guard() {        // whatever this node wrote is defined again
            if (s.cq_dead.empty() || is_noop(n)) return;
            const byte_range w = range_of(n);
            for (size_t q = 0; q < s.cq_dead.size(); ) { if (w.lo < s.cq_dead[q].hi && s.cq_dead[q].lo < w.hi) s.cq_dead.erase(s.cq_dead.begin() + q); else ++q; }
        } } dead_g{ s, g->nodes[i] };
        {   // pending copies leave before anything that is not itself a plain copy (the copy-shaped matchers below flush where they launch)
            const int op_ = g->nodes[i]->op;
            if (!s.cq.empty() && !is_noop(g->nodes[i]) && op_ != GGML_OP_CONT && op_ != GGML_OP_CONCAT && op_ != GGML_OP_CPY && op_ != GGML_OP_DUP) copy_flush(s);
        }
        if (s.gs.n && i != s.gs.consumer && !is_noop(g->nodes[i]) && g->nodes[i] != s.gs.n) { copy_flush(s); gs_materialise(s); }       // (somebody else runs before wo folds the attention slices)
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
            s.cq_owner = g->nodes[sink];
            compute_node(s, i);
            s.cq_owner = nullptr;
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
    if (s.vplain.t) { fprintf(stderr, "[mi355x] graph_compute: a V tensor written as rows was never read by its attention launch\n"); abort(); }
    copy_flush(s);
    gs_materialise(s);                                       // (the attention node was the graph's last launching node)
    if (launch_log && !s.capturing) { fprintf(launch_log, "== end of a graph of %d nodes\n", g->n_nodes); fflush(launch_log); }
}


} // namespace mi

// graph_exec_llm.cpp -- executor, text-decoder side: the fusion matchers of the Qwen3 / llama graphs (llm_build_qwen3, src/llama-model.cpp:9287-9406): sibling mat-muls
// as grouped GEMM / mat-vec launches, RMS_NORM -> MUL chains folded into their consumers, q / k norm + rope + KV store chains handed to the attention launch, split-K
// reductions folded into the norm behind them, the flash-attention-off chain of a prefill ubatch.  (Split out of graph_exec.cpp in round 6; no behaviour change.)
#include "graph_exec_internal.hpp"

namespace mi {

// Q4_K / Q6_K weights with NO resident F16 image (MI355X_NO_F16_SHADOW, the image budget spent, out of memory): the GEMM de-quantises the blocks
// inside its LDS staging (k_gemm_kq_glds) instead of running a de-quantise-to-scratch launch in front of every mat-mul.  With the image resident
// the F16 kernel is faster at every column count (the in-staging form spends ~900 VALU cycles per wave and K-step on nibbles, scales and f16
// rounding against 512 MFMA cycles: measured pp100 9.9 vs 7.6 ms, pp256 13.5 vs 9.4 ms), so otherwise it is only taken on request: MI355X_KQ_STAGING=1 / set_option("kq_staging") (<= MAX_COLS columns).
bool kq_in_staging(exec_state & s, const ggml_tensor * w, int64_t N) {
    static const bool off = getenv("MI355X_NO_KQ_STAGING") != nullptr;
    static const int64_t max_n = getenv("MI355X_KQ_STAGING_MAX_COLS") ? atoll(getenv("MI355X_KQ_STAGING_MAX_COLS")) : 256;
    if (off || !(w->type == GGML_TYPE_Q4_K || w->type == GGML_TYPE_Q6_K) || w->ne[0] % 256 != 0 || w->ne[2] != 1 || w->ne[3] != 1 ||
        w->nb[1] % (w->type == GGML_TYPE_Q4_K ? 16 : 2) != 0 || ((uintptr_t) w->data & 15) != 0) return false;
    if (s.c->opt_kq_staging) return N <= max_n;
    return weight_shadow(s, w, (const char *) w->data, w->ne[0], w->ne[1]) == nullptr;
}
bool gemm_operand(exec_state & s, const ggml_tensor * w, const uint16_t ** w16, size_t * rs) {
    if (w->ne[2] != 1 || w->ne[3] != 1) return false;
    if (w->type == GGML_TYPE_F16) { *w16 = (const uint16_t *) w->data; *rs = w->nb[1]; return true; }
    const uint16_t * sh = weight_shadow(s, w, (const char *) w->data, w->ne[0], w->ne[1]);
    if (!sh) return false;
    *w16 = sh; *rs = (size_t) w->ne[0] * 2;
    return true;
}
bool gemm_groupable(const ggml_tensor * c) {
    if (c->op != GGML_OP_MUL_MAT || is_empty(c) || !mm_uses_gemm(c)) return false;
    const ggml_tensor * x = c->src[1];
    return x->ne[2] == 1 && x->ne[3] == 1 && c->src[0]->ne[0] % 64 == 0 && c->nb[0] == 4 && c->type == GGML_TYPE_F32;
}
bool exec_gemm_group(exec_state & s, int i) {
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
                     is_contiguous(Cp) && Cp->ne[0] == N && Cp->ne[1] == S->ne[1] && Cp->ne[2] == S->ne[2] && Cp->ne[3] == 1) {       // [n_tokens, D, H] of PERMUTE(1, 2, 0, 3): V^T per head (an encoder's V CAST)
                ms = (size_t) N * 2; rs = 2;
                // ... unless its one reader is the K.Q -> SOFT_MAX -> V^T.P chain this executor runs as ONE flash-attention launch: then the tensor's bytes are written as V rows
                // ([D, n_tokens, H], the layout K has) and the launch is the plain-V form, whose head-size-64 kernel is the LDS-DMA ring (Whisper 1500 x 1500 x 16: 53.7 -> 39.5 us)
                static const bool no_vplain = getenv("MI355X_NO_ATTN_VPLAIN") != nullptr;
                const int64_t Dh = S->ne[1];
                int u2 = -1;                                               // Cp's one reader besides the CPY node itself (ggml_cast names its result as src[1])
                if (!no_vplain && Dh == 64 && !s.vplain.t) {
                    auto us = s.users.find(Cp);
                    int cnt = 0;
                    if (us != s.users.end()) for (int u : us->second) if (u != cj) { u2 = u; ++cnt; }
                    if (cnt != 1 || is_out(s, Cp)) u2 = -1;
                }
                if (u2 > cj && !s.done[u2] && g->nodes[u2]->op == GGML_OP_MUL_MAT && g->nodes[u2]->src[0] == Cp && g->nodes[u2]->src[1] && g->nodes[u2]->src[1]->op == GGML_OP_SOFT_MAX) {
                    const ggml_tensor * SMn = g->nodes[u2]->src[1];
                    const ggml_tensor * M1n = SMn->src[0];
                    auto i1 = M1n ? s.index.find(M1n) : s.index.end();
                    if (i1 != s.index.end() && i1->second > cj && M1n->op == GGML_OP_MUL_MAT && M * 2 % 16 == 0 && ((uintptr_t) Cp->data & 15) == 0 && exec_attn_sm_prefill(s, i1->second, true)) {
                        ms = 2; rs = (size_t) M * 2;
                        s.vplain.t = Cp;
                        s.vplain.v.p = Cp->data; s.vplain.v.ne[0] = Dh; s.vplain.v.ne[1] = N; s.vplain.v.ne[2] = S->ne[2]; s.vplain.v.ne[3] = 1;
                        s.vplain.v.nb[0] = 2; s.vplain.v.nb[1] = (size_t) M * 2; s.vplain.v.nb[2] = (size_t) Dh * 2; s.vplain.v.nb[3] = (size_t) M * 2 * (size_t) N;
                    }
                }
            }
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
void exec_mul_mat(exec_state & s, int i) {
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
                // ... and the GELU behind that ADD (a DiT block's FFN: linear -> + bias -> GELU -> linear, token2wav-impl.cpp DiT feed-forward): the ADD's only reader, the next launch,
                // same shape -- applied to the finished value in the same epilogue, with the element-wise kernel's arithmetic (act_dev.hpp op_gelu); the ADD's rows are never written
                static const bool no_act = getenv("MI355X_NO_GEMM_ANY_ACT") != nullptr;
                const int ui = no_act || nsib > 0 || is_out(s, A) ? -1 : sole_user(s, A);
                if (ui > ai && next_real_node(s, ai) == ui && g->nodes[ui]->op == GGML_OP_UNARY && op_param_i32(g->nodes[ui], 0) == GGML_UNARY_OP_GELU && g->nodes[ui]->src[0] == A) {
                    const ggml_tensor * U = g->nodes[ui];
                    bool oku = U->type == GGML_TYPE_F32 && U->data;
                    for (int d = 0; oku && d < 4; ++d) oku = U->ne[d] == A->ne[d] && U->nb[d] == A->nb[d];
                    oku = oku && !overlap(range_of(U), range_of(n->src[0])) && !overlap(range_of(U), range_of(n->src[1])) && !overlap(range_of(U), range_of(r));
                    if (oku) {
                        op_mul_mat(s, n, U, (const float *) r->data, nullptr, 0, nullptr, 1);
                        s.done[ai] = 1; s.done[ui] = 1; s.n_fused += 2;
                        note_write(s, U);
                        return;
                    }
                }
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
bool match_norm_rope(exec_state & s, int j, nr_chain & c) {
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
norm_rope_job chain_job(exec_state & s, const nr_chain & c) {
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
bool gemm_only_consumers(exec_state & s, const ggml_tensor * t, int64_t K, int64_t N, const ggml_tensor ** x_out) {
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
act_kind consumers_act_kind(exec_state & s, const ggml_tensor * x) {
    auto it = s.users.find(x);
    if (it == s.users.end()) return ACT_F16;
    for (int u : it->second) { const ggml_tensor * c = s.g->nodes[u]; if (c->op == GGML_OP_MUL_MAT && c->src[1] && c->src[1]->data == x->data) return gemm_act_kind(c); }
    return ACT_F16;
}
void seed_act_f16(exec_state & s, const ggml_tensor * x, bool quantised) {   // the f16 image of x now sits in act_scratch (quantised: the emitter wrote the Q8_K-quantised values already)
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
bool exec_norm(exec_state & s, int i) {
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
bool try_defer_qkv_to_attention(exec_state & s, const nr_chain & A, const nr_chain * B, int vj, const int * item, int ni) {
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
bool try_defer_qkv_to_softmax_attention(exec_state & s, const nr_chain & A, const nr_chain * B, int vsj, const int * item, int ni) {
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
bool match_rope_only(exec_state & s, int j, nr_chain & c) {
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
bool exec_rope_chain(exec_state & s, int i) {
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
bool exec_rms_norm(exec_state & s, int i) {
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
bool exec_attn_sm_prefill(exec_state & s, int i, bool dry) {       // dry: would this MUL_MAT be taken?  (no launches, no state)
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
    const bool vplain = s.vplain.t == fv;                              // (the V^T tensor's bytes hold V rows: exec_gemm_group wrote them that way for this launch)
    f.dst = td(C);
    f.dst.ne[0] = D; f.dst.ne[1] = H; f.dst.ne[2] = nq; f.dst.ne[3] = ns;
    f.dst.nb[0] = 4; f.dst.nb[1] = (size_t) D * 4; f.dst.nb[2] = (size_t) D * H * 4; f.dst.nb[3] = (size_t) D * H * nq * 4;
    f.mask = nullptr; f.sinks = nullptr; f.scale = op_param_f32(SM, 0); f.max_bias = 0.0f; f.logit_softcap = 0.0f;
    f.scratch = nullptr; f.scratch_bytes = 0;
    if (!fattn_sm_prefill_ok(f)) { if (vplain) { fprintf(stderr, "[mi355x] exec_attn_sm_prefill: the chain whose V was written as rows is refused\n"); abort(); } return false; }
    if (vplain && !dry) { f.v = s.vplain.v; f.v_transposed = false; s.vplain.t = nullptr; }
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
void materialise_vt(exec_state & s) {
    ggml_cgraph * g = s.g;
    const ggml_tensor * C = g->nodes[s.va.cont_i], * K = g->nodes[s.va.cast_i];
    s.va.cast = nullptr;
    prof_scope ps(s, "cpy", 0);
    cpy_strided(td(C->src[0]), C->src[0]->type, td(C), C->type, s.st);
    cpy_strided(td(K->src[0]), K->src[0]->type, td(K), K->type, s.st);
    s.n_kernels += 2;
}
bool try_alias_vt(exec_state & s, int i) {
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

} // namespace mi

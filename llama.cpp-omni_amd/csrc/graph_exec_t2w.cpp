// graph_exec_t2w.cpp -- executor, encoder / Token2Wav side: the matchers of the reference's omni modules (audition.cpp, vision.cpp, token2wav-impl.cpp): the f32 attention
// chain, element-wise chains and in-place sinks, lazy copies, the streaming causal convolution and its cache, the vocoder's conv1d, LayerNorm + modulation, gated norms.
// (Split out of graph_exec.cpp in round 6; no behaviour change.)
#include "graph_exec_internal.hpp"

namespace mi {

// `consecutive`: the norm's readers must be the launches right behind it (false: the caller checks with can_hoist that they may run at its own position)
bool match_norm_modulate(exec_state & s, int i, norm_mod_match & M, bool consecutive) {
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
bool exec_norm_modulate(exec_state & s, int i) {
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
bool exec_gate_norm(exec_state & s, int i) {
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
// K.Q -> [SCALE] -> SOFT_MAX (no mask) -> V^T.P -> [views -> CONT of the [D, H, nq, ns] permutation], everything f32 and nothing else in between: one attn_f32 launch
// (the reference's Token2Wav DiT attention, token2wav-impl.cpp:406-439).  `i` is the K.Q MUL_MAT.
bool exec_attn_f32(exec_state & s, int i) {
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
        copy_flush(s);
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

// The reference's Token2Wav builders put a ggml_cont behind most ops -- on tensors that are contiguous already (a third of a window's 15 000 launches are such
// copies).  When the producer is a plain element-wise / gather op, the copy is the very next launching node and the producer's only reader (directly or through
// RESHAPEs), the producer writes straight into the copy's buffer and the copy is not launched.  ggml-alloc may have placed the copy's buffer over memory that
// became free when the producer ran -- the producer's own sources -- so that overlap is checked.  Returns the CONT's node index or -1.
int cont_sink(exec_state & s, int i) {
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
int exec_ew_chain(exec_state & s, int i, int * taken) {
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
void lazy_materialise(exec_state & s, const ggml_tensor * t, int reader_op) {
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
        const copy_pair cp = { it->second.src, td(t), 4 };
        if (copy_queue_on(s) && copy_batch_ok(cp.src, cp.dst, 4)) copy_queue(s, &cp, 1, cp.dst, t);
        else { copy_flush(s); cpy_strided(it->second.src, GGML_TYPE_F32, td(t), GGML_TYPE_F32, s.st); ++s.n_kernels; }
    }
    s.lazy.erase(it);
    note_write(s, t);
}
// before node i runs outside the lazy-aware matchers: whatever it reads (through view chains) must exist, and nothing lazy may outlive its source
void lazy_net(exec_state & s, int i) {
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
    // a copy made real here was QUEUED (copy_queue): node i is about to launch a kernel that reads it -- or that writes over its source (the deadline) -- unless it is itself a
    // plain copy, whose own job is checked against the queue
    if (n->op != GGML_OP_CONT && n->op != GGML_OP_CONCAT && n->op != GGML_OP_CPY && n->op != GGML_OP_DUP) copy_flush(s);
}
bool lazy_try_register(exec_state & s, int i) {
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
bool exec_causal_conv(exec_state & s, int i) {
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
    copy_flush(s);                                                      // (copies met but not launched yet: this pattern's kernels read what they write)
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
bool exec_concat_tail(exec_state & s, int i) {
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
    if (s.pr.A || s.prm.n || (s.pn.m && s.pn.m == x)) copy_flush(s);
    if (s.pr.A) materialise_reduce(s);
    if (s.prm.n) materialise_group(s);
    if (s.pn.m && s.pn.m == x) materialise_norm(s);
    {
        prof_scope ps(s, "cpy", 0);
        tdesc src = td(x);
        src.p = (char *) x->data + (size_t) (f0 - P) * x->nb[1]; src.ne[1] = keep;
        const copy_pair cp = { src, td(n3), 4 };
        if (copy_queue_on(s) && copy_batch_ok(cp.src, cp.dst, 4)) copy_queue(s, &cp, 1, cp.dst, n3);
        else { cpy_strided(src, GGML_TYPE_F32, td(n3), GGML_TYPE_F32, s.st); ++s.n_kernels; }
    }
    note_write(s, n3);
    for (int k : { n0 ? j1 : -1, j2, j3 }) if (k >= 0) { s.done[k] = 1; ++s.n_fused; }
    return true;
}

// The HiFT vocoder's 1-D convolutions over a T-fastest signal (token2wav-impl.cpp:5136-5235): IM2COL(F32) -> [CONT] -> MUL_MAT against the reshaped kernel ->
// REPEAT(bias) -> ADD, five launches around a [KW*Cin, T] matrix of tens of megabytes.  One conv1d_tc launch (t2w_ops.hip) reads x itself; the kernel transposed once to
// [KW*Cin][Cout] is a resident image.  `i` is the IM2COL; the nodes must be the launches right behind one another.
bool exec_conv1d_tc(exec_state & s, int i) {
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

} // namespace mi

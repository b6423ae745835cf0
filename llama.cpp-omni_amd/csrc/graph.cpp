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
//
// Three translation units since round 4: graph_plan.cpp (supports_op, kernel-family rules, scratch sizing), graph_exec.cpp (node executors + fusion matchers),
// this file (fingerprints, capture records, graph_compute / graph_optimize, context, options); graph_internal.hpp holds what they share.
#include "graph_internal.hpp"

namespace mi {

// ------------------------------------------------------------------------------------------------ fingerprint
static inline uint64_t mix(uint64_t h, uint64_t v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); return h; }
// Identity of a cgraph for hipGraph replay: ops, types, shapes, strides, op_params, flags and the data pointers of nodes and sources.
// Runs on the host once per graph_compute, in front of the launch (the device idles meanwhile: ~1200 nodes per decode step), so the
// ~100 k word mixes are spread over four independent accumulators -- the serial xor-shift-add chain of one accumulator was ~50 us per
// token (tools/host_overhead.py), a quarter of that with four.
bool g_fp_collide = false;                       // test hook (set_option "fp_collide"): every graph gets the same fingerprint
static uint64_t fingerprint(const ggml_cgraph * g) {
    if (g_fp_collide) return 42;
    uint64_t h[4] = { 0xcbf29ce484222325ull, 0x84222325cbf29ce4ull, 0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full };
    h[0] = mix(h[0], (uint64_t) g->n_nodes);
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        h[0] = mix(h[0], ((uint64_t) n->op << 32) | (uint64_t) (uint32_t) n->type); h[1] = mix(h[1], (uint64_t) (uintptr_t) n->data); h[2] = mix(h[2], (uint64_t) (uint32_t) n->flags);
        for (int d = 0; d < 4; ++d) { h[d] = mix(h[d], (uint64_t) n->ne[d]); h[(d + 1) & 3] = mix(h[(d + 1) & 3], (uint64_t) n->nb[d]); }
        for (int p = 0; p < GGML_MAX_OP_PARAMS / 4; p += 2)                                  // two 32-bit words per mix
            h[(p >> 1) & 3] = mix(h[(p >> 1) & 3], ((uint64_t) (uint32_t) n->op_params[p] << 32) | (uint64_t) (uint32_t) n->op_params[p + 1]);
        for (int k = 0; k < GGML_MAX_SRC; ++k) {
            const ggml_tensor * sr = n->src[k];
            if (!sr) { h[k & 3] = mix(h[k & 3], 0x5bd1e995u + k); continue; }
            h[k & 3] = mix(h[k & 3], (uint64_t) (uintptr_t) sr->data); h[(k + 1) & 3] = mix(h[(k + 1) & 3], (uint64_t) sr->type);
            for (int d = 0; d < 4; ++d) { h[d] = mix(h[d], (uint64_t) sr->ne[d]); h[(d + 2) & 3] = mix(h[(d + 2) & 3], (uint64_t) sr->nb[d]); }
        }
    }
    return mix(mix(mix(h[0], h[1]), h[2]), h[3]);
}

// ------------------------------------------------------------------------------------------------ capture records
// whole-graph use count of t (ggml_hash_find, ggml-impl.h:257-270: pointer >> 4, linear probing), -1 when the graph carries none
static inline int whole_use_count(const ggml_cgraph * g, const ggml_tensor * t) {
    const ggml_hash_set & hs = g->visited_hash_set;
    const size_t h = ((size_t) (uintptr_t) t >> 4) % hs.size;
    size_t i = h;
    while ((hs.used[i >> 5] >> (i & 31)) & 1u) {
        if (hs.keys[i] == t) return g->use_counts[i];
        i = (i + 1) % hs.size;
        if (i == h) break;
    }
    return -1;
}
static inline bool graph_has_use_counts(const ggml_cgraph * g) { return g->use_counts && g->visited_hash_set.size > 0 && g->visited_hash_set.keys && g->visited_hash_set.used; }
static void make_recs(backend_ctx * c, const ggml_cgraph * g, graph_exec & e) {
    e.recs.assign((size_t) g->n_nodes, node_rec());
    std::unordered_map<const ggml_tensor *, int> direct;
    direct.reserve((size_t) g->n_nodes * 2);
    for (int i = 0; i < g->n_nodes; ++i)
        for (int k = 0; k < GGML_MAX_SRC; ++k) if (g->nodes[i]->src[k]) ++direct[g->nodes[i]->src[k]];
    e.with_ext = c->opt_fusion && graph_has_use_counts(g);          // (the condition under which run_nodes derives `external`)
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        node_rec & r = e.recs[(size_t) i];
        memset(&r, 0, sizeof r);
        r.data = n->data; r.op = (uint32_t) n->op; r.type = (uint32_t) n->type; r.flags = n->flags;
        for (int d = 0; d < 4; ++d) { r.ne[d] = n->ne[d]; r.nb[d] = n->nb[d]; }
        memcpy(r.op_params, n->op_params, sizeof r.op_params);
        for (int k = 0; k < GGML_MAX_SRC; ++k) r.src[k] = n->src[k] ? n->src[k]->data : nullptr;
        auto it = direct.find(n);
        r.direct = it == direct.end() ? 0 : it->second;
        r.ext = e.with_ext ? (whole_use_count(g, n) > r.direct ? 1 : 0) : 0;
    }
}
// does the submitted cgraph equal what capture e was built from?  (the hot path of a decode step: ~1200 nodes, a handful of compares each)
static bool recs_match(backend_ctx * c, const ggml_cgraph * g, const graph_exec & e) {
    if ((size_t) g->n_nodes != e.recs.size()) return false;
    if (e.with_ext != (c->opt_fusion && graph_has_use_counts(g))) return false;
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        const node_rec & r = e.recs[(size_t) i];
        if (r.data != n->data || r.op != (uint32_t) n->op || r.type != (uint32_t) n->type || r.flags != n->flags) return false;
        if (memcmp(r.ne, n->ne, sizeof r.ne) != 0 || memcmp(r.nb, n->nb, sizeof r.nb) != 0 || memcmp(r.op_params, n->op_params, sizeof r.op_params) != 0) return false;
        for (int k = 0; k < GGML_MAX_SRC; ++k) if (r.src[k] != (n->src[k] ? n->src[k]->data : nullptr)) return false;
    }
    // (the source SHAPES need no compare of their own: a source is a node of this graph -- compared above -- or a leaf, whose shape cannot change
    //  under an unchanged node that reads it; the reference compares the same set of fields)
    if (e.with_ext)
        for (int i = 0; i < g->n_nodes; ++i) {
            const node_rec & r = e.recs[(size_t) i];
            if ((whole_use_count(g, g->nodes[i]) > r.direct ? 1 : 0) != r.ext) return false;
        }
    return true;
}

// ------------------------------------------------------------------------------------------------ graph_compute
// MI355X_DUMP_GRAPH=<file>: every submitted cgraph is appended to <file>, one line per node (op id, sub-op, type and shape of the result and of each source) --
// how tools/graph_diff.py compares what the reference's graph builders submit with what this repo's Python mirrors build (debug facility; off by default)
static void dump_graph(const ggml_cgraph * g, const char * path) {
    static long serial = 0;
    FILE * f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "graph %ld nodes %d\n", serial++, g->n_nodes);
    static const bool detail = getenv("MI355X_DUMP_GRAPH_DETAIL") != nullptr;       // + per source: its node index in this graph (-1: leaf) and whether it is contiguous
    std::unordered_map<const ggml_tensor *, int> idx;
    if (detail) for (int i = 0; i < g->n_nodes; ++i) idx[g->nodes[i]] = i;
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        if (detail) {
            fprintf(f, "%d: op %d contig %d flags %d", i, (int) n->op, (int) is_contiguous(n), n->flags);
            for (int k = 0; k < GGML_MAX_SRC && n->src[k]; ++k) { auto it = idx.find(n->src[k]); fprintf(f, " src %d contig %d op %d", it == idx.end() ? -1 : it->second, (int) is_contiguous(n->src[k]), (int) n->src[k]->op); }
            fprintf(f, " ;; ");
        }
        fprintf(f, "%d %d t%d [%lld,%lld,%lld,%lld]", (int) n->op, (n->op == GGML_OP_UNARY || n->op == GGML_OP_GLU) ? n->op_params[0] : -1, (int) n->type,
                (long long) n->ne[0], (long long) n->ne[1], (long long) n->ne[2], (long long) n->ne[3]);
        for (int k = 0; k < GGML_MAX_SRC && n->src[k]; ++k)
            fprintf(f, " | t%d [%lld,%lld,%lld,%lld]", (int) n->src[k]->type, (long long) n->src[k]->ne[0], (long long) n->src[k]->ne[1], (long long) n->src[k]->ne[2], (long long) n->src[k]->ne[3]);
        fputc('\n', f);
    }
    fclose(f);
}
enum ggml_status graph_compute(backend_ctx * c, ggml_cgraph * g) {
    if (g->n_nodes == 0) return GGML_STATUS_SUCCESS;
    static const char * dump_path = getenv("MI355X_DUMP_GRAPH");
    if (dump_path) dump_graph(g, dump_path);
    shadow_reader image_hold(c->device);                           // weight images looked up / baked into captures below stay alive until the launches are enqueued (shadow.hpp)
    struct hold_guard { backend_ctx * c; ~hold_guard() { c->shadow_hold = nullptr; } } hg{ c };
    c->shadow_hold = &image_hold;
    // Replay fast path: a graph that was captured before is launched straight from its fingerprint -- the five scratch-size passes and
    // the eligibility scans below are per-node host work in front of the launch, with the device idle (decode: ~1200 nodes).  Safe
    // because a capture exists only for an eligible graph whose scratch was sized, and growing any scratch block drops every capture.
    // Identity is the per-node record of the capture compared field by field (recs_match), most recently used capture first; no hash on this path.
    uint64_t fp = 0; bool have_fp = false;
    if (c->opt_graphs && !c->opt_profile && !c->execs.empty()) {
        graph_exec * order[8]; int no = 0;
        for (auto & e : c->execs) if (e.exec && e.shadow_gen == shadow_generation() && no < 8) order[no++] = &e;
        for (int a = 1; a < no; ++a) for (int b = a; b > 0 && order[b]->last_use > order[b - 1]->last_use; --b) std::swap(order[b], order[b - 1]);
        for (int a = 0; a < no; ++a) {
            const auto t0 = std::chrono::steady_clock::now();
            const bool hit = recs_match(c, g, *order[a]);
            const auto t1 = std::chrono::steady_clock::now();
            c->host_ns_match += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
            if (hit) {
                graph_exec & e = *order[a];
                e.last_use = ++c->tick; e.seen++;
                HIP_CHECK(hipGraphLaunch(e.exec, c->stream));
                c->host_ns_launch += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t1).count();
                c->stat_replays++; c->stat_kernels_last = e.n_kernels;
                return GGML_STATUS_SUCCESS;
            }
        }
    }
    bool any_gemm_cols = false;                                   // a prefill graph: the GEMM that fuses SWIGLU writes its f16 result image next to its input image
    for (int i = 0; i < g->n_nodes && !any_gemm_cols; ++i) any_gemm_cols = g->nodes[i]->op == GGML_OP_GLU && g->nodes[i]->ne[1] > MI_MMVQ_MAX_COLS;
    if (!ensure_scratch(c, &c->act_scratch, &c->act_scratch_bytes, graph_act_scratch_need(g)) ||
        !ensure_scratch(c, &c->act_scratch_alt, &c->act_scratch_alt_bytes, any_gemm_cols ? graph_act_scratch_need(g) : 0) ||
        !ensure_scratch(c, &c->w_scratch, &c->w_scratch_bytes, graph_w_scratch_need(g)) ||
        !ensure_scratch(c, &c->fa_scratch, &c->fa_scratch_bytes, graph_fa_scratch_need(g)) ||
        !ensure_scratch(c, &c->rope_scratch, &c->rope_scratch_bytes, graph_rope_scratch_need(g)) ||
        !ensure_scratch(c, &c->gemm_partial, &c->gemm_partial_bytes, graph_gemm_partial_need(g))) {
        log_msg(GGML_LOG_LEVEL_ERROR, "[mi355x] graph_compute: out of device memory for scratch buffers\n");
        return GGML_STATUS_ALLOC_FAILED;
    }

    int n_real = 0;
    for (int i = 0; i < g->n_nodes; ++i) n_real += !is_noop(g->nodes[i]);

    // hipGraph replay pays for launch-bound graphs (decode: hundreds of ~5 us kernels).  A prefill ubatch runs 50-100 us kernels, the
    // host stays far ahead of the device when it simply enqueues them, and capture + instantiation would cost a submission several ms
    // ... but a graph of many SMALL wide mat-muls is launch-bound again (a Token2Wav DiT block: 77 launches of ~5 us; an omni encoder layer): those
    // replay too, judged by their total mat-mul work
    int64_t max_cols = 0; double mm_flops = 0.0;
    for (int i = 0; i < g->n_nodes; ++i)
        if (g->nodes[i]->op == GGML_OP_MUL_MAT) {
            const ggml_tensor * w = g->nodes[i]->src[0], * x = g->nodes[i]->src[1];
            if (x->ne[1] > max_cols) max_cols = x->ne[1];
            mm_flops += 2.0 * (double) w->ne[0] * (double) w->ne[1] * (double) x->ne[1] * (double) (x->ne[2] * x->ne[3]);
        }
    static const int64_t graph_max_cols = getenv("MI355X_GRAPH_MAX_COLS") ? atoll(getenv("MI355X_GRAPH_MAX_COLS")) : (mmq_max_cols() > 32 ? mmq_max_cols() : 32);
    static const double graph_max_flops = getenv("MI355X_GRAPH_MAX_GFLOP") ? atof(getenv("MI355X_GRAPH_MAX_GFLOP")) * 1e9 : 20e9;
    // ... and so is a very long graph of small products whatever their total (the reference's Token2Wav window: 10 flow-matching steps x 16 DiT blocks at ~200 columns,
    // 15 000 launches, ~13 MFLOP of mat-mul per node): eager submission is bound by the host's ~3.4 us per launch
    static const double graph_max_flops_per_node = getenv("MI355X_GRAPH_MAX_MFLOP_PER_NODE") ? atof(getenv("MI355X_GRAPH_MAX_MFLOP_PER_NODE")) * 1e6 : 50e6;
    const bool try_graph = c->opt_graphs && !c->opt_profile && n_real >= 8 &&
                           (max_cols <= graph_max_cols || mm_flops <= graph_max_flops || (n_real >= 512 && mm_flops <= (double) n_real * graph_max_flops_per_node));
    if (try_graph) {
        if (!have_fp) fp = fingerprint(g);
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
        if (ge->exec && ge->shadow_gen != shadow_generation()) {    // a weight image baked into this capture was dropped
            HIP_CHECK(hipGraphExecDestroy(ge->exec)); HIP_CHECK(hipGraphDestroy(ge->graph));
            ge->exec = nullptr; ge->graph = nullptr; ge->seen = 1;  // run eagerly once (rebuilds the images), capture next time
        }
        if (ge->exec && !recs_match(c, g, *ge)) {                  // same fingerprint, different graph (a collision, or other readers outside the cgraph): never replay it
            HIP_CHECK(hipGraphExecDestroy(ge->exec)); HIP_CHECK(hipGraphDestroy(ge->graph));
            ge->exec = nullptr; ge->graph = nullptr; ge->seen = 1; ge->recs.clear();
            c->stat_fp_mismatch++;
        }
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
                    ge->graph = graph; ge->exec = ex; ge->n_kernels = (int) s.n_kernels; ge->shadow_gen = shadow_generation();
                    make_recs(c, g, *ge);
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
    const auto t_run = std::chrono::steady_clock::now();
    run_nodes(s, g);
    // MI355X_GRAPH_SLICE="nodes:lo:hi[,lo:hi...]": behind the eager run of a graph of exactly `nodes` nodes, each node range [lo, hi) is captured on its own and replayed
    // 20 times between two events -- the in-graph device time of a block / a chain / one kernel (rocprofv3 cannot trace replayed graphs here).  The replays recompute
    // the range on the data the eager run left: harmless for timing, so only for a throw-away run.
    if (const char * sl = getenv("MI355X_GRAPH_SLICE")) {
        static bool done_once = false;
        int nodes = atoi(sl);
        if (!done_once && nodes == g->n_nodes) {
            done_once = true;
            HIP_CHECK(hipStreamSynchronize(c->stream));
            for (const char * p = strchr(sl, ':'); p; p = strchr(p, ',')) {
                ++p;
                int lo = 0, hi = 0;
                if (sscanf(p, "%d:%d", &lo, &hi) != 2 || lo < 0 || hi <= lo || hi > g->n_nodes) break;
                exec_state s2; s2.c = c; s2.st = c->stream; s2.capturing = true; s2.node_lo = lo; s2.node_hi = hi;
                HIP_CHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                for (int rep = 0; rep < 10; ++rep) run_nodes(s2, g);                 // ten copies of the range per replay: the ~10 us floor of a hipGraphLaunch amortised
                hipGraph_t graph = nullptr; hipGraphExec_t ex = nullptr;
                HIP_CHECK(hipStreamEndCapture(c->stream, &graph));
                HIP_CHECK(hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0));
                hipEvent_t e0, e1; HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
                for (int r = 0; r < 3; ++r) HIP_CHECK(hipGraphLaunch(ex, c->stream));
                HIP_CHECK(hipEventRecord(e0, c->stream));
                for (int r = 0; r < 20; ++r) HIP_CHECK(hipGraphLaunch(ex, c->stream));
                HIP_CHECK(hipEventRecord(e1, c->stream));
                HIP_CHECK(hipStreamSynchronize(c->stream));
                float ms = 0.0f; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                log_msg(GGML_LOG_LEVEL_INFO, "[mi355x] slice [%d, %d): %ld launches, %.2f us per pass (%.2f us per launch)\n", lo, hi, s2.n_kernels / 10, ms * 1e3 / 200, s2.n_kernels ? ms * 1e3 / 20 / s2.n_kernels : 0.0);
                (void) hipEventDestroy(e0); (void) hipEventDestroy(e1); (void) hipGraphExecDestroy(ex); (void) hipGraphDestroy(graph);
                if (!strchr(p, ',')) break;
            }
        }
    }
    c->host_ns_eager_run += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_run).count(); c->n_eager_kernels += s.n_kernels;
    c->stat_eager++; c->stat_kernels_last = s.n_kernels;
    if (c->opt_profile) prof_drain(c);
    return GGML_STATUS_SUCCESS;
}

// ggml_backend_i.graph_optimize: called by ggml_backend_sched on each split BEFORE ggml-alloc assigns addresses
// (ggml-backend.cpp:1304-1310), so a re-ordering here also shapes the tensor lifetimes the allocator sees.
// libllama emits the q / k / v projections interleaved with the q and k chains (wq, reshape, norm, mul, rope, wk, ...): by the
// time wk runs, ggml-alloc has already recycled the q chain's dead buffers for it, and an executor-side hoist of wk next to wq
// must be refused (its output aliases memory the q chain still writes).  Moving the sibling mat-muls -- same activation, weight
// operands -- directly behind the first one BEFORE allocation removes the aliasing and lets exec_mul_mat batch them, and leaves
// the q chain, k chain and v store adjacent for the single norm_rope launch.  A moved node depends only on a weight and on the
// shared activation, produces a fresh (non-view) tensor, and nothing between its old and new position can consume it, so every
// topological and memory dependency is preserved.
void graph_optimize(backend_ctx *, ggml_cgraph * g) {
    static const bool off = getenv("MI355X_NO_GRAPH_OPTIMIZE") != nullptr;
    if (off || !g || g->n_nodes < 3) return;
    auto is_weight = [](const ggml_tensor * t) { return t && t->op == GGML_OP_NONE && t->view_src == nullptr && t->buffer != nullptr; };
    auto root_of = [](const ggml_tensor * t) { while (t->view_src) t = t->view_src; return t; };
    for (int i = 0; i < g->n_nodes; ++i) {
        ggml_tensor * a = g->nodes[i];
        if (a->op != GGML_OP_MUL_MAT || a->view_src || !is_weight(a->src[0])) continue;
        int at = i + 1;                                                  // next free slot behind the group
        for (int j = i + 1; j < g->n_nodes && j < i + 64; ++j) {
            ggml_tensor * c = g->nodes[j];
            if (c->op != GGML_OP_MUL_MAT || c->view_src || c->src[1] != a->src[1] || !is_weight(c->src[0])) continue;
            // (an in-place op on the shared activation between the two would make the later mat-mul see different data: leave it)
            bool inplace_between = false;
            for (int k = at; k < j && !inplace_between; ++k) {
                const ggml_tensor * mnode = g->nodes[k];
                if (is_noop(mnode)) continue;
                for (const ggml_tensor * r = mnode->view_src; r; r = r->view_src) if (r == root_of(a->src[1])) { inplace_between = true; break; }
            }
            if (inplace_between) break;
            if (j != at) {                                               // rotate nodes[at .. j] right by one
                for (int k = j; k > at; --k) g->nodes[k] = g->nodes[k - 1];
                g->nodes[at] = c;
            }
            ++at;
        }
        i = at - 1;
    }
}

void backend_ctx_init(backend_ctx * c) {
    const char * e;
    if ((e = getenv("MI355X_GRAPHS")))  c->opt_graphs  = atoi(e) != 0;
    if ((e = getenv("MI355X_FUSION")))  c->opt_fusion  = atoi(e) != 0;
    if ((e = getenv("MI355X_PROFILE"))) c->opt_profile = atoi(e) != 0;
    if ((e = getenv("MI355X_NORM_IN_KERNEL"))) c->opt_norm_in_kernel = atoi(e) != 0;
    if ((e = getenv("MI355X_MV1")))     c->opt_mv1     = atoi(e) != 0;
    if ((e = getenv("MI355X_KQ_STAGING"))) c->opt_kq_staging = atoi(e) != 0;
}
void backend_ctx_release(backend_ctx * c) {
    drop_graph_execs(c);
    for (auto ev : c->prof_event_pool) (void) hipEventDestroy(ev);
    if (c->act_scratch) (void) hipFree(c->act_scratch);
    if (c->act_scratch_alt) (void) hipFree(c->act_scratch_alt);
    if (c->w_scratch) (void) hipFree(c->w_scratch);
    if (c->fa_scratch) (void) hipFree(c->fa_scratch);
    if (c->fa_counters) (void) hipFree(c->fa_counters);
    if (c->rope_scratch) (void) hipFree(c->rope_scratch);
    if (c->gemm_partial) (void) hipFree(c->gemm_partial);
    if (c->copy_event) (void) hipEventDestroy(c->copy_event);
    if (c->handoff_event) (void) hipEventDestroy(c->handoff_event);
    if (c->stream) (void) hipStreamDestroy(c->stream);
}

} // namespace mi

extern "C" {
int mi355x_set_option(struct ggml_backend * backend, const char * key, long value) {
    mi::backend_ctx * c = (mi::backend_ctx *) backend->context;
    if (!strcmp(key, "graphs"))  { c->opt_graphs = value != 0; return 0; }
    if (!strcmp(key, "fusion"))  { c->opt_fusion = value != 0; mi::drop_graph_execs(c); return 0; }
    if (!strcmp(key, "profile")) { c->opt_profile = value != 0; return 0; }
    if (!strcmp(key, "norm_in_kernel")) { c->opt_norm_in_kernel = value != 0; mi::drop_graph_execs(c); return 0; }
    if (!strcmp(key, "mv1")) { c->opt_mv1 = value != 0; mi::drop_graph_execs(c); return 0; }
    if (!strcmp(key, "batch_uploads")) { flush_uploads(c); c->opt_batch_uploads = value != 0; return 0; }
    if (!strcmp(key, "fp_collide")) { mi::g_fp_collide = value != 0; return 0; }
    if (!strcmp(key, "prefill_q8k")) { mi::prefill_q8k_set_mode((int) value); mi::drop_graph_execs(c); return 0; }      // (process-wide: prefill GEMMs of K-quant weights take the Q8_K-quantised activations; 0 = plain f16 rows, the round-4 arithmetic)
    if (!strcmp(key, "mmq_tile")) { mi::mmq_tile_set_mode((int) value); mi::drop_graph_execs(c); return 0; }      // (process-wide: the tiled int8-MFMA prefill kernel for Q4_K weights, mmq_tile.hip; 0 = the F16-image GEMM)
    if (!strcmp(key, "mv2")) { mi::mmv2_enable(value != 0); mi::drop_graph_execs(c); return 0; }       // (process-wide: the LDS-DMA engine form of the decode mat-vec)
    if (!strcmp(key, "kq_staging")) { c->opt_kq_staging = value != 0; mi::drop_graph_execs(c); return 0; }
    if (!strcmp(key, "fattn_gqa")) { mi::fattn_set_gqa(value != 0); mi::drop_graph_execs(c); return 0; }
    if (!strcmp(key, "fattn_one")) { mi::fattn_set_one(value != 0); mi::drop_graph_execs(c); return 0; }
    if (!strcmp(key, "copy_batch")) { c->opt_copy_batch = (int) value; mi::drop_graph_execs(c); return 0; }
    if (!strcmp(key, "fattn_gs")) { mi::fattn_set_gs((int) value); mi::drop_graph_execs(c); return 0; }       // (process-wide)
    if (!strcmp(key, "fattn_dma")) { mi::fattn_set_dma((int) value); mi::drop_graph_execs(c); return 0; }      // (process-wide)
    if (!strcmp(key, "f16_shadow")) { mi::shadow_set_enabled(value != 0); mi::drop_graph_execs(c); return 0; }
    if (!strcmp(key, "reset_stats")) { c->prof.clear(); c->stat_replays = c->stat_captures = c->stat_eager = 0; return 0; }
    return -1;
}
double mi355x_get_stat(struct ggml_backend * backend, const char * key) {
    mi::backend_ctx * c = (mi::backend_ctx *) backend->context;
    if (!strcmp(key, "graph_replays"))      return (double) c->stat_replays;
    if (!strcmp(key, "graph_fp_mismatch"))  return (double) c->stat_fp_mismatch;      // captures dropped because the graph differed under an equal fingerprint
    if (!strcmp(key, "graph_captures"))     return (double) c->stat_captures;
    if (!strcmp(key, "eager_graphs"))       return (double) c->stat_eager;
    if (!strcmp(key, "kernels_last_graph")) return (double) c->stat_kernels_last;
    if (!strcmp(key, "lazy_conts"))         return (double) c->stat_lazy_taken;
    if (!strcmp(key, "copies_batched"))     return (double) c->stat_copies_batched;
    if (!strcmp(key, "copies_forwarded"))   return (double) c->stat_copies_forwarded;
    if (!strcmp(key, "copies_dropped"))     return (double) c->stat_copies_dropped;
    if (!strcmp(key, "lazy_conts_materialised")) return (double) c->stat_lazy_materialised;
    if (!strncmp(key, "shadow_", 7))        return mi::shadow_stat(key);
    if (!strcmp(key, "gemm256_launches"))   return (double) mi::gemm_variant_launches(0);
    if (!strcmp(key, "gemm192_launches"))   return (double) mi::gemm_variant_launches(1);
    if (!strcmp(key, "gemm_glu_launches"))  return (double) mi::gemm_variant_launches(2);
    if (!strcmp(key, "gemm_glu96_launches")) return (double) mi::gemm_variant_launches(5);
    if (!strcmp(key, "norm_from_split_launches")) return (double) mi::norm_from_split_launches();
    if (!strcmp(key, "norm_rope_split_launches")) return (double) mi::norm_rope_split_launches();
    if (!strcmp(key, "mmq_tile_launches"))  return (double) mi::mmq_tile_launches();
    if (!strcmp(key, "gemm_kq_launches"))   return (double) mi::gemm_variant_launches(3);
    if (!strcmp(key, "fattn_dma_launches")) return (double) mi::fattn_dma_launches();
    if (!strcmp(key, "fattn_gs_launches"))  return (double) mi::fattn_gs_launches();
    if (!strcmp(key, "fattn_gs_far_launches")) return (double) mi::fattn_gs_far_launches();
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


// graph.hpp -- per-stream backend state and the graph executor interface.
#pragma once
#include "common.hpp"
#include <map>
#include <string>
#include <vector>

namespace mi {

void log_msg(int level, const char * fmt, ...);

struct prof_class { double us = 0; double bytes = 0; long n = 0; };

// what a captured hipGraph was built from, node by node (the reference keeps the same record for its CUDA graphs, ggml-cuda.cu:2721-2796 and
// compares it before every launch): a replay is only issued when EVERY field of EVERY node matches -- a 64-bit fingerprint alone would replay
// the wrong capture silently on a collision -- and when the same nodes have readers outside the cgraph (`ext`: the fusion decisions baked
// into the capture depend on it, and ggml_cgraph::use_counts can change while the nodes stay the same)
struct node_rec {
    const void * data; uint32_t op, type; int32_t flags; int32_t ext;
    int64_t ne[4]; size_t nb[4]; int32_t op_params[GGML_MAX_OP_PARAMS / 4]; const void * src[GGML_MAX_SRC]; int32_t direct, pad;
};
struct graph_exec {                    // one captured cgraph
    uint64_t        fingerprint = 0;   // (bookkeeping of graphs seen but not yet captured; identity of a capture is `recs`)
    int             seen = 0;          // times this fingerprint was submitted
    hipGraph_t      graph = nullptr;
    hipGraphExec_t  exec = nullptr;
    int             n_kernels = 0;
    uint64_t        last_use = 0;
    uint64_t        shadow_gen = 0;    // shadow_generation() at capture time
    bool            with_ext = false;  // recs[].ext was derived from use_counts (the graph carried them and fusion was on)
    std::vector<node_rec> recs;
};

struct backend_ctx {
    int          device = 0;
    std::string  name;
    hipStream_t  stream = nullptr;
    hipEvent_t   copy_event = nullptr;
    hipEvent_t   handoff_event = nullptr;   // mi355x_handoff's own event (cpy_tensor_async uses copy_event on the same stream)

    // scratch for quantised / converted activations (grown outside of graph capture only)
    void *  act_scratch = nullptr;  size_t act_scratch_bytes = 0;
    void *  act_scratch_alt = nullptr;  size_t act_scratch_alt_bytes = 0;      // second image buffer: a GEMM that emits the next GEMM's f16 activation image (SWIGLU epilogue) swaps the two
    // scratch for de-quantised weight tiles / f16 copies on the GEMM path
    void *  w_scratch = nullptr;    size_t w_scratch_bytes = 0;
    // mask tile map of the prefill flash-attention kernel
    void *  fa_scratch = nullptr;   size_t fa_scratch_bytes = 0;
    unsigned * fa_counters = nullptr;   // [1024] arrival counters of the sliced one-token attention (zero between launches: the last arriver resets its own)
    // (cos, sin) table of a prefill ubatch's rotary positions (fused.hip k_rope_table)
    void *  rope_scratch = nullptr; size_t rope_scratch_bytes = 0;
    // split-K partial sums of the prefill GEMM
    void *  gemm_partial = nullptr; size_t gemm_partial_bytes = 0;

    // small host -> device uploads (the per-token inputs of a decode step: embedding row, position, cache indices, mask row) are staged in
    // pinned memory and written by ONE launch in front of the next piece of stream work instead of one ~4 us copy each (backend.cpp)
    struct up_ent { void * dst; uint32_t off, size; };
    char *   up_host[2] = { nullptr, nullptr };      // pinned staging halves (device-visible)
    up_ent * up_ents[2] = { nullptr, nullptr };      // pinned descriptor tables
    hipEvent_t up_done[2] = { nullptr, nullptr };    // the half's last flush has run
    int      up_half = 0, up_n = 0; size_t up_used = 0;
    bool     opt_batch_uploads = true;

    struct shadow_reader * shadow_hold = nullptr;   // graph_compute's reader hold on the device's weight-image table (shadow.hpp), for the out-of-memory path that drops the images

    // options
    bool opt_graphs = true;
    bool opt_fusion = true;
    bool opt_profile = false;
    bool opt_norm_in_kernel = false;   // RMS_NORM+MUL computed inside the consuming mat-vec launches (mmvk.hip act_norm)
    bool opt_kq_staging = false;       // K-quant blocks de-quantised inside the GEMM's staging even when an F16 image could be used (<= 256 columns; measurement / tests)
    bool opt_mv1 = true;               // batch-1 decode mat-vecs on mmv1.hip (f32 activation in, image built in the prologue)

    // hipGraph cache
    std::vector<graph_exec> execs;
    uint64_t tick = 0;

    // statistics
    long stat_replays = 0, stat_captures = 0, stat_eager = 0, stat_kernels_last = 0, stat_fp_mismatch = 0;
    int  opt_copy_batch = -1;                                  // deferred layout copies (graph_exec.cpp copy_queue): -1 default (on), 0 off, 1 on
    long stat_copies_batched = 0, stat_copies_forwarded = 0, stat_copies_dropped = 0;   // copies that left in a batch of >= 2 / nodes forwarded to a pending node's sources / jobs never launched
    long stat_lazy_taken = 0, stat_lazy_materialised = 0;      // lazy CONTs (graph_exec_t2w.cpp lazy_try_register): left un-run / made real after all
    // host time spent inside the backend's entry points (ns; reported with MI355X_LOG_STATS): graph_compute, set/get_tensor_async, synchronize
    uint64_t host_ns_match = 0, host_ns_launch = 0;      // replay fast path: the record compare, hipGraphLaunch
    uint64_t host_ns_graph = 0, host_ns_set = 0, host_ns_get = 0, host_ns_sync = 0, host_ns_eager_run = 0; long n_eager_kernels = 0; long n_set = 0, n_get = 0, n_sync = 0, n_graph = 0;
    std::map<std::string, prof_class> prof;
    struct pending_prof { std::string cls; double bytes; hipEvent_t a, b; };
    std::vector<pending_prof> prof_pending;
    std::vector<hipEvent_t>   prof_event_pool;
};

void backend_ctx_init(backend_ctx * c);
void backend_ctx_release(backend_ctx * c);
void flush_uploads(backend_ctx * c);          // (backend.cpp) staged small uploads -> the stream; every entry point that enqueues stream work calls it first
void drop_graph_execs(backend_ctx * c);      // destroy every captured hipGraph of the context (option changes, scratch re-allocation)

bool             supports_op(const ggml_tensor * op);
enum ggml_status graph_compute(backend_ctx * c, ggml_cgraph * g);
void             graph_optimize(backend_ctx * c, ggml_cgraph * g);   // node re-ordering before allocation (ggml_backend_i.graph_optimize)

bool backend_is_mi355x(const struct ggml_backend * b);      // (backend.cpp) is this one of OUR backends -- its context a backend_ctx?

} // namespace mi

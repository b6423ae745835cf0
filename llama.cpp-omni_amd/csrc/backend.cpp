// backend.cpp -- the ggml plug-in boundary of the MI355X backend.
//
// Implements the five vtable structs of the reference's backend ABI (ggml/src/ggml-backend-impl.h:17-210)
// and exports `ggml_backend_init` / `ggml_backend_score`, the two symbols the reference's loader binds
// (ggml/src/ggml-backend-reg.cpp:257-285).  Semantics (ownership, error conventions, async behaviour) follow
// what the scheduler and allocator expect from a GPU backend; the behavioural template is the reference's
// own GPU backend host code (ggml-cuda.cu:552-757 buffers, :2541-2633 stream ops, :3271-3760 device/reg) --
// none of its code is reused: one process drives one or more gfx950 devices through the HIP runtime directly.
#include "common.hpp"
#include "ggml_util.hpp"
#include "kernels.hpp"
#include "graph.hpp"
#include <atomic>
#include <chrono>
#include "shadow.hpp"

#include <mutex>
#include <string>
#include <vector>
#include <stdarg.h>

#include "../../include/ggml-mi355x.h"

// ---------------------------------------------------------------------------------------------- host imports
// Resolved against the host process' libggml-base when the backend is loaded by the reference; absent
// (NULL) in the standalone harness, where the fallbacks below are used.
extern "C" {
__attribute__((weak)) ggml_backend_buffer_t ggml_backend_buffer_init(ggml_backend_buffer_type_t buft, struct ggml_backend_buffer_i iface, void * context, size_t size);
__attribute__((weak)) void ggml_log_internal(enum ggml_log_level level, const char * format, ...);
}

namespace mi {

void log_msg(int level, const char * fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (ggml_log_internal) ggml_log_internal((enum ggml_log_level) level, "%s", buf);
    else if (level >= GGML_LOG_LEVEL_WARN || getenv("MI355X_VERBOSE")) fputs(buf, stderr);
}

static ggml_backend_buffer_t make_buffer(ggml_backend_buffer_type_t buft, struct ggml_backend_buffer_i iface, void * ctx, size_t size) {
    if (ggml_backend_buffer_init) return ggml_backend_buffer_init(buft, iface, ctx, size);
    // standalone: same object the core would `new` (ggml-backend.cpp:85-99); freed by mi355x_host_buffer_free
    return new ggml_backend_buffer{iface, buft, ctx, size, GGML_BACKEND_BUFFER_USAGE_ANY};
}

// ---------------------------------------------------------------------------------------------- device table
struct device_ctx {
    int         index;
    std::string name;          // "MI355X<i>"
    std::string description;   // marketing name from the runtime
    std::string pci_id;        // dddd:bb:dd.f
    size_t      total_mem;
    ggml_backend_buffer_type  buft;
    ggml_backend_device       dev;
};

static std::vector<device_ctx *> g_devices;
static ggml_backend_reg          g_reg;
static ggml_backend_buffer_type  g_host_buft;
static std::mutex                g_mutex;
static bool                      g_init_done = false;

static void set_device(int d) { HIP_CHECK(hipSetDevice(d)); }

// ---------------------------------------------------------------------------------------------- device buffers
struct buffer_ctx { int device; void * base; size_t size; };

// ---- staged small uploads (be_set_async below) and the buffer-level entry points: a staged write sits in host memory until the owning backend's next entry point flushes it.
// The blocking buffer-level paths (set / get / memset / cpy / clear) and a buffer's release know nothing about backends, so they settle what is staged for the bytes
// they touch first: every live backend context of the device with a staged entry inside [lo, hi) is flushed and its stream drained.  (Before this, a compute buffer freed
// ahead of its backend was written by the flush in be_free, after the free; a blocking tensor_set issued behind a staged async one to the same bytes was overwritten by
// the older data at the next flush.)
void flush_uploads(backend_ctx * c);
static std::recursive_mutex g_live_mu;                 // (stage_upload re-enters itself after a flush)
static std::vector<backend_ctx *> g_live;
static void settle_staged(int device, const void * lo_, size_t n) {
    const char * lo = (const char *) lo_, * hi = lo + n;
    std::lock_guard<std::recursive_mutex> lk(g_live_mu);
    for (backend_ctx * c : g_live) {
        if (c->device != device || c->up_n == 0) continue;
        bool hit = false;
        for (int i = 0; i < c->up_n && !hit; ++i) {
            const backend_ctx::up_ent & e = c->up_ents[c->up_half][i];
            hit = (const char *) e.dst < hi && lo < (const char *) e.dst + e.size;
        }
        if (hit) {                                         // (the launch goes onto c's stream -- make c's device current for it, the caller's afterwards)
            int cur = 0; HIP_CHECK(hipGetDevice(&cur));
            if (cur != c->device) HIP_CHECK(hipSetDevice(c->device));
            flush_uploads(c); HIP_CHECK(hipStreamSynchronize(c->stream));
            if (cur != c->device) HIP_CHECK(hipSetDevice(cur));
        }
    }
}
static void buf_free(ggml_backend_buffer_t b) {
    buffer_ctx * c = (buffer_ctx *) b->context;
    set_device(c->device);
    settle_staged(c->device, c->base, c->size);
    shadow_invalidate(c->device, c->base, c->size);
    HIP_CHECK(hipFree(c->base));
    delete c;
}
static void * buf_get_base(ggml_backend_buffer_t b) { return ((buffer_ctx *) b->context)->base; }

static enum ggml_status buf_init_tensor(ggml_backend_buffer_t, struct ggml_tensor *) {
    // no per-tensor extras: kernels never read past the logical end of a row, so no row padding is needed
    // (the reference's GPU backend pads quantised rows to 512 elements instead, ggml-cuda.cu:604-622)
    return GGML_STATUS_SUCCESS;
}
static void buf_memset_tensor(ggml_backend_buffer_t b, struct ggml_tensor * t, uint8_t v, size_t off, size_t sz) {
    buffer_ctx * c = (buffer_ctx *) b->context; set_device(c->device);
    settle_staged(c->device, (char *) t->data + off, sz);
    shadow_invalidate(c->device, (char *) t->data + off, sz);
    HIP_CHECK(hipMemsetAsync((char *) t->data + off, v, sz, hipStreamPerThread));
    HIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
}
// blocking buffer-level transfers (ggml_backend_tensor_set / _get: model load, the omni encoders' inputs and outputs): bytes and host time, for the MI355X_LOG_STATS account
static std::atomic<uint64_t> g_buf_set_ns{0}, g_buf_set_bytes{0}, g_buf_set_n{0}, g_buf_get_ns{0}, g_buf_get_bytes{0}, g_buf_get_n{0};
struct buf_timer {
    std::atomic<uint64_t> & ns; std::chrono::steady_clock::time_point t0;
    buf_timer(std::atomic<uint64_t> & a, std::atomic<uint64_t> & bytes, std::atomic<uint64_t> & n, size_t sz) : ns(a), t0(std::chrono::steady_clock::now()) { bytes += sz; ++n; }
    ~buf_timer() { ns += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
static void buf_set_tensor(ggml_backend_buffer_t b, struct ggml_tensor * t, const void * data, size_t off, size_t sz) {
    buf_timer bt(g_buf_set_ns, g_buf_set_bytes, g_buf_set_n, sz);
    buffer_ctx * c = (buffer_ctx *) b->context; set_device(c->device);
    settle_staged(c->device, (char *) t->data + off, sz);
    shadow_invalidate(c->device, (char *) t->data + off, sz);
    HIP_CHECK(hipMemcpyAsync((char *) t->data + off, data, sz, hipMemcpyHostToDevice, hipStreamPerThread));
    HIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
}
static void buf_get_tensor(ggml_backend_buffer_t b, const struct ggml_tensor * t, void * data, size_t off, size_t sz) {
    buf_timer bt(g_buf_get_ns, g_buf_get_bytes, g_buf_get_n, sz);
    buffer_ctx * c = (buffer_ctx *) b->context; set_device(c->device);
    settle_staged(c->device, (const char *) t->data + off, sz);
    HIP_CHECK(hipMemcpyAsync(data, (const char *) t->data + off, sz, hipMemcpyDeviceToHost, hipStreamPerThread));
    HIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
}
static bool buffer_is_ours(ggml_backend_buffer_t b);
static bool buf_cpy_tensor(ggml_backend_buffer_t b, const struct ggml_tensor * src, struct ggml_tensor * dst) {
    ggml_backend_buffer_t sb = src->view_src ? src->view_src->buffer : src->buffer;
    if (!sb || !buffer_is_ours(sb)) return false;                    // "not handled": the core falls back to get+set
    buffer_ctx * sc = (buffer_ctx *) sb->context; buffer_ctx * dc = (buffer_ctx *) b->context;
    set_device(dc->device);
    const size_t n = nbytes(src);
    settle_staged(sc->device, src->data, n); settle_staged(dc->device, dst->data, n);
    shadow_invalidate(dc->device, dst->data, n);
    if (sc->device == dc->device) HIP_CHECK(hipMemcpyAsync(dst->data, src->data, n, hipMemcpyDeviceToDevice, hipStreamPerThread));
    else                          HIP_CHECK(hipMemcpyPeerAsync(dst->data, dc->device, src->data, sc->device, n, hipStreamPerThread));
    HIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
    return true;
}
static void buf_clear(ggml_backend_buffer_t b, uint8_t v) {
    buffer_ctx * c = (buffer_ctx *) b->context; set_device(c->device);
    settle_staged(c->device, c->base, c->size);
    shadow_invalidate(c->device, c->base, c->size);
    HIP_CHECK(hipMemsetAsync(c->base, v, c->size, hipStreamPerThread));
    HIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
}
static const ggml_backend_buffer_i k_buffer_iface = {
    buf_free, buf_get_base, buf_init_tensor, buf_memset_tensor, buf_set_tensor, buf_get_tensor, buf_cpy_tensor, buf_clear, /*reset*/ nullptr,
};
static bool buffer_is_ours(ggml_backend_buffer_t b) { return b->iface.free_buffer == buf_free; }

// ---------------------------------------------------------------------------------------------- buffer type (device)
static const char * buft_name(ggml_backend_buffer_type_t t) { return ((device_ctx *) t->context)->name.c_str(); }
static ggml_backend_buffer_t buft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    device_ctx * d = (device_ctx *) t->context;
    set_device(d->index);
    void * p = nullptr;
    const size_t asz = size < 1 ? 1 : size;
    hipError_t e = hipMalloc(&p, asz);
    if (e != hipSuccess && shadow_bytes(d->index) > 0) {              // the resident F16 weight images are droppable: give them back and retry
        (void) hipGetLastError();
        const size_t freed = shadow_drop_all(d->index);
        log_msg(GGML_LOG_LEVEL_WARN, "[mi355x] device %d short of memory: dropped %.1f MiB of F16 weight images\n", d->index, freed / 1048576.0);
        e = hipMalloc(&p, asz);
    }
    if (e != hipSuccess) {                                            // OOM is reported, not fatal (caller handles NULL)
        (void) hipGetLastError();
        log_msg(GGML_LOG_LEVEL_ERROR, "[mi355x] allocating %.2f MiB on device %d failed: %s\n", size / 1048576.0, d->index, hipGetErrorString(e));
        return nullptr;
    }
    return make_buffer(t, k_buffer_iface, new buffer_ctx{d->index, p, size}, size);
}
static size_t buft_alignment(ggml_backend_buffer_type_t) { return 128; }   // one HBM/L2 line; 16-B vector loads need 16
static bool   buft_is_host(ggml_backend_buffer_type_t) { return false; }
static const ggml_backend_buffer_type_i k_buft_iface = { buft_name, buft_alloc, buft_alignment, /*max_size*/ nullptr, /*alloc_size*/ nullptr, buft_is_host };

// ---------------------------------------------------------------------------------------------- pinned host buffers
static void hostbuf_free(ggml_backend_buffer_t b) { HIP_CHECK(hipHostFree(b->context)); }
static void * hostbuf_base(ggml_backend_buffer_t b) { return b->context; }
static void hostbuf_memset(ggml_backend_buffer_t, struct ggml_tensor * t, uint8_t v, size_t off, size_t sz) { memset((char *) t->data + off, v, sz); }
static void hostbuf_set(ggml_backend_buffer_t, struct ggml_tensor * t, const void * d, size_t off, size_t sz) { memcpy((char *) t->data + off, d, sz); }
static void hostbuf_get(ggml_backend_buffer_t, const struct ggml_tensor * t, void * d, size_t off, size_t sz) { memcpy(d, (const char *) t->data + off, sz); }
static void hostbuf_clear(ggml_backend_buffer_t b, uint8_t v) { memset(b->context, v, b->size); }
static const ggml_backend_buffer_i k_hostbuf_iface = { hostbuf_free, hostbuf_base, nullptr, hostbuf_memset, hostbuf_set, hostbuf_get, nullptr, hostbuf_clear, nullptr };
static const char * hostbuft_name(ggml_backend_buffer_type_t) { return "MI355X_Host"; }
static ggml_backend_buffer_t hostbuft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    void * p = nullptr;
    hipError_t e = hipHostMalloc(&p, size < 1 ? 1 : size, hipHostMallocDefault);
    if (e != hipSuccess) { (void) hipGetLastError(); return nullptr; }   // core falls back to plain CPU memory
    return make_buffer(t, k_hostbuf_iface, p, size);
}
static size_t hostbuft_alignment(ggml_backend_buffer_type_t) { return 64; }
static bool   hostbuft_is_host(ggml_backend_buffer_type_t) { return true; }
static const ggml_backend_buffer_type_i k_hostbuft_iface = { hostbuft_name, hostbuft_alloc, hostbuft_alignment, nullptr, nullptr, hostbuft_is_host };

// ---------------------------------------------------------------------------------------------- backend (stream)
static ggml_guid g_guid = { 0x4d, 0x49, 0x33, 0x35, 0x35, 0x58, 0x2d, 0x67, 0x66, 0x78, 0x39, 0x35, 0x30, 0x2d, 0x76, 0x31 };

// MI355X_GRAPH_GPU_TIME=1: an event pair around every graph on the backend's stream; the device time of each (nodes, ms) is printed when the backend is freed
// (how much of a caller's wall time per submission is the device -- the Token2Wav session builds and allocates a 27 000-node graph per window on the host)
struct graph_gpu_time { hipEvent_t a, b; int nodes; };
static std::vector<graph_gpu_time> g_graph_times;
static const char * be_name(ggml_backend_t b) { return ((backend_ctx *) b->context)->name.c_str(); }
static void be_free(ggml_backend_t b) {
    backend_ctx * c = (backend_ctx *) b->context;
    set_device(c->device);
    flush_uploads(c);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int h = 0; h < 2; ++h) { if (c->up_host[h]) (void) hipHostFree(c->up_host[h]); if (c->up_ents[h]) (void) hipHostFree(c->up_ents[h]); if (c->up_done[h]) (void) hipEventDestroy(c->up_done[h]); }
    std::vector<graph_gpu_time> times;
    { std::lock_guard<std::recursive_mutex> lk(g_live_mu); times.swap(g_graph_times); }      // (every backend's graph_compute pushes here: the first one freed drains the list)
    if (!times.empty()) {
        std::string line;
        for (const graph_gpu_time & t : times) { float ms = 0.0f; if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { char buf[64]; snprintf(buf, sizeof buf, " %d:%.2f", t.nodes, ms); line += buf; } (void) hipEventDestroy(t.a); (void) hipEventDestroy(t.b); }
        (void) hipGetLastError();
        log_msg(GGML_LOG_LEVEL_INFO, "[mi355x] %s: device time per graph (nodes:ms):%s\n", c->name.c_str(), line.c_str());
    }
    if (getenv("MI355X_LOG_STATS"))
        log_msg(GGML_LOG_LEVEL_INFO, "[mi355x] %s: graphs eager=%ld captured=%ld replayed=%ld, kernels in last graph=%ld\n", c->name.c_str(),
                c->stat_eager, c->stat_captures, c->stat_replays, c->stat_kernels_last);
    if (getenv("MI355X_LOG_STATS"))
        log_msg(GGML_LOG_LEVEL_INFO, "[mi355x] %s: host time inside the backend: graph_compute %ld calls %.1f us avg, set_tensor_async %ld calls %.2f us avg, get_tensor_async %ld calls %.2f us avg, synchronize %ld calls %.1f us avg (includes waiting for the device)\n",
                c->name.c_str(), c->n_graph, c->n_graph ? c->host_ns_graph * 1e-3 / c->n_graph : 0.0, c->n_set, c->n_set ? c->host_ns_set * 1e-3 / c->n_set : 0.0, c->n_get, c->n_get ? c->host_ns_get * 1e-3 / c->n_get : 0.0,
                c->n_sync, c->n_sync ? c->host_ns_sync * 1e-3 / c->n_sync : 0.0);
    if (getenv("MI355X_LOG_STATS"))
        log_msg(GGML_LOG_LEVEL_INFO, "[mi355x] blocking buffer transfers (process-wide): tensor_set %lu calls %.1f MB %.1f ms, tensor_get %lu calls %.1f MB %.1f ms\n",
                (unsigned long) g_buf_set_n.load(), g_buf_set_bytes.load() * 1e-6, g_buf_set_ns.load() * 1e-6, (unsigned long) g_buf_get_n.load(), g_buf_get_bytes.load() * 1e-6, g_buf_get_ns.load() * 1e-6);
    if (getenv("MI355X_LOG_STATS") && c->stat_replays)
        log_msg(GGML_LOG_LEVEL_INFO, "[mi355x] %s: replayed graphs: %.1f us avg comparing the node records, %.1f us avg in hipGraphLaunch (%ld replays)\n", c->name.c_str(),
                c->host_ns_match * 1e-3 / c->stat_replays, c->host_ns_launch * 1e-3 / c->stat_replays, c->stat_replays);
    if (getenv("MI355X_LOG_STATS") && c->stat_eager)
        log_msg(GGML_LOG_LEVEL_INFO, "[mi355x] %s: eager graphs: %.1f us avg walking the nodes and enqueueing (%ld launches, %.2f us per launch)\n", c->name.c_str(),
                c->host_ns_eager_run * 1e-3 / c->stat_eager, c->n_eager_kernels, c->n_eager_kernels ? c->host_ns_eager_run * 1e-3 / c->n_eager_kernels : 0.0);
    if (getenv("MI355X_LOG_STATS"))
        for (auto & kv : c->prof)                                         // per-class event timing (only filled in "profile" mode)
            log_msg(GGML_LOG_LEVEL_INFO, "[mi355x]   %-22s n=%8ld  total %10.1f us  avg %8.2f us\n", kv.first.c_str(), kv.second.n, kv.second.us, kv.second.n ? kv.second.us / kv.second.n : 0.0);
    { std::lock_guard<std::recursive_mutex> lk(g_live_mu); g_live.erase(std::remove(g_live.begin(), g_live.end(), c), g_live.end()); }
    backend_ctx_release(c);
    delete c;
    delete b;
}
static bool backend_is_ours(ggml_backend_t b);
void flush_uploads(backend_ctx * c);
// ---- small uploads: staged, then written by one launch (ggml_backend_tensor_set_async is how llama_decode hands over the five per-token inputs)
static const size_t UP_HALF = 256 * 1024, UP_SMALL = 64 * 1024; static const int UP_MAX = 64;
struct host_timer { uint64_t & acc; std::chrono::steady_clock::time_point t0; host_timer(uint64_t & a) : acc(a), t0(std::chrono::steady_clock::now()) {} ~host_timer() { acc += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } };
void flush_uploads(backend_ctx * c) {                     // everything staged so far goes onto the stream, in order, in front of what follows
    std::lock_guard<std::recursive_mutex> lk(g_live_mu);  // (the owner thread's flush and a settle_staged() from another thread's buffer call both reach this)
    if (c->up_n == 0) return;
    const int h = c->up_half;
    void * dents = nullptr, * dbase = nullptr;
    HIP_CHECK(hipHostGetDevicePointer(&dents, c->up_ents[h], 0));
    HIP_CHECK(hipHostGetDevicePointer(&dbase, c->up_host[h], 0));
    upload_small((const upload_ent *) dents, (const char *) dbase, c->up_n, c->stream);
    HIP_CHECK(hipEventRecord(c->up_done[h], c->stream));
    c->up_half = h ^ 1; c->up_n = 0; c->up_used = 0;
    HIP_CHECK(hipEventSynchronize(c->up_done[c->up_half]));          // the other half's launch (two flushes ago) has long run
}
static bool stage_upload(backend_ctx * c, void * dst, const void * data, size_t sz) {
    if (!c->opt_batch_uploads || sz == 0 || sz > UP_SMALL) return false;
    std::lock_guard<std::recursive_mutex> lk(g_live_mu);                      // (settle_staged walks the entries from whatever thread frees / sets a buffer)
    if (!c->up_host[0]) {
        for (int h = 0; h < 2; ++h) {
            if (hipHostMalloc((void **) &c->up_host[h], UP_HALF, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **) &c->up_ents[h], sizeof(backend_ctx::up_ent) * UP_MAX, hipHostMallocDefault) != hipSuccess) {
                (void) hipGetLastError(); c->opt_batch_uploads = false; return false;
            }
            HIP_CHECK(hipEventCreateWithFlags(&c->up_done[h], hipEventDisableTiming));
        }
    }
    // entries of one launch are written by concurrent workgroups: two writes to overlapping bytes must be in different launches to keep their order
    for (int i = 0; i < c->up_n; ++i) {
        const backend_ctx::up_ent & e = c->up_ents[c->up_half][i];
        if ((char *) dst < (char *) e.dst + e.size && (char *) e.dst < (char *) dst + sz) { flush_uploads(c); break; }
    }
    const size_t at = (c->up_used + 15) & ~(size_t) 15;
    if (at + sz > UP_HALF || c->up_n == UP_MAX) { flush_uploads(c); return stage_upload(c, dst, data, sz); }
    memcpy(c->up_host[c->up_half] + at, data, sz);
    c->up_ents[c->up_half][c->up_n++] = { dst, (uint32_t) at, (uint32_t) sz };
    c->up_used = at + sz;
    return true;
}
static void be_set_async(ggml_backend_t b, struct ggml_tensor * t, const void * data, size_t off, size_t sz) {
    backend_ctx * c = (backend_ctx *) b->context; host_timer ht(c->host_ns_set); ++c->n_set; set_device(c->device);
    shadow_invalidate(c->device, (char *) t->data + off, sz);
    if (stage_upload(c, (char *) t->data + off, data, sz)) return;
    flush_uploads(c);
    HIP_CHECK(hipMemcpyAsync((char *) t->data + off, data, sz, hipMemcpyHostToDevice, c->stream));
}
static void be_get_async(ggml_backend_t b, const struct ggml_tensor * t, void * data, size_t off, size_t sz) {
    backend_ctx * c = (backend_ctx *) b->context; host_timer ht(c->host_ns_get); ++c->n_get; set_device(c->device);
    flush_uploads(c);
    HIP_CHECK(hipMemcpyAsync(data, (const char *) t->data + off, sz, hipMemcpyDeviceToHost, c->stream));
}
static bool be_cpy_async(ggml_backend_t bs, ggml_backend_t bd, const struct ggml_tensor * src, struct ggml_tensor * dst) {
    if (!backend_is_ours(bs) || !backend_is_ours(bd)) return false;
    ggml_backend_buffer_t sb = src->view_src ? src->view_src->buffer : src->buffer;
    ggml_backend_buffer_t db = dst->view_src ? dst->view_src->buffer : dst->buffer;
    if (!sb || !db || !buffer_is_ours(sb) || !buffer_is_ours(db)) return false;
    backend_ctx * cs = (backend_ctx *) bs->context; backend_ctx * cd = (backend_ctx *) bd->context;
    if (((buffer_ctx *) sb->context)->device != cs->device || ((buffer_ctx *) db->context)->device != cd->device) return false;
    flush_uploads(cs); if (cd != cs) flush_uploads(cd);
    const size_t n = nbytes(dst);
    shadow_invalidate(cd->device, dst->data, n);
    set_device(cs->device);
    if (bs == bd) {
        HIP_CHECK(hipMemcpyAsync(dst->data, src->data, n, hipMemcpyDeviceToDevice, cs->stream));
        return true;
    }
    // copy on the source stream (over xGMI when the devices differ), then make the destination stream wait for it
    if (cs->device == cd->device) HIP_CHECK(hipMemcpyAsync(dst->data, src->data, n, hipMemcpyDeviceToDevice, cs->stream));
    else                          HIP_CHECK(hipMemcpyPeerAsync(dst->data, cd->device, src->data, cs->device, n, cs->stream));
    if (!cs->copy_event) HIP_CHECK(hipEventCreateWithFlags(&cs->copy_event, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(cs->copy_event, cs->stream));
    set_device(cd->device);
    HIP_CHECK(hipStreamWaitEvent(cd->stream, cs->copy_event, 0));
    return true;
}
static void be_sync(ggml_backend_t b) {
    backend_ctx * c = (backend_ctx *) b->context; host_timer ht(c->host_ns_sync); ++c->n_sync; set_device(c->device);
    flush_uploads(c);
    HIP_CHECK(hipStreamSynchronize(c->stream));
}
static enum ggml_status be_graph_compute(ggml_backend_t b, struct ggml_cgraph * g) {
    backend_ctx * c = (backend_ctx *) b->context; host_timer ht(c->host_ns_graph); ++c->n_graph; set_device(c->device);
    flush_uploads(c);
    static const bool gpu_time = getenv("MI355X_GRAPH_GPU_TIME") != nullptr;
    if (!gpu_time) return graph_compute(c, g);
    graph_gpu_time t; t.nodes = g->n_nodes;
    HIP_CHECK(hipEventCreate(&t.a)); HIP_CHECK(hipEventCreate(&t.b));
    HIP_CHECK(hipEventRecord(t.a, c->stream));
    const enum ggml_status st = graph_compute(c, g);
    HIP_CHECK(hipEventRecord(t.b, c->stream));
    { std::lock_guard<std::recursive_mutex> lk(g_live_mu); g_graph_times.push_back(t); }
    return st;
}
static void be_graph_optimize(ggml_backend_t b, struct ggml_cgraph * g) { graph_optimize((backend_ctx *) b->context, g); }
static void be_event_record(ggml_backend_t b, ggml_backend_event_t e) {
    backend_ctx * c = (backend_ctx *) b->context; set_device(c->device);
    flush_uploads(c);
    HIP_CHECK(hipEventRecord((hipEvent_t) e->context, c->stream));
}
static void be_event_wait(ggml_backend_t b, ggml_backend_event_t e) {
    backend_ctx * c = (backend_ctx *) b->context; set_device(c->device);
    HIP_CHECK(hipStreamWaitEvent(c->stream, (hipEvent_t) e->context, 0));
}
static const ggml_backend_i k_backend_iface = {
    be_name, be_free, be_set_async, be_get_async, be_cpy_async, be_sync,
    /*plan_create*/ nullptr, /*plan_free*/ nullptr, /*plan_update*/ nullptr, /*plan_compute*/ nullptr,
    be_graph_compute, be_event_record, be_event_wait, be_graph_optimize,
};
static bool backend_is_ours(ggml_backend_t b) { return b && b->iface.graph_compute == be_graph_compute; }
bool backend_is_mi355x(const struct ggml_backend * b) { return backend_is_ours((ggml_backend_t) b); }

// ---------------------------------------------------------------------------------------------- device
static const char * dev_name(ggml_backend_dev_t d) { return ((device_ctx *) d->context)->name.c_str(); }
static const char * dev_desc(ggml_backend_dev_t d) { return ((device_ctx *) d->context)->description.c_str(); }
static void dev_memory(ggml_backend_dev_t d, size_t * free, size_t * total) {
    const int idx = ((device_ctx *) d->context)->index;
    set_device(idx);
    HIP_CHECK(hipMemGetInfo(free, total));
    *free += shadow_bytes(idx);                                       // (images are dropped on demand: buft_alloc / ensure_scratch)
    if (*free > *total) *free = *total;
}
static enum ggml_backend_dev_type dev_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }
static void dev_props(ggml_backend_dev_t d, struct ggml_backend_dev_props * p) {
    device_ctx * c = (device_ctx *) d->context;
    p->name = c->name.c_str(); p->description = c->description.c_str();
    dev_memory(d, &p->memory_free, &p->memory_total);
    p->type = GGML_BACKEND_DEVICE_TYPE_GPU;
    p->device_id = c->pci_id.empty() ? nullptr : c->pci_id.c_str();
    p->caps = { /*async*/ true, /*host_buffer*/ true, /*buffer_from_host_ptr*/ false, /*events*/ true };
}
static ggml_backend_t dev_init_backend(ggml_backend_dev_t d, const char *) {
    device_ctx * dc = (device_ctx *) d->context;
    set_device(dc->index);
    backend_ctx * c = new backend_ctx();
    c->device = dc->index;
    c->name   = dc->name;
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    backend_ctx_init(c);
    { std::lock_guard<std::recursive_mutex> lk(g_live_mu); g_live.push_back(c); }
    return new ggml_backend{ &g_guid, k_backend_iface, d, c };
}
static ggml_backend_buffer_type_t dev_buft(ggml_backend_dev_t d) { return &((device_ctx *) d->context)->buft; }
static ggml_backend_buffer_type_t dev_host_buft(ggml_backend_dev_t) { return &g_host_buft; }
static bool dev_supports_op(ggml_backend_dev_t, const struct ggml_tensor * op) { return supports_op(op); }
static bool dev_supports_buft(ggml_backend_dev_t d, ggml_backend_buffer_type_t t) {
    if (t == &g_host_buft) return false;                              // host memory is not dereferenced by kernels
    return t->iface.get_name == buft_name && t->context == d->context;
}
static bool dev_offload_op(ggml_backend_dev_t, const struct ggml_tensor * op) {
    // worth shipping CPU-resident weights over PCIe only for real batches (same policy as the reference's GPU backend, ggml-cuda.cu:3701-3705)
    const int min_batch = 32;
    return (op->ne[1] >= min_batch && op->op != GGML_OP_GET_ROWS) || (op->ne[2] >= min_batch && op->op == GGML_OP_MUL_MAT_ID);
}
static ggml_backend_event_t dev_event_new(ggml_backend_dev_t d) {
    set_device(((device_ctx *) d->context)->index);
    hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return new ggml_backend_event{ d, e };
}
static void dev_event_free(ggml_backend_dev_t, ggml_backend_event_t e) { HIP_CHECK(hipEventDestroy((hipEvent_t) e->context)); delete e; }
static void dev_event_sync(ggml_backend_dev_t, ggml_backend_event_t e) { HIP_CHECK(hipEventSynchronize((hipEvent_t) e->context)); }
static const ggml_backend_device_i k_device_iface = {
    dev_name, dev_desc, dev_memory, dev_type, dev_props, dev_init_backend, dev_buft, dev_host_buft, /*from_host_ptr*/ nullptr,
    dev_supports_op, dev_supports_buft, dev_offload_op, dev_event_new, dev_event_free, dev_event_sync,
};

// ---------------------------------------------------------------------------------------------- registry
static const char * reg_name(ggml_backend_reg_t) { return GGML_MI355X_NAME; }
static size_t reg_dev_count(ggml_backend_reg_t) { return g_devices.size(); }
static ggml_backend_dev_t reg_get_dev(ggml_backend_reg_t, size_t i) { return i < g_devices.size() ? &g_devices[i]->dev : nullptr; }

struct feature { const char * name; const char * value; };
static feature g_features[] = { {"ARCH", "gfx950"}, {"WAVE", "64"}, {"ACT_QUANT", "Q8_K/Q8_0 (CPU-parity)"}, {nullptr, nullptr} };
static feature * get_features(ggml_backend_reg_t) { return g_features; }

} // namespace mi

// extension entry points are reachable through reg->iface.get_proc_address (the reference's mechanism for
// backend-specific functions, ggml-backend.h:197-212)
namespace mi {
static void * reg_proc(ggml_backend_reg_t, const char * name) {
    if (!strcmp(name, "ggml_backend_get_features")) return (void *) get_features;
    if (!strcmp(name, "mi355x_timed_event_new"))     return (void *) mi355x_timed_event_new;
    if (!strcmp(name, "mi355x_timed_event_record"))  return (void *) mi355x_timed_event_record;
    if (!strcmp(name, "mi355x_timed_event_elapsed_ms")) return (void *) mi355x_timed_event_elapsed_ms;
    if (!strcmp(name, "mi355x_timed_event_free"))    return (void *) mi355x_timed_event_free;
    if (!strcmp(name, "mi355x_set_option"))          return (void *) mi355x_set_option;
    if (!strcmp(name, "mi355x_get_stat"))            return (void *) mi355x_get_stat;
    if (!strcmp(name, "mi355x_debug_quantize"))      return (void *) mi355x_debug_quantize;
    if (!strcmp(name, "mi355x_handoff_init"))        return (void *) mi355x_handoff_init;
    if (!strcmp(name, "mi355x_handoff"))             return (void *) mi355x_handoff;
    if (!strcmp(name, "mi355x_handoff_tensor"))      return (void *) mi355x_handoff_tensor;
    if (!strcmp(name, "mi355x_handoff_count"))       return (void *) mi355x_handoff_count;
    if (!strcmp(name, "mi355x_handoff_shutdown"))    return (void *) mi355x_handoff_shutdown;
    if (!strcmp(name, "mi355x_module_device"))       return (void *) mi355x_module_device;
    return nullptr;
}
static const ggml_backend_reg_i k_reg_iface = { reg_name, reg_dev_count, reg_get_dev, reg_proc };
// A host that calls ggml_backend_load_all() more than once (token2wav-impl.cpp does, after omni_init already did) makes the reference's loader bind and call
// ggml_backend_init again, and ggml-backend-reg.cpp:226-239 appends whatever devices the returned registry reports: MI355X0 appeared three times in
// ggml_backend_dev_count() under the omni runtime.  The loader's entry point therefore hands out the device-bearing registry ONCE per process; later calls get this
// second registry object -- same name, same proc addresses, no devices -- so the global device list stays what the first registration made it.
static size_t reg_dev_count_none(ggml_backend_reg_t) { return 0; }
static ggml_backend_dev_t reg_get_dev_none(ggml_backend_reg_t, size_t) { return nullptr; }
static const ggml_backend_reg_i k_reg_again_iface = { reg_name, reg_dev_count_none, reg_get_dev_none, reg_proc };
static ggml_backend_reg g_reg_again = { GGML_BACKEND_API_VERSION, k_reg_again_iface, nullptr };

static void init_once() {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_init_done) return;
    g_init_done = true;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void) hipGetLastError(); n = 0; }
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) != hipSuccess) continue;
        const std::string arch = p.gcnArchName;
        if (arch.rfind("gfx950", 0) != 0 && !getenv("MI355X_ALLOW_ANY_ARCH")) {
            log_msg(GGML_LOG_LEVEL_WARN, "[mi355x] device %d is %s, not gfx950: skipped\n", i, p.gcnArchName);
            continue;
        }
        device_ctx * d = new device_ctx();
        d->index = i;
        d->name = std::string(GGML_MI355X_NAME) + std::to_string(g_devices.size());
        d->description = p.name[0] ? std::string(p.name) : std::string("AMD Instinct MI355X (") + p.gcnArchName + ")";   // some containers report no marketing name
        char id[32]; snprintf(id, sizeof(id), "%04x:%02x:%02x.0", p.pciDomainID, p.pciBusID, p.pciDeviceID);
        d->pci_id = id;
        d->total_mem = p.totalGlobalMem;
        d->buft = { k_buft_iface, &d->dev, d };
        d->dev  = { k_device_iface, &g_reg, d };
        g_devices.push_back(d);
    }
    g_host_buft = { k_hostbuft_iface, g_devices.empty() ? nullptr : &g_devices[0]->dev, nullptr };
    g_reg = { GGML_BACKEND_API_VERSION, k_reg_iface, nullptr };
}

} // namespace mi

// ---------------------------------------------------------------------------------------------- exported C ABI
extern "C" {

ggml_backend_reg_t ggml_backend_mi355x_reg(void) { mi::init_once(); return &mi::g_reg; }
ggml_backend_reg_t ggml_backend_init(void) {          // the loader's entry (ggml-backend-reg.cpp:265-285): idempotent towards the registry's device list
    static std::atomic<int> calls{0};
    static const bool always_full = getenv("MI355X_INIT_ALWAYS_FULL") != nullptr;
    ggml_backend_reg_t full = ggml_backend_mi355x_reg();
    return calls.fetch_add(1) == 0 || always_full ? full : &mi::g_reg_again;
}
int ggml_backend_score(void) {
    mi::init_once();
    return mi::g_devices.empty() ? 0 : 100;
}

// standalone harness only: what the core's ggml_backend_buffer_free does (ggml-backend.cpp:108-117)
void mi355x_host_buffer_free(ggml_backend_buffer_t b) {
    if (!b) return;
    if (b->iface.free_buffer) b->iface.free_buffer(b);
    delete b;
}

long mi355x_debug_quantize(ggml_backend_t backend, int kind, const float * host_x, long K, long nrows, void * host_images) {
    mi::backend_ctx * c = (mi::backend_ctx *) backend->context;
    HIP_CHECK(hipSetDevice(c->device));
    const size_t img = kind == 0 ? q8k_image_bytes(K) : q80_image_bytes(K);
    float * dx = nullptr; void * di = nullptr;
    HIP_CHECK(hipMalloc((void **) &dx, (size_t) K * nrows * 4));
    HIP_CHECK(hipMalloc(&di, img * nrows));
    HIP_CHECK(hipMemcpyAsync(dx, host_x, (size_t) K * nrows * 4, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemsetAsync(di, 0, img * nrows, c->stream));
    if (kind == 0) mi::quantize_q8k_image(dx, (size_t) K * 4, di, K, nrows, c->stream);
    else           mi::quantize_q80_image(dx, (size_t) K * 4, di, K, nrows, c->stream);
    HIP_CHECK(hipMemcpyAsync(host_images, di, img * nrows, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    HIP_CHECK(hipFree(dx)); HIP_CHECK(hipFree(di));
    return (long) img;
}

void * mi355x_timed_event_new(void) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return e; }
void   mi355x_timed_event_record(void * ev, ggml_backend_t backend) {
    mi::backend_ctx * c = (mi::backend_ctx *) backend->context;
    HIP_CHECK(hipSetDevice(c->device));
    mi::flush_uploads(c);
    HIP_CHECK(hipEventRecord((hipEvent_t) ev, c->stream));
}
float  mi355x_timed_event_elapsed_ms(void * start, void * stop) {
    HIP_CHECK(hipEventSynchronize((hipEvent_t) stop));
    float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t) start, (hipEvent_t) stop)); return ms;
}
void   mi355x_timed_event_free(void * ev) { HIP_CHECK(hipEventDestroy((hipEvent_t) ev)); }

} // extern "C"

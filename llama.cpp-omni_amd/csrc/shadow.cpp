// shadow.cpp -- registry of resident F16 weight images (see shadow.hpp)
#include "shadow.hpp"
#include <atomic>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <vector>

namespace mi {

struct shadow_entry {
    int device; const char * src; size_t src_bytes; int type; int64_t K, M; size_t src_rs;
    void * f16; size_t bytes;
    hipEvent_t ready; hipStream_t owner; bool recorded, complete;     // fill kernel: event recorded after it on `owner`; complete once the event was seen done
};

static std::mutex                g_mu;
static std::vector<shadow_entry> g_entries;
static std::atomic<uint64_t>     g_gen { 0 };
static std::atomic<bool>         g_on { true };
static size_t                    g_total = 0;
// bounding box of every registered source range: the common case (writes to activations / KV) is rejected with two compares
static const char * g_lo = nullptr; static const char * g_hi = nullptr;

static const size_t RESERVE_BYTES = (size_t) 8 << 30;        // never take the device below 8 GiB free

static void recompute_box() {
    g_lo = g_hi = nullptr;
    for (const shadow_entry & e : g_entries) {
        if (!g_lo || e.src < g_lo) g_lo = e.src;
        if (!g_hi || e.src + e.src_bytes > g_hi) g_hi = e.src + e.src_bytes;
    }
}

static size_t max_total_bytes() {
    static const size_t v = getenv("MI355X_F16_SHADOW_MAX_GB") ? (size_t) atoll(getenv("MI355X_F16_SHADOW_MAX_GB")) << 30 : (size_t) -1;
    return v;
}
// (g_mu held) make the image safe to read on `st`; false: not usable here
static bool entry_usable(shadow_entry & e, hipStream_t st, bool capturing) {
    if (e.complete) return true;
    if (!e.recorded) return false;                                   // its creator has not even launched the fill yet
    if (capturing) return st == e.owner;                             // (no event query / cross-stream wait inside a stream capture: both invalidate it)
    if (hipEventQuery(e.ready) == hipSuccess) { e.complete = true; return true; }
    (void) hipGetLastError();
    if (st == e.owner) return true;                                  // same stream: ordered behind the fill
    HIP_CHECK(hipStreamWaitEvent(st, e.ready, 0));
    return true;
}

const uint16_t * shadow_find(int device, const void * src, int type, int64_t K, int64_t M, size_t src_rs, hipStream_t st, bool capturing) {
    if (!g_on.load(std::memory_order_relaxed)) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    for (shadow_entry & e : g_entries)
        if (e.src == (const char *) src && e.device == device && e.type == type && e.K == K && e.M == M && e.src_rs == src_rs)
            return entry_usable(e, st, capturing) ? (const uint16_t *) e.f16 : nullptr;
    return nullptr;
}

uint16_t * shadow_get_or_create(int device, const void * src, size_t src_bytes, int type, int64_t K, int64_t M, size_t src_rs, hipStream_t st, bool capturing, bool * created) {
    static const bool env_off = getenv("MI355X_NO_F16_SHADOW") != nullptr;
    *created = false;
    if (!g_on.load(std::memory_order_relaxed) || env_off) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    for (shadow_entry & e : g_entries)
        if (e.src == (const char *) src && e.device == device && e.type == type && e.K == K && e.M == M && e.src_rs == src_rs)
            return entry_usable(e, st, capturing) ? (uint16_t *) e.f16 : nullptr;
    if (capturing) return nullptr;                                   // (no allocation inside a stream capture)
    const size_t bytes = (size_t) K * (size_t) M * 2;
    if (g_total + bytes > max_total_bytes()) return nullptr;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    if (free_b < bytes + RESERVE_BYTES) return nullptr;
    void * p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); (void) hipFree(p); return nullptr; }
    g_entries.push_back({ device, (const char *) src, src_bytes, type, K, M, src_rs, p, bytes, ev, st, false, false });
    g_total += bytes;
    recompute_box();
    *created = true;
    return (uint16_t *) p;
}

void shadow_mark_ready(const void * image, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (shadow_entry & e : g_entries)
        if (e.f16 == image) { HIP_CHECK(hipEventRecord(e.ready, st)); e.recorded = true; return; }
}

static std::shared_mutex & use_lock(int device) {
    static std::shared_mutex locks[16];
    return locks[device >= 0 && device < 16 ? device : 0];
}
shadow_reader::shadow_reader(int d) : device(d), held(false) { lock(); }
shadow_reader::~shadow_reader() { unlock(); }
void shadow_reader::lock()   { if (!held) { use_lock(device).lock_shared(); held = true; } }
void shadow_reader::unlock() { if (held) { use_lock(device).unlock_shared(); held = false; } }

size_t shadow_drop_all(int device, shadow_reader * mine) {
    const bool had = mine && mine->held;
    if (had) mine->unlock();
    struct relock { shadow_reader * m; bool on; ~relock() { if (on) m->lock(); } } rl{ mine, had };
    std::unique_lock<std::shared_mutex> ex(use_lock(device));          // no submission of any context of this device is between an image look-up and its launches
    std::vector<shadow_entry> dead;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_entries.size();) {
            if (g_entries[i].device == device) { dead.push_back(g_entries[i]); g_total -= g_entries[i].bytes; g_entries[i] = g_entries.back(); g_entries.pop_back(); }
            else ++i;
        }
        if (!dead.empty()) { recompute_box(); g_gen.fetch_add(1); }
    }
    size_t n = 0;
    for (shadow_entry & d : dead) { n += d.bytes; (void) hipFree(d.f16); (void) hipEventDestroy(d.ready); }   // hipFree waits for kernels still reading the image
    return n;
}
size_t shadow_bytes(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    size_t n = 0;
    for (const shadow_entry & e : g_entries) if (e.device == device) n += e.bytes;
    return n;
}

void shadow_invalidate(int device, const void * p, size_t n) {
    const char * lo = (const char *) p, * hi = lo + n;
    std::vector<shadow_entry> dead;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_lo || hi <= g_lo || lo >= g_hi) return;
        for (size_t i = 0; i < g_entries.size();) {
            shadow_entry & e = g_entries[i];
            if (e.device == device && lo < e.src + e.src_bytes && e.src < hi) {
                dead.push_back(e); g_total -= e.bytes;
                g_entries[i] = g_entries.back(); g_entries.pop_back();
            } else ++i;
        }
        if (!dead.empty()) { recompute_box(); g_gen.fetch_add(1); }
    }
    for (shadow_entry & d : dead) { HIP_CHECK(hipFree(d.f16)); (void) hipEventDestroy(d.ready); }   // hipFree waits for kernels that may still read the image
}

uint64_t shadow_generation() { return g_gen.load(); }
void     shadow_set_enabled(bool on) { g_on.store(on); }
bool     shadow_enabled() { return g_on.load(); }
double   shadow_stat(const char * name) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!strcmp(name, "shadow_bytes"))   return (double) g_total;
    if (!strcmp(name, "shadow_tensors")) return (double) g_entries.size();
    return -1.0;
}

} // namespace mi

// shadow.cpp -- registry of resident F16 weight images (see shadow.hpp)
#include "shadow.hpp"
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

namespace mi {

struct shadow_entry {
    int device; const char * src; size_t src_bytes; int type; int64_t K, M; size_t src_rs;
    void * f16; size_t bytes;
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

const uint16_t * shadow_find(int device, const void * src, int type, int64_t K, int64_t M, size_t src_rs) {
    if (!g_on.load(std::memory_order_relaxed)) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    for (const shadow_entry & e : g_entries)
        if (e.src == (const char *) src && e.device == device && e.type == type && e.K == K && e.M == M && e.src_rs == src_rs) return (const uint16_t *) e.f16;
    return nullptr;
}

uint16_t * shadow_create(int device, const void * src, size_t src_bytes, int type, int64_t K, int64_t M, size_t src_rs) {
    static const bool env_off = getenv("MI355X_NO_F16_SHADOW") != nullptr;
    if (!g_on.load(std::memory_order_relaxed) || env_off) return nullptr;
    const size_t bytes = (size_t) K * (size_t) M * 2;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    if (free_b < bytes + RESERVE_BYTES) return nullptr;
    void * p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(g_mu);
    g_entries.push_back({ device, (const char *) src, src_bytes, type, K, M, src_rs, p, bytes });
    g_total += bytes;
    recompute_box();
    return (uint16_t *) p;
}

void shadow_invalidate(int device, const void * p, size_t n) {
    const char * lo = (const char *) p, * hi = lo + n;
    std::vector<void *> dead;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_lo || hi <= g_lo || lo >= g_hi) return;
        for (size_t i = 0; i < g_entries.size();) {
            shadow_entry & e = g_entries[i];
            if (e.device == device && lo < e.src + e.src_bytes && e.src < hi) {
                dead.push_back(e.f16); g_total -= e.bytes;
                g_entries[i] = g_entries.back(); g_entries.pop_back();
            } else ++i;
        }
        if (!dead.empty()) { recompute_box(); g_gen.fetch_add(1); }
    }
    for (void * d : dead) HIP_CHECK(hipFree(d));               // hipFree waits for kernels that may still read the image
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

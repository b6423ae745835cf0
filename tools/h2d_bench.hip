// tools/h2d_bench.hip -- what a blocking 8 MB host -> device copy from PAGEABLE memory costs (the inter-split copy of a 512-token embedding block in libllama), against
// staging it through pinned memory with 1 / 2 / 4 / 8 copying threads.   hipcc -O2 -pthread tools/h2d_bench.hip -o /tmp/h2d && /tmp/h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t sizes[] = { 64 << 10, 608 << 10, 2 << 20, 8 << 20, 32 << 20 };
    void * dev; hipMalloc(&dev, 64 << 20);
    char * pin; hipHostMalloc((void **) &pin, 64 << 20, hipHostMallocDefault);
    hipStream_t st; hipStreamCreate(&st);
    {   // host time of the NON-waiting form for small blocks: copy into a pinned slot, hipMemcpyAsync on a copy stream, record an event (what a later submission waits on)
        hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        for (size_t sz : { (size_t) 64, (size_t) 2048, (size_t) 16384, (size_t) 65536 }) {
            char * src = (char *) malloc(sz); memset(src, 3, sz);
            double best = 1e9, best_sync = 1e9;
            for (int rep = 0; rep < 50; ++rep) {
                double t0 = now(); memcpy(pin, src, sz); hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, st); hipEventRecord(ev, st); best = std::min(best, now() - t0);
                hipStreamSynchronize(st);
                t0 = now(); hipMemcpyAsync(dev, src, sz, hipMemcpyHostToDevice, hipStreamPerThread); hipStreamSynchronize(hipStreamPerThread); best_sync = std::min(best_sync, now() - t0);
            }
            printf("%8zu B: pinned slot + async copy + event record %6.1f us of host time | pageable async + stream sync (today) %6.1f us\n", sz, best, best_sync);
            free(src);
        }
    }
    for (size_t sz : sizes) {
        char * src = (char *) malloc(sz); memset(src, 1, sz);
        double best[6] = { 1e9, 1e9, 1e9, 1e9, 1e9, 1e9 };
        for (int rep = 0; rep < 8; ++rep) {
            memset(src, rep, sz);                                   // (touched by the CPU just before, like a CPU split's output)
            double t0 = now(); hipMemcpy(dev, src, sz, hipMemcpyHostToDevice); best[0] = std::min(best[0], now() - t0);
            t0 = now(); hipMemcpyAsync(dev, src, sz, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); best[1] = std::min(best[1], now() - t0);
            int k = 2;
            for (int nt : { 1, 2, 4, 8 }) {
                t0 = now();
                if (nt == 1) memcpy(pin, src, sz);
                else {
                    std::vector<std::thread> th;
                    const size_t per = (sz / nt + 4095) & ~(size_t) 4095;
                    for (int i = 0; i < nt; ++i) th.emplace_back([=] { const size_t a = i * per; if (a < sz) memcpy(pin + a, src + a, std::min(per, sz - a)); });
                    for (auto & t : th) t.join();
                }
                hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
                best[k] = std::min(best[k], now() - t0); ++k;
            }
        }
        printf("%8zu KB: hipMemcpy pageable %7.1f us | async+sync pageable %7.1f | pinned staging 1 thread %7.1f, 2 threads %7.1f, 4 threads %7.1f, 8 threads %7.1f us\n", sz >> 10, best[0], best[1], best[2], best[3], best[4], best[5]);
        free(src);
    }
    return 0;
}

// tools/launch_rate.hip -- host cost of one kernel launch on this box (eager stream launches, what graph_compute pays per node when a graph cannot be replayed)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct big { void * p[8]; int v[16]; };
__global__ void k_empty(float * p) { if (p && threadIdx.x == 9999) *p = 0; }
__global__ void k_big(const big b) { if (b.p[0] && threadIdx.x == 9999) *(float *) b.p[0] = 0; }
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int pass = 0; pass < 2; ++pass)
        for (int n : {256, 2048, 20000}) {
            big b = {};
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) { if (pass) k_big<<<64, 256, 0, st>>>(b); else k_empty<<<64, 256, 0, st>>>(nullptr); }
            auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(st);
            auto t2 = std::chrono::steady_clock::now();
            printf("%s x %5d: host enqueue %.2f us/launch, enqueue+drain %.2f us/launch\n", pass ? "128-byte arg" : "pointer arg ", n,
                   std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
        }
    return 0;
}

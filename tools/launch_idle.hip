// tools/launch_idle.hip -- host cost of the FIRST kernel launch after the stream went idle (hipStreamSynchronize, a blocking copy on another stream, a sleep): what an
// eager graph_compute pays on its first node when the caller has just synchronised / uploaded inputs (the omni encoders do both before every chunk)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_empty(float * p) { if (p && threadIdx.x == 9999) *p = 0; }
static double us_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t).count(); }
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    float * d; hipMalloc(&d, 1 << 20); std::vector<float> h(1 << 18);
    for (int i = 0; i < 100; ++i) k_empty<<<64, 256, 0, st>>>(nullptr);
    hipStreamSynchronize(st);
    const char * what[] = { "right after hipStreamSynchronize", "after sync + 1 MB hipMemcpyAsync on hipStreamPerThread + its sync", "after sync + 200 us sleep", "after sync + 5 ms sleep", "after sync + blocking hipMemcpy H2D (null stream)" };
    for (int mode = 0; mode < 5; ++mode) {
        double first = 0, second = 0;
        const int R = 20;
        for (int r = 0; r < R; ++r) {
            for (int i = 0; i < 50; ++i) k_empty<<<64, 256, 0, st>>>(nullptr);
            hipStreamSynchronize(st);
            if (mode == 1) { hipMemcpyAsync(d, h.data(), 1 << 20, hipMemcpyHostToDevice, hipStreamPerThread); hipStreamSynchronize(hipStreamPerThread); }
            if (mode == 2) std::this_thread::sleep_for(std::chrono::microseconds(200));
            if (mode == 3) std::this_thread::sleep_for(std::chrono::milliseconds(5));
            if (mode == 4) hipMemcpy(d, h.data(), 1 << 20, hipMemcpyHostToDevice);
            auto t0 = std::chrono::steady_clock::now();
            k_empty<<<64, 256, 0, st>>>(nullptr);
            first += us_since(t0);
            t0 = std::chrono::steady_clock::now();
            k_empty<<<64, 256, 0, st>>>(nullptr);
            second += us_since(t0);
        }
        printf("%-80s first launch %.1f us, second %.1f us\n", what[mode], first / R, second / R);
    }
    return 0;
}

// tools/overlap_bench.hip -- measurement (not part of the product): can two dependent-by-flag kernels overlap on MI355X?
//   (a) two kernels in ONE stream, plain launches                         -> serial by the AQL barrier bit
//   (b) the same with hipExtAnyOrderLaunch                                -> overlapped if the runtime drops the barrier bit on gfx950
//   (c) two capture streams forked / joined into one hipGraph              -> overlapped if parallel graph branches get their own queues
// Each kernel: `grid` workgroups of 256 threads spinning ~T us on the realtime counter (s_memrealtime, 100 MHz).
//   build: hipcc --offload-arch=gfx950 -O3 tools/overlap_bench.hip -o /tmp/overlap_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_spin(unsigned long long ticks, unsigned * sink) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks && n < 100000000u) { __builtin_amdgcn_s_sleep(8); ++n; }
    if (n == 0xffffffffu) *sink = n;
}
__global__ void __launch_bounds__(256) k_spin2(unsigned long long ticks, unsigned * sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks && n < 100000000u) { __builtin_amdgcn_s_sleep(8); ++n; }
    if (n == 0xffffffffu) *sink = n;
}

int main() {
    hipStream_t s0, s1; CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
    hipEvent_t e0, e1, ef, ej; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&ef)); CHECK(hipEventCreate(&ej));
    unsigned * sink; CHECK(hipMalloc(&sink, 4));
    const unsigned long long ticks = 100 * 50;          // 100 MHz wall clock: 50 us
    for (int grid : { 64, 256, 1024 }) {
        float ms_a = 1e30f, ms_b = 1e30f, ms_c = 1e30f, ms_1 = 1e30f;
        for (int r = 0; r < 5; ++r) {
            float ms;
            CHECK(hipEventRecord(e0, s0)); k_spin2<<<grid, 256, 0, s0>>>(ticks, sink); CHECK(hipEventRecord(e1, s0)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < ms_1) ms_1 = ms;
            CHECK(hipEventRecord(e0, s0)); k_spin2<<<grid, 256, 0, s0>>>(ticks, sink); k_spin2<<<grid, 256, 0, s0>>>(ticks, sink); CHECK(hipEventRecord(e1, s0)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < ms_a) ms_a = ms;
            CHECK(hipEventRecord(e0, s0));
            hipExtLaunchKernelGGL(k_spin2, dim3(grid), dim3(256), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink);
            hipExtLaunchKernelGGL(k_spin2, dim3(grid), dim3(256), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink);
            CHECK(hipEventRecord(e1, s0)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < ms_b) ms_b = ms;
        }
        hipGraph_t graph; hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        CHECK(hipEventRecord(ef, s0)); CHECK(hipStreamWaitEvent(s1, ef, 0));
        k_spin2<<<grid, 256, 0, s0>>>(ticks, sink);
        k_spin2<<<grid, 256, 0, s1>>>(ticks, sink);
        CHECK(hipEventRecord(ej, s1)); CHECK(hipStreamWaitEvent(s0, ej, 0));
        CHECK(hipStreamEndCapture(s0, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int r = 0; r < 5; ++r) {
            float ms;
            CHECK(hipEventRecord(e0, s0)); CHECK(hipGraphLaunch(exec, s0)); CHECK(hipEventRecord(e1, s0)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < ms_c) ms_c = ms;
        }
        printf("grid %5d: one 50-us kernel %.1f us | two in a stream %.1f us | two with hipExtAnyOrderLaunch %.1f us | two graph branches %.1f us\n", grid, ms_1 * 1e3f, ms_a * 1e3f, ms_b * 1e3f, ms_c * 1e3f);
        CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
    }
    return 0;
}

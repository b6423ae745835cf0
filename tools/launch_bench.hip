// tools/launch_bench.hip -- measurement (not part of the product): what does ONE dependent kernel cost on this box, as a function of what the
// kernel does before its first useful instruction?  Separates the launch boundary (MI355X_MICROARCH.md "boundary": ~1.45 us between trivial
// kernels) from a kernel's own serial latency chain (kernarg fetch, a dependent read of what the previous kernel wrote, LDS reduction, store).
//   build: hipcc --offload-arch=gfx950 -O3 tools/launch_bench.hip -o gpurun_out/launch_bench      run: gpurun_out/launch_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct big_args { const float * in; float * out; int n; int pad; uint64_t filler[28]; };   // 256-byte kernarg block

__global__ void __launch_bounds__(256) k_empty(const float * in, float * out, int n) {}
__global__ void __launch_bounds__(256) k_touch(const float * in, float * out, int n) {          // one lane per workgroup: read one word, write one
    if (threadIdx.x == 0) out[blockIdx.x] = in[blockIdx.x] + 1.0f;
}
__global__ void __launch_bounds__(256) k_touch_big(const big_args a) {
    if (threadIdx.x == 0) a.out[blockIdx.x] = a.in[blockIdx.x] + 1.0f;
}
// every workgroup reads the whole 16 KB vector the previous kernel wrote (one 16-B load per lane x 4), sums it on the DPP network +
// one LDS exchange, writes its slice
__global__ void __launch_bounds__(256) k_vec(const float * __restrict__ in, float * __restrict__ out, int n) {
    __shared__ float red[4];
    const f32x4 * p = (const f32x4 *) in;
    f32x4 v0 = p[threadIdx.x], v1 = p[threadIdx.x + 256], v2 = p[threadIdx.x + 512], v3 = p[threadIdx.x + 768];
    float s = (v0[0] + v0[1] + v0[2] + v0[3]) + (v1[0] + v1[1] + v1[2] + v1[3]) + (v2[0] + v2[1] + v2[2] + v2[3]) + (v3[0] + v3[1] + v3[2] + v3[3]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const int per = n / gridDim.x;
    for (int i = threadIdx.x; i < per; i += 256) out[blockIdx.x * per + i] = tot * 1e-9f + 1.0f;
}
// the same plus a weight stream: each workgroup also streams `wbytes / gridDim.x` bytes of a private weight slice (16-B loads, 8 in flight
// per lane) issued BEFORE the dependent read -- what a mat-vec with an in-kernel activation prologue does
__global__ void __launch_bounds__(256) k_vec_stream(const float * __restrict__ in, float * __restrict__ out, int n, const f32x4 * __restrict__ W, long per_wg16) {
    __shared__ float red[4];
    const f32x4 * w = W + (long) blockIdx.x * per_wg16;
    f32x4 acc = {0, 0, 0, 0};
    const f32x4 * p = (const f32x4 *) in;
    f32x4 v0 = p[threadIdx.x], v1 = p[threadIdx.x + 256], v2 = p[threadIdx.x + 512], v3 = p[threadIdx.x + 768];
    for (long i = threadIdx.x; i < per_wg16; i += 256 * 8) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { long k = i + u * 256; t[u] = k < per_wg16 ? __builtin_nontemporal_load(w + k) : f32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u];
    }
    float s = (v0[0] + v0[1] + v0[2] + v0[3]) + (v1[0] + v1[1] + v1[2] + v1[3]) + (v2[0] + v2[1] + v2[2] + v2[3]) + (v3[0] + v3[1] + v3[2] + v3[3]);
    s += acc[0] + acc[1] + acc[2] + acc[3];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const int per = n / gridDim.x;
    for (int i = threadIdx.x; i < per; i += 256) out[blockIdx.x * per + i] = tot * 1e-9f + 1.0f;
}

static hipStream_t st;
static hipEvent_t e0, e1;

template <typename F> static void run(const char * name, int grid, F launch) {
    const int N = 400;
    // eager
    float best_e = 1e30f, best_g = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0, st));
        for (int s = 0; s < N; ++s) launch(s);
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_e) best_e = ms;
    }
    hipGraph_t graph; hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < N; ++s) launch(s);
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int r = 0; r < 6; ++r) {
        CHECK(hipEventRecord(e0, st)); CHECK(hipGraphLaunch(exec, st)); CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_g) best_g = ms;
    }
    printf("%-44s grid %5d : %6.2f us / kernel eager, %6.2f us / kernel in a replayed hipGraph\n", name, grid, best_e * 1e3f / N, best_g * 1e3f / N);
    CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
}

int main() {
    const int n = 4096;
    float * a, * b; f32x4 * W;
    const long wbytes_max = 64l << 20;
    CHECK(hipMalloc(&a, 1 << 20)); CHECK(hipMalloc(&b, 1 << 20)); CHECK(hipMalloc(&W, wbytes_max * 8));
    CHECK(hipMemset(a, 0, 1 << 20)); CHECK(hipMemset(b, 0, 1 << 20)); CHECK(hipMemset(W, 0, wbytes_max * 8));
    CHECK(hipStreamCreate(&st));
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grids[] = { 1, 256, 1024, 2048 };
    for (int g : grids) {
        run("empty", g, [&](int s) { k_empty<<<g, 256, 0, st>>>((s & 1) ? b : a, (s & 1) ? a : b, n); });
        run("touch (1 word in, 1 word out per WG)", g, [&](int s) { k_touch<<<g, 256, 0, st>>>((s & 1) ? b : a, (s & 1) ? a : b, n); });
        run("touch, 256-byte kernarg", g, [&](int s) { big_args x; x.in = (s & 1) ? b : a; x.out = (s & 1) ? a : b; x.n = n; k_touch_big<<<g, 256, 0, st>>>(x); });
        run("read 16 KB + reduce + write slice", g, [&](int s) { k_vec<<<g, 256, 0, st>>>((s & 1) ? b : a, (s & 1) ? a : b, n); });
    }
    // dependent read + weight stream of 9.4 / 28 / 56 MB (8 distinct slices in rotation so nothing is cache-resident)
    const long sizes[] = { 9437184, 28311552, 56623104 };
    for (long wb : sizes) for (int g : { 256, 512, 1024 }) {
        char nm[96]; snprintf(nm, sizeof nm, "16 KB dependent read + %.1f MB stream", wb / 1e6);
        const long per16 = wb / 16 / g;
        run(nm, g, [&](int s) { k_vec_stream<<<g, 256, 0, st>>>((s & 1) ? b : a, (s & 1) ? a : b, n, W + (long) (s % 8) * (wbytes_max / 16), per16); });
    }
    return 0;
}

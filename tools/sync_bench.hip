// tools/sync_bench.hip -- measurement for DESIGN.md section 7 (not part of the product): what does a dependency between two stages of
// the decode step cost on MI355X when it is (a) a kernel boundary inside a replayed hipGraph, (b) a grid-wide barrier inside one
// persistent launch?  Both variants move a small "activation" (16 KB) through memory between stages, like the real step does.
//   build: hipcc --offload-arch=gfx950 -O3 tools/sync_bench.hip -o gpurun_out/sync_bench      run: gpurun_out/sync_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// one stage: every workgroup reads the 4096-float vector produced by the previous stage, does a token amount of work, and the
// workgroups together write the next vector (each owns a slice)
__device__ __forceinline__ float stage_work(const float * in, int n, int tid, int nt) {
    float s = 0.0f;
    for (int i = tid; i < n; i += nt) s += in[i];
    return s;
}

__global__ void __launch_bounds__(256) k_stage(const float * __restrict__ in, float * __restrict__ out, int n) {
    __shared__ float red[4];
    float s = stage_work(in, n, threadIdx.x, 256);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const int per = (n + gridDim.x - 1) / gridDim.x;
    for (int i = threadIdx.x; i < per; i += 256) { const int j = blockIdx.x * per + i; if (j < n) out[j] = tot * 1e-6f + (float) j; }
}

// persistent variant: the same stages separated by a grid barrier (monotonic counter, agent-scope release / acquire, one lane per
// workgroup spins; cdna_hip_programming.md section 6 G16).  All workgroups must be co-resident (grid <= resident capacity).
__global__ void __launch_bounds__(256) k_persistent(float * __restrict__ bufa, float * __restrict__ bufb, int n, int nstages, unsigned * __restrict__ counter) {
    __shared__ float red[4];
    const int per = (n + gridDim.x - 1) / gridDim.x;
    for (int st = 0; st < nstages; ++st) {
        const float * in = (st & 1) ? bufb : bufa;
        float * out = (st & 1) ? bufa : bufb;
        float s = 0.0f;
        for (int i = threadIdx.x; i < n; i += 256) s += __builtin_nontemporal_load(in + i);      // (fresh from memory: written by other workgroups)
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        const float tot = red[0] + red[1] + red[2] + red[3];
        for (int i = threadIdx.x; i < per; i += 256) { const int j = blockIdx.x * per + i; if (j < n) out[j] = tot * 1e-6f + (float) j; }
        // ---- grid barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned) (st + 1) * gridDim.x;
            long spins = 0;                                                  // bounded: a workgroup that is not resident must not hang the GPU
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 50000000L) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

int main() {
    const int n = 4096, nstages = 256;
    float * a, * b; unsigned * cnt;
    CHECK(hipMalloc(&a, n * 4)); CHECK(hipMalloc(&b, n * 4)); CHECK(hipMalloc(&cnt, 4));
    CHECK(hipMemset(a, 0, n * 4)); CHECK(hipMemset(b, 0, n * 4));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grids[] = { 8, 64, 256, 512, 1024 };
    for (int g : grids) {
        // (a) hipGraph of nstages dependent kernels
        hipGraph_t graph; hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int s = 0; s < nstages; ++s) k_stage<<<g, 256, 0, st>>>((s & 1) ? b : a, (s & 1) ? a : b, n);
        CHECK(hipStreamEndCapture(st, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        float best_a = 1e30f, best_b = 1e30f;
        for (int r = 0; r < 5; ++r) {
            CHECK(hipEventRecord(e0, st)); CHECK(hipGraphLaunch(exec, st)); CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_a) best_a = ms;
        }
        // (b) one persistent launch with grid barriers (only when every workgroup can be resident: 256 CUs x 4)
        if (g <= 1024) {
            for (int r = 0; r < 5; ++r) {
                CHECK(hipMemsetAsync(cnt, 0, 4, st));
                CHECK(hipEventRecord(e0, st));
                k_persistent<<<g, 256, 0, st>>>(a, b, n, nstages, cnt);
                CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_b) best_b = ms;
            }
        }
        printf("workgroups %5d: dependent stage = %6.2f us as a hipGraph kernel node, %6.2f us behind a grid barrier in one launch\n",
               g, best_a * 1e3f / nstages, best_b * 1e3f / nstages);
        CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
    }
    return 0;
}

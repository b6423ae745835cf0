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


// (c) flag-array variant: no atomics, no full-L2 fences.  Every workgroup owns one flag word; data and flags travel as agent-scope
// (sc1) write-through stores / L2-bypassing loads (buffer aux bit 4), ordered by s_waitcnt vmcnt(0).  One wave polls all flags (one
// coalesced load), so nothing serialises on a single address.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void * p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *) p, 0, (int) bytes, 0x00020000); }
#define AUX_AGENT 16
template <int MODE>   // bit 0: x through the L2 (acquire = buffer_inv sc1, then cached loads) instead of sc1 loads; bit 1: s_sleep between polls
__global__ void __launch_bounds__(256) k_flags(float * __restrict__ bufa, float * __restrict__ bufb, int n, int nstages, unsigned * __restrict__ flags, unsigned epoch0, unsigned * __restrict__ err) {
    __shared__ float red[4];
    const int G = gridDim.x, per = (n + G - 1) / G;
    const __amdgpu_buffer_rsrc_t rf = rsrc_of(flags, (unsigned) G * 4);
    for (int st = 0; st < nstages; ++st) {
        float * in = (st & 1) ? bufb : bufa; float * out = (st & 1) ? bufa : bufb;
        const __amdgpu_buffer_rsrc_t ri = rsrc_of(in, (unsigned) n * 4), ro = rsrc_of(out, (unsigned) n * 4);
        if (st > 0) {                                                     // wait: every flag >= epoch0 + st
            if (threadIdx.x < 64) {
                const unsigned want = epoch0 + (unsigned) st;
                long spins = 0; bool ok;
                do {
                    ok = true;
                    for (int j = threadIdx.x; j < G; j += 64) { const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rf, j * 4, 0, AUX_AGENT); ok = ok && (int) (v - want) >= 0; }
                    ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
                    if (!ok && (MODE & 2)) __builtin_amdgcn_s_sleep(2);
                } while (!ok && ++spins < 2000000L);
                if (!ok && threadIdx.x == 0) *err = 1;
            }
            __syncthreads();
            if (MODE & 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        float s = 0.0f;
        for (int i = threadIdx.x; i < n; i += 256) s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, i * 4, 0, (MODE & 1) ? 0 : AUX_AGENT));
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        const float tot = red[0] + red[1] + red[2] + red[3];
        for (int i = threadIdx.x; i < per; i += 256) { const int j = blockIdx.x * per + i; if (j < n) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tot * 1e-6f + (float) j), ro, j * 4, 0, AUX_AGENT); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b32(epoch0 + (unsigned) st + 1, rf, blockIdx.x * 4, 0, AUX_AGENT);
    }
}

int main() {
    const int n = 4096, nstages = 256;
    float * a, * b; unsigned * cnt, * flags, * err;
    CHECK(hipMalloc(&a, n * 4)); CHECK(hipMalloc(&b, n * 4)); CHECK(hipMalloc(&cnt, 4)); CHECK(hipMalloc(&flags, 4096 * 4)); CHECK(hipMalloc(&err, 4));
    CHECK(hipMemset(flags, 0, 4096 * 4)); CHECK(hipMemset(err, 0, 4));
    std::vector<float> ha(n), hc(n); unsigned epoch = 0;
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
        // (c) flag array; checked against the graph variant's result (same arithmetic, same start)
        float best_c[4] = { 1e30f, 1e30f, 1e30f, 1e30f }; int bad[4] = { -1, -1, -1, -1 };
        if (g <= 1024) {
            CHECK(hipMemset(a, 0, n * 4)); CHECK(hipMemset(b, 0, n * 4));
            CHECK(hipGraphLaunch(exec, st)); CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
            for (int m = 0; m < 4; ++m) {
                for (int r = 0; r < 5; ++r) {
                    CHECK(hipMemsetAsync(a, 0, n * 4, st)); CHECK(hipMemsetAsync(b, 0, n * 4, st));
                    CHECK(hipEventRecord(e0, st));
                    if (m == 0) k_flags<0><<<g, 256, 0, st>>>(a, b, n, nstages, flags, epoch, err);
                    if (m == 1) k_flags<1><<<g, 256, 0, st>>>(a, b, n, nstages, flags, epoch, err);
                    if (m == 2) k_flags<2><<<g, 256, 0, st>>>(a, b, n, nstages, flags, epoch, err);
                    if (m == 3) k_flags<3><<<g, 256, 0, st>>>(a, b, n, nstages, flags, epoch, err);
                    CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
                    epoch += nstages + 1;
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_c[m]) best_c[m] = ms;
                }
                CHECK(hipMemcpy(hc.data(), a, n * 4, hipMemcpyDeviceToHost));
                unsigned herr; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CHECK(hipMemset(err, 0, 4));
                bad[m] = herr ? -2 : 0; for (int i = 0; i < n && !herr; ++i) if (ha[i] != hc[i]) ++bad[m];
            }
        }
        printf("workgroups %5d: dependent stage = %6.2f us as a hipGraph kernel node, %6.2f us behind a grid barrier in one launch, flag array [sc1 x | L2 x | sc1 x + sleep | L2 x + sleep] = %5.2f %5.2f %5.2f %5.2f us (mismatches vs graph result: %d %d %d %d)\n",
               g, best_a * 1e3f / nstages, best_b * 1e3f / nstages, best_c[0] * 1e3f / nstages, best_c[1] * 1e3f / nstages, best_c[2] * 1e3f / nstages, best_c[3] * 1e3f / nstages, bad[0], bad[1], bad[2], bad[3]);
        CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
    }
    return 0;
}

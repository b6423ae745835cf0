// tools/dma_align.hip -- measurement: does LDS-DMA (buffer_load_dwordx4 ... lds) accept global addresses that are only 2-byte aligned, and at what cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
extern __shared__ char lds[];
typedef __attribute__((address_space(3))) void * lds_ptr;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// one wave: lane L fetches 16 bytes from global byte offset L * 16 + mis into LDS slot L; then the slots are copied out
__global__ void k_one(const char * W, int mis, uint32_t * out) {
    const int lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) W, (short) 0, 1 << 20, 0x00020000);
    const uint32_t v = lane * 16 + mis;
    const uint32_t m = (uint32_t) (uintptr_t) (lds_ptr) lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" :: "s"(m), "v"(v), "s"(rs) : "memory", "m0");
    __builtin_amdgcn_s_barrier();
    const u32x4 r = *(const u32x4 *) (lds + lane * 16);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = r[i];
}
template <int MIS>
__global__ void __launch_bounds__(64) k_stream(const char * W, size_t per_wave, float * out) {
    const int lane = threadIdx.x & 63;
    const char * base = W + (size_t) blockIdx.x * per_wave;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) base, (short) 0, (int) per_wave, 0x00020000);
    const uint32_t v16 = lane * 16 + MIS;
    const uint32_t ring = (uint32_t) (uintptr_t) (lds_ptr) lds;
    const int n = (int) (per_wave / 4096) - 1;
    for (int i = 0; i < n; ++i) {
        const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t) i * 4096u);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 nt lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:2048 nt lds\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen offset:3072 nt lds"
                     :: "s"(ring + (i & 1) * 4096), "v"(v16), "s"(rs), "s"(so) : "memory", "m0");
        asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) out[blockIdx.x] = (float) lds[0];
}
int main() {
    const size_t total = (size_t) 1 << 30;
    char * W; uint32_t * out; float * fo;
    CHECK(hipMalloc(&W, total)); CHECK(hipMalloc(&out, 4096)); CHECK(hipMalloc(&fo, 1 << 20));
    std::vector<unsigned char> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char) (i * 7 + (i >> 8));
    CHECK(hipMemcpy(W, h.data(), h.size(), hipMemcpyHostToDevice));
    for (int mis : { 0, 2, 4, 6, 10, 14 }) {
        k_one<<<1, 64, 4096>>>(W, mis, out);
        std::vector<uint32_t> r(256);
        CHECK(hipMemcpy(r.data(), out, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int b = 0; b < 1024; ++b) if (((unsigned char *) r.data())[b] != h[b + mis]) ++bad;
        printf("misalignment %2d bytes: %s (%d bytes differ)\n", mis, bad ? "WRONG" : "data correct", bad);
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t per_wave = total / 256 / 4096 * 4096;
    auto run = [&](auto kern, int mis) {
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) { CHECK(hipEventRecord(e0)); kern<<<256, 64, 16384>>>(W, per_wave, fo); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
        printf("stream, one loader wave per CU, misalignment %2d: %7.1f us  %5.2f TB/s\n", mis, best * 1e3, (double) per_wave * 256 / (best * 1e3) / 1e6);
    };
    run(k_stream<0>, 0); run(k_stream<2>, 2); run(k_stream<6>, 6); run(k_stream<14>, 14);
    return 0;
}

// tools/dma_bench.hip -- measurement (not part of the product): what limits ONE wave streaming memory into LDS by LDS-DMA?
// One workgroup per CU, W loader waves each issuing 1 KiB `buffer_load_dwordx4 ... lds` instructions over its own slice; variants:
//   M0 written once / before every instruction / every 4th; window = instructions kept in flight (s_waitcnt vmcnt).
//   build: hipcc --offload-arch=gfx950 -O3 tools/dma_bench.hip -o build/dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
extern __shared__ char lds[];
typedef __attribute__((address_space(3))) void * lds_ptr;

// MODE 0: M0 set once per 4 instructions (offsets 0..3072); MODE 1: M0 set before every instruction; MODE 2: M0 constant for all (same 4 KiB overwritten)
template <int MODE, int WIN>
__global__ void __launch_bounds__(1024) k_dma(const char * W, size_t per_wave, float * out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + wave);
    const char * base = W + (size_t) gw * per_wave;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) base, (short) 0, (int) per_wave, 0x00020000);
    const uint32_t v16 = lane * 16;
    const uint32_t ring = (uint32_t) (uintptr_t) (lds_ptr) lds + wave * 16384;        // (16 KiB per wave: the stride-2048 modes write 8 KiB per group, everything stays inside the allocation)
    const int n = (int) (per_wave / (MODE == 3 ? 2304 : 4096));
    for (int i = 0; i < n; ++i) {
        const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t) i * 4096u);
        const uint32_t m = ring + (MODE == 2 ? 0 : (i & 1) * ((MODE == 1 || MODE == 10) ? 8192 : 4096));
        if (MODE == 3) {
            const uint32_t so3 = __builtin_amdgcn_readfirstlane((uint32_t) i * 2304u);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen nt lds\n\t"
                         "buffer_load_dwordx4 %1, %3, %4 offen offset:1024 nt lds\n\t"
                         "buffer_load_dword %2, %3, %4 offen offset:2048 nt lds"
                         :: "s"(m), "v"(v16), "v"(lane * 4), "s"(rs), "s"(so3) : "memory", "m0");
        } else if (MODE == 1) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds\n\t"
                         "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:1024 nt lds\n\t"
                         "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:2048 nt lds\n\t"
                         "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:3072 nt lds"
                         :: "s"(m - 0), "v"(v16), "s"(rs), "s"(so) : "memory", "m0");
        } else if (MODE == 8) {                                    // mode 0 with the scalar padding of mode 1 between the loads (pacing only: same M0, same LDS addresses)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds\n\t"
                         "s_nop 0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:1024 nt lds\n\t"
                         "s_nop 0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:2048 nt lds\n\t"
                         "s_nop 0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:3072 nt lds"
                         :: "s"(m), "v"(v16), "s"(rs), "s"(so) : "memory", "m0");
        } else if (MODE == 9) {                                    // M0 rewritten before every load but the SAME LDS addresses as mode 0 (global offset through soffset)
            const uint32_t so1 = so + 1024u, so2 = so + 2048u, so3 = so + 3072u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds\n\t"
                         "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen nt lds\n\t"
                         "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %5 offen nt lds\n\t"
                         "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %6 offen nt lds"
                         :: "s"(m), "v"(v16), "s"(rs), "s"(so), "s"(so1), "s"(so2), "s"(so3) : "memory", "m0");
        } else if (MODE == 10) {                                   // LDS stride 2048 like mode 1 but written with ONE M0 per group?  not expressible (imm offset moves both): M0 stride 2048, global through soffset
            const uint32_t so1 = so + 1024u, so2 = so + 2048u, so3 = so + 3072u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds\n\t"
                         "s_add_u32 m0, m0, 2048\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen nt lds\n\t"
                         "s_add_u32 m0, m0, 2048\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %5 offen nt lds\n\t"
                         "s_add_u32 m0, m0, 2048\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %6 offen nt lds"
                         :: "s"(m), "v"(v16), "s"(rs), "s"(so), "s"(so1), "s"(so2), "s"(so3) : "memory", "m0");
        } else {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 nt lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %3 offen offset:2048 nt lds\n\t"
                         "buffer_load_dwordx4 %1, %2, %3 offen offset:3072 nt lds"
                         :: "s"(m), "v"(v16), "s"(rs), "s"(so) : "memory", "m0");
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WIN) : "memory");
        if (MODE == 4 && lane == 0) *(volatile uint32_t *) (lds + 65536) = (uint32_t) i;                        // generic pointer: flat_store + vmcnt(0)
        if (MODE == 5 && lane == 0) *(volatile __attribute__((address_space(3))) uint32_t *) (lds + 65536) = (uint32_t) i;     // ds_write_b32
        if (MODE == 6) { const uint32_t f = *(const volatile __attribute__((address_space(3))) uint32_t *) (lds + 65540); if (__builtin_amdgcn_readfirstlane(f) == 0x12345u) break; }   // ds_read_b32 + wait
        if (MODE == 7) { if (lane == 0) *(volatile __attribute__((address_space(3))) uint32_t *) (lds + 65536) = (uint32_t) i;
                         const uint32_t f = *(const volatile __attribute__((address_space(3))) uint32_t *) (lds + 65540); if (__builtin_amdgcn_readfirstlane(f) == 0x12345u) break; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) out[gw] = (float) lds[ring & 0xffff];
}
static hipStream_t st; static hipEvent_t e0, e1;
template <typename F> static double timeit(F f) {
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) { CHECK(hipEventRecord(e0, st)); f(); CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    return best * 1e3;
}
template <int MODE, int WIN> static void run(const char * W, float * out, int waves, size_t total) {
    const size_t per_wave = total / (256 * waves) / 4096 * 4096;
    const size_t ldsb = (size_t) waves * 16384;
    CHECK(hipFuncSetAttribute((const void *) k_dma<MODE, WIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    const double us = timeit([&] { k_dma<MODE, WIN><<<256, 64 * waves, ldsb, st>>>(W, per_wave, out); });
    printf("  mode %d (M0 %s) window %2d instr, %2d loader wave(s)/CU: %8.1f us  %6.2f TB/s  (%5.1f GB/s per wave)\n", MODE, MODE == 0 ? "per 4 KiB" : MODE == 1 ? "per instr" : MODE == 2 ? "constant " : MODE == 3 ? "q4k step " : MODE == 4 ? "+flat st " : MODE == 5 ? "+ds_write" : MODE == 6 ? "+ds_read " : MODE == 7 ? "+ds rd/wr" : MODE == 8 ? "m0 + nops " : MODE == 9 ? "per instr, LDS stride 1024" : "per instr, LDS stride 2048", WIN, waves, us,
           (double) per_wave * 256 * waves / us / 1e6, (double) per_wave / us / 1e3);
}
int main() {
    const size_t total = (size_t) 1 << 30;
    char * W; float * out;
    CHECK(hipMalloc(&W, total)); CHECK(hipMalloc(&out, 1 << 20)); CHECK(hipMemset(W, 1, total));
    CHECK(hipStreamCreate(&st)); CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int waves : { 1, 2 }) {
        printf("%d loader wave(s) per CU, 1 GiB total:\n", waves);
        run<0, 4>(W, out, waves, total); run<0, 12>(W, out, waves, total); run<0, 28>(W, out, waves, total); run<0, 56>(W, out, waves, total);
        run<1, 28>(W, out, waves, total); run<1, 56>(W, out, waves, total);
        run<8, 28>(W, out, waves, total); run<9, 28>(W, out, waves, total); run<10, 28>(W, out, waves, total); run<0, 28>(W, out, waves, total); run<1, 28>(W, out, waves, total);
        run<2, 28>(W, out, waves, total); run<2, 56>(W, out, waves, total);
        run<3, 27>(W, out, waves, total); run<3, 57>(W, out, waves, total); run<4, 28>(W, out, waves, total); run<5, 28>(W, out, waves, total); run<5, 56>(W, out, waves, total); run<6, 28>(W, out, waves, total); run<6, 56>(W, out, waves, total); run<7, 56>(W, out, waves, total);
    }
    return 0;
}

// tools/tr_probe.hip -- what ds_read_b64_tr_b16 returns on gfx950: LDS holds the index of every 16-bit cell; every lane passes an address, and each of the four
// 16-bit values it gets back is printed as (lane whose address it came from, cell within that lane's 8 bytes).
//   hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_probe2(uint16_t * out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t) i;
    __syncthreads();
    const int lane = threadIdx.x;
    const uint32_t addr = (uint32_t) (uintptr_t) lds + (mode == 0 ? lane * 8 : lane * 256);
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    u2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[lane * 4 + 0] = (uint16_t) (r[0] & 0xffff); out[lane * 4 + 1] = (uint16_t) (r[0] >> 16);
    out[lane * 4 + 2] = (uint16_t) (r[1] & 0xffff); out[lane * 4 + 3] = (uint16_t) (r[1] >> 16);
}

int main() {
    uint16_t * d; hipMalloc(&d, 64 * 4 * 2);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k_probe2, dim3(1), dim3(64), 0, 0, d, mode);
        std::vector<uint16_t> h(256);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("mode %d (lane l passes %s): value i of lane l = (source lane, cell)\n", mode, mode == 0 ? "byte 8 l" : "byte 256 l");
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int i = 0; i < 4; ++i) {
                const int cell = h[l * 4 + i];
                const int src = mode == 0 ? cell / 4 : cell / 128, c = mode == 0 ? cell % 4 : cell % 128;
                printf("  (%2d,%d)", src, c);
            }
            printf("\n");
        }
    }
    return 0;
}

// tools/mmv_lab.hip -- measurement bench (not part of the product) for the batch-1 decode mat-vec kernels: the mmvk.hip launches of round 1
// (stand-alone norm / quantise kernel + mat-vec) against the mmv1.hip family (f32 activation in, image built in the prologue) in every
// decode shape of Qwen3-8B Q4_K_M, as nodes of a replayed hipGraph with rotating weights (nothing cache-resident), plus a result check.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/mmv_lab.hip -o build/mmv_lab      run: build/mmv_lab
#include "../llama.cpp-omni_amd/csrc/kernels/quantize.hip"
#include "../llama.cpp-omni_amd/csrc/kernels/mmvk.hip"
#include "../llama.cpp-omni_amd/csrc/kernels/mmv1.hip"
#include "../llama.cpp-omni_amd/csrc/kernels/mmv1q.hip"
#include <vector>
#include <string>
#include <functional>
#include <cmath>

using namespace mi;

__global__ void k_fill(uint32_t * p, size_t n32, uint32_t seed) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t) i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = h;
    }
}
// make the f16 scale fields of every block finite and small: Q4_K d, dmin at bytes 0..3; Q6_K d at byte 208
__global__ void k_fix_scales(char * p, size_t nblk, int bs, int off, int nf16) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < nblk; i += (size_t) gridDim.x * blockDim.x) {
        uint16_t * d = (uint16_t *) (p + i * bs + off);
        for (int k = 0; k < nf16; ++k) d[k] = (uint16_t) (0x1c00 + ((i * 7 + k * 13) & 0x3ff));     // ~ 2^-8 .. 2^-7
    }
}
__global__ void k_fill_f32(float * p, size_t n, uint32_t seed, float amp) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t) i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = amp * ((float) (h & 0xffffff) / 8388608.0f - 1.0f);
    }
}

static hipStream_t st;
static hipEvent_t e0, e1;

static double time_graph(int N, const std::function<void(int)> & launch) {
    for (int s = 0; s < 3; ++s) launch(s);                                // function attributes, warm-up
    HIP_CHECK(hipStreamSynchronize(st));
    hipGraph_t graph; hipGraphExec_t exec;
    HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < N; ++s) launch(s);
    HIP_CHECK(hipStreamEndCapture(st, &graph));
    HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        HIP_CHECK(hipEventRecord(e0, st)); HIP_CHECK(hipGraphLaunch(exec, st)); HIP_CHECK(hipEventRecord(e1, st)); HIP_CHECK(hipEventSynchronize(e1));
        float ms; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    HIP_CHECK(hipGraphExecDestroy(exec)); HIP_CHECK(hipGraphDestroy(graph));
    return best * 1e3 / N;
}

template <int NW, int U, int DEPTH, int TM, bool PAIR, bool NT>
static void go_mv1(const mv1_dev & d, int grid) {
    const size_t lds = mv1_image_bytes(d.K);
    constexpr int XA = (16 + NW - 1) / NW, XC = NW >= 8 ? (48 + NW - 1) / NW : 1;
    if (d.K <= 4096) k_mv1<NW, XA, U, DEPTH, TM, PAIR, NT><<<dim3(grid), dim3(64 * NW), lds, st>>>(d);
    else if (NW >= 8) k_mv1<NW, XC, U, DEPTH, TM, PAIR, NT><<<dim3(grid), dim3(64 * NW), lds, st>>>(d);
    else { fprintf(stderr, "no instance\n"); abort(); }
    HIP_CHECK(hipGetLastError());
}

struct shape { const char * name; int type; int K; int nmat; int nrows[3]; int types[3]; bool pair; bool norm; bool resid; };

int main(int argc, char ** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    HIP_CHECK(hipStreamCreate(&st));
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    const size_t ARENA = (size_t) 768 << 20;
    char * a4, * a6;
    HIP_CHECK(hipMalloc(&a4, ARENA)); HIP_CHECK(hipMalloc(&a6, ARENA));
    k_fill<<<4096, 256, 0, st>>>((uint32_t *) a4, ARENA / 4, 1u); k_fill<<<4096, 256, 0, st>>>((uint32_t *) a6, ARENA / 4, 2u);
    k_fix_scales<<<4096, 256, 0, st>>>(a4, ARENA / 144, 144, 0, 2); k_fix_scales<<<4096, 256, 0, st>>>(a6, ARENA / 210, 210, 208, 1);
    float * x, * nw, * resid, * out_a, * out_b; char * img;
    HIP_CHECK(hipMalloc(&x, 12288 * 4)); HIP_CHECK(hipMalloc(&nw, 12288 * 4)); HIP_CHECK(hipMalloc(&resid, 12288 * 4 * 4));
    HIP_CHECK(hipMalloc(&out_a, 160000 * 4)); HIP_CHECK(hipMalloc(&out_b, 160000 * 4)); HIP_CHECK(hipMalloc(&img, 65536));
    k_fill_f32<<<64, 256, 0, st>>>(x, 12288, 11u, 1.0f); k_fill_f32<<<64, 256, 0, st>>>(nw, 12288, 12u, 1.0f); k_fill_f32<<<64, 256, 0, st>>>(resid, 12288 * 4, 13u, 1.0f);
    HIP_CHECK(hipStreamSynchronize(st));

    const int Q4 = GGML_TYPE_Q4_K, Q6 = GGML_TYPE_Q6_K;
    const shape shapes[] = {
        { "gate/up pair Q4_K 12288x4096 (56.6 MB) + norm",   Q4, 4096, 1, { 12288, 0, 0 }, { Q4, 0, 0 }, true,  true,  false },
        { "qkv Q4_K 4096+1024+1024 x4096 (14.2 MB) + norm",  Q4, 4096, 3, { 4096, 1024, 1024 }, { Q4, Q4, Q4 }, false, true, false },
        { "qkv Q4_K/Q6_K v (15.2 MB) + norm",                Q4, 4096, 3, { 4096, 1024, 1024 }, { Q4, Q4, Q6 }, false, true, false },
        { "wo Q4_K 4096x4096 (9.4 MB) + resid",              Q4, 4096, 1, { 4096, 0, 0 }, { Q4, 0, 0 }, false, false, true },
        { "down Q4_K 4096x12288 (28.3 MB) + resid",          Q4, 12288, 1, { 4096, 0, 0 }, { Q4, 0, 0 }, false, false, true },
        { "down Q6_K 4096x12288 (41.3 MB) + resid",          Q6, 12288, 1, { 4096, 0, 0 }, { Q6, 0, 0 }, false, false, true },
        { "lm-head Q6_K 151936x4096 (510 MB) + norm",        Q6, 4096, 1, { 151936, 0, 0 }, { Q6, 0, 0 }, false, true, false },
    };
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    int si = -1;
    for (const shape & S : shapes) {
        ++si;
        if (only >= 0 && si != only) continue;
        const int K = S.K, nb = K / 256;
        size_t mbytes[3] = { 0, 0, 0 }, total = 0;
        for (int i = 0; i < S.nmat; ++i) { mbytes[i] = (size_t) S.nrows[i] * nb * (S.types[i] == Q4 ? 144 : 210); total += mbytes[i]; }
        if (S.pair) total *= 2;
        // rotation: slices of the arenas, stride = the launch's bytes rounded up to 1 MB; at least 600 MB before anything repeats
        const size_t stride = ((total + (1 << 20) - 1) >> 20) << 20;
        int nrot = (int) std::max<size_t>(1, std::min<size_t>(ARENA / stride, 64));
        if (getenv("LAB_NROT")) nrot = std::min(nrot, atoi(getenv("LAB_NROT")));     // few sets: the weights stay in the 256 MB Infinity Cache
        const int N = total > (100u << 20) ? 8 : 48;
        printf("\n== %s : %.1f MB per launch, %d rotating weight sets\n", S.name, total / 1e6, nrot);
        auto wptr = [&](int s, int i, bool second) -> const char * {
            size_t off = (size_t) (s % nrot) * stride;
            for (int k = 0; k < i; ++k) off += mbytes[k];
            if (second) off += mbytes[0];
            const int t = S.types[i];
            off = off / (t == Q4 ? 144 : 210) * (t == Q4 ? 144 : 210);           // block-aligned inside the arena
            if (t == Q4) off = off / 16 * 16;
            return (t == Q4 ? a4 : a6) + off;
        };
        // NOTE: Q4_K slices must start 16-B aligned AND on a block boundary: 144 = 9 * 16, fine.  Q6_K rows (nb * 210) keep 2-B alignment.

        // ---- baseline: round-1 launches
        auto base = [&](int s, float * out) {
            if (S.norm) rms_norm_mul_quant(x, K * 4, nw, nullptr, 0, img, K, 1, 1e-6f, st);
            else        quantize_q8k_image(x, K * 4, img, K, 1, st);
            if (S.pair) mmv_kquant_pair_swiglu(S.types[0], wptr(s, 0, false), wptr(s, 0, true), (size_t) nb * 144, img, q8k_image_bytes(K), out, 0, K, S.nrows[0], 1, st);
            else {
                mmv_multi_args a; a.nmat = S.nmat; a.act = img; a.act_cs = q8k_image_bytes(K); a.K = K; a.ncols = 1;
                size_t o = 0;
                for (int i = 0; i < S.nmat; ++i) {
                    a.m[i] = { wptr(s, i, false), (size_t) nb * (S.types[i] == Q4 ? 144 : 210), out + o, 0, S.resid ? resid + o : nullptr, 0, S.nrows[i], S.types[i] };
                    o += S.nrows[i];
                }
                mmv_kquant_multi(a, st);
            }
        };
        auto mk = [&](int s, float * out, int nwaves) {
            mv1_dev d; d.nmat = S.nmat; d.K = K; d.W1 = S.pair ? wptr(s, 0, true) : nullptr;
            d.src = { x, S.norm ? nw : nullptr, 1e-6f, nullptr };
            size_t o = 0; double acc_b = 0; int acc_w = 0; double tb = 0;
            for (int i = 0; i < S.nmat; ++i) tb += (double) mbytes[i];
            for (int i = 0; i < 3; ++i) {
                if (i >= S.nmat) { d.m[i] = d.m[0]; d.m[i].wave_end = nwaves; continue; }
                acc_b += (double) mbytes[i];
                int end = i == S.nmat - 1 ? nwaves : (int) (nwaves * (acc_b / tb) + 0.5);
                if (end <= acc_w) end = acc_w + 1;
                d.m[i] = { wptr(s, i, false), (size_t) nb * (S.types[i] == Q4 ? 144 : 210), (char *) (out + o), S.resid ? (const char *) (resid + o) : nullptr, S.nrows[i], S.types[i], end };
                acc_w = end; o += S.nrows[i];
            }
            return d;
        };
        int ntot = 0; for (int i = 0; i < S.nmat; ++i) ntot += S.nrows[i];
        std::vector<float> ha(ntot), hb(ntot);
        auto check = [&](const char * nm, const std::function<void(int, float *)> & f) {
            HIP_CHECK(hipMemsetAsync(out_a, 0, ntot * 4, st)); HIP_CHECK(hipMemsetAsync(out_b, 0xff, ntot * 4, st));
            base(1, out_a); f(1, out_b);
            HIP_CHECK(hipMemcpyAsync(ha.data(), out_a, ntot * 4, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipMemcpyAsync(hb.data(), out_b, ntot * 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            double num = 0, den = 0; int nbad = 0;
            for (int i = 0; i < ntot; ++i) { const double dlt = (double) ha[i] - hb[i]; num += dlt * dlt; den += (double) ha[i] * ha[i]; if (!(std::fabs(dlt) <= 1e-3 * (std::fabs(ha[i]) + 1e-2))) ++nbad; }
            char b[64]; snprintf(b, sizeof b, "%s nmse %.1e", nbad == 0 ? "ok" : "MISMATCH", num / (den + 1e-30)); return std::string(b);
        };
        const double tb = time_graph(N, [&](int s) { base(s, out_a); });
        printf("   %-58s %7.2f us  (%.2f TB/s)\n", "round-1: norm/quantise launch + mmvk launch", tb, total / tb / 1e6);

#define VAR(NW, U, DEPTH, NT, WAVES)                                                                                           \
        do {                                                                                                                   \
            const int waves = (WAVES);                                                                                         \
            auto f = [&](int s, float * out) {                                                                                 \
                const mv1_dev d = mk(s, out, waves);                                                                           \
                if (S.pair) { if (S.types[0] == Q4) go_mv1<NW, U, DEPTH, 1, true, NT>(d, waves / NW); else go_mv1<NW, U, DEPTH, 2, true, false>(d, waves / NW); } \
                else if (S.nmat == 1 && S.types[0] == Q4) go_mv1<NW, U, DEPTH, 1, false, NT>(d, waves / NW);                     \
                else if (S.nmat == 1) go_mv1<NW, U, DEPTH, 2, false, false>(d, waves / NW);                                      \
                else go_mv1<NW, U, DEPTH, 3, false, NT>(d, waves / NW);                                                          \
            };                                                                                                                 \
            if ((K > 4096 && NW < 8) || (S.pair && U == 1)) break;                                                                         \
            const std::string c = check("", f);                                                                                \
            const double t = time_graph(N, [&](int s) { f(s, out_b); });                                                        \
            char nm[96]; snprintf(nm, sizeof nm, "mv1 NW=%d U=%d DEPTH=%d NT=%d waves=%d", NW, U, DEPTH, NT, waves);            \
            printf("   %-58s %7.2f us  (%.2f TB/s)  %s\n", nm, t, total / t / 1e6, c.c_str());                                  \
        } while (0)

        const bool big = ntot > 20000;
#if defined(LAB_FEW)       // fewer resident waves, deeper register prefetch (one workgroup per CU keeps 256 VGPRs per lane)
        VAR(8, 2, 1, 0, 4096);
        VAR(8, 2, 2, 0, 2048);
        VAR(8, 2, 3, 0, 2048);
        VAR(8, 2, 4, 0, 2048);
        VAR(8, 2, 5, 0, 2048);
        VAR(16, 1, 1, 0, 4096);
        VAR(16, 1, 2, 0, 2048);
        VAR(16, 1, 4, 0, 2048);
        VAR(16, 1, 6, 0, 2048);
        VAR(8, 1, 4, 0, 2048);
        VAR(8, 1, 6, 0, 1024);
#elif defined(LAB_DEPTH)     // prefetch depth with the default cache policy
        VAR(8, 2, 1, 0, 4096);
        VAR(8, 2, 2, 0, 4096);
        VAR(8, 2, 3, 0, 4096);
        VAR(16, 1, 1, 0, 4096);
        VAR(16, 1, 2, 0, 4096);
        VAR(16, 1, 3, 0, 4096);
#elif defined(LAB_ONE)
        VAR(8, 2, 1, 0, 4096);
        VAR(8, 2, 1, 1, 4096);
        VAR(8, 2, 2, 1, 4096);
        VAR(8, 2, 3, 1, 4096);
        VAR(16, 1, 1, 0, 4096);
        VAR(16, 1, 1, 1, 4096);
        VAR(16, 1, 3, 1, 4096);
#else
        VAR(8, 2, 1, 0, 4096);
        VAR(8, 2, 2, 0, 4096);
        VAR(8, 2, 3, 0, 4096);
        VAR(16, 2, 3, 0, 4096);
        VAR(4, 2, 3, 0, 4096);
        VAR(8, 2, 3, 1, 4096);
        VAR(8, 1, 1, 0, 4096);
        VAR(8, 1, 3, 0, 4096);
        VAR(16, 1, 1, 0, 4096);
        VAR(16, 1, 2, 0, 4096);
        VAR(16, 1, 3, 0, 4096);
        VAR(16, 1, 3, 0, 8192);
        VAR(8, 1, 3, 0, 8192);
#endif
        (void) big;
    }
    return 0;
}

// tools/mmv2_lab.hip -- measurement bench (not part of the product): the dense-load LDS-DMA decode mat-vec (mmv2.hip) against the register-load
// family (mmv1.hip) in every decode shape of Qwen3-8B Q4_K_M, as nodes of a replayed hipGraph with rotating weights (nothing cache-resident),
// with a result check, and -- built with -DMV2_TRACE -- a per-wave time line of one launch (s_memrealtime stamps).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DMV2_TRACE] tools/mmv2_lab.hip -o build/mmv2_lab      run: build/mmv2_lab [shape]
#include "../llama.cpp-omni_amd/csrc/kernels/quantize.hip"
#include "../llama.cpp-omni_amd/csrc/kernels/mmv1.hip"
#include "../llama.cpp-omni_amd/csrc/kernels/mmv1q.hip"
#include "../llama.cpp-omni_amd/csrc/kernels/mmv2.hip"
#include "../llama.cpp-omni_amd/csrc/kernels/fattn_one.hip"
#include <vector>
#include <string>
#include <functional>
#include <algorithm>
#include <cmath>

using namespace mi;

__global__ void k_fill(uint32_t * p, size_t n32, uint32_t seed) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t) i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = h;
    }
}
__global__ void k_fix_scales(char * p, size_t nblk, int bs, int off, int nf16) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < nblk; i += (size_t) gridDim.x * blockDim.x) {
        uint16_t * d = (uint16_t *) (p + i * bs + off);
        for (int k = 0; k < nf16; ++k) d[k] = (uint16_t) (0x1c00 + ((i * 7 + k * 13) & 0x3ff));
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
    for (int s = 0; s < 3; ++s) launch(s);
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

struct shape { const char * name; int K; int nmat; int nrows[3]; int types[3]; bool pair; bool norm; bool resid; };

#ifdef MV2_TRACE
static unsigned long long * trace_dev = nullptr;
static void trace_report(const char * nm, int nwg, int nl, int NW) {
    const int nwaves = nwg * NW;
    std::vector<unsigned long long> h((size_t) nwaves * 8);
    HIP_CHECK(hipMemcpy(h.data(), trace_dev, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < nwaves; ++w) if (h[(size_t) w * 8]) t0 = std::min(t0, h[(size_t) w * 8]);
    static const char * labl[8] = { "start", "setup done", "first issue", "rows landed", "-", "-", "-", "end" };
    static const char * labr[8] = { "start", "setup done", "before 1st DMA", "1st DMA issued", "(consume starts)", "all DMA issued", "-", "end" };
    static const char * labc[8] = { "start", "setup done", "rows seen", "scale known", "image done", "my blocks done", "-", "end" };
    { printf("      wave -> SIMD of workgroup 0 (HW_ID bits 4-5):"); for (int w = 0; w < NW; ++w) printf(" %d", (int) ((h[(size_t) w * 8 + 6] >> 4) & 3)); printf("   CU ids of WG 0..3: %d %d %d %d\n", (int) ((h[6] >> 8) & 15), (int) ((h[NW * 8 + 6] >> 8) & 15), (int) ((h[2 * NW * 8 + 6] >> 8) & 15), (int) ((h[3 * NW * 8 + 6] >> 8) & 15)); }
    if (getenv("MV2_DUMP")) {                    // per-workgroup: XCC id, loader start / first issue / end, last consumer end (us after the first wave's start)
        printf("      per-workgroup dump of %s: wg xcc se cu  loader_start first_issue loader_end  last_consumer_end\n", nm);
        for (int g = 0; g < nwg; ++g) {
            const unsigned long long * L = &h[(size_t) g * NW * 8];
            double cend = 0; for (int w = 1; w < NW; ++w) cend = std::max(cend, (double) (h[((size_t) g * NW + w) * 8 + 7] - t0) * 0.01);
            printf("      WG %3d %2d %2d %2d  %6.2f %6.2f %6.2f  %6.2f\n", g, (int) (L[5] & 15), (int) ((L[6] >> 13) & 7), (int) ((L[6] >> 8) & 15), (double) (L[0] - t0) * 0.01, (double) (L[2] - t0) * 0.01, (double) (L[7] - t0) * 0.01, cend);
        }
    }
    if (getenv("MV2_WG0")) {                     // every wave of workgroups 0 and 1: the eight stamps (us after the workgroup's first wave)
        for (int g = 0; g < 2; ++g) {
            unsigned long long w0 = ~0ull; for (int w = 0; w < NW; ++w) if (h[((size_t) g * NW + w) * 8]) w0 = std::min(w0, h[((size_t) g * NW + w) * 8]);
            printf("      workgroup %d, per wave (stamps 0 1 2 3 4 5 7; loader = wave 0, row waves = the last four):\n", g);
            for (int w = 0; w < NW; ++w) { printf("        wave %2d:", w); for (int i = 0; i < 8; ++i) { if (i == 6) continue; const unsigned long long v = h[((size_t) g * NW + w) * 8 + i]; if (v && i != 5) printf(" %6.2f", (double) (v - w0) * 0.01); else if (i == 5 && w > 0 && v) printf(" %6.2f", (double) (v - w0) * 0.01); else printf("      -"); } printf("\n"); }
        }
    }
    printf("      time line of %s (us after the first wave's start; min / median / max)\n", nm);
    for (int role = 0; role < 3; ++role) for (int i = 0; i < 8; ++i) {
        const char * const * lab = role == 0 ? labl : role == 1 ? labc : labr;
        if (lab[i][0] == '-') continue;
        std::vector<double> v;
        for (int w = 0; w < nwaves; ++w) if ((role == 0 ? (w % NW) < nl : role == 1 ? ((w % NW) >= nl && (w % NW) < 5) : (w % NW) >= NW - 4) && h[(size_t) w * 8 + i]) v.push_back((double) (h[(size_t) w * 8 + i] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("        %-9s %-20s %6.2f / %6.2f / %6.2f\n", role == 0 ? "loader" : role == 1 ? "consumer" : "row wave", lab[i], v[0], v[v.size() / 2], v[v.size() - 1]);
    }
}
#endif

int main(int argc, char ** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    HIP_CHECK(hipStreamCreate(&st));
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    const size_t ARENA = (size_t) 768 << 20;
    char * a4, * a6;
    HIP_CHECK(hipMalloc(&a4, ARENA)); HIP_CHECK(hipMalloc(&a6, ARENA));
    k_fill<<<4096, 256, 0, st>>>((uint32_t *) a4, ARENA / 4, 1u); k_fill<<<4096, 256, 0, st>>>((uint32_t *) a6, ARENA / 4, 2u);
    k_fix_scales<<<4096, 256, 0, st>>>(a4, ARENA / 144, 144, 0, 2); k_fix_scales<<<4096, 256, 0, st>>>(a6, ARENA / 210, 210, 208, 1);
    float * x, * nw, * resid, * out_a, * out_b;
    HIP_CHECK(hipMalloc(&x, 12288 * 4)); HIP_CHECK(hipMalloc(&nw, 12288 * 4)); HIP_CHECK(hipMalloc(&resid, 230000 * 4));
    HIP_CHECK(hipMalloc(&out_a, 230000 * 4)); HIP_CHECK(hipMalloc(&out_b, 230000 * 4));
    k_fill_f32<<<64, 256, 0, st>>>(x, 12288, 11u, 1.0f); k_fill_f32<<<64, 256, 0, st>>>(nw, 12288, 12u, 1.0f); k_fill_f32<<<64, 256, 0, st>>>(resid, 230000, 13u, 1.0f);
#ifdef MV2_TRACE
    HIP_CHECK(hipMalloc(&trace_dev, 8192 * 8 * 8));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(mv2_trace_buf), &trace_dev, sizeof trace_dev));
#endif
    HIP_CHECK(hipStreamSynchronize(st));
    mmv2_enable(false);                 // the baseline is the register-load family

    const int Q4 = GGML_TYPE_Q4_K, Q6 = GGML_TYPE_Q6_K;
    const shape shapes[] = {
        { "gate/up pair Q4_K 12288x4096 (56.6 MB) + norm",   4096, 1, { 12288, 0, 0 }, { Q4, 0, 0 }, true,  true,  false },
        { "qkv Q4_K 4096+1024+1024 x4096 (14.2 MB) + norm",  4096, 3, { 4096, 1024, 1024 }, { Q4, Q4, Q4 }, false, true, false },
        { "qkv Q4_K/Q6_K v (15.2 MB) + norm",                4096, 3, { 4096, 1024, 1024 }, { Q4, Q4, Q6 }, false, true, false },
        { "wo Q4_K 4096x4096 (9.4 MB) + resid",              4096, 1, { 4096, 0, 0 }, { Q4, 0, 0 }, false, false, true },
        { "down Q4_K 4096x12288 (28.3 MB) + resid",          12288, 1, { 4096, 0, 0 }, { Q4, 0, 0 }, false, false, true },
        { "big Q4_K 151936x4096 (350 MB) + norm",            4096, 1, { 151936, 0, 0 }, { Q4, 0, 0 }, false, true, false },
        { "huge Q4_K 225280x4096 (519 MB) + norm",           4096, 1, { 225280, 0, 0 }, { Q4, 0, 0 }, false, true, false },
        { "down Q6_K 4096x12288 (41.3 MB) + resid",          12288, 1, { 4096, 0, 0 }, { Q6, 0, 0 }, false, false, true },
        { "lm-head Q6_K 151936x4096 (510 MB) + norm",        4096, 1, { 151936, 0, 0 }, { Q6, 0, 0 }, false, true, false },
    };
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    if (only == 101) {      // row-wave sweep: ffn_down (K = 12288, Q4_K / Q6_K, 10 waves), wo (K = 4096, 10 waves)
        mmv2_enable(true);
        for (int ty = 0; ty < 3; ++ty) {
            const int K = ty == 2 ? 4096 : 12288, M = 4096, nb = K / 256, bs = ty == 1 ? 210 : 144; const int T = ty == 1 ? GGML_TYPE_Q6_K : GGML_TYPE_Q4_K;
            const size_t wbytes = (size_t) M * nb * bs, stride = ((wbytes + (1 << 20) - 1) >> 20) << 20; const int nrot = (int) std::min<size_t>(ARENA / stride, 48);
            auto dev = [&](int s, float * out) {
                mv2_dev d{}; d.nmat = 1; d.K = K; d.W1 = nullptr; d.src = { x, nullptr, 0.0f, nullptr };
                const char * W = (ty == 1 ? a6 : a4) + (size_t) (s % nrot) * stride / (bs * 16) * (bs * 16);
                d.m[0] = { W, (char *) out, (const char *) resid, (uint32_t) (nb * bs), M, T, 0, M / 256, M % 256 };
                d.m[1] = d.m[0]; d.m[1].wg0 = 256; d.m[2] = d.m[1];
                return d;
            };
#define RWSWEEP(TMv, NITv, RW) { const double t = time_graph(48, [&](int s) { mv2_launch<TMv, NITv, false, true, 10, false, RW>(dev(s, out_b), 256, st); }); printf("   %s K %d, 10 waves, %d row waves: %.2f us\n", ty == 1 ? "Q6_K" : "Q4_K", K, RW, t); }
            if (ty == 0) { RWSWEEP(1, 3, 4) RWSWEEP(1, 3, 8) RWSWEEP(1, 3, 2) }
            if (ty == 1) { RWSWEEP(2, 3, 4) RWSWEEP(2, 3, 8) RWSWEEP(2, 3, 2) }
            if (ty == 2) { RWSWEEP(1, 1, 4) RWSWEEP(1, 1, 8) RWSWEEP(1, 1, 2) }
        }
        return 0;
    }
    if (only == 100) {      // wo (4096 x 4096 Q4_K + resid) on attention slices' partial states (k_mv2 PARTS) against merge kernel + the plain launch
        const int K = 4096, M = 4096, NH = 32, D = 128, NSL = fattn_gs_nslice();
        float * parts, * xm; HIP_CHECK(hipMalloc(&parts, fattn_gs_parts_bytes(NH, D))); HIP_CHECK(hipMalloc(&xm, K * 4));
        std::vector<float> hp(fattn_gs_parts_bytes(NH, D) / 4);
        uint32_t rng = 12345u; auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return (float) (rng >> 8) / 16777216.0f; };
        for (int s = 0; s < NSL; ++s) for (int i = 0; i < K; ++i) hp[(size_t) s * K + i] = (rnd() - 0.5f) * (s == 3 ? 0.0f : 4.0f);
        for (int s = 0; s < NSL; ++s) for (int h = 0; h < NH; ++h) { float * ms = &hp[(size_t) NSL * K + ((size_t) s * NH + h) * 2]; ms[0] = s == 3 ? -INFINITY : (rnd() - 0.5f) * 6.0f; ms[1] = s == 3 ? 0.0f : 1.0f + 20.0f * rnd(); if (h == 5) { ms[0] = -INFINITY; ms[1] = 0.0f; } }
        HIP_CHECK(hipMemcpy(parts, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
        const size_t wbytes = (size_t) M * 16 * 144, stride = ((wbytes + (1 << 20) - 1) >> 20) << 20; const int nrot = 48;
        auto launch = [&](int s, float * out, bool use_parts) {
            mv1_args v; v.nmat = 1; v.K = K; v.eps = 0.0f;
            v.m[0] = { a4 + (size_t) (s % nrot) * stride / 2304 * 2304, (size_t) 16 * 144, out, 0, resid, 0, M, GGML_TYPE_Q4_K };
            if (use_parts) { v.parts = parts; v.nslice = NSL; } else v.x = xm;
            mmv2(v, st);
        };
        mmv2_enable(true);
        fattn_gs_merge(parts, xm, NH, D, st);
        launch(1, out_a, false); launch(1, out_b, true);
        std::vector<float> ha(M), hb(M);
        HIP_CHECK(hipMemcpyAsync(ha.data(), out_a, M * 4, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipMemcpyAsync(hb.data(), out_b, M * 4, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipStreamSynchronize(st));
        double num = 0, den = 0; int nexact = 0; for (int i = 0; i < M; ++i) { const double d = (double) ha[i] - hb[i]; num += d * d; den += (double) ha[i] * ha[i]; nexact += ha[i] == hb[i]; }
        printf("wo on partial states vs merge launch + wo: nmse %.2e (%d / %d identical)\n", num / (den + 1e-30), nexact, M);
        const double t0 = time_graph(48, [&](int s) { launch(s, out_a, false); });
        const double t1 = time_graph(48, [&](int s) { launch(s, out_b, true); });
        printf("   wo 4096 x 4096 Q4_K + resid, plain row %.2f us | attention partial states folded in the prologue %.2f us\n", t0, t1);
#ifdef MV2_TRACE
        HIP_CHECK(hipMemsetAsync(trace_dev, 0, 8192 * 64, st)); for (int s_ = 0; s_ < 4; ++s_) launch(s_ + 7, out_b, false); HIP_CHECK(hipStreamSynchronize(st)); trace_report("wo, plain row (10 waves)", 256, 1, 10);
        HIP_CHECK(hipMemsetAsync(trace_dev, 0, 8192 * 64, st)); for (int s_ = 0; s_ < 4; ++s_) launch(s_ + 7, out_b, true); HIP_CHECK(hipStreamSynchronize(st)); trace_report("wo, partial states (10 waves, 8 row waves)", 256, 1, 10);
#endif
        return 0;
    }
    int si = -1;
    for (const shape & S : shapes) {
        ++si;
        if (only >= 0 && si != only) continue;
        const int K = S.K, nb = K / 256;
        size_t mbytes[3] = { 0, 0, 0 }, total = 0;
        for (int i = 0; i < S.nmat; ++i) { mbytes[i] = (size_t) S.nrows[i] * nb * (S.types[i] == Q4 ? 144 : 210); total += mbytes[i]; }
        if (S.pair) total *= 2;
        const size_t stride = ((total + (1 << 20) - 1) >> 20) << 20;
        int nrot = (int) std::max<size_t>(1, std::min<size_t>(ARENA / stride, 64));
        if (getenv("MV2_NROT")) nrot = std::max(1, std::min(nrot, atoi(getenv("MV2_NROT"))));      // 1: the same weights every launch (L2 / Infinity-Cache resident)
        const int N = total > (100u << 20) ? 8 : 48;
        printf("\n== %s : %.1f MB per launch, %d rotating weight sets\n", S.name, total / 1e6, nrot);
        auto wptr = [&](int s, int i, bool second) -> const char * {
            size_t off = (size_t) (s % nrot) * stride;
            for (int k = 0; k < i; ++k) off += mbytes[k];
            if (second) off += mbytes[0];
            const int t = S.types[i];
            off = off / (t == Q4 ? 144 : 210) * (t == Q4 ? 144 : 210);
            if (t == Q4) off = off / 2304 * 2304;                                  // 128-B aligned rows
            else         off = off / 3360 * 3360;
            return (t == Q4 ? a4 : a6) + off;
        };
        int ntot = 0; for (int i = 0; i < S.nmat; ++i) ntot += S.nrows[i];
        // the product's launch (mmv1)
        auto base = [&](int s, float * out) {
            mv1_args v; v.nmat = S.nmat; v.K = K; v.W_up = S.pair ? wptr(s, 0, true) : nullptr;
            v.x = x; v.norm_w = S.norm ? nw : nullptr; v.eps = 1e-6f;
            size_t o = 0;
            for (int i = 0; i < S.nmat; ++i) { v.m[i] = { wptr(s, i, false), (size_t) nb * (S.types[i] == Q4 ? 144 : 210), out + o, 0, S.resid ? resid + o : nullptr, 0, S.nrows[i], S.types[i] }; o += S.nrows[i]; }
            mmv1(v, st);
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
        auto mk2 = [&](int s, float * out, int grid) {
            const mv1_dev o = mk(s, out, grid);
            mv2_dev d; d.nmat = o.nmat; d.K = o.K; d.W1 = o.W1; d.src = o.src;
            int acc = 0;
            for (int i = 0; i < 3; ++i) {
                const int nwg = o.m[i].wave_end - acc;
                if (i >= S.nmat || nwg <= 0) { d.m[i] = d.m[0]; d.m[i].wg0 = grid; continue; }
                d.m[i] = { o.m[i].W, o.m[i].dst, o.m[i].resid, (uint32_t) o.m[i].w_rs, o.m[i].nrows, o.m[i].type, acc, o.m[i].nrows / nwg, o.m[i].nrows % nwg };
                acc = o.m[i].wave_end;
            }
            return d;
        };
        std::vector<float> ha(ntot), hb(ntot);
        auto check = [&](const std::function<void(int, float *)> & f) {
            HIP_CHECK(hipMemsetAsync(out_a, 0, ntot * 4, st)); HIP_CHECK(hipMemsetAsync(out_b, 0xff, ntot * 4, st));
            base(1, out_a); f(1, out_b);
            HIP_CHECK(hipMemcpyAsync(ha.data(), out_a, ntot * 4, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipMemcpyAsync(hb.data(), out_b, ntot * 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            double num = 0, den = 0; int nbad = 0, nexact = 0;
            for (int i = 0; i < ntot; ++i) { const double dlt = (double) ha[i] - hb[i]; num += dlt * dlt; den += (double) ha[i] * ha[i]; if (!(std::fabs(dlt) <= 1e-3 * (std::fabs(ha[i]) + 1e-2))) ++nbad; if (ha[i] == hb[i]) ++nexact; }
            char b[96]; snprintf(b, sizeof b, "%s nmse %.1e (%d/%d identical)", nbad == 0 ? "ok" : "MISMATCH", num / (den + 1e-30), nexact, ntot); return std::string(b);
        };
        const double tb = time_graph(N, [&](int s) { base(s, out_a); });
        printf("   %-58s %7.2f us  (%.2f TB/s)\n", "mmv1 (product launch)", tb, total / tb / 1e6);

#define VAR3(NT, NW)                                                                                                               \
        do {                                                                                                                   \
            const int grid = 256;                                                                                              \
            auto f = [&](int s, float * out) {                                                                                 \
                const mv2_dev d = mk2(s, out, grid);                                                                           \
                const bool q4 = S.types[0] == Q4 && (S.nmat == 1 || (S.types[1] == Q4 && S.types[2] == Q4));                    \
                const bool q6 = S.types[0] == Q6 && (S.nmat == 1 || (S.types[1] == Q6 && S.types[2] == Q6));                    \
                if (S.pair)         mv2_launch<1, 1, true, NT, NW>(d, grid, st);                                                    \
                else if (K == 4096) { if (q4) mv2_launch<1, 1, false, NT, NW>(d, grid, st); else if (q6) mv2_launch<2, 1, false, NT, NW>(d, grid, st); else mv2_launch<3, 1, false, NT, NW>(d, grid, st); } \
                else                { if (q4) mv2_launch<1, 3, false, NT, NW>(d, grid, st); else if (q6) mv2_launch<2, 3, false, NT, NW>(d, grid, st); else mv2_launch<3, 3, false, NT, NW>(d, grid, st); } \
            };                                                                                                                 \
            const std::string c = check(f);                                                                                    \
            const double t = time_graph(N, [&](int s) { f(s, out_b); });                                                        \
            char nm[96]; snprintf(nm, sizeof nm, "mv2 engine NT=%d waves=%d", NT, NW);                                                       \
            printf("   %-58s %7.2f us  (%.2f TB/s)  %s\n", nm, t, total / t / 1e6, c.c_str());                                  \
            TRACE_REPORT(nm, grid, 1, f, NW);                                                                                      \
        } while (0)
#ifdef MV2_TRACE
// (the stamps of the LAST launch of the replayed graph that was just timed: in-graph behaviour; MV2_TRACE_EAGER=1: four eager launches instead)
#define TRACE_REPORT(nm, grid, nl, f, NW) do { if (getenv("MV2_TRACE_EAGER")) { HIP_CHECK(hipMemsetAsync(trace_dev, 0, 8192 * 64, st)); for (int s_ = 0; s_ < 4; ++s_) f(s_ + 7, out_b); } HIP_CHECK(hipStreamSynchronize(st)); trace_report(nm, grid, nl, NW); } while (0)
#else
#define TRACE_REPORT(nm, grid, nl, f, NW) do { } while (0)
#endif
#ifdef MV2_LAB_ONE
        if (S.pair) VAR3(true, 16); else if (S.nmat > 1) VAR3(true, 12); else if (S.nrows[0] <= 16384) VAR3(true, 10); else VAR3(true, 16);
#else
        VAR3(true, 16);
#endif
#ifndef MV2_LAB_ONE
        VAR3(true, 13);
        VAR3(true, 12);
        VAR3(true, 10);
        VAR3(true, 9);
        VAR3(true, 8);
        VAR3(true, 6);
#endif
    }
    return 0;
}

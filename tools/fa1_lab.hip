// tools/fa1_lab.hip -- measurement bench (not part of the product): the one-token attention launch of a Qwen3-8B decode layer (k_fattn_one<128>: q / k chains,
// cache stores, 256 cache rows, 32 heads over 8 KV heads) as a node of a replayed hipGraph BEHIND a producer launch that writes its q / k / v rows (what the qkv
// mat-vec launch does in the model: the rows are cold in every L2), rotating over 36 layers' caches; built with -DFA1_TRACE: the per-wave time line of one launch.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=14 [-DFA1_TRACE] tools/fa1_lab.hip -o tools/bin/fa1_lab
#include "../llama.cpp-omni_amd/csrc/kernels/fattn_one.hip"
#include <vector>
#include <functional>
#include <algorithm>
#include <cmath>

using namespace mi;

__global__ void k_fill_f32(float * p, size_t n, uint32_t seed, float amp) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t) i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = amp * ((float) (h & 0xffffff) / 8388608.0f - 1.0f);
    }
}
__global__ void k_fill_f16(uint16_t * p, size_t n, uint32_t seed) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t) i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (uint16_t) (0x3000 + (h & 0x3ff) + ((h >> 10 & 1) << 15));      // +-[0.125, 0.25)
    }
}
// stands for the qkv mat-vec launch: 256 workgroups, each writes its 24 rows of the 6144-row q / k / v vector (values depend on the step so nothing is folded away)
__global__ void __launch_bounds__(256) k_producer(float * qkv, int step) {
    if (threadIdx.x < 24) qkv[blockIdx.x * 24 + threadIdx.x] = 0.01f * (float) ((blockIdx.x * 24 + threadIdx.x + step) % 97) - 0.4f;
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

int main(int argc, char ** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    HIP_CHECK(hipStreamCreate(&st)); HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    const int D = 128, NH = 32, NKVH = 8, NL = 36, NCTX = 512;
    const int nkv = argc > 1 ? atoi(argv[1]) : 256, cur = nkv > 100 ? 100 : nkv - 1;
    float * qkv, * qw, * kw, * tab, * dst; uint16_t * kc, * vc, * mask; long long * idx;
    HIP_CHECK(hipMalloc(&qkv, 6144 * 4)); HIP_CHECK(hipMalloc(&qw, D * 4)); HIP_CHECK(hipMalloc(&kw, D * 4)); HIP_CHECK(hipMalloc(&tab, D * 4)); HIP_CHECK(hipMalloc(&dst, 4096 * 4));
    const size_t cache_l = (size_t) NCTX * NKVH * D;
    HIP_CHECK(hipMalloc(&kc, cache_l * 2 * NL * 2)); vc = kc + cache_l * NL;      /* one buffer, as the model's KV cache: the kernel takes V as a 32-bit offset (x 16 B) from K */ HIP_CHECK(hipMalloc(&mask, 4096 * 2 * 64)); HIP_CHECK(hipMalloc(&idx, 16));
    k_fill_f32<<<8, 256, 0, st>>>(qw, D, 3u, 1.0f); k_fill_f32<<<8, 256, 0, st>>>(kw, D, 4u, 1.0f);
    k_fill_f16<<<1024, 256, 0, st>>>(kc, cache_l * NL, 5u); k_fill_f16<<<1024, 256, 0, st>>>(vc, cache_l * NL, 6u);
    std::vector<float> htab(D); for (int i = 0; i < D / 2; ++i) { htab[2 * i] = cosf(0.01f * i * cur); htab[2 * i + 1] = sinf(0.01f * i * cur); }
    std::vector<uint16_t> hmask(4096 * 64, 0xfc00); for (int r = 0; r < 64; ++r) for (int i = 0; i <= cur; ++i) hmask[(size_t) r * 4096 + i] = 0;
    long long hidx[2] = { cur, cur };
    HIP_CHECK(hipMemcpy(tab, htab.data(), D * 4, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(mask, hmask.data(), hmask.size() * 2, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(idx, hidx, 16, hipMemcpyHostToDevice));
#ifdef FA1_TRACE
    unsigned long long * trace_dev; HIP_CHECK(hipMalloc(&trace_dev, 1024 * 4 * 8 * 8)); HIP_CHECK(hipMemset(trace_dev, 0, 1024 * 4 * 8 * 8));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(fa1_trace_buf), &trace_dev, sizeof trace_dev));
#endif
    HIP_CHECK(hipStreamSynchronize(st));
    auto args = [&](int layer) {
        fa1_dev a{};
        a.qraw = (const char *) qkv; a.kraw = (const char *) (qkv + 4096); a.vraw = (const char *) (qkv + 5120); a.qw = qw; a.kw = kw; a.tab = tab;
        a.kcache = (char *) (kc + cache_l * layer); a.vcache = (char *) (vc + cache_l * layer); a.kidx = (const char *) idx; a.vidx = (const char *) (idx + 1);
        a.k = a.kcache; a.v = a.vcache; a.mask = (const char *) mask; a.sinks = nullptr; a.dst = (char *) dst;
        a.q_hs = D * 4; a.k_hs = D * 4; a.v_hs = D * 4; a.kc_rs = NKVH * D * 2; a.vc_rs = NKVH * D * 2;
        a.knb1 = NKVH * D * 2; a.knb2 = D * 2; a.vnb1 = NKVH * D * 2; a.vnb2 = D * 2; a.mnb2 = 0; a.mne2 = 1; a.dnb1 = D * 4;
        a.nkv = nkv; a.gq = NH / NKVH; a.neox = 1; a.n_head_log2 = 32; a.n_head = NH; a.nkvh_log2 = 3; a.vidx_st = 0; a.vidx_n = 0; a.has_norm = 1;
        a.nsplit = (nkv + FA1_NKV - 1) / FA1_NKV; a.part = nullptr; a.cnt = nullptr;
        a.eps = 1e-6f; a.scale = 1.0f / sqrtf((float) D); a.max_bias = 0.0f; a.logit_softcap = 0.0f; a.m0 = 1.0f; a.m1 = 1.0f;
        return a;
    };
    if (nkv > FA1_NKV) { printf("this lab runs one slice (n_kv <= 256)\n"); return 1; }
    auto attn = [&](int s) { const fa1_dev a = args(s % NL); k_fattn_one<128><<<dim3(NH), dim3(256), 0, st>>>(FA1_LEAD_ARGS(a), a); };
    float * parts, * dst2; HIP_CHECK(hipMalloc(&parts, fattn_gs_parts_bytes(NH, D))); HIP_CHECK(hipMalloc(&dst2, 4096 * 4));
    if (getenv("FA1_PTRS")) fprintf(stderr, "qkv %p qw %p kw %p tab %p dst %p kc %p vc %p mask %p idx %p parts %p (%zu B) dst2 %p\n", qkv, qw, kw, tab, dst, kc, vc, mask, idx, parts, fattn_gs_parts_bytes(NH, D), dst2);
    auto gs = [&](int s) {
        fa1_dev a = args(s % NL); a.nsplit = FGS_NSL; a.part = parts;
        const uint32_t pk = (uint32_t) NKVH | (1u << 8) | (1u << 9) | ((uint32_t) nkv << 10) | (((uint32_t) (int32_t) ((a.vidx - a.kidx) / 8) & 0xfffu) << 19) | (getenv("FA1_LAB_FAR") ? 0x80000000u : 0u);
        k_fattn_gs<128><<<dim3(NKVH * FGS_NSL), dim3(64 * FGS_W), 0, st>>>(a.qraw, a.qw, a.k, (int) (((const char *) a.tab - a.qraw) / 16), (int) ((a.kidx - a.qraw) / 8), (uint32_t) (uint16_t) (int16_t) ((a.kraw - a.qraw) / 16) | ((uint32_t) (uint16_t) (int16_t) ((a.vraw - a.qraw) / 16) << 16), (int) ((a.mask - a.qraw) / 16), (int) ((const char *) a.kw - (const char *) a.qw), (int) ((a.v - a.k) / 16), a.eps, pk, a);
    };
    {   // result check: the group-slice form + merge against the one-workgroup-per-head form, same cache (the new rows are written by both: identical values)
        k_producer<<<256, 256, 0, st>>>(qkv, 5); attn(3); HIP_CHECK(hipStreamSynchronize(st)); fprintf(stderr, "one-per-head form ran\n");
        std::vector<float> r1(4096), r2(4096); HIP_CHECK(hipMemcpy(r1.data(), dst, 4096 * 4, hipMemcpyDeviceToHost));
        gs(3); HIP_CHECK(hipStreamSynchronize(st)); fprintf(stderr, "group-slice form ran\n");
        fattn_gs_merge(parts, dst2, NH, D, st); HIP_CHECK(hipStreamSynchronize(st)); fprintf(stderr, "merge ran\n");
        HIP_CHECK(hipMemcpy(r2.data(), dst2, 4096 * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0, mx = 0; for (int i = 0; i < 4096; ++i) { const double d = (double) r1[i] - r2[i]; num += d * d; den += (double) r1[i] * r1[i]; mx = std::max(mx, std::fabs(d)); }
        printf("group-slice form vs one-per-head form: nmse %.2e, max abs diff %.2e (|out| rms %.3e)\n", num / (den + 1e-30), mx, std::sqrt(den / 4096));
    }
    const int N = 72;
    const double tp  = time_graph(N, [&](int s) { k_producer<<<256, 256, 0, st>>>(qkv, s); });
    const double tpa = time_graph(N, [&](int s) { k_producer<<<256, 256, 0, st>>>(qkv, s); attn(s); });
    const double ta  = time_graph(N, [&](int s) { attn(s); });
    const double tpg = time_graph(N, [&](int s) { k_producer<<<256, 256, 0, st>>>(qkv, s); gs(s); });
    printf("n_kv %d: group-slice form behind a producer %.2f us\n", nkv, tpg - tp);
    printf("n_kv %d: producer alone %.2f us | producer + attention %.2f us -> attention behind a producer %.2f us | attention back to back (its inputs warm) %.2f us\n", nkv, tp, tpa, tpa - tp, ta);
#ifdef FA1_TRACE
    HIP_CHECK(hipMemset(trace_dev, 0, 1024 * 4 * 8 * 8));
    for (int s = 0; s < 4; ++s) { k_producer<<<256, 256, 0, st>>>(qkv, s); attn(s + 7); }
    HIP_CHECK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h((size_t) NH * 4 * 8);
    HIP_CHECK(hipMemcpy(h.data(), trace_dev, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull; for (size_t w = 0; w < (size_t) NH * 4; ++w) if (h[w * 8]) t0 = std::min(t0, h[w * 8]);
    static const char * lab[8] = { "start", "every load requested", "q / k chain done (raw rows, norm, rope, stores)", "at barrier 1", "past barrier 1", "scores done (K rows used)", "soft-max done", "end (P.V done, output stored)" };
    printf("   time line (us after the first wave's start; min / median / max over the %d waves)\n", NH * 4);
    for (int i = 0; i < 8; ++i) {
        std::vector<double> v; for (size_t w = 0; w < (size_t) NH * 4; ++w) if (h[w * 8 + i]) v.push_back((double) (h[w * 8 + i] - t0) * 0.01);
        if (v.empty()) continue; std::sort(v.begin(), v.end());
        printf("      %-50s %6.2f / %6.2f / %6.2f\n", lab[i], v[0], v[v.size() / 2], v[v.size() - 1]);
    }
    for (int wv = 0; wv < 4; ++wv) { printf("      wave %d of workgroup 0:", wv); for (int i = 0; i < 8; ++i) printf(" %6.2f", (double) (h[(size_t) wv * 8 + i] - t0) * 0.01); printf("\n"); }
    {
        HIP_CHECK(hipMemset(trace_dev, 0, 1024 * 4 * 8 * 8));
        for (int s = 0; s < 4; ++s) { k_producer<<<256, 256, 0, st>>>(qkv, s); gs(s + 7); }
        HIP_CHECK(hipStreamSynchronize(st));
        const size_t nwv = (size_t) NKVH * FGS_NSL * FGS_W;
        std::vector<unsigned long long> hg(nwv * 8);
        HIP_CHECK(hipMemcpy(hg.data(), trace_dev, hg.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long tg = ~0ull; for (size_t w = 0; w < nwv; ++w) if (hg[w * 8]) tg = std::min(tg, hg[w * 8]);
        static const char * labg[8] = { "start", "K / V DMA + every load requested", "chains done", "past barrier 1 (K / V tiles landed)", "K / V DMA issued (slot 4)", "soft-max done, past barrier 3", "P.V done, past barrier 4", "end (partial state stored)" };
        printf("   group-slice form, time line (us after the first wave's start; min / median / max over the %d waves)\n", (int) nwv);
        for (int i = 0; i < 8; ++i) {
            std::vector<double> v; for (size_t w = 0; w < nwv; ++w) if (hg[w * 8 + i]) v.push_back((double) (hg[w * 8 + i] - tg) * 0.01);
            if (v.empty()) continue; std::sort(v.begin(), v.end());
            printf("      %-50s %6.2f / %6.2f / %6.2f\n", labg[i], v[0], v[v.size() / 2], v[v.size() - 1]);
        }
    }
#endif
    return 0;
}

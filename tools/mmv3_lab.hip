// tools/mmv3_lab.hip -- measurement bench (not part of the product): the stages of a Qwen3-8B decode layer (wo + resid -> ffn_norm gate / up + SwiGLU ->
// down + resid -> attn_norm wq / wk / wv) as four k_mv2 launches (the product's launch form) against ONE persistent k_mv3 launch (mmv3.hip), as nodes of a
// replayed hipGraph with rotating weights, every output compared bit for bit; built with -DMV3_TRACE: the per-stage time line of one engine launch.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=14 [-DMV3_TRACE] tools/mmv3_lab.hip -o build/mmv3_lab
//   run:   build/mmv3_lab [first_stage last_stage]      (default 0 3: the whole chain; e.g. 1 2 = gate/up -> down only)
#include "../llama.cpp-omni_amd/csrc/kernels/quantize.hip"
#include "../llama.cpp-omni_amd/csrc/kernels/mmv1.hip"
#include "../llama.cpp-omni_amd/csrc/kernels/mmv1q.hip"
#include "../llama.cpp-omni_amd/csrc/kernels/mmv2.hip"
#include "mmv3_engine.hip"
#include <vector>
#include <string>
#include <functional>
#include <algorithm>
#include <cmath>
#include <cstring>

using namespace mi;
namespace mi { bool mmv3(const mv1_args * st, int n, hipStream_t stream); uint32_t mmv3_error(); void mmv3_set_trace(uint32_t * buf); }

__global__ void k_fill(uint32_t * p, size_t n32, uint32_t seed) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (size_t) gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t) i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = h;
    }
}
__global__ void k_fix_scales(char * p, size_t nblk, int bs, int off, int nf16) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < nblk; i += (size_t) gridDim.x * blockDim.x) {
        uint16_t * d = (uint16_t *) (p + i * bs + off);
        for (int k = 0; k < nf16; ++k) d[k] = (uint16_t) (0x1400 + ((i * 7 + k * 13) & 0x3ff));
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

int main(int argc, char ** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int s_first = argc > 2 ? atoi(argv[1]) : 0, s_last = argc > 2 ? atoi(argv[2]) : 3;
    HIP_CHECK(hipStreamCreate(&st));
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    const size_t ARENA = (size_t) 1024 << 20;
    char * a4, * a6;
    HIP_CHECK(hipMalloc(&a4, ARENA)); HIP_CHECK(hipMalloc(&a6, ARENA));
    k_fill<<<4096, 256, 0, st>>>((uint32_t *) a4, ARENA / 4, 1u); k_fill<<<4096, 256, 0, st>>>((uint32_t *) a6, ARENA / 4, 2u);
    k_fix_scales<<<4096, 256, 0, st>>>(a4, ARENA / 144, 144, 0, 2); k_fix_scales<<<4096, 256, 0, st>>>(a6, ARENA / 210, 210, 208, 1);
    // activations: attn (input of wo), xin (residual of wo), two norm weight rows; outputs of either form: x2 [4096], h [12288], x3 [4096], qkv [6144]
    float * attn, * xin, * nw1, * nw2, * hin, * x2in;
    HIP_CHECK(hipMalloc(&attn, 4096 * 4)); HIP_CHECK(hipMalloc(&xin, 4096 * 4)); HIP_CHECK(hipMalloc(&nw1, 4096 * 4)); HIP_CHECK(hipMalloc(&nw2, 4096 * 4));
    HIP_CHECK(hipMalloc(&hin, 12288 * 4)); HIP_CHECK(hipMalloc(&x2in, 4096 * 4));
    k_fill_f32<<<64, 256, 0, st>>>(attn, 4096, 11u, 1.0f); k_fill_f32<<<64, 256, 0, st>>>(xin, 4096, 12u, 1.0f);
    k_fill_f32<<<64, 256, 0, st>>>(nw1, 4096, 13u, 1.0f); k_fill_f32<<<64, 256, 0, st>>>(nw2, 4096, 14u, 1.0f);
    k_fill_f32<<<64, 256, 0, st>>>(hin, 12288, 15u, 1.0f); k_fill_f32<<<64, 256, 0, st>>>(x2in, 4096, 16u, 1.0f);
    const int NOUT = 4096 + 12288 + 4096 + 6144;
    float * out_a, * out_b;
    HIP_CHECK(hipMalloc(&out_a, NOUT * 4)); HIP_CHECK(hipMalloc(&out_b, NOUT * 4));
    uint32_t * trace_dev = nullptr;
#ifdef MV3_TRACE
    HIP_CHECK(hipMalloc(&trace_dev, 256 * 16 * 64 * 4));
    mmv3_set_trace(trace_dev);
#endif
    HIP_CHECK(hipStreamSynchronize(st));

    const int Q4 = GGML_TYPE_Q4_K, Q6 = GGML_TYPE_Q6_K;
    for (int variant = 0; variant < 2; ++variant) {           // 0: Q4_K down / Q4_K v;  1: Q6_K down / Q6_K v (the use_more_bits layers)
        const int tdown = variant ? Q6 : Q4, tv = variant ? Q6 : Q4;
        // byte sizes of the layer's matrices in stream order: wo, gate, up, down, wq, wk, wv
        const size_t b_wo = (size_t) 4096 * 16 * 144, b_g = (size_t) 12288 * 16 * 144, b_down = (size_t) 4096 * 48 * (tdown == Q4 ? 144 : 210);
        const size_t b_q = (size_t) 4096 * 16 * 144, b_k = (size_t) 1024 * 16 * 144, b_v = (size_t) 1024 * 16 * (tv == Q4 ? 144 : 210);
        const size_t per_layer4 = b_wo + 2 * b_g + (tdown == Q4 ? b_down : 0) + b_q + b_k + (tv == Q4 ? b_v : 0);
        const size_t per_layer6 = (tdown == Q6 ? b_down : 0) + (tv == Q6 ? b_v : 0);
        const size_t stride4 = ((per_layer4 + (2 << 20) - 1) >> 20) << 20, stride6 = ((per_layer6 + (2 << 20)) >> 20) << 20;
        const int nrot = (int) std::min<size_t>(ARENA / stride4, per_layer6 ? ARENA / stride6 : 64);
        const double layer_bytes = (double) (b_wo + 2 * b_g + b_down + b_q + b_k + b_v);
        printf("\n== layer variant %d (%s down / v): %.1f MB per layer, %d rotating weight sets; stages %d..%d\n", variant, variant ? "Q6_K" : "Q4_K", layer_bytes / 1e6, nrot, s_first, s_last);
        auto al = [](size_t off, size_t unit) { return off / unit * unit; };
        // stage descriptions for weight set s, outputs into `out`
        auto stages = [&](int s, float * out, mv1_args * v) {
            char * p4 = a4 + (size_t) (s % nrot) * stride4, * p6 = a6 + (size_t) (s % nrot) * stride6;
            size_t o4 = 0, o6 = 0;
            auto take = [&](size_t bytes, int type) -> const char * {      // (offsets are multiples of a step FROM THE ARENA'S BASE: the blocks' f16 scales were fixed there)
                if (type == Q4) { const char * r = a4 + al((size_t) (p4 - a4) + o4 + 2303, 2304); o4 = (size_t) (r - p4) + bytes; return r; }
                const char * r = a6 + al((size_t) (p6 - a6) + o6 + 3359, 3360); o6 = (size_t) (r - p6) + bytes; return r;
            };
            float * x2 = out, * h = out + 4096, * x3 = h + 12288, * q = x3 + 4096, * k = q + 4096, * vv = k + 1024;
            // 0: wo + resid
            v[0] = mv1_args(); v[0].nmat = 1; v[0].K = 4096; v[0].x = attn;
            v[0].m[0] = { take(b_wo, Q4), (size_t) 16 * 144, x2, 0, xin, 0, 4096, Q4 };
            // 1: gate / up pair + norm
            v[1] = mv1_args(); v[1].nmat = 1; v[1].K = 4096; v[1].x = x2; v[1].norm_w = nw1; v[1].eps = 1e-6f;
            { const char * g = take(b_g, Q4); const char * u = take(b_g, Q4); v[1].m[0] = { g, (size_t) 16 * 144, h, 0, nullptr, 0, 12288, Q4 }; v[1].W_up = u; }
            // 2: down + resid (x2)
            v[2] = mv1_args(); v[2].nmat = 1; v[2].K = 12288; v[2].x = h;
            v[2].m[0] = { take(b_down, tdown), (size_t) 48 * (tdown == Q4 ? 144 : 210), x3, 0, x2, 0, 4096, tdown };
            // 3: wq / wk / wv + norm
            v[3] = mv1_args(); v[3].nmat = 3; v[3].K = 4096; v[3].x = x3; v[3].norm_w = nw2; v[3].eps = 1e-6f;
            v[3].m[0] = { take(b_q, Q4), (size_t) 16 * 144, q, 0, nullptr, 0, 4096, Q4 };
            v[3].m[1] = { take(b_k, Q4), (size_t) 16 * 144, k, 0, nullptr, 0, 1024, Q4 };
            v[3].m[2] = { take(b_v, tv), (size_t) 16 * (tv == Q4 ? 144 : 210), vv, 0, nullptr, 0, 1024, tv };
            // a partial chain starts from ready-made inputs
            if (s_first == 1) v[1].x = x2in;
            if (s_first == 2) v[2].x = hin;
            if (s_first == 3) v[3].x = x2in;
            if (s_first >= 1) v[2].m[0].resid = x2in;                  // the chain does not contain wo: down's residual is an external row
        };
        double chain_bytes = 0;
        { const double b[4] = { (double) b_wo, 2.0 * b_g, (double) b_down, (double) (b_q + b_k + b_v) }; for (int i = s_first; i <= s_last; ++i) chain_bytes += b[i]; }
        auto launches = [&](int s, float * out) { mv1_args v[4]; stages(s, out, v); for (int i = s_first; i <= s_last; ++i) mmv2(v[i], st); };
        auto engine   = [&](int s, float * out) { mv1_args v[4]; stages(s, out, v); if (!mmv3(v + s_first, s_last - s_first + 1, st)) { fprintf(stderr, "mmv3 refused the chain\n"); exit(1); } };
        // ---- correctness: several weight sets, outputs bit for bit
        std::vector<float> ha(NOUT), hb(NOUT);
        int bad_total = 0;
        for (int s = 0; s < 3; ++s) {
            HIP_CHECK(hipMemsetAsync(out_a, 0, NOUT * 4, st)); HIP_CHECK(hipMemsetAsync(out_b, 0xff, NOUT * 4, st));
            launches(s, out_a); engine(s, out_b);
            HIP_CHECK(hipMemcpyAsync(ha.data(), out_a, NOUT * 4, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipMemcpyAsync(hb.data(), out_b, NOUT * 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            const uint32_t e = mmv3_error();
            if (e) { printf("   ENGINE GAVE UP A WAIT: code %u, workgroup %u\n", e >> 16, (e & 0xffff) - 1); return 2; }
            const int lo[4] = { 0, 4096, 4096 + 12288, 4096 + 12288 + 4096 }, hi[4] = { 4096, 4096 + 12288, 4096 + 12288 + 4096, NOUT };
            static const char * nm[4] = { "x2 = wo + resid", "h = silu(gate) up", "x3 = down + resid", "q, k, v" };
            for (int t = s_first; t <= s_last; ++t) {
                int nbad = 0, first = -1; double num = 0, den = 0;
                for (int i = lo[t]; i < hi[t]; ++i) { if (memcmp(&ha[i], &hb[i], 4) != 0) { ++nbad; if (first < 0) first = i - lo[t]; } const double dl = (double) ha[i] - hb[i]; num += dl * dl; den += (double) ha[i] * ha[i]; }
                printf("   set %d  %-20s %s (%d / %d differ%s, nmse %.1e)\n", s, nm[t], nbad ? "MISMATCH" : "identical", nbad, hi[t] - lo[t], nbad ? (" first at " + std::to_string(first)).c_str() : "", num / (den + 1e-30));
                bad_total += nbad;
            }
        }
        // ---- timing
        const int N = 36;
        const double tl = time_graph(N, [&](int s) { launches(s, out_a); });
        printf("   %-46s %7.2f us per layer-chain  (%.2f TB/s)\n", "launch form (k_mv2 per stage)", tl, chain_bytes / tl / 1e6);
        const double te = time_graph(N, [&](int s) { engine(s, out_b); });
        { const uint32_t e = mmv3_error(); if (e) { printf("   ENGINE GAVE UP A WAIT (timing): code %u, workgroup %u\n", e >> 16, (e & 0xffff) - 1); return 2; } }
        printf("   %-46s %7.2f us per layer-chain  (%.2f TB/s)   %s\n", "engine (one k_mv3 launch)", te, chain_bytes / te / 1e6, bad_total ? "RESULTS DIFFER" : "results identical");
        // per-stage launch times for reference
        for (int i = s_first; i <= s_last; ++i) {
            const double t1 = time_graph(N, [&](int s) { mv1_args v[4]; stages(s, out_a, v); mmv2(v[i], st); });
            printf("      stage %d alone as a launch: %6.2f us\n", i, t1);
        }
#ifdef MV3_TRACE
        {
            HIP_CHECK(hipMemsetAsync(trace_dev, 0, 256 * 16 * 64 * 4, st));
            for (int s = 0; s < 4; ++s) engine(s + 5, out_b);
            HIP_CHECK(hipStreamSynchronize(st));
            std::vector<uint32_t> h((size_t) 256 * 16 * 64);
            HIP_CHECK(hipMemcpy(h.data(), trace_dev, h.size() * 4, hipMemcpyDeviceToHost));
            uint32_t t0 = 0xffffffffu;
            for (int w = 0; w < 256 * 16; ++w) if (h[(size_t) w * 64 + 63]) t0 = std::min(t0, h[(size_t) w * 64 + 63]);
            auto rep = [&](const char * what, int slot, std::function<bool(int)> sel) {
                std::vector<double> v;
                for (int w = 0; w < 256 * 16; ++w) if (sel(w % 16) && h[(size_t) w * 64 + slot]) v.push_back((double) (h[(size_t) w * 64 + slot] - t0) * 0.01);
                if (v.empty()) return;
                std::sort(v.begin(), v.end());
                printf("        %-44s %6.2f / %6.2f / %6.2f   (%zu waves)\n", what, v[0], v[v.size() / 2], v[v.size() - 1], v.size());
            };
            printf("      time line of the last engine launch (us after the first wave's start; min / median / max over the workgroups)\n");
            const int nst = s_last - s_first + 1;
            for (int s = 0; s < nst; ++s) {
                char b[96];
                snprintf(b, sizeof b, "stage %d  loader: first request", s); rep(b, 8 * s + 0, [](int w) { return w == 0; });
                snprintf(b, sizeof b, "stage %d  loader: last request issued", s); rep(b, 8 * s + 1, [](int w) { return w == 0; });
                snprintf(b, sizeof b, "stage %d  consumers enter", s); rep(b, 8 * s + 0, [](int w) { return w >= 1; });
                snprintf(b, sizeof b, "stage %d  gather: inputs published (go)", s); rep(b, 8 * s + 1, [](int w) { return w >= 1 && w <= 12; });
                snprintf(b, sizeof b, "stage %d  gather: vector in registers", s); rep(b, 8 * s + 2, [](int w) { return w >= 1 && w <= 12; });
                snprintf(b, sizeof b, "stage %d  gather: scale known", s); rep(b, 8 * s + 3, [](int w) { return w >= 1 && w <= 12; });
                snprintf(b, sizeof b, "stage %d  image complete (consume starts)", s); rep(b, 8 * s + 4, [](int w) { return w >= 1; });
                snprintf(b, sizeof b, "stage %d  consumers done", s); rep(b, 8 * s + 5, [](int w) { return w >= 1; });
                snprintf(b, sizeof b, "stage %d  workgroup's rows published", s); rep(b, 8 * s + 6, [](int w) { return w >= 1; });
            }
            rep("loader: drained", 62, [](int w) { return w == 0; });
            rep("consumers: end", 62, [](int w) { return w >= 1; });
        }
#endif
    }
    return 0;
}

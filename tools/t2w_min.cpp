// tools/t2w_min.cpp -- TEST / MEASUREMENT INFRASTRUCTURE (not part of the product library).
//
// Drives the REFERENCE's Token2Wav module -- tools/omni/token2wav/token2wav-impl.cpp + token2wav.cpp compiled from /root/reference by
// oracle/Makefile.ref `omni` (Token2WavSession: conformer token encoder -> 10-step flow-matching DiT -> HiFT vocoder, streaming in windows of
// 25 + 3 look-ahead tokens = 1 s of 24 kHz audio each) -- on the synthetic module set of tools/make_synth_omni_gguf.py --module t2w, on the CPU
// backend ("cpu") or on this repo's plug-in ("gpu": ggml_backend_init_by_type(GPU) after ggml_backend_load_all() picked up $GGML_BACKEND_PATH;
// the module has NO scheduler -- token2wav-impl.cpp:6287-6345 -- so every node of its graphs must run on the one backend).
//   t2w-min DIR out.f32 cpu|gpu [--windows N]
// Feeds N windows of deterministic pseudo-random tokens, writes the concatenated waveform, prints one JSON line (ms per window, real-time factor).
#include "token2wav.h"
#include "ggml-backend.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char ** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s DIR out.f32 cpu|gpu [--windows N]\n", argv[0]); return 2; }
    const std::string dir = argv[1], out = argv[2], dev = argv[3];
    int windows = 3;
    for (int i = 4; i < argc; ++i) if (!strcmp(argv[i], "--windows") && i + 1 < argc) windows = atoi(argv[++i]);
    ggml_backend_load_all();                                   // (the flow loader looks its GPU backend up before the vocoder's own load_all: have the plug-in registered first, as a build with a linked-in GPU backend has)
    omni::flow::Token2WavSession s;
    const auto t_init0 = std::chrono::steady_clock::now();
    if (!s.init_from_prompt_bundle(dir + "/encoder.gguf", dir + "/flow_matching.gguf", dir + "/flow_extra.gguf", dir + "/prompt", dir + "/hifigan2.gguf", dev, dev, 10, 1.0f)) {
        fprintf(stderr, "init_from_prompt_bundle failed\n"); return 1;
    }
    const double init_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_init0).count();
    uint32_t lcg = 12345u;
    std::vector<float> all; std::string per;
    double total_ms = 0, steady_ms = 0;
    for (int w = 0; w < windows; ++w) {
        std::vector<int32_t> tok(omni::flow::Token2Mel::kDt);
        for (auto & t : tok) { lcg = lcg * 1664525u + 1013904223u; t = (int32_t) ((lcg >> 8) % 6561u); }
        std::vector<float> wave;
        const auto t0 = std::chrono::steady_clock::now();
        if (!s.feed_window(tok, w == windows - 1, wave)) { fprintf(stderr, "feed_window %d failed\n", w); return 1; }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        total_ms += ms; if (w > 0) steady_ms += ms;
        char b[48]; snprintf(b, sizeof b, "%s%.2f", w ? ", " : "", ms); per += b;
        all.insert(all.end(), wave.begin(), wave.end());
    }
    FILE * f = fopen(out.c_str(), "wb"); fwrite(all.data(), sizeof(float), all.size(), f); fclose(f);
    const double audio_s = (double) all.size() / omni::flow::Token2Wav::kSampleRate;
    printf("{\"module\": \"t2w\", \"device\": \"%s\", \"windows\": %d, \"samples\": %zu, \"audio_s\": %.3f, \"init_ms\": %.1f, \"ms_windows\": [%s], \"rtf\": %.5f, \"rtf_after_first_window\": %.5f}\n",
           dev.c_str(), windows, all.size(), audio_s, init_ms, per.c_str(), total_ms / 1e3 / audio_s, windows > 1 ? steady_ms / 1e3 / (audio_s * (windows - 1) / windows) : 0.0);
    return 0;
}

// tools/omni_enc_min.cpp -- TEST / MEASUREMENT INFRASTRUCTURE (not part of the product library).
//
// Drives the REFERENCE's own omni encoder code -- tools/omni/audition.cpp (build_whisper :341-715, the loader :790-1135, audition_audio_encode)
// compiled from /root/reference by oracle/Makefile.ref `omni` -- on a synthetic module GGUF (tools/make_synth_omni_gguf.py), on the CPU backend or,
// with GGML_BACKEND_PATH + MTMD_BACKEND_DEVICE=MI355X0 (audition.cpp:241-266), on this repo's plug-in through ggml_backend_sched.
// and tools/omni/vision.cpp (build_minicpmv :292-380, build_vit, the loader :787-1090, vision_image_encode) likewise.
//   omni-enc-min apm model.gguf out.bin [--gpu] [--chunks N] [--frames F] [--threads T]
//   omni-enc-min vpm model.gguf out.bin [--gpu] [--chunks N] [--size WxH] [--threads T]
// apm: feeds N chunks of F mel frames (deterministic pseudo-random values; 100 frames = 1 s of audio = 50 positions = 10 embeddings) through the
// streaming KV cache audition_init sets up (chunk i attends to chunks < i, as stream_prefill drives it); vpm: N slices of WxH pixels (default 448x448 =
// 1024 patches -> 64 embeddings each). Every chunk's embeddings go to out.bin; one JSON line on stdout.
#include "audition.h"
#include "vision.h"
#include "ggml-backend.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static uint32_t g_lcg = 2463534242u;
static float lcg_unit() { g_lcg = g_lcg * 1664525u + 1013904223u; return (float) ((g_lcg >> 8) & 0xffff) / 32768.0f - 1.0f; }

int main(int argc, char ** argv) {
    if (argc < 4 || (strcmp(argv[1], "apm") != 0 && strcmp(argv[1], "vpm") != 0)) {
        fprintf(stderr, "usage: %s apm|vpm model.gguf out.bin [--gpu] [--chunks N] [--frames F] [--size WxH] [--threads T]\n", argv[0]); return 2;
    }
    const bool is_vpm = !strcmp(argv[1], "vpm");
    const std::string model = argv[2], out = argv[3];
    bool gpu = false; int chunks = 1, frames = 100, threads = 8, iw = 448, ih = 448;
    for (int i = 4; i < argc; ++i) {
        if (!strcmp(argv[i], "--gpu")) gpu = true;
        else if (!strcmp(argv[i], "--size") && i + 1 < argc) { if (sscanf(argv[++i], "%dx%d", &iw, &ih) != 2) return 2; }
        else if (!strcmp(argv[i], "--chunks") && i + 1 < argc) chunks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
    }
    ggml_backend_load_all();                                   // picks up $GGML_BACKEND_PATH
    if (is_vpm) {
        vision_context_params p; p.use_gpu = gpu; p.verbosity = GGML_LOG_LEVEL_INFO; p.coreml_model_path = nullptr;
        vision_ctx * ctx = vision_init(model.c_str(), p);
        if (!ctx) { fprintf(stderr, "vision_init failed\n"); return 1; }
        const int n_embd = vision_n_mmproj_embd(ctx), n_tok = vision_n_output_tokens(ctx);
        FILE * f = fopen(out.c_str(), "wb");
        double ms_total = 0, ms_last = 0;
        for (int c = 0; c < chunks; ++c) {
            vision_image_f32 * img = vision_image_f32_init();
            img->nx = iw; img->ny = ih; img->buf.resize((size_t) 3 * iw * ih);
            for (float & x : img->buf) x = lcg_unit();
            std::vector<float> vec((size_t) n_tok * (size_t) n_embd);
            const auto t0 = std::chrono::steady_clock::now();
            if (!vision_image_encode(ctx, threads, img, vec.data())) { fprintf(stderr, "vision_image_encode failed\n"); return 1; }
            ms_last = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); ms_total += ms_last;
            fwrite(vec.data(), sizeof(float), vec.size(), f);
            vision_image_f32_free(img);
        }
        fclose(f);
        printf("{\"module\": \"vpm\", \"gpu\": %s, \"chunks\": %d, \"size\": \"%dx%d\", \"n_embd\": %d, \"tokens\": %d, \"ms_per_chunk\": %.3f, \"ms_last_chunk\": %.3f}\n",
               gpu ? "true" : "false", chunks, iw, ih, n_embd, n_tok * chunks, ms_total / chunks, ms_last);
        vision_free(ctx);
        return 0;
    }
    audition_context_params p; p.use_gpu = gpu; p.verbosity = GGML_LOG_LEVEL_INFO;
    audition_ctx * ctx = audition_init(model.c_str(), p);
    if (!ctx) { fprintf(stderr, "audition_init failed\n"); return 1; }
    const int n_embd = audition_n_mmproj_embd(ctx);
    FILE * f = fopen(out.c_str(), "wb");
    double ms_total = 0, ms_last = 0; int n_tok_total = 0; std::string per_chunk;
    for (int c = 0; c < chunks; ++c) {
        audition_audio_f32 audio; audio.nx = frames; audio.ny = 80; audio.buf.resize((size_t) frames * 80);
        for (float & x : audio.buf) x = lcg_unit();
        const int n_tok = audition_n_output_tokens(ctx, &audio);
        std::vector<float> vec((size_t) n_tok * (size_t) n_embd);
        const auto t0 = std::chrono::steady_clock::now();
        if (!audition_audio_encode(ctx, threads, &audio, vec.data())) { fprintf(stderr, "audition_audio_encode failed\n"); return 1; }
        ms_last = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); ms_total += ms_last;
        fwrite(vec.data(), sizeof(float), vec.size(), f);
        char b[32]; snprintf(b, sizeof b, "%s%.2f", c ? ", " : "", ms_last); per_chunk += b;
        n_tok_total += n_tok;
    }
    fclose(f);
    printf("{\"module\": \"apm\", \"gpu\": %s, \"chunks\": %d, \"frames\": %d, \"n_embd\": %d, \"tokens\": %d, \"ms_per_chunk\": %.3f, \"ms_last_chunk\": %.3f, \"ms_chunks\": [%s]}\n",
           gpu ? "true" : "false", chunks, frames, n_embd, n_tok_total, ms_total / chunks, ms_last, per_chunk.c_str());
    audition_free(ctx);
    return 0;
}

// tools/omni_min.cpp -- TEST / MEASUREMENT INFRASTRUCTURE (not part of the product library).
//
// the REFERENCE's omni runtime as the caller of the plug-in.  tools/omni/omni.cpp (omni_init :3472, stream_prefill :8637,
// stream_decode :8950, its LLM / TTS / Token2Wav threads), audition.cpp, vision.cpp, token2wav-impl.cpp, libllama and common/{common,sampling,log}.cpp
// are compiled from /root/reference by oracle/Makefile.ref `omnirt`; nothing of their orchestration is restated here.  This file is only what
// tools/omni/omni-cli.cpp:198-380 is -- argument handling, one omni_init, the `--test <prefix> <n>` loop (synchronous stream_prefill per wav, then one
// stream_decode), the wait for generation_done.flag, thread shutdown -- written on the public omni.h API because omni-cli.cpp itself calls
// common_init(), whose body references the cmake-generated common/build-info.cpp (LLAMA_BUILD_NUMBER / LLAMA_COMMIT / LLAMA_COMPILER /
// LLAMA_BUILD_TARGET).  No stand-in for that file exists here: the harness never reaches common_init and the link drops its section (--gc-sections).
//
//   omni-min -m LLM.gguf [--test PREFIX N] [-c CTX] [-ngl N] [--ref-audio WAV] [--no-tts] [--omni] [--out DIR] [--t2w-device gpu:N|cpu] [--max-tgt N] [-mg N] [-sm none]
// The other module paths follow omni-cli's directory convention ({dir}/audio/MiniCPM-o-4_5-audio-F16.gguf, {dir}/tts/..., {dir}/vision/...,
// {dir}/token2wav-gguf/*).  Prints one JSON line: the devices every module's backend landed on (from the registry), prefill / decode wall times and
// the timestamps the reference itself writes (TTFT of the first wav chunk from the output directory's mtime).
#include "omni.h"
#include "common/common.h"
#include "ggml-backend.h"
#include "llama.h"

#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static bool file_exists(const std::string & p) { struct stat st; return stat(p.c_str(), &st) == 0; }
static std::string parent_dir(const std::string & p) { const size_t s = p.find_last_of('/'); return s == std::string::npos ? "." : p.substr(0, s); }
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char ** argv) {
    std::string llm, prefix, ref_audio, out_dir = "./omni_out", t2w_dev = "gpu:0";
    int n = 1, n_ctx = 4096, ngl = 99, max_tgt = -1, main_gpu = -1;
    bool use_tts = true, omni_mode = false, split_none = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-m" && i + 1 < argc) llm = argv[++i];
        else if (a == "--test" && i + 2 < argc) { prefix = argv[++i]; n = atoi(argv[++i]); }
        else if ((a == "-c" || a == "--ctx-size") && i + 1 < argc) n_ctx = atoi(argv[++i]);
        else if (a == "-ngl" && i + 1 < argc) ngl = atoi(argv[++i]);
        else if (a == "--ref-audio" && i + 1 < argc) ref_audio = argv[++i];
        else if (a == "--out" && i + 1 < argc) out_dir = argv[++i];
        else if (a == "--t2w-device" && i + 1 < argc) t2w_dev = argv[++i];
        else if (a == "--max-tgt" && i + 1 < argc) max_tgt = atoi(argv[++i]);
        else if ((a == "-mg" || a == "--main-gpu") && i + 1 < argc) main_gpu = atoi(argv[++i]);    // common/arg.cpp:2955 (with -sm none: the one device of the LLM -- and of the TTS model, omni.cpp:3457)
        else if ((a == "-sm" || a == "--split-mode") && i + 1 < argc) split_none = std::string(argv[++i]) == "none";
        else if (a == "--no-tts") use_tts = false;
        else if (a == "--omni") omni_mode = true;
        else { fprintf(stderr, "usage: %s -m LLM.gguf --test PREFIX N [-c CTX] [-ngl N] [--ref-audio WAV] [--no-tts] [--omni] [--out DIR] [--t2w-device D] [--max-tgt N]\n", argv[0]); return 2; }
    }
    if (llm.empty() || prefix.empty()) { fprintf(stderr, "need -m and --test\n"); return 2; }
    ggml_time_init();
    ggml_backend_load_all();                                   // $GGML_BACKEND_PATH: the plug-in registers before any module asks for a GPU device
    const std::string dir = parent_dir(llm);
    common_params params;
    params.model.path = llm;
    params.vpm_model = dir + "/vision/MiniCPM-o-4_5-vision-F16.gguf";
    params.apm_model = dir + "/audio/MiniCPM-o-4_5-audio-F16.gguf";
    params.tts_model = dir + "/tts/MiniCPM-o-4_5-tts-F16.gguf";
    params.n_ctx = n_ctx;
    params.n_gpu_layers = ngl;
    if (main_gpu >= 0) params.main_gpu = main_gpu;
    if (split_none) params.split_mode = LLAMA_SPLIT_MODE_NONE;
    if (max_tgt > 0) params.n_predict = max_tgt;           // stream_decode's max_tgt_len (omni.cpp:9112): a random-weight LLM never emits <|tts_eos|>
    if (use_tts && !file_exists(params.tts_model)) { fprintf(stderr, "TTS model missing: %s\n", params.tts_model.c_str()); return 1; }
    const std::string tts_bin_dir = parent_dir(params.tts_model);

    const double t_init0 = now_s();
    omni_context * ctx = omni_init(&params, omni_mode ? 2 : 1, use_tts, tts_bin_dir, -1, t2w_dev, false, nullptr, nullptr, out_dir);
    if (!ctx) { fprintf(stderr, "omni_init failed\n"); return 1; }
    const double init_s = now_s() - t_init0;
    ctx->async = true;
    ctx->ref_audio_path = ref_audio;

    // omni-cli.cpp:158-196 test_case(): synchronous prefill of every input, then one decode in the async (threaded) form
    ctx->system_prompt_initialized = false;
    const bool orig_async = ctx->async;
    ctx->async = false;
    std::string per;
    const double t_pf0 = now_s();
    for (int il = 0; il < n; ++il) {
        char idx[16]; snprintf(idx, sizeof idx, "%04d", il);
        const std::string aud = prefix + idx + ".wav", img_c = prefix + idx + ".jpg";
        const std::string img = file_exists(img_c) ? img_c : "";
        const double t0 = now_s();
        if (!stream_prefill(ctx, aud, img, il)) { fprintf(stderr, "stream_prefill %d failed\n", il); return 1; }
        char b[48]; snprintf(b, sizeof b, "%s%.4f", il ? ", " : "", now_s() - t0); per += b;
    }
    const double prefill_s = now_s() - t_pf0;
    const int n_past_prefill = ctx->n_past;
    ctx->async = orig_async;
    const double t_dec0 = now_s();
    if (!stream_decode(ctx, "./")) { fprintf(stderr, "stream_decode failed\n"); return 1; }
    const double decode_call_s = now_s() - t_dec0;

    // omni-cli.cpp:362-372: wait for the Token2Wav thread's completion flag (at most 120 s, as there); the first wav chunk's appearance is the pipeline's
    // time to first audio.  A random-weight LLM stops at max_tgt_len without an end token, so the flag may never be written: the wait also ends when the
    // output directory has been quiet for 3 s after its first wav.
    double first_wav_s = -1, done_s = -1, last_change = now_s();
    int n_wav = 0;
    if (use_tts) {
        const std::string wav_dir = out_dir + "/round_000/tts_wav", done = wav_dir + "/generation_done.flag";
        for (int i = 0; i < 6000; ++i) {
            int k = n_wav;
            while (file_exists(wav_dir + "/wav_" + std::to_string(k) + ".wav")) ++k;
            if (k != n_wav) { if (n_wav == 0) first_wav_s = now_s() - t_dec0; n_wav = k; last_change = now_s(); }
            if (file_exists(done)) { done_s = now_s() - t_dec0; break; }
            if (n_wav > 0 && now_s() - last_change > 3.0) break;
            usleep(20000);
        }
    }
    const double quiet_s = last_change - t_dec0;
    omni_stop_threads(ctx);
    if (ctx->llm_thread.joinable()) ctx->llm_thread.join();
    if (use_tts && ctx->tts_thread.joinable()) ctx->tts_thread.join();
    if (use_tts && ctx->t2w_thread.joinable()) ctx->t2w_thread.join();

    // where did every module land?  (the registry is the reference's: ggml_backend_dev_*)
    std::string devs;
    for (size_t i = 0; i < ggml_backend_dev_count(); ++i) {
        ggml_backend_dev_t d = ggml_backend_dev_get(i);
        devs += std::string(i ? ", " : "") + "\"" + ggml_backend_dev_name(d) + "\"";
    }
    printf("{\"harness\": \"omni-min\", \"registry_devices\": [%s], \"n_inputs\": %d, \"init_s\": %.3f, \"prefill_s\": %.4f, \"prefill_each_s\": [%s], \"n_past_after_prefill\": %d, "
           "\"n_past_after_decode\": %d, \"stream_decode_call_s\": %.4f, \"first_wav_s\": %.4f, \"generation_done_s\": %.4f, \"n_wav\": %d, \"last_wav_s\": %.4f, \"tts\": %s}\n",
           devs.c_str(), n, init_s, prefill_s, per.c_str(), n_past_prefill, ctx->n_past, decode_call_s, first_wav_s, done_s, n_wav, quiet_s, use_tts ? "true" : "false");
    fflush(stdout);
    llama_perf_context_print(ctx->ctx_llama);
    omni_free(ctx);
    return 0;
}

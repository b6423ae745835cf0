#!/usr/bin/env python3
"""tools/make_synth_omni_set.py -- the synthetic MiniCPM-o module set in the directory layout tools/omni/omni-cli.cpp:97-150 resolves and
tools/omni/omni.cpp omni_init (:3472-3990) loads, for the reference-runtime-as-caller run of SURVEY.md 8 row g1 (tools/omni_min.cpp).  TEST INFRASTRUCTURE.

    ROOT/gguf/MiniCPM-o-4_5-Q4_K_M.gguf                  LLM: Qwen3-8B shapes (--llm-layers to shrink for CPU runs), byte-level BPE tokenizer with omni's special tokens
    ROOT/gguf/audio/MiniCPM-o-4_5-audio-F16.gguf         Whisper-medium encoder + audio projector           (make_synth_omni_gguf.apm)
    ROOT/gguf/vision/MiniCPM-o-4_5-vision-F16.gguf       SigLip2 tower + resampler (only with --vision)    (make_synth_omni_gguf.vpm)
    ROOT/gguf/tts/MiniCPM-o-4_5-tts-F16.gguf             20-layer 768-wide llama decoder + emb_code / emb_text / projector_semantic / head_code
    ROOT/gguf/tts/MiniCPM-o-4_5-projector-F16.gguf       linear1 4096->768, linear2 768->768 (omni.cpp projector_init :1068-1175)
    ROOT/gguf/token2wav-gguf/{encoder,flow_matching,flow_extra,hifigan2}.gguf                               (make_synth_omni_gguf.t2w)
    ROOT/tools/omni/assets/default_ref_audio/{default_ref_audio.wav, spk_f32.bin, prompt_tokens_i32.bin, prompt_mel_btc_f32.bin}
                                                         (omni.cpp reads these relative to the working directory: run the harness with cwd = ROOT)
    ROOT/case/audio_0000.wav ...                         user turns: 16 kHz mono PCM16, deterministic band-limited noise

Everything is random weights / synthetic audio: the run demonstrates placement and timing, not content."""
import argparse
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_synth_omni_gguf as mo  # noqa: E402


def write_wav(path, seconds, seed, rate=16000):
    rng = np.random.default_rng(seed)
    n = int(seconds * rate)
    t = np.arange(n) / rate
    x = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in zip(rng.uniform(0.02, 0.2, 12), rng.uniform(90, 3400, 12), rng.uniform(0, 6.28, 12)))
    x = x * (0.5 + 0.5 * np.sin(2 * np.pi * 3.1 * t)) + 0.01 * rng.standard_normal(n)
    pcm = np.clip(x / np.abs(x).max() * 0.6 * 32767, -32768, 32767).astype("<i2")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16) + b"data" + struct.pack("<I", pcm.nbytes))
        f.write(pcm.tobytes())


def write_png(path, w, h, seed):
    """an 8-bit RGB PNG (stb_image, which omni.cpp's vision_image_load_from_bytes :518 uses, detects the format from the bytes, so the `.jpg` name omni-cli
    looks for -- <prefix>NNNN.jpg -- may hold it): smooth colour gradients + blocks, deterministic"""
    import zlib
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 120 * np.sin(xx / rng.uniform(20, 90) + yy / rng.uniform(30, 120) + c) for c in (0.0, 2.1, 4.2)], axis=-1)
    for _ in range(12):
        x0, y0 = int(rng.integers(0, w - 40)), int(rng.integers(0, h - 40))
        img[y0:y0 + int(rng.integers(10, 120)), x0:x0 + int(rng.integers(10, 120))] = rng.integers(0, 255, 3)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), np.clip(img, 0, 255).astype(np.uint8).reshape(h, w * 3)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--llm-layers", type=int, default=0, help="0 = the 8B's 36")
    ap.add_argument("--llm-types", default="q4_k_m")
    ap.add_argument("--tts-layers", type=int, default=0, help="0 = 20")
    ap.add_argument("--apm-layers", type=int, default=24)
    ap.add_argument("--vision", action="store_true", help="also the image module and one 448 x 448 picture beside every user turn after the first (omni mode, media_type 2)")
    ap.add_argument("--vpm-layers", type=int, default=27)
    ap.add_argument("--turns", type=int, default=1)
    ap.add_argument("--turn-seconds", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=11)
    a = ap.parse_args()
    root = os.path.abspath(a.out)
    g = os.path.join(root, "gguf")
    for d in ("audio", "vision", "tts", "token2wav-gguf"):
        os.makedirs(os.path.join(g, d), exist_ok=True)
    ref_dir = os.path.join(root, "tools", "omni", "assets", "default_ref_audio")
    os.makedirs(ref_dir, exist_ok=True)
    os.makedirs(os.path.join(root, "case"), exist_ok=True)

    synth = os.path.join(HERE, "make_synth_gguf.py")
    cmd = [sys.executable, synth, "--config", "8b", "--types", a.llm_types, "--vocab", "omni", "--n-ctx", "8192", "-o", os.path.join(g, "MiniCPM-o-4_5-Q4_K_M.gguf"), "--seed", str(a.seed)]
    if a.llm_layers:
        cmd += ["--layers", str(a.llm_layers)]
    subprocess.check_call(cmd)
    cmd = [sys.executable, synth, "--config", "tts", "--types", "f16", "--omni-tts-extra", "--n-ctx", "8192", "-o", os.path.join(g, "tts", "MiniCPM-o-4_5-tts-F16.gguf"), "--seed", str(a.seed + 1)]
    if a.tts_layers:
        cmd += ["--layers", str(a.tts_layers)]
    subprocess.check_call(cmd)

    rng = np.random.default_rng(a.seed + 2)
    h = np.float16
    mo.write_gguf(os.path.join(g, "tts", "MiniCPM-o-4_5-projector-F16.gguf"), [mo.kv_str("general.architecture", "omni-projector")],
                  [("linear1.weight", (rng.standard_normal((768, 4096)) / 64).astype(h)), ("linear1.bias", np.zeros(768, np.float32)),
                   ("linear2.weight", (rng.standard_normal((768, 768)) / 28).astype(h)), ("linear2.bias", np.zeros(768, np.float32))])
    mo.apm(os.path.join(g, "audio", "MiniCPM-o-4_5-audio-F16.gguf"), a.apm_layers, a.seed + 3)
    if a.vision:
        mo.vpm(os.path.join(g, "vision", "MiniCPM-o-4_5-vision-F16.gguf"), a.vpm_layers, a.seed + 4)
    t2w_dir = os.path.join(g, "token2wav-gguf")
    mo.t2w(t2w_dir, a.seed + 5)
    for fn in ("spk_f32.bin", "prompt_tokens_i32.bin", "prompt_mel_btc_f32.bin"):
        os.replace(os.path.join(t2w_dir, "prompt", fn), os.path.join(ref_dir, fn))
    os.rmdir(os.path.join(t2w_dir, "prompt"))
    write_wav(os.path.join(ref_dir, "default_ref_audio.wav"), 3.0, a.seed + 6)
    for i in range(a.turns):
        write_wav(os.path.join(root, "case", f"audio_{i:04d}.wav"), a.turn_seconds, a.seed + 7 + i)
        if a.vision and i > 0:                                   # (index 0 only initialises the system prompt: omni.cpp:8766-8800)
            write_png(os.path.join(root, "case", f"audio_{i:04d}.jpg"), 448, 448, a.seed + 40 + i)
    print("omni set in", root)


if __name__ == "__main__":
    main()

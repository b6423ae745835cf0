"""SURVEY.md 8 row g1 (north_star: "omni_init / stream_prefill / stream_decode see a drop-in backend"): the REFERENCE's omni runtime -- tools/omni/omni.cpp
with its LLM / TTS / Token2Wav threads, audition.cpp, token2wav-impl.cpp, libllama, common/{common,sampling,log}.cpp, all compiled from /root/reference by
oracle/Makefile.ref `omnirt` -- as the CALLER of the plug-in.  tools/omni_min.cpp is omni-cli.cpp's main() on the public omni.h API (one omni_init, the
`--test` loop of synchronous stream_prefill calls, one stream_decode, the wait for generation_done.flag); nothing of the orchestration is restated.
The module set is synthetic (tools/make_synth_omni_set.py: full-size shapes, random weights, a byte-level BPE tokenizer carrying omni's special tokens, synthetic
speech-band audio), so what is asserted is placement, completion and the reference's own timestamps -- not content."""
import json
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "omni-min")
LIB = os.path.join(ROOT, "llama.cpp-omni_amd", "lib", "libggml-mi355x.so")


def _read_wav(path):
    b = open(path, "rb").read()
    i = b.find(b"data")
    assert b[:4] == b"RIFF" and i > 0
    return np.frombuffer(b[i + 8:], "<i2").astype(np.float32) / 32768.0


def run_omni_min(root, out_dir, max_tgt=24, plug=True, turns=1, timeout=1500, omni=False):
    env = dict(os.environ)
    env.pop("GGML_BACKEND_PATH", None)
    env.pop("MTMD_BACKEND_DEVICE", None)
    if plug:
        env["GGML_BACKEND_PATH"] = LIB
        env["MI355X_LOG_STATS"] = "1"
    cmd = [BIN, "-m", "gguf/MiniCPM-o-4_5-Q4_K_M.gguf", "--test", "case/audio_", str(turns), "-ngl", "99" if plug else "0", "--t2w-device", "gpu:0" if plug else "cpu",
           "--max-tgt", str(max_tgt), "--out", out_dir, "-c", "4096"] + (["--omni"] if omni else [])
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout, errors="replace")
    return r, r.stdout + "\n" + r.stderr


def summarise(log):
    """the reference's own timestamps: first-audio time (omni.cpp's "First Audio Response" line) and the Token2Wav thread's per-window lines"""
    j = json.loads([l for l in log.splitlines() if l.startswith('{"harness"')][-1])
    m = re.search(r"First Audio Response\): (\d+)ms", log)
    j["reference_first_audio_ms"] = int(m.group(1)) if m else None
    w = [(float(a), float(b)) for a, b in re.findall(r"wav_\d+\.wav \| ([0-9.]+)s audio \| ([0-9.]+)ms inference", log)]
    j["t2w_windows"] = len(w)
    j["t2w_ms_per_window_median"] = float(np.median([b for _, b in w])) if w else None
    j["t2w_rtf_median"] = float(np.median([b / 1e3 / a for a, b in w])) if w else None
    tm = [(float(a), float(b)) for a, b in re.findall(r"\[timing\] call=\d+ tokens=\d+ final=\d token2mel=([0-9.]+)ms vocoder=([0-9.]+)ms", log)]
    j["t2w_token2mel_ms_median"] = float(np.median([a for a, _ in tm])) if tm else None       # the flow model: on the device omni_init was given ("gpu:0")
    j["t2w_vocoder_ms_median"] = float(np.median([b for _, b in tm])) if tm else None         # the vocoder: omni.cpp:3779 pins it to "cpu" whatever the device
    m = re.search(r"prompt eval time =\s*([0-9.]+) ms /\s*(\d+) tokens", log)
    j["llm_prompt_tok_s"] = int(m.group(2)) / float(m.group(1)) * 1e3 if m else None
    m = re.search(r"\n[^\n]*eval time =\s*([0-9.]+) ms /\s*(\d+) runs", log.split("prompt eval time")[-1])
    j["llm_decode_tok_s"] = int(m.group(2)) / float(m.group(1)) * 1e3 if m else None
    return j


def test_reference_omni_runtime_drives_the_plugin(tmp_path):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/omni-min not built (make -f oracle/Makefile.ref omnirt)")
    if shutil.disk_usage(str(tmp_path)).free < 9e9:
        pytest.skip("needs 7 GB of scratch disk for the synthetic module set")
    root = str(tmp_path / "set")
    # all five modules: the second user turn carries a 448 x 448 picture, so media_type 2 ("omni") also runs vision.cpp's SigLip2 tower + resampler
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_omni_set.py"), "-o", root, "--vision", "--turns", "2"], check=True, timeout=1800, capture_output=True)
    try:
        out_dir = str(tmp_path / "out")
        r, log = run_omni_min(root, out_dir, turns=2, omni=True)
        assert r.returncode == 0, log[-4000:]
        m = re.search(r"vision using (\S+) backend", log)                                                   # vision.cpp's own line
        assert m and m.group(1).startswith("MI355X"), m and m.group(0)
        assert re.search(r"prefilled \d+ vision chunks \(64 tokens each\)", log), "the picture did not reach the LLM"
        # ---- placement: every module that asks the registry for a GPU got the plug-in's device
        assert len(re.findall(r"offloaded 37/37 layers to GPU", log)) >= 1, log[-3000:]                   # the LLM (36 layers + output)
        assert len(re.findall(r"offloaded 21/21 layers to GPU", log)) >= 1, log[-3000:]                   # the TTS decoder (20 layers + output)
        assert "init_backend device=gpu:0, gpu_idx=0, backend=MI355X0" in log                              # Token2Wav's flow model (token2wav-impl.cpp:1951)
        # the plug-in's own account, printed as each backend context is freed: the LLM, the TTS decoder, the audio encoder (audition.cpp:241-249 asks the
        # registry for a GPU backend; its logger is silent at this level) and Token2Wav's flow model each computed graphs on it (omni_free does not free the image encoder's backend: its evidence is vision.cpp's own line above)
        stats = [(int(a), int(b), int(c), int(d)) for a, b, c, d in re.findall(r"\[mi355x\] MI355X0: graphs eager=(\d+) captured=(\d+) replayed=(\d+), kernels in last graph=(\d+)", log)]
        live = [s for s in stats if s[0] + s[1] + s[2] > 0]
        assert len(live) >= 4 and any(s[3] > 2000 for s in live) and any(s[2] > 500 for s in live), stats        # (Token2Wav's 4 000-launch window graph; the TTS decoder's replays)
        j = summarise(log)
        assert j["registry_devices"].count("MI355X0") == 1, j["registry_devices"]      # (round 6: the loader entry reports its devices once, however often the modules call ggml_backend_load_all)
        # ---- completion: the three threads ran to the end and wrote audio
        assert j["first_wav_s"] > 0 and j["n_wav"] >= 2 and j["n_past_after_decode"] > j["n_past_after_prefill"] > 100
        assert j["reference_first_audio_ms"] is not None and j["t2w_token2mel_ms_median"] is not None
        wav_dir = os.path.join(out_dir, "round_000", "tts_wav")
        wavs = sorted(f for f in os.listdir(wav_dir) if f.endswith(".wav"))
        assert len(wavs) >= 2 and j["t2w_windows"] >= 2
        for f in wavs[:4]:
            x = _read_wav(os.path.join(wav_dir, f))
            assert x.size >= 12000 and np.isfinite(x).all() and float(x.std()) > 1e-3, (f, x.size, float(x.std()))
        print("omni runtime on the plug-in:", json.dumps(j))
    finally:
        shutil.rmtree(root, ignore_errors=True)


def test_pinned_form_one_module_per_gpu_through_the_reference_knobs(tmp_path):
    """BASELINE C4 / C5's pinned form with the reference runtime as caller: tools/omni_runtime.sh --map puts the LLM (+ TTS: omni.cpp:3457 loads it with the LLM's
    common_params) on device 0 and the audio encoder, the image encoder and Token2Wav's flow model on device 1, spelled with the reference's own knobs (-mg / -sm none,
    MTMD_BACKEND_DEVICE, Omni_BACKEND_DEVICE, --t2w-device gpu:1).  Needs two MI355X: skipped on the one-GPU boxes this round's tests ran on (the translation itself is
    covered without a GPU by tests/test_host_mirror.py::test_omni_runtime_device_map_uses_the_reference_knobs)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two MI355X")
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/omni-min not built (make -f oracle/Makefile.ref omnirt)")
    if shutil.disk_usage(str(tmp_path)).free < 9e9:
        pytest.skip("needs 7 GB of scratch disk for the synthetic module set")
    root = str(tmp_path / "set")
    env = dict(os.environ, OMNI_SET=root)
    try:
        r = subprocess.run(["bash", os.path.join(ROOT, "tools", "omni_runtime.sh"), "--map", "llm=0,t2w=1,apm=1,vpm=1", "24", "2", "omni"], env=env, capture_output=True, text=True, timeout=3000)
        assert r.returncode == 0 and "pass 2 exit 0" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
        log = open(os.path.join(ROOT, "gpurun_out", "omni_runtime_pass2.log"), errors="replace").read()
        assert "init_backend device=gpu:1, gpu_idx=1, backend=MI355X1" in log                                # Token2Wav's flow model on the second device
        m = re.search(r"vision using (\S+) backend", log)
        assert m and m.group(1).startswith("MI355X1"), m and m.group(0)
        assert len(re.findall(r"offloaded 37/37 layers to GPU", log)) >= 1 and len(re.findall(r"offloaded 21/21 layers to GPU", log)) >= 1
        used = {d for d, a, b, c in re.findall(r"\[mi355x\] (MI355X\d): graphs eager=(\d+) captured=(\d+) replayed=(\d+)", log) if int(a) + int(b) + int(c) > 0}
        assert used == {"MI355X0", "MI355X1"}, used
    finally:
        shutil.rmtree(root, ignore_errors=True)

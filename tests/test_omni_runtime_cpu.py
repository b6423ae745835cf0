"""Row g1's harness on the CPU tier: the synthetic module-set writer (tools/make_synth_omni_set.py) produces files the REFERENCE's loaders accept -- libllama's
GGUF / BPE-vocabulary loader, audition.cpp, omni.cpp's load_tts_weights_from_gguf and projector_init -- and the reference's omni runtime (oracle/_ref/omni-min) runs
omni_init -> stream_prefill -> stream_decode on its own CPU backend over a shrunken set (one layer per model, TTS off so the run stays inside seconds).  No GPU, no
plug-in: this pins the harness, not the product."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "omni-min")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_special_token_ids_follow_the_reference_constants():
    """ids omni.cpp hard-codes (g_special_token_ids, tools/omni/omni.cpp:4432-4441) and the strings omni_init looks up by text (:3964-3982)"""
    import make_synth_gguf as m
    S = m.OMNI_SPECIALS
    assert S[151667] == "<think>" and S[151668] == "</think>" and S[151704] == "<|tts_eos|>" and S[151705] == "<|listen|>" and S[151706] == "<|speak|>"
    assert S[151717] == "<|turn_eos|>" and S[151718] == "<|chunk_eos|>" and S[151721] == "<|chunk_tts_eos|>"
    for t in ("<|im_start|>", "<|im_end|>", "<|audio_start|>", "<|audio_end|>", "<|tts_bos|>", "<|tts_pad|>", "<unit>", "</unit>", "<image>", "</image>", "<slice>", "</slice>"):
        assert t in S.values(), t
    b2c = m.gpt2_byte_chars()
    assert len(set(b2c.values())) == 256 and b2c[ord("A")] == "A" and b2c[32] == "Ġ" and b2c[10] == "Ċ"       # byte-level BPE's space / newline stand-ins


def test_reference_omni_runtime_on_its_cpu_backend(tmp_path):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/omni-min not built (make -f oracle/Makefile.ref omnirt)")
    root = str(tmp_path / "set")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_omni_set.py"), "-o", root, "--llm-layers", "1", "--tts-layers", "1", "--apm-layers", "1"],
                   check=True, timeout=600, capture_output=True)
    env = dict(os.environ)
    env.pop("GGML_BACKEND_PATH", None)
    r = subprocess.run([BIN, "-m", "gguf/MiniCPM-o-4_5-Q4_K_M.gguf", "--test", "case/audio_", "1", "-ngl", "0", "--no-tts", "--max-tgt", "4", "--out", str(tmp_path / "out"), "-c", "1024"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600, errors="replace")
    log = r.stdout + "\n" + r.stderr
    assert r.returncode == 0, log[-3000:]
    j = json.loads([ln for ln in log.splitlines() if ln.startswith('{"harness"')][-1])
    assert j["registry_devices"] == ["CPU"] and j["n_past_after_prefill"] > 100 and j["n_past_after_decode"] > j["n_past_after_prefill"]
    # the tokenizer: the special tokens came through as single CONTROL tokens (libllama prints the EOG set it found)
    assert re.search(r"EOG token\s+= 151645 '<\|im_end\|>'", log), log[:3000]
    assert "system prompt ref_audio embedding: n_pos=" in log          # the reference voice went through audition.cpp into the LLM
